"""TorchMelSpectrogram with the reference's signature (/root/reference/utils.py:97-162) on
libgenvc_hip (gvc_mel_forward)."""
import os

import numpy as np
import torch
from torch import nn

from .engine import MelEngine

DEFAULT_MEL_NORM_FILE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "mel_stats.npy")


def load_mel_norms(path):
    """mel_stats.pth of the reference (a float32[80] tensor) or the .npy copy shipped in assets/."""
    if path is None:
        return None
    if path.endswith(".npy"):
        return np.load(path).astype(np.float32)
    return torch.load(path, map_location="cpu").numpy().astype(np.float32)


class TorchMelSpectrogram(nn.Module):
    def __init__(self, filter_length=1024, hop_length=256, win_length=1024, n_mel_channels=80, mel_fmin=0,
                 mel_fmax=8000, sampling_rate=22050, normalize=False, mel_norm_file=DEFAULT_MEL_NORM_FILE):
        super().__init__()
        if normalize:
            raise NotImplementedError("normalized=True STFT is not used by GenVC (hifigan_trainer.py:105-115)")
        norms = load_mel_norms(mel_norm_file)
        self.mel_norms = None if norms is None else torch.from_numpy(norms)
        self.args = dict(n_fft=filter_length, hop=hop_length, win=win_length, sample_rate=sampling_rate,
                         f_min=float(mel_fmin), f_max=float(mel_fmax), n_mels=n_mel_channels)
        self._engine = None

    @torch.inference_mode()
    def forward(self, inp, frames_major=False):
        if inp.dim() == 3:
            inp = inp.squeeze(1)
        assert inp.dim() == 2
        if self._engine is None:
            norms = np.ones(self.args["n_mels"], np.float32) if self.mel_norms is None else self.mel_norms.numpy()
            self._engine = MelEngine(norms, **self.args)
        return self._engine.forward(inp.to(torch.float32).contiguous(), frames_major=frames_major)
