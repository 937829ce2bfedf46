"""ContentVec boundary (reference layers/content_processor.py:7-34).

The reference calls fairseq's HuBERT-base (`extract_features(output_layer=12)` + `final_proj`), a
third-party model that is neither vendored in the reference nor installed here, and whose weights
(`contentVec.pt`) are not available: SURVEY.md section 8a row 4 marks it as the boundary INPUT of this
build (a PyTorch-ROCm restatement is the "next" row f3).  What this file provides:

  * `contentvec_frames(T)`: the exact frame count of the HuBERT conv stack (k=[10,3,3,3,3,2,2],
    s=[5,2,2,2,2,2,2]) so shapes downstream are right (16000 samples -> 49 frames, 96000 -> 299);
  * `SyntheticContentExtractor`: a deterministic stand-in with the reference's interface
    (`extract_content_features(wavs[B,T]) -> [B,T50,256]`) used for plumbing and benchmarks.
    It is NOT ContentVec and makes no parity claim.
"""
import torch
from torch import nn

_K = (10, 3, 3, 3, 3, 2, 2)
_S = (5, 2, 2, 2, 2, 2, 2)


def contentvec_frames(n_samples):
    n = n_samples
    for k, s in zip(_K, _S):
        n = (n - k) // s + 1
    return n


class SyntheticContentExtractor(nn.Module):
    def __init__(self, dim=256, seed=0):
        super().__init__()
        from .. import synth
        self.dim = dim
        # 400-sample receptive field / 320-sample hop of the real conv stack
        self.register_buffer("proj", synth.uniform(seed, "synthetic_contentvec.proj", (dim, 400), 0.25))

    @property
    def model(self):
        """the reference's model_init touches `.content_extractor.model` (model_init.py:29-30)"""
        return self

    @torch.inference_mode()
    def extract_content_features(self, wavs):
        B, T = wavs.shape
        n = contentvec_frames(T)
        frames = wavs.unfold(1, 400, 320)[:, :n]                      # [B,n,400]
        return torch.tanh(frames.to(torch.float32) @ self.proj.t().to(wavs.device) * 8.0)

    def forward(self, wavs):
        return self.extract_content_features(wavs)
