"""ContentVec (reference layers/content_processor.py:7-34) on libgenvc_hip.

The reference calls fairseq's HuBERT-base checkpoint `contentVec.pt`: `extract_features(source, padding_mask,
output_layer=12)[0]` followed by `final_proj`.  fairseq is neither vendored in the reference nor installed here, so
the forward is restated (oracle/genvc_oracle.py:hubert_extract_features, pinned to HuggingFace's HubertModel) and
built as HIP kernels behind `gvc_hubert_*` (genvc_amd/csrc/hubert.hip).  This file provides

  * `contentvec_frames(T)`: the frame count of the conv stack (k=[10,3,3,3,3,2,2], s=[5,2,2,2,2,2,2]);
  * `ContentvecExtractor`: the reference's interface (`.model`, `extract_content_features(wavs[B,T]) -> [B,T50,256]`)
    whose `.model` holds the fairseq-named parameters, so a GenVC checkpoint's `content_extractor.model.*` keys load
    with `load_state_dict`.  The reference's `padding_mask = (wav == 0)` (:24) is applied inside `gvc_hubert_forward` with
    fairseq's frame reduction (a frame whose whole chunk of samples is exactly zero is padding: zeroed ahead of the positional
    conv, excluded as an attention key) -- it matters for the harness's zero-padded tail segment and for digital silence.
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


class _Holder(nn.Module):
    pass


def _register(root, dotted, shape):
    parts = dotted.split(".")
    m = root
    for p in parts[:-1]:
        if p not in m._modules:
            m.add_module(p, _Holder())
        m = m._modules[p]
    m.register_parameter(parts[-1], nn.Parameter(torch.zeros(*shape), requires_grad=False))


class ContentvecExtractor(nn.Module):
    def __init__(self, cfg=None, max_batch=2, max_samples=16000 * 30):
        super().__init__()
        from .. import config as gcfg, synth
        self.cfg = dict(cfg or gcfg.DEFAULT_HUBERT)
        self.model = _Holder()
        for name, (shape, _) in synth.hubert_weight_spec(self.cfg).items():
            _register(self.model, name, shape)
        self.max_batch, self.max_samples = max_batch, max_samples
        self._engine = None

    def bind(self):
        from ..engine import HubertEngine
        if self._engine is not None:
            self._engine.close()
        self._engine = HubertEngine(self.cfg, max_batch=self.max_batch, max_samples=self.max_samples)
        self._engine.bind(dict(self.model.state_dict()))
        return self

    @torch.inference_mode()
    def extract_content_features(self, wavs):
        """wavs [B,T] at 16 kHz -> [B,T50,256]"""
        if self._engine is None:
            self.bind()
        device = next(self.model.parameters()).device
        return self._engine.forward(wavs.to(device=device, dtype=torch.float32).contiguous())

    def forward(self, wavs):
        return self.extract_content_features(wavs)
