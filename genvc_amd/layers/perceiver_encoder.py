"""PerceiverResampler with the reference's constructor and forward signature
(/root/reference/layers/perceiver_encoder.py:225-276); the arithmetic runs in libgenvc_hip
(gvc_perceiver_forward).  Parameter names match the reference state dict, so
`load_state_dict(ckpt)` works unchanged."""
import torch
from torch import nn

from ..engine import PerceiverEngine


class _Holder(nn.Module):
    """Parameter container (the forward math lives in the HIP library)."""


def _linear(out_f, in_f, bias=True):
    m = _Holder()
    m.weight = nn.Parameter(torch.empty(out_f, in_f).normal_(std=0.02), requires_grad=False)
    if bias:
        m.bias = nn.Parameter(torch.zeros(out_f), requires_grad=False)
    return m


class PerceiverResampler(nn.Module):
    def __init__(self, *, dim, depth=2, dim_context=None, num_latents=32, dim_head=64, heads=8, ff_mult=4,
                 use_flash_attn=False):
        super().__init__()
        dim_context = dim if dim_context is None else dim_context
        self.cfg = dict(dim=dim, depth=depth, dim_context=dim_context, num_latents=num_latents, dim_head=dim_head,
                        heads=heads, ff_mult=ff_mult)
        inner = dim_head * heads
        ffi = int(dim * ff_mult * 2 / 3)
        self.proj_context = _linear(dim, dim_context) if dim_context != dim else nn.Identity()
        self.latents = nn.Parameter(torch.empty(num_latents, dim).normal_(std=0.02), requires_grad=False)
        self.layers = nn.ModuleList()
        for _ in range(depth):
            attn = _Holder()
            attn.to_q, attn.to_kv, attn.to_out = _linear(inner, dim, False), _linear(2 * inner, dim, False), _linear(dim, inner, False)
            ff = nn.ModuleList([_linear(2 * ffi, dim), _Holder(), _linear(dim, ffi)])     # indices 0 and 2 as in nn.Sequential
            self.layers.append(nn.ModuleList([attn, ff]))
        self.norm = _Holder()
        self.norm.gamma = nn.Parameter(torch.ones(dim), requires_grad=False)
        self._engine = None

    def bind(self, max_batch=8, max_frames=2816):
        """(re)upload the current parameters to the HIP context"""
        if self._engine is not None:
            self._engine.close()
        self._engine = PerceiverEngine(max_batch=max_batch, max_frames=max_frames, **self.cfg)
        self._engine.bind({k: v for k, v in self.state_dict().items()})
        return self

    @torch.inference_mode()
    def forward(self, x, mask=None):
        """x [B,F,dim_context] -> [B,num_latents,dim]"""
        if mask is not None:
            raise NotImplementedError("PerceiverResampler mask is a training-only path (reference gpt.py:362-367)")
        if self._engine is None:
            self.bind()
        return self._engine.forward(x.to(torch.float32).contiguous())
