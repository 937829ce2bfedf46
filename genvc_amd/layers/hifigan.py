"""HiFi-GAN generator with the reference's constructor and `forward(x[B,d,T]) -> [B,1,256T]`
(/root/reference/layers/hifigan.py:160-233) on libgenvc_hip (gvc_hifigan_forward).  Holds the
weight-norm parametrised tensors under the reference's names (conv_pre.weight_g/_v, ups.{i}.*, resblocks.*)."""
import torch
from torch import nn

from ..engine import HifiganEngine


class _Holder(nn.Module):
    pass


def _wn(shape, bias_n):
    m = _Holder()
    m.weight_g = nn.Parameter(torch.ones(shape[0], 1, 1), requires_grad=False)
    m.weight_v = nn.Parameter(torch.empty(*shape).normal_(std=0.01), requires_grad=False)
    m.bias = nn.Parameter(torch.zeros(bias_n), requires_grad=False)
    return m


class HiFiGAN(nn.Module):
    def __init__(self, input_feat_dim, upsample_initial_channel, resblock_kernel_sizes, resblock_dilation_sizes,
                 upsample_rates, upsample_kernel_sizes, resblock_type="1"):
        super().__init__()
        if str(resblock_type) != "2":
            raise NotImplementedError("GenVC's vocoder uses ResBlock2 (configs/vocoder_configs.py:20)")
        self.cfg = dict(input_feat_dim=input_feat_dim, upsample_initial_channel=upsample_initial_channel,
                        resblock_kernel_sizes=list(resblock_kernel_sizes),
                        resblock_dilation_sizes=[list(d) for d in resblock_dilation_sizes],
                        upsample_rates=list(upsample_rates), upsample_kernel_sizes=list(upsample_kernel_sizes))
        ch = upsample_initial_channel
        self.conv_pre = _wn((ch, input_feat_dim, 7), ch)
        self.ups = nn.ModuleList()
        self.resblocks = nn.ModuleList()
        for r, k in zip(upsample_rates, upsample_kernel_sizes):
            self.ups.append(_wn((ch, ch // 2, k), ch // 2))
            ch //= 2
            for kk in resblock_kernel_sizes:
                rb = _Holder()
                rb.convs = nn.ModuleList([_wn((ch, ch, kk), ch), _wn((ch, ch, kk), ch)])
                self.resblocks.append(rb)
        self.conv_post = _wn((1, ch, 7), 1)
        self._engine = None

    def bind(self, max_batch=2, max_frames=2560):
        if self._engine is not None:
            self._engine.close()
        self._engine = HifiganEngine(self.cfg, max_batch=max_batch, max_frames=max_frames)
        self._engine.bind(dict(self.state_dict()))
        return self

    # frames of context kept on each side of a window of a long input: the generator's receptive field is +-13 input frames
    # (conv_pre 3, the k = 7 / dilation 12 ResBlock of the first upsampling stage 6, the rest < 1 each)
    def receptive_frames(self):
        """input frames on either side that can reach one output sample (reference hifigan.py:119-157,218-233: conv_pre k = 7, per
        stage a transposed conv of kernel k_up at the stage's output rate and ResBlock2s whose two dilated convs reach
        (k - 1) / 2 * d samples each; conv_post k = 7): 11.5 frames for the trained configuration"""
        rf, cum = 3.0, 1.0
        for r, ku in zip(self.cfg["upsample_rates"], self.cfg["upsample_kernel_sizes"]):
            cum *= r
            reach = max(sum((k - 1) // 2 * d for d in ds) for k, ds in zip(self.cfg["resblock_kernel_sizes"], self.cfg["resblock_dilation_sizes"]))
            rf += (ku / 2.0 + reach) / cum
        return rf + 3.0 / cum

    @property
    def window_overlap(self):
        """frames re-computed on either side of a window's kept interior: the receptive field rounded up to a multiple of 8 plus
        a margin (32 for the trained configuration)"""
        import math
        return int(math.ceil((self.receptive_frames() + 8) / 8.0)) * 8 + 8

    @torch.inference_mode()
    def forward(self, x):
        """x [B,d,T] -> [B,1,T * prod(upsample_rates)].  The reference has no length limit (non-streaming conversion vocodes the
        latents of ALL segments in one call, inference_utils.py:79-87): inputs beyond the engine's buffers go through it in
        overlapping windows whose interiors are stitched (every kept sample sees its full receptive field)."""
        if self._engine is None:
            self.bind()
        x = x.to(torch.float32)
        T, cap, ov = x.shape[-1], self._engine.max_frames, self.window_overlap
        if T <= cap:
            return self._engine.forward(x.contiguous())
        if cap <= 2 * ov + 8:
            raise ValueError(f"HiFiGAN: {T} frames exceed the engine's buffers ({cap} frames) and those are too small for overlapping "
                             f"windows (receptive field {self.receptive_frames():.1f} frames, overlap {ov}): bind(max_frames=...) larger")
        up = 1
        for r in self.cfg["upsample_rates"]:
            up *= r
        core = cap - 2 * ov
        out = []
        for s in range(0, T, core):
            lo, hi = max(0, s - ov), min(T, s + core + ov)
            y = self._engine.forward(x[:, :, lo:hi].contiguous())
            n = min(core, T - s)
            out.append(y[:, :, (s - lo) * up:(s - lo + n) * up].clone())
        return torch.cat(out, dim=-1)

    @torch.inference_mode()
    def forward_latents(self, latents, scale=4):
        """latents [B,n,d] -> wav; fuses the harness's F.interpolate(scale_factor=4, mode='linear')"""
        if self._engine is None:
            self.bind()
        latents = latents.to(torch.float32)
        if latents.shape[1] * scale > self._engine.max_frames:      # long input: the reference's own interpolation call, then windows
            mel = torch.nn.functional.interpolate(latents.transpose(1, 2), scale_factor=[float(scale)], mode="linear")
            return self.forward(mel)
        return self._engine.forward_latents(latents.contiguous(), scale)
