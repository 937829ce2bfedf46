"""Content DiscreteVAE tokenizer: `get_codebook_indices` of the reference
(/root/reference/layers/dvae.py:203-331, inference part only) on libgenvc_hip (gvc_dvae_encode)."""
import torch
from torch import nn

from ..engine import DvaeEngine


class _Holder(nn.Module):
    pass


def _conv(cout, cin, k):
    m = _Holder()
    m.weight = nn.Parameter(torch.empty(cout, cin, k).normal_(std=0.02), requires_grad=False)
    m.bias = nn.Parameter(torch.zeros(cout), requires_grad=False)
    return m


class DiscreteVAE(nn.Module):
    def __init__(self, positional_dims=1, num_tokens=512, codebook_dim=512, num_layers=3, num_resnet_blocks=0,
                 hidden_dim=64, channels=3, stride=2, kernel_size=4, use_transposed_convs=True, encoder_norm=False,
                 activation="relu", normalization=None, **_unused):
        super().__init__()
        if positional_dims != 1 or stride != 2 or encoder_norm or activation != "relu" or normalization is not None:
            raise NotImplementedError("only the 1-D, stride-2, ReLU, un-normalised content DVAE of GenVC is supported")
        self.cfg = dict(num_channels=channels, num_tokens=num_tokens, codebook_dim=codebook_dim, hidden_dim=hidden_dim,
                        num_resnet_blocks=num_resnet_blocks, kernel_size=kernel_size, num_layers=num_layers)
        self.num_tokens = num_tokens
        layers = []
        cin = channels
        for i in range(num_layers):
            blk = nn.ModuleList([_conv(hidden_dim * 2 ** i, cin, kernel_size)])      # encoder.{i}.0.*
            layers.append(blk)
            cin = hidden_dim * 2 ** i
        for _ in range(num_resnet_blocks):
            rb = _Holder()
            rb.net = nn.ModuleList([_conv(cin, cin, 3), _Holder(), _conv(cin, cin, 3), _Holder(), _conv(cin, cin, 1)])
            layers.append(rb)                                                          # encoder.{i}.net.{0,2,4}.*
        layers.append(_conv(codebook_dim, cin, 1))                                     # encoder.{last}.*
        self.encoder = nn.ModuleList(layers)
        self.codebook = _Holder()
        self.codebook.register_buffer("embed", torch.randn(codebook_dim, num_tokens))
        self._engine = None

    def bind(self, max_batch=8, max_frames=1504):
        if self._engine is not None:
            self._engine.close()
        self._engine = DvaeEngine(self.cfg, max_batch=max_batch, max_frames=max_frames)
        self._engine.bind(dict(self.state_dict()))
        return self

    @torch.inference_mode()
    def get_codebook_indices(self, images):
        """images [B,channels,T] -> int64 [B,Tc]"""
        if self._engine is None:
            self.bind()
        images = images.to(torch.float32)
        if not images.is_contiguous() and images.transpose(1, 2).is_contiguous():
            # the harness passes `content_feat.transpose(1, 2)`: hand the frame-major storage over as it is
            return self._engine.encode(images.transpose(1, 2), frames_major=True).long()
        return self._engine.encode(images.contiguous()).long()
