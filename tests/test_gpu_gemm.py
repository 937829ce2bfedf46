"""fp32 MFMA GEMM kernels under the GPT prefill (csrc/gemm.hip) against an fp64 matmul, through the C-ABI measurement hook
gvc_gemm_probe: the strip kernel (batched prefill, > 128 rows), the 64x64x32 tiled kernel and the skinny kernel (<= 128 rows)
compute C = A W^T + bias for the reference's projection shapes (GPT2Block c_attn / c_proj / c_fc / mlp c_proj,
/root/reference/layers/gpt_inference.py:81-91 drives them through transformers' GPT2Model)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _probe(variant, A, W, b, sk_max):
    from genvc_amd import _lib
    out = torch.empty(A.shape[0], W.shape[0], device="cuda")
    us = C.c_float(0)
    _lib.check(_lib.lib().gvc_gemm_probe(variant, _lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), A.shape[0], W.shape[0],
                                         A.shape[1], sk_max, 0, C.byref(us), _lib.stream()), "gvc_gemm_probe")
    return out


@pytest.mark.parametrize("M", [1, 17, 110, 129, 437, 550, 881, 2048])
def test_gemm_kernels_vs_fp64(M):
    g = torch.Generator(device="cpu").manual_seed(M)
    # (N, K): tiny model (d = 256), d = 512 / 768 variants, the full model's four projections
    for N, K in ((768, 256), (256, 1024), (1536, 512), (768, 3072), (3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)):
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) * 0.05).cuda()
        b = torch.randn(N, generator=g).cuda()
        ref = (A.double() @ W.double().T + b.double()).float()
        tol = 2e-5 * (K / 256) ** 0.5 * max(1.0, ref.abs().max().item())
        for variant, sk in ((0, 1), (1, 1), (1, 8)) + (((2, 1),) if M <= 128 else ()):
            out = _probe(variant, A, W, b, sk)
            err = (out - ref).abs().max().item()
            assert err < tol, (M, N, K, variant, sk, err)


def test_probe_rejects_unsupported_shapes():
    from genvc_amd import _lib
    A = torch.zeros(4, 24, device="cuda")
    W = torch.zeros(16, 24, device="cuda")
    out = torch.empty(4, 16, device="cuda")
    us = C.c_float(0)
    rc = _lib.lib().gvc_gemm_probe(1, _lib.ptr(A), _lib.ptr(W), None, _lib.ptr(out), 4, 16, 24, 1, 0, C.byref(us), _lib.stream())
    assert rc != 0          # K % 16 != 0: refused, not computed wrong
