"""GEMM kernels alone (gvc_gemm_probe): time and check the tiled / strip / skinny kernels at the prefill shapes.
usage: time_gemm.py [M ...]   (default: the GPT prefill row counts 48 110 550 880)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import _lib

PEAK = 157.3e12
lib = _lib.lib()


def probe(variant, A, W, bias, sk_max=8, iters=20):
    M, K = A.shape
    N = W.shape[0]
    out = torch.empty(M, N, device="cuda")
    us = C.c_float(0)
    rc = lib.gvc_gemm_probe(variant, _lib.ptr(A), _lib.ptr(W), _lib.ptr(bias), _lib.ptr(out), M, N, K, sk_max, iters,
                            C.byref(us), _lib.stream())
    _lib.check(rc, "gemm_probe")
    return out, us.value


Ms = [int(a) for a in sys.argv[1:]] or [48, 110, 550, 880]
g = torch.Generator(device="cpu").manual_seed(1)
for M in Ms:
    for name, N, K in (("c_attn", 3072, 1024), ("attn c_proj", 1024, 1024), ("c_fc", 4096, 1024), ("mlp c_proj", 1024, 4096)):
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) * 0.02).cuda()
        b = torch.randn(N, generator=g).cuda()
        ref = (A.double() @ W.double().T + b.double()).float()
        line = f"M={M:4d} {name:12s} N={N} K={K}:"
        for v, vn in ((0, "tiled"), (1, "strip"), (2, "skinny")):
            if v == 2 and M > 128:
                continue
            out, us = probe(v, A, W, b)
            err = (out - ref).abs().max().item()
            tf = 2.0 * M * N * K / (us * 1e-6) / 1e12
            line += f"  {vn} {us:7.1f} us {tf:5.1f} TF ({tf / PEAK * 1e14:4.1f}%) err {err:.1e}"
        print(line)
