"""Strip / tiled GEMM kernels against an fp64 matmul over a grid of shapes (gvc_gemm_probe)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import _lib

lib = _lib.lib()
g = torch.Generator(device="cpu").manual_seed(3)
bad = 0
for M in (17, 129, 437, 550, 880, 1000, 2048):
    for N, K in ((768, 256), (256, 256), (1024, 256), (256, 1024), (1536, 512), (2304, 768), (768, 3072), (3072, 1024)):
        A = torch.randn(M, K, generator=g).cuda()
        W = (torch.randn(N, K, generator=g) * 0.05).cuda()
        b = torch.randn(N, generator=g).cuda()
        ref = (A.double() @ W.double().T + b.double()).float()
        for sk in (1, 8):
            out = torch.empty(M, N, device="cuda")
            us = C.c_float(0)
            _lib.check(lib.gvc_gemm_probe(1, _lib.ptr(A), _lib.ptr(W), _lib.ptr(b), _lib.ptr(out), M, N, K, sk, 0, C.byref(us), _lib.stream()), "probe")
            err = (out - ref).abs().max().item()
            if not err < 1e-4:
                bad += 1
                print(f"M={M} N={N} K={K} sk_max={sk}: err {err:.3e}  rows wrong: {((out - ref).abs().amax(1) > 1e-4).nonzero().flatten()[:8].tolist()} cols wrong: {((out - ref).abs().amax(0) > 1e-4).nonzero().flatten()[:8].tolist()}")
print("bad cases:", bad)
