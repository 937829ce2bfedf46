"""Prefill timing and fp32-MFMA utilisation: rows T = P+1 per stream, B streams (2*T*L*12d^2 + 4*L*d*T(T+1)/2 FLOP)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import GptEngine

PEAK_F32_MFMA = 157.3e12
dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
d, L = dims["d_model"], dims["n_layer"]
w = synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda")
eng = GptEngine(dims, max_slots=8, max_rows=4096)
eng.bind(w)
cases = [(1, 13), (1, 75), (5, 75), (8, 75), (1, 402)] if len(sys.argv) < 2 else [(int(sys.argv[1]), int(sys.argv[2]))]
for B, Tc in cases:
    cond = synth.uniform(1, "c", (B, 32, d), 1.0).to("cuda")
    codes = synth.integers(1, "k", (B, Tc), 256).to("cuda").int()
    slots = torch.arange(B, device="cuda", dtype=torch.int32)
    prefix = eng.prefix_embeddings(cond, codes)
    T = prefix.shape[1] + 1
    for _ in range(2):
        eng.prefill(slots, prefix, want_outputs=False)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 5
    e0.record()
    for _ in range(n):
        eng.prefill(slots, prefix, want_outputs=False)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    flops = B * (2 * T * L * 12 * d * d + 4 * L * d * T * (T + 1) / 2)
    print(f"prefill B={B} T={T}: {ms:.3f} ms  {flops / 1e9:.1f} GFLOP  {flops / ms / 1e9:.2f} TFLOP/s = "
          f"{flops / (ms * 1e-3) / PEAK_F32_MFMA * 100:.1f}% of the 157.3 TF fp32-MFMA peak; weight floor {1.516e9 / 6.3e12 * 1e3:.2f} ms")
