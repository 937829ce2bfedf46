"""Cached (16 uncached rows per stream) vs full prefill of a 1 s chunk: python scripts/time_prefill_cached.py [streams] [WEIGHTS env: fp32 | bf16_kv | bf16_act]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import GptEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
eng = GptEngine(dims, max_slots=max(8, B), max_rows=4096, weight_dtype=os.environ.get("WEIGHTS", "fp32"))
eng.bind(synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda"))
cond = synth.uniform(1, "c", (B, 32, 1024), 1.0).cuda()
codes = synth.integers(1, "k", (B, 13), 256).cuda().int()
slots = torch.arange(B, device="cuda", dtype=torch.int32)
prefix = eng.prefix_embeddings(cond, codes)
for n_cached in (0, 32):
    for _ in range(3):
        eng.prefill(slots, prefix, want_outputs=False, n_cached=n_cached)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        eng.prefill(slots, prefix, want_outputs=False, n_cached=n_cached)
    e1.record()
    torch.cuda.synchronize()
    print(f"B={B} {os.environ.get('WEIGHTS', 'fp32')} prefill n_cached={n_cached}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us  ({B * (48 - n_cached)} rows)")
