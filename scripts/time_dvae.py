"""DVAE + VQ of one streaming chunk: 49 ContentVec frames -> 13 codes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import DvaeEngine

T = int(sys.argv[1]) if len(sys.argv) > 1 else 49
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = gcfg.DEFAULT_CONTENT_DVAE
eng = DvaeEngine(cfg, max_batch=max(B, 2), max_frames=max(T, 64))
eng.bind(synth.make_weights(3, synth.dvae_weight_spec(cfg), device="cuda"))
feat = synth.uniform(5, "f", (B, cfg["num_channels"], T), 1.0).cuda()
for _ in range(3):
    eng.encode(feat)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 50
e0.record()
for _ in range(n):
    eng.encode(feat)
e1.record()
torch.cuda.synchronize()
print(f"dvae B={B} T={T}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per call")
