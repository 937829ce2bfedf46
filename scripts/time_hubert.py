"""ContentVec forward timing (1 s chunk by default)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import HubertEngine

T = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
c = gcfg.DEFAULT_HUBERT
eng = HubertEngine(c, max_batch=max(B, 2), max_samples=max(T, 16000))
eng.bind(synth.make_weights(17, synth.hubert_weight_spec(c), device="cuda"))
wav = torch.cat([synth.synth_audio(3 + b, "w", T) for b in range(B)], 0).cuda()
for _ in range(3):
    eng.forward(wav)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 20
e0.record()
for _ in range(n):
    eng.forward(wav)
e1.record()
torch.cuda.synchronize()
print(f"hubert B={B} T={T}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per forward ({eng.frames(T)} frames)")
