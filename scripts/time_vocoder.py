"""HiFi-GAN call of the streaming loop: 8 latents -> x4 interpolation -> 8192 samples."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import HifiganEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cfg = gcfg.DEFAULT_VOCODER
eng = HifiganEngine(cfg, max_batch=max(B, 2), max_frames=max(4 * n, 64))
eng.bind(synth.make_weights(5, synth.hifigan_weight_spec(cfg), device="cuda"))
lat = synth.uniform(7, "lat", (B, n, 1024), 1.0).cuda()
for _ in range(3):
    eng.forward_latents(lat, 4)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 50
import time
torch.cuda.synchronize()
e0.record()
h0 = time.perf_counter()
for _ in range(reps):
    eng.forward_latents(lat, 4)
h1 = time.perf_counter()
e1.record()
torch.cuda.synchronize()
print(f"hifigan B={B} n={n}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us per call (host enqueue {(h1 - h0) / reps * 1e6:.1f} us per call)")
# one call at a time, as the streaming loop issues it (the host waits for the tokens before it can enqueue the vocoder)
t = 0.0
for _ in range(reps):
    torch.cuda.synchronize()
    h0 = time.perf_counter()
    eng.forward_latents(lat, 4)
    torch.cuda.synchronize()
    t += time.perf_counter() - h0
print(f"  isolated call, enqueue -> synchronized: {t / reps * 1e6:.1f} us")
