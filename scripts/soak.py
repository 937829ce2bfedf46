"""Soak: N utterances of the bench workload back to back; free HBM and host RSS must stay flat, tokens must repeat."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import psutil
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
streams = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = bench.Workload("cuda:0", 0, streams, os.environ.get("WEIGHTS", "fp32"), max_slots=max(8, streams))      # WEIGHTS=bf16_act: the bf16-activation rows step
proc = psutil.Process()
ref = {}
t0 = time.time()
for u in range(n):
    toks = wl.utterance(u).clone()
    torch.cuda.synchronize()
    key = u % 4
    if key in ref:
        assert torch.equal(ref[key], toks), f"utterance {u}: tokens differ from the first run of input {key}"
    else:
        ref[key] = toks
    if u % 20 == 0 or u == n - 1:
        free, total = torch.cuda.mem_get_info()
        print(f"utt {u:4d}  free HBM {free / 2**20:10.1f} MiB  host RSS {proc.memory_info().rss / 2**20:8.1f} MiB  {time.time() - t0:6.1f} s", flush=True)
wl.eng.health()
print("ok")
