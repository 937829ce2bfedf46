"""Where the first chunk's latency goes: HIP events at the stage marks of bench.Workload.utterance (chunk 0 only) and the HOST clock at which
each mark was enqueued (host ahead of the device = the device never waits for a launch).  python scripts/first_chunk_stages.py [utterances]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
wl = bench.Workload(torch.device("cuda:0"), 0)
for u in range(3):
    wl.utterance(u)
torch.cuda.synchronize()
rows = []
for u in range(n):
    host = []
    wl.stage_ev = []
    orig_append = wl.stage_ev.append
    class L(list):
        def append(self, x):
            host.append(time.perf_counter())
            list.append(self, x)
    wl.stage_ev = L()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.utterance(u)
    torch.cuda.synchronize()
    ev, wl.stage_ev = list(wl.stage_ev), None
    first = ev[:6]                 # start, contentvec+dvae, prefill, decode, vocoder
    line = []
    for i in range(1, len(first)):
        line.append((first[i][0][:28], ev[0][1].elapsed_time(first[i][1]), (host[i] - host[0]) * 1e3))
    rows.append(line)
for line in rows:
    print("  ".join(f"{n}: dev {d:6.3f} host {h:6.3f}" for n, d, h in line))
