"""A/B of bench.py's configs[3] leg (8 concurrent streams) over the storage modes: python scripts/time_streams_leg.py [modes...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
modes = sys.argv[1:] or ["bf16_kv", "bf16_act"]
for m in modes:
    r = bench.streams_leg("cuda:0", 0, 8, m, steps=3)
    print(m, json.dumps({k: r[k] for k in ("utts_per_s", "first_chunk_latency_ms", "decode_step_us", "decode_variant")}), "frac", round(r["roofline"]["frac"], 4), flush=True)
