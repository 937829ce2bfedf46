"""BASELINE configs[2]: batched offline conversion (tokens only, as parallel_offline.convert_offline produces them) of 64
synthetic 10 s utterances (segments 6 s + 4 s), top_k = 1, fixed token budget 141 per segment (synthetic weights rarely stop),
one GPU, micro-batches of 8 / 16 / 32 utterances.  Not the headline benchmark (bench.py)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.inference.model_init import model_init_synthetic
from genvc_amd.parallel_offline import convert_offline

N_UTT = int(os.environ.get("N_UTT", "64"))
m, cfg = model_init_synthetic(gcfg.default_config(), seed=1, device="cuda", max_slots=32)
m.config.top_k = 1
srcs = [synth.synth_audio(500 + i, "src", 160000) for i in range(N_UTT)]
ref = synth.synth_audio(7, "ref", 72000)
out = {}
for mb in (8, 16, 32):
    convert_offline(m, srcs[:mb], ref, seg_len=6.0, micro_batch=mb, top_k=1, max_new_tokens=141)       # warm-up / graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    toks = convert_offline(m, srcs, ref, seg_len=6.0, micro_batch=mb, top_k=1, max_new_tokens=141)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out[f"micro_batch_{mb}"] = {"utterances_per_s": N_UTT / dt, "seconds": dt, "tokens": int(toks.shape[1] * 141 * N_UTT)}
print(json.dumps({"workload": f"{N_UTT} x 10 s utterances, segments 6 s + 4 s, 141 tokens per segment, top_k=1, tokens only",
                  "n_gpus": 1, **out}))
