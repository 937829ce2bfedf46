"""Decode step time per weight/KV dtype (fp32, bf16 weights, bf16 weights + bf16 KV cache) and batch size.

    python scripts/time_decode_dtype.py [modes] [batches]      e.g.  fp32,bf16,bf16_kv 1,8,16
Under rocprofv3 use one mode and one batch (GVC_TD_DTYPE / GVC_TD_B still work)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import GptEngine, sample_params

modes = (sys.argv[1] if len(sys.argv) > 1 else os.environ.get("GVC_TD_DTYPE", "fp32,bf16,bf16_kv")).split(",")
batches = [int(b) for b in (sys.argv[2] if len(sys.argv) > 2 else os.environ.get("GVC_TD_B", "1,8,16")).split(",")]
dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
w = synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda")
Tc, n = 13, 64
sp = sample_params(dict(gcfg.DEFAULT_SAMPLING, top_k=1), 1026, -1)
for mode in modes:
    eng = GptEngine(dims, max_slots=max(batches + [8]), max_rows=4096, weight_dtype=mode)
    eng.bind(w)
    for B in batches:
        cond = synth.uniform(1, "c", (B, 32, 1024), 1.0).cuda()
        codes = synth.integers(1, "k", (B, Tc), 256).cuda().int()
        slots = torch.arange(B, device="cuda", dtype=torch.int32)
        prefix = eng.prefix_embeddings(cond, codes)
        P = prefix.shape[1]

        def run():
            ids = torch.ones(B, P + 1 + n + 8, device="cuda", dtype=torch.int32)
            ids[:, P] = 1024
            il = torch.full((B,), P + 1, device="cuda", dtype=torch.int32)
            fin = torch.zeros(B, device="cuda", dtype=torch.int32)
            toks = torch.zeros(B, n, device="cuda", dtype=torch.int32)
            lats = torch.zeros(B, n, 1024, device="cuda")
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            e[0].record()
            eng.prefill(slots, prefix, want_outputs=False)
            e[1].record()
            eng.generate(slots, ids, il, fin, sp, 0, n, toks, lats)
            e[2].record()
            torch.cuda.synchronize()
            return e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2]) / n * 1000

        run()
        r = [run() for _ in range(3)]
        print(f"{mode:8s} B={B:3d}  prefill({P + 1} rows/stream) {min(x[0] for x in r):7.3f} ms   decode {min(x[1] for x in r):7.1f} us/step", flush=True)
    eng.close()
