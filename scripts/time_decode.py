"""Quick timing probe: decode step (graph), prefill, per-kernel via rocprof if wrapped."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import GptEngine, sample_params

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Tc = int(sys.argv[2]) if len(sys.argv) > 2 else 13
nsteps = int(sys.argv[3]) if len(sys.argv) > 3 else 64
dims = gcfg.gpt_dims(dict(gcfg.DEFAULT_MODEL_ARGS, gpt_n_heads=int(os.environ.get("HEADS", "4"))))
w = synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda")
eng = GptEngine(dims, max_slots=max(B, 8), max_rows=4096, weight_dtype=os.environ.get("WEIGHTS", "fp32"))    # WEIGHTS=bf16 / bf16_kv: BASELINE configs[3] storage
eng.bind(w)
dev = "cuda"
cond = synth.uniform(1, "c", (B, 32, 1024), 1.0).to(dev)
codes = synth.integers(1, "k", (B, Tc), 256).to(dev).int()
slots = torch.arange(B, device=dev, dtype=torch.int32)
prefix = eng.prefix_embeddings(cond, codes)
P = prefix.shape[1]
sp = sample_params(dict(gcfg.DEFAULT_SAMPLING, top_k=int(os.environ.get("TOPK", "1"))), 1026, 1025)

def run(n):
    ids = torch.ones(B, P + 1 + n + 8, device=dev, dtype=torch.int32); ids[:, P] = 1024
    ids_len = torch.full((B,), P + 1, device=dev, dtype=torch.int32)
    fin = torch.zeros(B, device=dev, dtype=torch.int32)
    toks = torch.zeros(B, n, device=dev, dtype=torch.int32)
    lats = torch.zeros(B, n, 1024, device=dev)
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    eng.prefill(slots, prefix, want_outputs=False)
    e1.record()
    eng.generate(slots, ids, ids_len, fin, sp, 0, n, toks, lats)
    e2.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1), e1.elapsed_time(e2)

run(8)
for _ in range(3):
    tp, tg = run(nsteps)
    print(f"B={B} P={P} prefill({P+1} rows) {tp:.3f} ms ; decode {tg/nsteps*1000:.1f} us/step ({nsteps} steps) "
          f"-> {1.5159e9/(tg/nsteps*1e-3)/1e12:.2f} TB/s weights-only")
# eager decode steps for comparison
tok = torch.zeros(B, device=dev, dtype=torch.int32)
lg = torch.empty(B, 1026, device=dev); lt = torch.empty(B, 1024, device=dev)
torch.cuda.synchronize(); t = time.time()
for _ in range(32): eng.decode_step(slots, tok, lg, lt)
torch.cuda.synchronize(); print(f"eager decode_step {(time.time()-t)/32*1e6:.1f} us/step")
