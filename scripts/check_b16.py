"""dev check of the bf16-activation rows step (weight_dtype 3) against the oracle's act_bf16 mode: prints agreement and deviations"""
import sys, os, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import GptEngine
from oracle import genvc_oracle as O
from test_gpu_gpt import run_generate, _round_bf16

GREEDY = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
L = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
Tc = int(sys.argv[3]) if len(sys.argv) > 3 else 13
n = int(sys.argv[4]) if len(sys.argv) > 4 else 24
H = int(sys.argv[5]) if len(sys.argv) > 5 else 4
margs = dict(gcfg.DEFAULT_MODEL_ARGS, gpt_layers=L, gpt_n_heads=H)
dims = gcfg.gpt_dims(margs)
w = synth.make_weights(5, synth.gpt_weight_spec(dims), device="cuda")
wr = _round_bf16({k: v.cpu() for k, v in w.items()})
cond = synth.uniform(100, "cond_latents", (B, 32, 1024), 1.0)
codes = synth.integers(100, "content_codes", (B, Tc), 256)
res = {}
for mode in ("bf16_kv", "bf16_act"):
    eng = GptEngine(dims, max_slots=max(B, 8), max_rows=2048, weight_dtype=mode)
    eng.bind(w)
    t0 = time.time()
    _, toks, lats = run_generate(eng, dims, cond, codes, n)
    torch.cuda.synchronize()
    print(mode, "variant", eng.decode_variant(), f"{time.time() - t0:.2f}s")
    eng.health()
    res[mode] = (toks, lats)
    eng.close()
for mode, od in (("bf16_kv", dict(dims, kv_bf16=True)), ("bf16_act", dict(dims, kv_bf16=True, act_bf16=True))):
    ref_t, ref_l, ref_logits = O.generate(wr, od, cond, codes, GREEDY, max_new=n, stop_on_eos=False)
    toks, lats = res[mode]
    agree = (toks.long() == ref_t)
    first = min(int((~agree[b]).nonzero()[0]) if (~agree[b]).any() else n for b in range(B))
    pen = [O.process_logits(ref_logits[i], torch.cat([torch.ones(B, 32 + Tc + 2, dtype=torch.long), torch.full((B, 1), 1024), ref_t[:, :i]], 1), 2.0, 1.0, 0, 1.0) for i in range(n)]
    margins = torch.stack([p.topk(2, -1)[0][:, 0] - p.topk(2, -1)[0][:, 1] for p in pen], 1)
    d = (lats[:, :first] - ref_l[:, :first]).abs()
    print(f"{mode}: agreement {agree.float().mean():.4f}, first divergence step {first}/{n}, oracle min margin {margins.min():.2e}, "
          f"latents max |d| {d.max():.3e} mean {d.mean():.3e} (|lat| mean {ref_l.abs().mean():.3f})")
    dd = (lats - ref_l).abs()
    print("   per step median:", " ".join(f"{float(dd[:, j].median()):.1e}" for j in range(min(n, 8))), " max:", " ".join(f"{float(dd[:, j].max()):.1e}" for j in range(min(n, 8))))
    res[mode + "_ref"] = ref_l
    if first < n:
        for b in range(B):
            if (~agree[b]).any():
                j = int((~agree[b]).nonzero()[0]); print("   stream", b, "step", j, "oracle margin there", float(margins[b, j]))
oo = (res["bf16_kv_ref"] - res["bf16_act_ref"]).abs()
print("oracle mode 2 vs oracle mode 3: per step median:", " ".join(f"{float(oo[:, j].median()):.1e}" for j in range(min(n, 8))), " mean", float(oo[:, :4].mean()), "max", float(oo[:, :4].max()))
a, bq = res["bf16_kv"][1], res["bf16_act"][1]
print("mode 2 vs mode 3 latents (first 4 steps): max", float((a[:, :4] - bq[:, :4]).abs().max()))
