"""Phase timeline of the one-launch rows step (GVC_PERSIST_STAMPS=1): in-kernel wall-clock stamps (wave 0, lane 0).
    python scripts/stamps_rows.py B [Tc] [fp32|bf16|bf16_kv]
stamp k of (layer, phase): 0 input gathered, 1 output published, 2 LayerNorm statistics merged (B: own gathers done), 3 weight fills
waited for, 4 MFMA loop (B: score loop) done, 5 partials in LDS, 6 barrier passed, 7 final values ready (csrc/persist_rows.h)"""
import os, sys, ctypes as C
os.environ["GVC_PERSIST_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvc_amd import config as gcfg, synth, _lib
from genvc_amd.engine import GptEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
Tc = int(sys.argv[2]) if len(sys.argv) > 2 else 13
wd = sys.argv[3] if len(sys.argv) > 3 else "fp32"
dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
w = synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda")
eng = GptEngine(dims, max_slots=max(8, B), max_rows=8192, weight_dtype=wd); eng.bind(w)
dev = "cuda"
cond = synth.uniform(1, "c", (B, 32, 1024), 1.0).to(dev)
codes = synth.integers(1, "k", (B, Tc), 256).to(dev).int()
slots = torch.arange(B, device=dev, dtype=torch.int32)
eng.prefill(slots, eng.prefix_embeddings(cond, codes), want_outputs=False)
tok = torch.zeros(B, device=dev, dtype=torch.int32); lg = torch.empty(B, 1026, device=dev); lt = torch.empty(B, 1024, device=dev)
L = _lib.lib(); L.gvc_gpt_debug_stamps.restype = C.c_int; L.gvc_gpt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
for _ in range(4): eng.decode_step(slots, tok, lg, lt)
nl = dims["n_layer"]
hb = np.zeros(40 * (nl + 2) + 40 * 256, dtype=np.uint64)
n = L.gvc_gpt_debug_stamps(eng._h, hb.ctypes.data_as(C.c_void_p), -2)
assert n > 0, "no stamps: did the call run on the one-launch rows step?"
t0 = int(hb[0])
us = lambda v: (int(v) - t0) / 100.0 if v else np.nan
names = ["A qkv", "B attn", "C proj", "D fc", "E mlp"]
W = lambda l, p, k: us(hb[(l * 5 + p) * 8 + k])
for l in list(range(3)) + [nl - 1]:
    print(f"layer {l}: " + "  ".join(f"{names[p]} in {W(l,p,0):7.2f} out {W(l,p,1):7.2f}" for p in range(5)))
d = np.array([[W(l, p, k) for p in range(5) for k in range(2)] for l in range(1, nl)])
per = np.diff(np.concatenate([d[:-1, -1:], d[1:, :]], axis=1), axis=1).mean(axis=0)
print(f"B={B} keys~{32 + Tc + 3} {wd}: workgroup 0, mean us per stage (layers 2..): " +
      "  ".join(f"{names[i//2]}{' wait' if i%2==0 else ' work'} {per[i]:.2f}" for i in range(10)))
print(f"mean per layer {np.diff(d[:, -1]).mean():.2f} us;  waits {per[0::2].sum():.2f}  work {per[1::2].sum():.2f}")
# attribution of the work term, workgroup 0, mean over layers 2..: consecutive stamps in program order
order = {0: [0, 3, 4, 5, 6, 7, 1], 1: [0, 4, 5, 6, 7, 1], 2: [0, 3, 4, 5, 6, 7, 1], 3: [0, 3, 4, 5, 6, 7, 1], 4: [0, 3, 4, 5, 6, 7, 1]}
label = {0: "gathered", 1: "published", 2: "LN merged", 3: "fills ok", 4: "MFMA done", 5: "partials in LDS", 6: "barrier", 7: "final ready"}
labelB = {2: "own gathers", 0: "barrier (q ready)", 4: "scores+fold", 5: "partials in LDS", 6: "barrier", 7: "merged", 1: "published"}
tot = {}
for p in range(5):
    ks = [k for k in order[p] if not np.isnan(W(2, p, k))]          # (the bf16-activation kernel stamps 0, 3, 4, 6, 1 only)
    seg = []
    for a, b in zip(ks[:-1], ks[1:]):
        v = np.nanmean([W(l, p, b) - W(l, p, a) for l in range(2, nl)])
        lab = (labelB if p == 1 else label)[b]
        seg.append(f"->{lab} {v:.2f}")
        tot[lab] = tot.get(lab, 0.0) + v
    print(f"  {names[p]:7s} " + "  ".join(seg))
print("  sum over the five phases by kind: " + "  ".join(f"{k} {v:.2f}" for k, v in tot.items()))
# every workgroup at layer 2
b2 = 40 * (nl + 2)
a2 = np.array([[us(hb[b2 + (w * 5 + p) * 8 + k]) for p in range(5) for k in range(8)] for w in range(256)])
ref = np.nanmin(a2[:, 0])
for p in range(5):
    for k in order[p]:
        col = a2[:, p * 8 + k]
        col = col[~np.isnan(col)]
        if len(col):
            lab = (labelB if p == 1 else label)[k]
            print(f"layer 2 {names[p]:7s} {lab:18s}: min {col.min()-ref:6.2f} median {np.median(col)-ref:6.2f} p90 {np.percentile(col, 90)-ref:6.2f} max {col.max()-ref:6.2f}  (n={len(col)})")
