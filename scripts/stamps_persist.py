"""Phase timeline of the one-launch decode step (GVC_PERSIST_STAMPS=1): workgroup 0's wall-clock stamps."""
import os, sys, ctypes as C
os.environ["GVC_PERSIST_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from genvc_amd import config as gcfg, synth, _lib
from genvc_amd.engine import GptEngine

Tc = int(sys.argv[1]) if len(sys.argv) > 1 else 13
dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
w = synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda")
eng = GptEngine(dims, max_slots=8, max_rows=4096, weight_dtype=os.environ.get("GVC_WD", "fp32")); eng.bind(w)
dev = "cuda"
cond = synth.uniform(1, "c", (1, 32, 1024), 1.0).to(dev)
codes = synth.integers(1, "k", (1, Tc), 256).to(dev).int()
slots = torch.arange(1, device=dev, dtype=torch.int32)
eng.prefill(slots, eng.prefix_embeddings(cond, codes), want_outputs=False)
tok = torch.zeros(1, device=dev, dtype=torch.int32); lg = torch.empty(1, 1026, device=dev); lt = torch.empty(1, 1024, device=dev)
L = _lib.lib(); L.gvc_gpt_debug_stamps.restype = C.c_int; L.gvc_gpt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
for _ in range(4): eng.decode_step(slots, tok, lg, lt)
nl = dims["n_layer"]
hb = np.zeros(20 * (nl + 2) + 10 * 256 + 8 * 256, dtype=np.uint64)
n = L.gvc_gpt_debug_stamps(eng._h, hb.ctypes.data_as(C.c_void_p), -1)
t0 = int(hb[20 * (nl + 1)])
us = lambda v: (int(v) - t0) / 100.0
names = ["A qkv", "B attn", "C proj", "D fc", "E mlp"]
W = lambda l, p, k: us(hb[(l * 5 + p) * 4 + k])
for l in list(range(3)) + [nl - 1]:
    print(f"layer {l}: " + "  ".join(f"{names[p]} in {W(l,p,0):7.2f} out {W(l,p,1):7.2f}" for p in range(5)))
print(f"head in {W(nl,0,0):.2f} out {W(nl,0,1):.2f}")
d = np.array([[W(l, p, k) for p in range(5) for k in range(2)] for l in range(1, nl)])
per = np.diff(np.concatenate([d[:-1, -1:], d[1:, :]], axis=1), axis=1).mean(axis=0)
print("workgroup 0, mean us per stage (layers 2..): " + "  ".join(f"{names[i//2]}{' wait' if i%2==0 else ' work'} {per[i]:.2f}" for i in range(10)))
print(f"mean per layer {np.diff(d[:, -1]).mean():.2f} us")
ws = np.diff(np.concatenate([d[:-1, -1:], d[1:, :]], axis=1), axis=1)[:, 0::2]       # workgroup 0's five waits, layers 2..
print("workgroup 0, waits per layer (A B C D E):")
for i in range(0, ws.shape[0], 4):
    print("   " + "   ".join(f"l{i+2+j:2d}: " + " ".join(f"{v:4.1f}" for v in ws[i + j]) for j in range(min(4, ws.shape[0] - i))))
LS = 2          # the layer every workgroup stamps (csrc/persist_kernel.h: kPStampLayer)
ex = np.array([[W(l, 0, 2) - W(l, 0, 0), W(l, 0, 3) - W(l, 0, 2), W(l, 0, 1) - W(l, 0, 3),
                W(l, 1, 2) - W(l, 1, 0), W(l, 1, 3) - W(l, 1, 2), W(l, 1, 1) - W(l, 1, 3)] for l in range(2, nl)]).mean(axis=0)
print("A: LN %.2f rows %.2f publish %.2f | B: scores+fold %.2f combine barrier %.2f merge+publish %.2f" % tuple(ex))
# every workgroup at layer 2
b2 = 20 * (nl + 2)
a2 = np.array([[us(hb[b2 + (w * 5 + p) * 2 + k]) for p in range(5) for k in range(2)] for w in range(256)])
ref = a2[:, 0].min()
lab = [f"{names[p]} {'in' if k == 0 else 'out'}" for p in range(5) for k in range(2)]
for j in range(10):
    col = a2[:, j]
    col = col[col > 0] if j in (2, 3) else col
    print(f"layer {LS} {lab[j]:11s}: min {col.min()-ref:6.2f} median {np.median(col)-ref:6.2f} max {col.max()-ref:6.2f}  (n={len(col)})")

# phase D's gather of wave 0 in every workgroup at layer 2 (four partial planes of x' in the fused short-context variant), and per-XCD view
b3 = b2 + 10 * 256
gd = hb[b3:b3 + 8 * 256].reshape(256, 8)
if gd[:, 0].any():
    t_in, t_sent, t_done = [np.array([us(v) for v in gd[:, k]]) - ref for k in range(3)]
    c_out = a2[:, 5] - ref
    print(f"layer {LS} phase D gather (wave 0 of each workgroup): enters {np.median(t_in):.2f} (median), sentinel seen {np.median(t_sent):.2f}, done {np.median(t_done):.2f}; "
          f"last C publish {c_out.max():.2f}; sentinel polls median {np.median(gd[:, 3]):.0f} max {gd[:, 3].max()}, sweep passes median {np.median(gd[:, 4]):.0f} max {gd[:, 4].max()}")
    xcd = (gd[:, 5] >> 8).astype(int)
    if (gd[:, 5] != 0xffff).all():
        print(f"per XCD (layer {LS}, us after the first workgroup's phase-A input): workgroups | A out | BC in | C out (median / max) | D sentinel | D gathered | D out | E out")
        for x in range(8):
            m = xcd == x
            print(f"  XCD {x}: {m.sum():3d} | {np.median(a2[m,1])-ref:5.2f} | {np.median(a2[m,2])-ref:5.2f} | {np.median(a2[m,5])-ref:5.2f} / {a2[m,5].max()-ref:5.2f} | "
                  f"{np.median(t_sent[m]):5.2f} | {np.median(t_done[m]):5.2f} | {np.median(a2[m,7])-ref:5.2f} | {np.median(a2[m,9])-ref:5.2f}")
