import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GVC_DEBUG_STAMPS"] = "1"
import numpy as np, torch
from genvc_amd import config as gcfg, synth, _lib
from genvc_amd.engine import GptEngine
dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
w = synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda")
eng = GptEngine(dims, max_slots=8, max_rows=1024); eng.bind(w)
dev = "cuda"
cond = synth.uniform(1, "c", (1, 32, 1024), 1.0).to(dev); codes = synth.integers(1, "k", (1, 13), 256).to(dev).int()
slots = torch.zeros(1, device=dev, dtype=torch.int32)
prefix = eng.prefix_embeddings(cond, codes); eng.prefill(slots, prefix, want_outputs=False)
tok = torch.zeros(1, device=dev, dtype=torch.int32)
lg = torch.empty(1, 1026, device=dev); lt = torch.empty(1, 1024, device=dev)
L = _lib.lib(); L.gvc_gpt_debug_stamps.restype = C.c_int
buf = np.zeros((4096, 8), np.uint64)
for _ in range(3): eng.decode_step(slots, tok, lg, lt)
L.gvc_gpt_debug_stamps(eng._h, buf.ctypes.data_as(C.c_void_p), 4096)
# capture one step inside a torch CUDA graph-free eager run is host-bound; use the library graph instead
ids = torch.ones(1, 200, device=dev, dtype=torch.int32); ids_len = torch.full((1,), 48, device=dev, dtype=torch.int32)
fin = torch.zeros(1, device=dev, dtype=torch.int32); toks = torch.zeros(1, 16, device=dev, dtype=torch.int32); lats = torch.zeros(1, 16, 1024, device=dev)
from genvc_amd.engine import sample_params
sp = sample_params(dict(gcfg.DEFAULT_SAMPLING, top_k=1), 1026, -1)
eng.generate(slots, ids, ids_len, fin, sp, 0, 4, toks, lats)      # graph capture happens here (stamps pointers baked)
n = L.gvc_gpt_debug_stamps(eng._h, buf.ctypes.data_as(C.c_void_p), 4096)
eng.generate(slots, ids, ids_len, fin, sp, 4, 2, toks, lats)      # replays overwrite the same stamp slots
torch.cuda.synchronize()
import ctypes
hb = np.zeros((4096, 8), np.uint64)
# read the raw buffer again (dbg_n was reset; copy first 160 launches manually)
L.gvc_gpt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
# hack: bump by running one eager step so dbg_n > 0, then read everything
eng.decode_step(slots, tok, lg, lt)
n = L.gvc_gpt_debug_stamps(eng._h, hb.ctypes.data_as(C.c_void_p), 4096)
print("launches", n)
names = ["qkv", "proj", "fc", "p2"]
t = hb[:n].astype(np.int64)
t0 = t[4, 0]
for i in range(4, 4 + 12):
    r = t[i]
    k = names[i % 4]
    nxt = t[i + 1, 0] if i + 1 < n else 0
    print(f"{i:3d} {k:5s} first-wg: entry {(r[0]-t0)/100:8.2f}us  prologue +{(r[1]-r[0])/100:5.2f}  barrier +{(r[2]-r[0])/100:5.2f}  reduced +{(r[3]-r[0])/100:5.2f} | last-wg entry +{(r[4]-r[0])/100:5.2f} prologue +{(r[5]-r[0])/100:5.2f} barrier +{(r[6]-r[0])/100:5.2f} reduced +{(r[7]-r[0])/100:5.2f} | next entry +{(nxt-r[0])/100:5.2f}")
