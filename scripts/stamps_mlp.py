import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["GVC_DEBUG_STAMPS"] = "1"
import numpy as np, torch
from genvc_amd import config as gcfg, synth, _lib
from genvc_amd.engine import GptEngine, sample_params
dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
w = synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda")
eng = GptEngine(dims, max_slots=8, max_rows=1024); eng.bind(w)
dev = "cuda"
cond = synth.uniform(1, "c", (1, 32, 1024), 1.0).to(dev); codes = synth.integers(1, "k", (1, 13), 256).to(dev).int()
slots = torch.zeros(1, device=dev, dtype=torch.int32)
prefix = eng.prefix_embeddings(cond, codes); eng.prefill(slots, prefix, want_outputs=False)
L = _lib.lib(); L.gvc_gpt_debug_stamps.restype = C.c_int
L.gvc_gpt_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
ids = torch.ones(1, 120, device=dev, dtype=torch.int32); ids_len = torch.full((1,), 48, device=dev, dtype=torch.int32)
fin = torch.zeros(1, device=dev, dtype=torch.int32); toks = torch.zeros(1, 16, device=dev, dtype=torch.int32); lats = torch.zeros(1, 16, 1024, device=dev)
sp = sample_params(dict(gcfg.DEFAULT_SAMPLING, top_k=1), 1026, -1)
eng.generate(slots, ids, ids_len, fin, sp, 0, 4, toks, lats)      # capture: stamp slots are baked into the graph
eng.generate(slots, ids, ids_len, fin, sp, 4, 4, toks, lats)
torch.cuda.synchronize()
hb = np.zeros((4096, 8), np.uint64)
n = L.gvc_gpt_debug_stamps(eng._h, hb.ctypes.data_as(C.c_void_p), 4096)
print("stamped launches", n)
t = hb[:n].astype(np.int64)
# launches per layer in the graph: qkv gemv, (attn_proj: unstamped), mlp fused  -> stamped: qkv, mlp alternate
for i in range(2, min(n, 14)):
    r = t[i]
    nxt = t[i + 1, 0] if i + 1 < n else 0
    print(f"{i:3d} wg0: entry 0  +{(r[1]-r[0])/100:5.2f}  +{(r[2]-r[0])/100:5.2f}  +{(r[3]-r[0])/100:5.2f} | last wg: entry +{(r[4]-r[0])/100:5.2f}  +{(r[5]-r[0])/100:5.2f}  +{(r[6]-r[0])/100:5.2f}  +{(r[7]-r[0])/100:5.2f} | next stamped entry +{(nxt-r[0])/100:5.2f}")
