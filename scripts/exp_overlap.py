"""Experiment: does a ContentVec forward on a second stream overlap with the one-launch decode steps of the main stream?
(the next chunk's front end is independent of the current chunk's decode steps)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.engine import GptEngine, HubertEngine, sample_params

dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
eng = GptEngine(dims, max_slots=8, max_rows=4096)
eng.bind(synth.make_weights(1, synth.gpt_weight_spec(dims), device="cuda"))
hc = gcfg.DEFAULT_HUBERT
hub = HubertEngine(hc, max_batch=2, max_samples=16000)
hub.bind(synth.make_weights(17, synth.hubert_weight_spec(hc), device="cuda"))
dev = "cuda"
wav = synth.synth_audio(3, "w", 16000).cuda()
cond = synth.uniform(1, "c", (1, 32, 1024), 1.0).to(dev)
codes = synth.integers(1, "k", (1, 13), 256).to(dev).int()
slots = torch.arange(1, device=dev, dtype=torch.int32)
prefix = eng.prefix_embeddings(cond, codes)
P = prefix.shape[1]
sp = sample_params(dict(gcfg.DEFAULT_SAMPLING, top_k=1), 1026, 1025)
n = 24
side = torch.cuda.Stream()

def decode():
    ids = torch.ones(1, P + 1 + n + 8, device=dev, dtype=torch.int32); ids[:, P] = 1024
    ids_len = torch.full((1,), P + 1, device=dev, dtype=torch.int32)
    fin = torch.zeros(1, device=dev, dtype=torch.int32)
    toks = torch.zeros(1, n, device=dev, dtype=torch.int32)
    lats = torch.zeros(1, n, 1024, device=dev)
    eng.prefill(slots, prefix, want_outputs=False)
    for g in range(0, n, 8):
        eng.generate(slots, ids, ids_len, fin, sp, g, 8, toks, lats, max_keys=P + 1 + g + 8)
    return toks

for _ in range(2):
    decode(); hub.forward(wav)
torch.cuda.synchronize()

def timed(fn, reps=5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

def seq():
    hub.forward(wav); decode()

def par():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        hub.forward(wav)
    decode()
    torch.cuda.current_stream().wait_stream(side)

ref = decode().clone(); torch.cuda.synchronize()
print(f"decode of one chunk alone       : {timed(decode):.3f} ms")
print(f"ContentVec alone                : {timed(lambda: hub.forward(wav)):.3f} ms")
print(f"ContentVec then decode (1 stream): {timed(seq):.3f} ms")
print(f"ContentVec || decode (2 streams) : {timed(par):.3f} ms")
par(); torch.cuda.synchronize()
print("tokens equal after an overlapped run:", bool(torch.equal(decode(), ref)))
eng.health()
