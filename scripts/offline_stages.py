"""Where a micro-batch of the batched offline path (BASELINE configs[2]: 8 utterances = 16 segment-streams) spends its time:
wall clock per stage with a device synchronisation after each (so the figures add up to a little more than the pipelined run)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.inference.model_init import model_init_synthetic
from genvc_amd.inference.inference_utils import segments, _sampling_kwargs
from genvc_amd.parallel_offline import convert_batch

m, cfg = model_init_synthetic(gcfg.default_config(), seed=1, device="cuda", max_slots=16)
m.config.top_k = 1
srcs = [synth.synth_audio(500 + i, "src", 160000) for i in range(8)]
ref = synth.synth_audio(7, "ref", 72000)
cond = m.get_gpt_cond_latents(ref.to(m.device), 24000)
kw = dict(seg_len=6.0, top_k=1, max_new_tokens=141, tokens_per_second=23.4375)
convert_batch(m, srcs, cond, **kw)
torch.cuda.synchronize()
t0 = time.perf_counter(); convert_batch(m, srcs, cond, **kw); torch.cuda.synchronize(); whole = time.perf_counter() - t0

def T(f):
    torch.cuda.synchronize(); t = time.perf_counter(); r = f(); torch.cuda.synchronize(); return r, (time.perf_counter() - t) * 1e3

seg = int(6.0 * 16000)
per = [list(segments(w.to(m.device), seg, 5120)) for w in srcs]
out = {}
groups = []
for s in range(2):
    wav = torch.cat([p[s] for p in per], 0)
    feat, out[f"contentvec_{s}"] = T(lambda: m.content_extractor.extract_content_features(wav))
    codes, out[f"dvae_{s}"] = T(lambda: m.content_dvae.get_codebook_indices(feat.transpose(1, 2)))
    groups.append((cond.expand(8, -1, -1).contiguous(), codes))
g = m.gpt
k2 = dict(_sampling_kwargs(m)); k2.update(top_k=1, max_new_tokens=[141, 94])
eng = g.engine
prefixes = []
for i, (c, t) in enumerate(groups):
    p, out[f"prefix_{i}"] = T(lambda: eng.prefix_embeddings(c.float().contiguous(), t.int().contiguous()))
    prefixes.append(p)
slots = torch.arange(16, device="cuda", dtype=torch.int32)
for i, p in enumerate(prefixes):
    _, out[f"prefill_{i}_{p.shape[1] + 1}rows"] = T(lambda: eng.prefill(slots[8 * i:8 * i + 8].contiguous(), p, want_outputs=False))
_, out["generate_groups_total"] = T(lambda: g.generate_groups(groups, **k2))
print(f"convert_batch of 8 utterances: {whole * 1e3:.1f} ms  ->  {8 / whole:.1f} utterances/s")
for k, v in out.items():
    print(f"  {k:28s} {v:8.2f} ms")
print(f"  decode inside generate_groups ~ {out['generate_groups_total'] - sum(v for k, v in out.items() if k.startswith(('prefix_', 'prefill_'))):.1f} ms for 94 steps x 16 streams + 47 steps x 8 streams")
