"""End-to-end harness timing with the CLI's defaults (seg_len 6 s, top_k 15) on synthetic weights."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.inference.inference_utils import synthesize_utt, synthesize_utt_streaming
from genvc_amd.inference.model_init import model_init_synthetic

m, cfg = model_init_synthetic(gcfg.default_config(), seed=1, device="cuda")
m.config.top_k = int(os.environ.get("TOPK", "15"))
m.gpt.max_gen_mel_tokens = 150            # synthetic weights rarely stop: cap like a 6 s segment (141 tokens)
src = synth.synth_audio(1, "src", 160000)
ref = synth.synth_audio(2, "ref", 72000)
for name, fn in (("non-streaming seg 6 s", lambda: synthesize_utt(m, src, ref, seg_len=6.0, return_details=True)),
                 ("streaming seg 6 s", lambda: synthesize_utt_streaming(m, src, ref, seg_len=6.0, verbose=False, return_details=True)),
                 ("streaming seg 1 s", lambda: synthesize_utt_streaming(m, src, ref, seg_len=1.0, verbose=False, return_details=True))):
    fn()
    torch.cuda.synchronize()
    t0 = time.time()
    out = fn()
    torch.cuda.synchronize()
    dt = time.time() - t0
    n = sum(t.shape[-1] for t in (out["codes"] if "codes" in out else out["tokens"]))
    print(f"{name}: {dt * 1e3:.1f} ms for a 10 s utterance, {n} tokens, {dt * 1e6 / n:.0f} us per token all-in, RTF {dt / 10:.4f}"
          + (f", first chunk {out['latency'] * 1e3:.1f} ms" if "latency" in out else ""))
