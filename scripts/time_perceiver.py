"""mel + Perceiver of the reference speaker (reference trainers/hifigan_trainer.py:438-455 for one chunk): time per call.
    python scripts/time_perceiver.py [ref_seconds=3] [reps=50]
F = 1 + T // 256 mel frames of a `ref_seconds` reference at 24 kHz (3 s: 282 frames; 6 s: 563)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from genvc_amd import config as gcfg, synth
from genvc_amd.inference.model_init import model_init_synthetic

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 3.0
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
m, cfg = model_init_synthetic(gcfg.default_config(), seed=1, device="cuda", max_slots=2)
ref = synth.synth_audio(100, "ref", int(secs * 24000)).cuda()
mel_mod = m.torch_mel_spectrogram_style_encoder
for _ in range(3):
    out = m.get_gpt_cond_latents(ref, 24000)
torch.cuda.synchronize()
e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tm = tp = 0.0
for _ in range(reps):
    e[0].record()
    mel, fm = mel_mod(ref, frames_major=True)
    e[1].record()
    emb = m.gpt.conditioning_perceiver(fm)
    e[2].record()
    torch.cuda.synchronize()
    tm += e[0].elapsed_time(e[1]); tp += e[1].elapsed_time(e[2])
e[0].record()
for _ in range(reps):
    out = m.get_gpt_cond_latents(ref, 24000)
e[1].record()
torch.cuda.synchronize()
F = mel.shape[-1]
d = 1024
macs = 4 * ((32 + F) * d * 1024 + 32 * d * 512 + 2 * 8 * 32 * (32 + F) * 64 + 32 * 512 * d + 32 * d * 5460 + 32 * 2730 * d) + F * 80 * d
print(f"ref {secs:g} s -> F = {F} frames: mel {tm / reps * 1e3:.1f} us, Perceiver {tp / reps * 1e3:.1f} us (events around each call), "
      f"get_gpt_cond_latents {e[0].elapsed_time(e[1]) / reps * 1e3:.1f} us per call back to back; Perceiver {2 * macs / 1e9:.2f} GFLOP (SURVEY 8d) -> "
      f"{2 * macs / (tp / reps * 1e-3) / 1e12:.1f} TFLOP/s = {2 * macs / (tp / reps * 1e-3) / 1e12 / 157.3:.3f} of the fp32 MFMA peak")
