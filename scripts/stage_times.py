"""Per-stage device time of one bench utterance (HIP events on the launch stream, averaged over repeats)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    torch.cuda.set_device(0)
    wl = bench.Workload("cuda:0", 0)
    m, eng = wl.model, wl.eng
    for u in range(2):
        wl.utterance(u)
    torch.cuda.synchronize()
    acc = {}

    def timed(name, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn()
        e1.record()
        acc.setdefault(name, []).append((e0, e1))
        return out

    reps = 3
    for r in range(reps):
        cond = timed("mel+perceiver (once)", lambda: m.get_gpt_cond_latents(wl.ref[r % 4], 24000))
        src = wl.src[r % 4]
        for c in range(wl.n_chunks):
            feat = timed("contentvec", lambda: m.content_extractor.extract_content_features(src[c]))
            codes = timed("dvae+vq", lambda: m.content_dvae._engine.encode(feat, frames_major=True))
            prefix = timed("prefix_emb", lambda: eng.prefix_embeddings(cond, codes))

            def glue():
                wl.ids.fill_(1)
                wl.ids[:, wl.P] = wl.dims["start_audio_token"]
                wl.ids_len.fill_(wl.P + 1)
                wl.fin.zero_()
            timed("torch glue", glue)
            timed("prefill", lambda: eng.prefill(wl.slots, prefix, want_outputs=False))
            base = c * bench.STEPS_PER_CHUNK
            tv = wl.toks[:, base:base + bench.STEPS_PER_CHUNK]
            lv = wl.lats[:, base:base + bench.STEPS_PER_CHUNK]
            for g in range(0, bench.STEPS_PER_CHUNK, bench.GROUP):
                timed("decode x8", lambda: eng.generate(wl.slots, wl.ids, wl.ids_len, wl.fin, wl.sp, g, bench.GROUP, tv, lv))
                timed("vocoder", lambda: m.hifigan.forward_latents(lv[:, g:g + bench.GROUP], 4))
    torch.cuda.synchronize()
    tot = 0.0
    for k, evs in acc.items():
        ms = sum(a.elapsed_time(b) for a, b in evs) / reps
        tot += ms
        print(f"{k:24s} {len(evs) // reps:4d} calls/utt  {ms:8.3f} ms/utt  {ms / (len(evs) // reps) * 1e3:9.1f} us/call")
    print(f"sum of stages {tot:.2f} ms per utterance")


if __name__ == "__main__":
    main()
