"""Headline benchmark of the GenVC codec-token generation hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1 without a launcher (no WORLD_SIZE in the environment): bench.py re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1` -- one rank per GPU over RCCL; it
refuses to start when fewer than N GPUs are visible.  Launched BY torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE.

Workload (BASELINE.json configs[1]): GenVC_small streaming, 1 s chunks, top_k=1, batch 1 per GPU.
A "step" = one synthetic utterance (10 s source @16 kHz, 3 s reference @24 kHz) pushed through the hot
path exactly as inference_utils.synthesize_utt_streaming orders it:
    reference wav -> log-mel -> Perceiver -> 32 conditioning latents                     (once; on a second stream beside the
                                                                                          first chunk's ContentVec + DVAE: independent chains)
    per 1 s chunk: ContentVec (HuBERT-base, 16000 samples -> 49 frames) -> DVAE encoder + VQ -> prefix embeddings
                   -> prefill (48 rows)
                   -> 24 x (sample, KV-cached decode step) in groups of 8 tokens, each group followed by the
                      vocoder call (x4 interpolation + HiFi-GAN -> 8192 samples)
All inputs (source and reference waveforms) are resident in HBM before the clock starts; nothing of the reference's
per-utterance device work is outside the timed path.  Synthetic weights rarely emit EOS, so the token budget is
fixed: round(1 s * 23.4375) = 23 tokens + the EOS step = 24 decode steps per chunk.
N > 1: one process per GPU, utterances sharded by rank, no collective on the data path; the generated
token ids are all-gathered (RCCL) inside the timed region.  value = utterances/s of the whole job.

Extra legs, outside the K timed steps (their numbers are extra keys of the same JSON line):
  * `offline` (every rank): BASELINE configs[2] -- 64 synthetic 10 s utterances (segments 6 s + 4 s, 141 + 94 tokens, top_k=1)
    through parallel_offline.convert_offline: utterances sharded over the ranks, a FIXED micro-batch of 8 utterances per
    GPU, no collective while converting, ONE all_gather of the token ids at the end.  N = 1 also reports the fully batched
    figure (a decode step streams the weights once whatever the batch, so only the fixed-micro-batch number
    scales with GPUs).  `--no-offline` skips it.
  * `harness` (rank 0): the same utterance through inference_utils.synthesize_utt_streaming with the inputs in HOST memory --
    the reference's latency window (clock before the host->device copy, inference_utils.py:148) next to the device-only one.

  * `streams8_bf16_kv` (N = 1): BASELINE configs[3] -- 8 concurrent streams stepped together, bf16 weights + bf16 KV cache.
  * `prefill_5x110` (N = 1): the batched prefill of BASELINE configs[4] (5 segments x 110 rows) with its own roofline entry
    (`bound: mfma`, achieved TFLOP/s against the 157.3 TFLOP/s fp32-MFMA peak; FLOPs per SURVEY.md 8d).
  * `config4` (every N): BASELINE configs[4] -- 30 s source + 10 s reference, top_k=50: mel + Perceiver on 563 + 376 frames, the five
    6 s segments as ONE batch (ContentVec, DVAE, 5 x 110-row prefill, 5-stream sampled decode of 141 steps, latent re-pass, vocoder).

`--streams B` (default 1 = the headline configuration) steps B concurrent streams per GPU together (BASELINE
configs[3] shape: shared launches, one decode step for all streams; the MFMA rows path from 5 streams up); a step is
then B utterances and the JSON line says so in `config.workload`.
"""
import argparse
import json
import os
import platform
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from genvc_amd import config as gcfg          # noqa: E402
from genvc_amd import synth                   # noqa: E402

SRC_SECONDS, REF_SECONDS, CHUNK_SECONDS = 10.0, 3.0, 1.0
STEPS_PER_CHUNK = 24                          # 23 tokens + EOS step
GROUP = 8                                     # stream_chunk_size of the reference harness
KERNEL_NAMES = ["c_attn_gemv(ln1+qkv)", "attention+attn_c_proj(fused,head-split)", "attn_c_proj_gemv(unfused path only)",
                "mlp_c_fc_gemv(resid-sum+ln2+gelu)", "mlp_c_proj_gemv(resid)", "head_gemv(2xln+mel_head)"]
HBM_PEAK_GBS = 8000.0                         # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
OFFLINE_UTTS, OFFLINE_MICRO_BATCH = 64, 8     # BASELINE configs[2]; fixed per-GPU micro-batch
MFMA_F32_PEAK_TFLOPS = 157.3                  # MI355X_MICROARCH.md: dense fp32 matrix (v_mfma_f32_32x32x2_f32) peak
OFFLINE_SEGMENTS = 2                          # a 10 s utterance at seg_len 6 s = a 6 s and a 4 s segment


def kernel_bytes(dims, which, S, wb=4, kvb=4):
    """algorithmic bytes of one launch of a decode-step kernel class at B=1; wb / kvb = bytes per
    streamed weight / KV-cache element (4, or 2 with --weights bf16 / bf16_kv); vectors and biases are always fp32"""
    d, V, H = dims["d_model"], dims["num_audio_tokens"], dims["n_head"]
    f = 4
    if which == 0:
        return 3 * d * d * wb + (3 * d + 2 * d + d + d) * f + 2 * d * kvb
    if which == 1:     # fused attention + head-split attn c_proj (short-context variant): K/V rows + q + c_proj weights
        return 2 * S * d * kvb + (d + H * d) * f + d * d * wb
    if which == 2:
        return d * d * wb + (d + 8 * H * (d // H + 4) + 2 * d) * f
    if which == 3:
        return 4 * d * d * wb + (4 * d + 2 * d + d + 4 * d) * f
    if which == 4:
        return 4 * d * d * wb + (d + 4 * d + 2 * d) * f
    return V * d * wb + (V + 4 * d + d + V + d) * f


def step_bytes(dims, S, wb=4, kvb=4):
    """algorithmic bytes of one whole decode step of one stream with S cached positions: every weight once,
    the K/V rows of S positions read, one position written"""
    d, V, L = dims["d_model"], dims["num_audio_tokens"], dims["n_layer"]
    w_dec = L * (12 * d * d + 13 * d) + 4 * d + (d * V + V) + 2 * d
    return w_dec * wb + (2 * L * S * d + 2 * L * d) * kvb


class Workload:
    def __init__(self, device, rank, streams=1, weight_dtype="fp32", max_slots=8):
        from genvc_amd.inference.model_init import model_init_synthetic
        self.dev = device
        self.S = S = streams
        self.model, self.config = model_init_synthetic(gcfg.default_config(), seed=1, device=device, max_slots=max(max_slots, S),
                                                       weight_dtype=weight_dtype)
        m = self.model
        self.dims = m.gpt.dims()
        self.eng = m.gpt.engine
        self.n_chunks = int(SRC_SECONDS / CHUNK_SECONDS)
        from genvc_amd.layers.content_processor import contentvec_frames
        self.t50 = contentvec_frames(int(CHUNK_SECONDS * 16000))                       # 49
        # resident inputs: 4 distinct utterances per rank, cycled
        self.ref = [synth.synth_audio(100 + rank * 16 + u, "ref", int(REF_SECONDS * 24000)).to(device) for u in range(4)]
        # src[u]: [n_chunks, S, 16000] -- S concurrent streams, chunk-major
        self.src = [torch.stack([synth.synth_audio(200 + rank * 16 + u + 977 * j, "src", int(SRC_SECONDS * 16000))
                                 .view(self.n_chunks, -1) for j in range(S)], 1).contiguous().to(device) for u in range(4)]
        from genvc_amd.engine import sample_params
        self.sp = sample_params(dict(gcfg.DEFAULT_SAMPLING, top_k=1), self.dims["num_audio_tokens"], -1, 0)
        self.slots = torch.arange(S, device=device, dtype=torch.int32)
        n_tok = self.n_chunks * STEPS_PER_CHUNK
        self.toks = torch.zeros(S, n_tok, device=device, dtype=torch.int32)
        self.lats = torch.zeros(S, n_tok, self.dims["d_model"], device=device)
        self.Tc = m.content_dvae._engine.out_frames(self.t50)                          # 13
        self.P = 32 + self.Tc + 2
        self.ids = torch.ones(S, self.P + 1 + STEPS_PER_CHUNK + 8, device=device, dtype=torch.int32)
        self.ids_len = torch.zeros(S, device=device, dtype=torch.int32)
        self.fin = torch.zeros(S, device=device, dtype=torch.int32)
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        self.keep_codes = None                        # a list: utterance() appends the content codes of every chunk (parity_in_bench)
        self.stage_ev = None                          # a list: utterance() appends (name, event) marks around its stages (stage_times)
        self.t_first = None                           # host clock at the first 8-token group of a `sync_first` utterance

    def utterance(self, u, record=False, sync_first=False, registered=False):
        m, eng = self.model, self.eng
        cond = None
        if registered:
            # a streaming SESSION (genvc_amd/streaming.py): the target speaker is registered before the source starts to arrive -- conditioning
            # latents and their 32 rows of KV cache (gvc_gpt_prefill_cond) are there when the clock starts.  NOT the reference's window
            # (inference_utils.py:148-154 starts its clock before get_gpt_cond_latents): reported beside it, never as the headline
            cond = m.get_gpt_cond_latents(self.ref[u % 4], 24000)
            if self.S > 1:
                cond = cond.expand(self.S, -1, -1).contiguous()
            eng.prefill_cond(self.slots, cond)
        if record:
            self.ev[0].record()
        # mel + Perceiver of the reference speaker on a second stream, beside the first chunk's ContentVec + DVAE (as the harness does:
        # inference_utils.synthesize_utt_streaming; the two chains are independent)
        # (enqueued BEHIND the first chunk's ContentVec + DVAE launches, as the harness does: the side stream only waits for `ready`)
        ready = torch.cuda.Event()
        ready.record()
        cond_future = None
        src = self.src[u % 4]
        def mark(name):
            if self.stage_ev is not None:
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                self.stage_ev.append((name, e))
        mark("start")
        for c in range(self.n_chunks):
            feat = m.content_extractor.extract_content_features(src[c])                # ContentVec [S,49,256]
            codes = m.content_dvae._engine.encode(feat, frames_major=True)             # DVAE + VQ (int32 [1,13])
            mark("contentvec+dvae")
            if self.keep_codes is not None:
                self.keep_codes.append(codes.clone())
            if cond is None:
                if cond_future is None:
                    cond_future = m.get_gpt_cond_latents_async(self.ref[u % 4], 24000, after=ready)
                cond = cond_future.result()
                if self.S > 1:
                    cond = cond.expand(self.S, -1, -1).contiguous()                    # one reference speaker for the batch
            prefix = eng.prefix_embeddings(cond, codes)
            self.ids.fill_(1)
            self.ids[:, self.P] = self.dims["start_audio_token"]
            self.ids_len.fill_(self.P + 1)
            self.fin.zero_()
            # prefix caching: the 32 conditioning rows of chunk 0 stay in the KV cache for the utterance's other chunks
            eng.prefill(self.slots, prefix, want_outputs=False, n_cached=32 if (c > 0 or registered) else 0)
            mark("cond join + prefix + prefill (first chunk)" if c == 0 else "prefix + cached prefill")
            base = c * STEPS_PER_CHUNK
            tok_view = self.toks[:, base:base + STEPS_PER_CHUNK]
            lat_view = self.lats[:, base:base + STEPS_PER_CHUNK]
            for g in range(0, STEPS_PER_CHUNK, GROUP):
                eng.generate(self.slots, self.ids, self.ids_len, self.fin, self.sp, g, GROUP, tok_view, lat_view,
                             max_keys=self.P + 1 + g + GROUP)
                mark("decode steps")
                # vocoder every 8 tokens (x4 interpolation + HiFi-GAN -> 8192 samples), inference_utils.py:195-205
                self.wav = m.hifigan.forward_latents(lat_view[:, g:g + GROUP], 4)
                mark("vocoder")
                if record and c == 0 and g == 0:
                    self.ev[1].record()                                                # first 8-token group done
                if sync_first and c == 0 and g == 0:
                    torch.cuda.synchronize()
                    self.t_first = time.perf_counter()
        if record:
            self.ev[2].record()
        return self.toks


def _guarded(name, fn):
    """an auxiliary leg must never cost the headline line: its failure is recorded in its place (and on stderr)"""
    try:
        return fn()
    except Exception as e:                                   # noqa: BLE001
        import traceback
        print(f"bench.py: leg `{name}` failed: {type(e).__name__}: {e}\n{traceback.format_exc()}", file=sys.stderr)
        try:
            torch.cuda.synchronize()
        except Exception:                                    # noqa: BLE001
            pass
        return {"error": f"{type(e).__name__}: {e}"}


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return platform.processor() or platform.machine()


def stage_times(wl, u=0):
    """ms per stage of one utterance (HIP events between the stages of Workload.utterance, summed by stage)"""
    wl.stage_ev = []
    wl.utterance(u)
    torch.cuda.synchronize()
    ev, wl.stage_ev = wl.stage_ev, None
    out = {}
    for (_, e0), (name, e1) in zip(ev[:-1], ev[1:]):
        out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
    out["total"] = ev[0][1].elapsed_time(ev[-1][1])
    return out


def cpu_baseline(wl, budget_s=10.0, gpu=None):
    """The oracle (port) on the host cores, one 1 s chunk at a time (ContentVec, DVAE+VQ, prefix, prefill, 24 steps).
    gpu = (token ids [n_chunks * 24], [content codes per chunk]) of the SAME utterance (src[0], ref[0], greedy) from the timed HIP
    path: the ids the oracle generates here anyway are compared with them -> `parity_in_bench` (the record checks what it computed)."""
    from oracle import genvc_oracle as O
    m = wl.model
    w = {k[len("gpt."):]: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith("gpt.")}
    wd = {k[len("content_dvae."):]: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith("content_dvae.")}
    pre = "content_extractor.model."
    wh = {k[len(pre):]: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith(pre)}
    hcfg = m.content_extractor.cfg
    dims = wl.dims
    norms = m.torch_mel_spectrogram_style_encoder.mel_norms
    greedy = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
    # batch-1 GEMVs scale badly past a few dozen threads: give the CPU its best thread count
    cond0 = torch.zeros(1, 32, dims["d_model"])
    codes0 = torch.zeros(1, 13, dtype=torch.long)
    best = None
    sweep = []
    for nt in sorted({8, 16, 32, 64, min(os.cpu_count() or 8, 128)}):      # (all 256 hardware threads of the GPU box's host: 150 s for three steps)
        if nt > (os.cpu_count() or 8):
            continue
        if best is not None and sweep[-1]["seconds_prefill48_plus_2_steps"] > 3.0 * best[0]:
            break                                     # past the knee: more threads only get slower
        torch.set_num_threads(nt)
        t0 = time.time()
        O.generate(w, dims, cond0, codes0, greedy, max_new=3, stop_on_eos=False)
        t = time.time() - t0
        sweep.append({"threads": nt, "seconds_prefill48_plus_2_steps": round(t, 3)})
        if best is None or t < best[0]:
            best = (t, nt)
    cores = best[1]
    torch.set_num_threads(cores)
    t0 = time.time()
    cond = O.get_gpt_cond_latents(w, wl.ref[0].cpu(), norms)
    t_ref = time.time() - t0
    src = wl.src[0][:, 0].cpu()
    # prefill alone (48 rows) and the CPU path's first-chunk latency: mel + Perceiver, ContentVec, DVAE + VQ, prefill, 8 decode
    # steps, x4 interpolation + HiFi-GAN on the 8 latents -- the reference's clock window (inference_utils.py:148, 208-211) on the host cores
    wv = {k[len("hifigan."):]: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith("hifigan.")}
    t0 = time.time()
    feat0 = O.hubert_extract_features(wh, hcfg, src[0:1])
    codes_0 = O.dvae_get_codebook_indices(wd, feat0.transpose(1, 2))
    t_front = time.time() - t0
    prefix0, _ = O.compute_embeddings(w, dims, cond, codes_0)
    O.gpt_prefill(w, dims, prefix0)
    t0 = time.time()
    O.gpt_prefill(w, dims, prefix0)
    t_prefill = time.time() - t0
    t0 = time.time()
    _, lat8, _ = O.generate(w, dims, cond, codes_0, greedy, max_new=GROUP, stop_on_eos=False)
    t_gen8 = time.time() - t0
    t0 = time.time()
    O.vocode_latents(wv, m.hifigan.cfg, lat8)
    t_voc = time.time() - t0
    first_chunk_ms = (t_ref + t_front + t_gen8 + t_voc) * 1e3

    kept = {}

    def chunk(c):
        feat = O.hubert_extract_features(wh, hcfg, src[c:c + 1])
        codes = O.dvae_get_codebook_indices(wd, feat.transpose(1, 2))
        kept[c] = (codes,) + tuple(O.generate(w, dims, cond, codes, greedy, max_new=STEPS_PER_CHUNK, stop_on_eos=False))

    chunk(0)                                   # warm-up
    times = []
    t_all = time.time()
    for c in range(1, wl.n_chunks):
        t0 = time.time()
        chunk(c)
        times.append(time.time() - t0)
        if time.time() - t_all > budget_s:
            break
    t_chunk = sum(times) / len(times)
    utt_s = t_ref + wl.n_chunks * t_chunk
    parity = None
    if gpu is not None:
        # (outside the timed chunks) the oracle's ids of every chunk it generated against the HIP path's ids of the same chunk; at a
        # divergence: the oracle's own top-1 / top-2 margin at that step (penalised scores, as tests/screen_rows_seeds.py measures it)
        g_toks, g_codes = gpu
        chunks, eq_tok, eq_codes, first, min_margin = sorted(kept), True, True, None, float("inf")
        for c in chunks:
            codes, toks, _lats, logits = kept[c]
            _, ids0 = O.compute_embeddings(w, dims, cond, codes)
            got = g_toks[c * STEPS_PER_CHUNK:(c + 1) * STEPS_PER_CHUNK].long()
            same_codes = bool(torch.equal(codes[0].long(), g_codes[c][0].long().cpu()))
            eq_codes = eq_codes and same_codes
            for i in range(toks.shape[1]):
                sc = O.process_logits(logits[i], torch.cat([ids0, toks[:, :i]], 1), greedy["repetition_penalty"], 1.0, 0, 1.0)
                t2 = sc.topk(2, -1)[0]
                mg = float(t2[0, 0] - t2[0, 1])
                min_margin = min(min_margin, mg)
                if int(toks[0, i]) != int(got[i]):
                    eq_tok = False
                    if first is None:
                        first = {"chunk": c, "step": i, "oracle_margin": mg, "codes_equal": same_codes}
                    break
        parity = {"chunks": len(chunks), "tokens_compared": len(chunks) * STEPS_PER_CHUNK, "tokens_equal": eq_tok, "codes_equal": eq_codes,
                  "first_divergence": first, "oracle_min_margin": min_margin,
                  "what": "token ids of utterance 0 from the timed HIP path (ContentVec -> DVAE -> prefill -> 24 greedy steps per chunk) vs the "
                          "ids the CPU oracle generated for the same chunks in this very run; unscreened input: a divergence at an oracle "
                          "margin below 1e-3 is a near-tie flip, anything else is a bug"}
    return {"parity": parity, "value": 1.0 / utt_s, "unit": "utterances/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} of {wl.n_chunks} one-second chunks of one utterance (ContentVec, DVAE+VQ, prefill 48 rows, "
                      f"{STEPS_PER_CHUNK} decode steps each) + mel/Perceiver once; extrapolated to the full utterance",
            "ms_per_chunk": t_chunk * 1e3, "ms_per_chunk_min_max": [min(times) * 1e3, max(times) * 1e3],
            "note": "host load moves this figure by +-20 % from run to run (0.05-0.07 utterances/s observed); context, not a target",
            "ms_per_decode_token_est": t_chunk * 1e3 / (STEPS_PER_CHUNK + 2),
            "rtf": utt_s / SRC_SECONDS, "host_cpus": os.cpu_count(), "cpu": _cpu_model(),
            "prefill_48_rows_ms": t_prefill * 1e3, "first_chunk_ms": first_chunk_ms,
            "first_chunk_stages_ms": {"mel + Perceiver": t_ref * 1e3, "ContentVec + DVAE/VQ": t_front * 1e3,
                                      "prefill + 8 decode steps": t_gen8 * 1e3, "x4 interpolation + HiFi-GAN (8 tokens)": t_voc * 1e3},
            "threads_sweep": sweep, "cores_note": f"`cores` = the fastest of the swept thread counts ({cores}); batch-1 GEMVs collapse beyond a few dozen threads"}


def offline_leg(wl, rank, world, dist, device):
    """BASELINE configs[2] through parallel_offline.convert_offline (see the module docstring).  Every rank runs it."""
    from genvc_amd.parallel_offline import convert_offline
    m = wl.model
    top_k = m.config.top_k
    m.config.top_k = 1
    srcs = [synth.synth_audio(500 + i, "src", int(SRC_SECONDS * 16000)) for i in range(OFFLINE_UTTS)]
    ref = synth.synth_audio(7, "ref", int(REF_SECONDS * 24000))
    # fixed token budgets by segment duration (23.4375 tokens per second: 141 for the 6 s class, 94 for the 4 s class)
    kw = dict(seg_len=6.0, top_k=1, max_new_tokens=141, tokens_per_second=23.4375, group=int(os.environ.get("GVC_BENCH_OFFLINE_GROUP", "48")))     # decode steps between two host looks at the finished flags (library default 16: +0.5 %)

    def run(mb, r, w, rolling=False):
        convert_offline(m, srcs[:mb * w], ref, micro_batch=mb, rank=r, world=w, rolling=rolling, **kw)        # graph capture / warm-up, one wave
        torch.cuda.synchronize()
        if dist is not None and w > 1:
            dist.barrier()
        t0 = time.perf_counter()
        toks = convert_offline(m, srcs, ref, micro_batch=mb, rank=r, world=w, rolling=rolling, **kw)          # ONE all_gather at its end
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if dist is not None and w > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert toks.shape[0] == OFFLINE_UTTS
        return OFFLINE_UTTS / dt

    m.gpt.groups_stats = {"joint": 0, "separate": 0}
    rate_waves = run(OFFLINE_MICRO_BATCH, rank, world)                     # one joint decode per micro-batch, drained to its longest class
    rate = run(OFFLINE_MICRO_BATCH, rank, world, rolling=True)             # the same 16 streams per step, kept full across micro-batches
    stats = dict(m.gpt.groups_stats)
    variant = wl.eng.decode_variant()
    # the micro-batch must have been decoded JOINTLY (its 6 s and 4 s classes in one step over 16 streams) on the multi-stream
    # decode path, on every rank and at every world size: otherwise the N > 1 figure is not the N = 1 path times N
    assert stats["separate"] == 0 and stats["joint"] > 0, f"offline leg: micro-batches were not decoded jointly ({stats})"
    assert variant in (4, 5), f"offline leg: decode variant {variant}, expected the multi-stream step (4 rows path / 5 one-launch rows)"
    out = {"workload": f"{OFFLINE_UTTS} synthetic 10 s utterances (segments 6 s + 4 s: 141 + 94 tokens, fixed budget), top_k=1, tokens "
                       "only (BASELINE configs[2]); utterances sharded over the ranks, one all_gather of the token ids at the end",
           "micro_batch_utterances_per_gpu": OFFLINE_MICRO_BATCH, "n_gpus": world,
           "streams_per_decode_step": OFFLINE_SEGMENTS * OFFLINE_MICRO_BATCH, "joint_decodes": stats["joint"], "decode_variant": variant,
           "offline_utts_per_s": rate, "offline_utts_per_s_wave_by_wave": rate_waves,
           "decode": "rolling: a (micro-batch, segment) class joins the 16-stream decode step when KV slots free up and leaves it when its "
                     "token budget is spent (GPT.generate_rolling); `offline_utts_per_s_wave_by_wave` = one joint decode per micro-batch, "
                     "drained to its 6 s class (round 3's figure).  With one micro-batch per rank (N = 8) the two coincide"}
    if world == 1:
        out["offline_utts_per_s_fully_batched_1gpu"] = run(OFFLINE_UTTS, 0, 1)
        out["note"] = ("a decode step streams the weights once whatever the batch: the fully batched figure is what one GPU can do, the "
                       "fixed-micro-batch figure is the one that scales with the GPU count (SURVEY.md 8e)")
    m.config.top_k = top_k
    return out


def harness_leg(wl, reps=3):
    """the headline utterance through the reference-shaped harness, inputs in host memory (reference latency window)"""
    from genvc_amd.inference.inference_utils import synthesize_utt_streaming
    m = wl.model
    top_k = m.config.top_k
    m.config.top_k = 1
    cap = m.gpt.max_gen_mel_tokens
    m.gpt.max_gen_mel_tokens = STEPS_PER_CHUNK          # fixed token budget per 1 s chunk, as in the timed workload
    src = synth.synth_audio(200, "src", int(SRC_SECONDS * 16000))
    ref = synth.synth_audio(100, "ref", int(REF_SECONDS * 24000))
    synthesize_utt_streaming(m, src, ref, seg_len=CHUNK_SECONDS, stream_chunk_size=GROUP, verbose=False, return_details=True)
    lat, rtf = [], []
    for _ in range(reps):
        r = synthesize_utt_streaming(m, src, ref, seg_len=CHUNK_SECONDS, stream_chunk_size=GROUP, verbose=False, return_details=True)
        lat.append(r["latency"] * 1e3)
        rtf.append(r["rtf"])
    m.gpt.max_gen_mel_tokens = cap
    m.config.top_k = top_k
    return {"path": "inference_utils.synthesize_utt_streaming(model, src_wav[host], ref[host], seg_len=1.0, stream_chunk_size=8)",
            "rtf": sum(rtf) / len(rtf), "first_chunk_latency_ms": sum(lat) / len(lat), "first_chunk_latency_ms_min_max": [min(lat), max(lat)],
            "window": "host clock from before the host->device copies to the first vocoder chunk, device synchronised "
                      "(the reference reads its clock without a sync, inference_utils.py:148,208-211)", "runs": reps}


def prefill_flops(dims, B, T):
    """prefill of T rows per stream = 2 T L 12 d^2 (the four projections) + 4 L d T (T + 1) / 2 (causal attention)"""
    d, L = dims["d_model"], dims["n_layer"]
    return B * (2 * T * L * 12 * d * d + 4 * L * d * T * (T + 1) / 2)


def _timed(fn, reps, warm=2):
    """mean ms of fn() over `reps` back-to-back calls (events on the launch stream)"""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def prefill_leg(wl, B=5, Tc=75, reps=6):
    """BASELINE configs[4]'s GPT prefill: B segments x (32 + Tc + 2 + 1) rows in one batched call (reference
    layers/gpt_inference.py:81-91), fp32 MFMA.  roofline: achieved = SURVEY 8d FLOPs / mean call time."""
    dev, eng, dims = wl.dev, wl.eng, wl.dims
    cond = synth.uniform(1, "c", (B, 32, dims["d_model"]), 1.0).to(dev)
    codes = synth.integers(1, "k", (B, Tc), 256).to(dev).int()
    slots = torch.arange(B, device=dev, dtype=torch.int32)
    prefix = eng.prefix_embeddings(cond, codes)
    T = prefix.shape[1] + 1
    ms = _timed(lambda: eng.prefill(slots, prefix, want_outputs=False), reps)
    fl = prefill_flops(dims, B, T)
    tf = fl / (ms * 1e-3) / 1e12
    out = {"workload": f"batched GPT prefill, {B} segments x {T} rows (6 s segments: 32 conditioning + {Tc} + 2 text rows + start), fp32 "
                       "(BASELINE configs[4] prefill shape; GenVC_large := GenVC_small dims, no checkpoint ships)",
           "ms": ms, "gflop": fl / 1e9,
           "roofline": {"bound": "mfma", "achieved": tf, "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F32_PEAK_TFLOPS,
                        "traffic": None, "flops_per_call": fl,
                        "kernel": "k_gemm_strip (c_attn / attn c_proj / c_fc / mlp c_proj) + k_attention_tile + k_ln_sum_rows, 30 layers"}}
    # the single-stream shapes of the streaming path beside it (48 rows: first 1 s chunk; 110 rows: a 6 s segment)
    for b1, tc1 in ((1, 13), (1, 75)):
        c1, k1 = cond[:b1].contiguous(), synth.integers(2, "k", (b1, tc1), 256).to(dev).int()
        p1 = eng.prefix_embeddings(c1, k1)
        s1 = slots[:b1].contiguous()
        m1 = _timed(lambda: eng.prefill(s1, p1, want_outputs=False), reps)
        t1 = p1.shape[1] + 1
        out[f"prefill_1x{t1}_ms"] = m1
        out[f"prefill_1x{t1}_mfma_frac"] = prefill_flops(dims, b1, t1) / (m1 * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS
    # the 16 uncached rows of a later 1 s chunk (conditioning rows cached): the one-launch rows step
    c1, k1 = cond[:1].contiguous(), synth.integers(2, "k", (1, 13), 256).to(dev).int()
    p1, s1 = eng.prefix_embeddings(c1, k1), slots[:1].contiguous()
    eng.prefill(s1, p1, want_outputs=False)
    out["prefill_cached_16_rows_ms"] = _timed(lambda: eng.prefill(s1, p1, want_outputs=False, n_cached=32), reps)
    return out


def _round_bf16_weights(w):
    """the matrices a bf16-weights context rounds at bind time (include/genvc_hip.h: weight_dtype >= 1)"""
    out = dict(w)
    for k, v in w.items():
        if k.endswith(("attn.c_attn.weight", "attn.c_proj.weight", "mlp.c_fc.weight", "mlp.c_proj.weight")) or k == "mel_head.weight":
            out[k] = v.to(torch.bfloat16).to(torch.float32)
    return out


def streams_parity(wl, weights):
    """chunk 0 of the leg's utterance 0, all streams: the ids / latents the timed path produced against the CPU oracle with the same storage
    and rounding points (bf16-rounded weights, k / v rounded into the cache, and for bf16_act the four hand-off activations of the rows step),
    fed the HIP path's own conditioning latents and content codes.  bf16 modes: an agreement rate and a tolerance, not bit-exactness."""
    from oracle import genvc_oracle as O
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    m, S, n = wl.model, wl.S, STEPS_PER_CHUNK
    wl.keep_codes = []
    wl.utterance(0)
    torch.cuda.synchronize()
    codes = wl.keep_codes[0].long().cpu()
    wl.keep_codes = None
    toks, lats = wl.toks[:, :n].long().cpu(), wl.lats[:, :n].cpu()
    cond = m.get_gpt_cond_latents(wl.ref[0], 24000).cpu().expand(S, -1, -1).contiguous()
    w = {k[len("gpt."):]: v.detach().cpu() for k, v in m.state_dict().items() if k.startswith("gpt.")}
    dims = dict(wl.dims)
    if weights != "fp32":
        w = _round_bf16_weights(w)
        dims.update(kv_bf16=weights in ("bf16_kv", "bf16_act"), act_bf16=weights == "bf16_act")
    greedy = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
    t0 = time.time()
    ref_t, ref_l, ref_logits = O.generate(w, dims, cond, codes, greedy, max_new=n, stop_on_eos=False)
    _, ids0 = O.compute_embeddings(w, dims, cond, codes)
    margins = []
    for i in range(n):
        sc = O.process_logits(ref_logits[i], torch.cat([ids0, ref_t[:, :i]], 1), greedy["repetition_penalty"], 1.0, 0, 1.0)
        t2 = sc.topk(2, -1)[0]
        margins.append(t2[:, 0] - t2[:, 1])
    margins = torch.stack(margins, 1)
    agree = toks == ref_t
    first = None
    common = n
    for b in range(S):
        bad = (~agree[b]).nonzero()
        if len(bad):
            j = int(bad[0])
            common = min(common, j)
            if first is None or j < first["step"]:
                first = {"stream": b, "step": j, "oracle_margin": float(margins[b, j])}
    d = (lats[:, :max(common, 1)] - ref_l[:, :max(common, 1)]).abs()
    return {"streams": S, "tokens_compared": S * n, "agreement": float(agree.float().mean()), "first_divergence": first,
            "oracle_min_margin": float(margins.min()), "latents_abs_dev_median": float(d.median()), "latents_abs_dev_max": float(d.max()),
            "latents_mean_abs": float(ref_l.abs().mean()), "oracle_seconds": time.time() - t0,
            "what": "chunk 0 of utterance 0, every stream: ids and latents of the timed HIP path vs the CPU oracle with the same rounding points "
                    "(oracle/genvc_oracle.py: kv_bf16" + (", act_bf16" if weights == "bf16_act" else "") + "); unscreened input.  bf16 activations "
                    "make the map discontinuous: two correct implementations sit ~2e-3 (median) apart after one layer "
                    "(tests/test_gpu_round6.py measures the oracle against itself under a 2e-7 input perturbation), so the claim is the "
                    "agreement rate and a first divergence only at a small oracle margin"}


def streams_leg(device, rank, streams=8, weights="bf16_act", steps=3, parity=True):
    """BASELINE configs[3]: `streams` concurrent streams on one GPU, bf16 weights + bf16 KV cache (+ bf16 activations across the hand-offs of
    the one-launch rows step with weights = "bf16_act": csrc/persist_rows_b16.h; fp32 accumulation everywhere), every stream converted as
    synthesize_utt_streaming converts it alone; the streams share the launches (one decode step for all)."""
    wl = Workload(device, rank, streams, weights, max_slots=max(8, streams))
    wl.utterance(0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for u in range(steps):
        wl.utterance(u + 1)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    wl.utterance(0, record=True)
    torch.cuda.synchronize()
    first_ms = wl.ev[0].elapsed_time(wl.ev[1])
    # decode-step time of the batch: graph-replayed generation steps at the chunk's contexts
    n = STEPS_PER_CHUNK
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    cond = wl.model.get_gpt_cond_latents(wl.ref[0], 24000).expand(streams, -1, -1).contiguous()
    wl.eng.prefill(wl.slots, wl.eng.prefix_embeddings(cond, torch.zeros(streams, wl.Tc, device=device, dtype=torch.int32)), want_outputs=False)
    wl.ids.fill_(1)
    wl.ids[:, wl.P] = wl.dims["start_audio_token"]
    wl.ids_len.fill_(wl.P + 1)
    wl.fin.zero_()
    tv, lv = wl.toks[:, :n], wl.lats[:, :n]
    wl.eng.generate(wl.slots, wl.ids, wl.ids_len, wl.fin, wl.sp, 0, GROUP, tv, lv, max_keys=wl.P + 1 + n)
    ev[0].record()
    wl.eng.generate(wl.slots, wl.ids, wl.ids_len, wl.fin, wl.sp, GROUP, n - GROUP, tv, lv, max_keys=wl.P + 1 + n)
    ev[1].record()
    torch.cuda.synchronize()
    step_us = ev[0].elapsed_time(ev[1]) / (n - GROUP) * 1e3
    variant = wl.eng.decode_variant()
    wb, kvb = (4, 4) if weights == "fp32" else (2, 2 if weights in ("bf16_kv", "bf16_act") else 4)
    s_mid = wl.P + 1 + n // 2
    by = step_bytes(wl.dims, s_mid, wb, kvb) + (streams - 1) * (2 * wl.dims["n_layer"] * (s_mid + 1) * wl.dims["d_model"]) * kvb
    kname = {"bf16_act": "k_rows_persist_b16<8>", "bf16_kv": "k_rows_persist<8,1,1>", "bf16": "k_rows_persist<8,1,0>", "fp32": "k_rows_persist<8>"}[weights]
    traffic, traffic_source = None, None
    pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(pmc) and streams == 8:
        tj = json.load(open(pmc))
        if kname in tj:
            traffic = tj[kname]
            traffic_source = ("imported from profiles/pmc_traffic.json (a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run of this kernel, "
                              + tj.get("_source_" + kname, "see the file's notes") + "); `achieved` / `frac` are this run's timing")
    out = {"workload": f"GenVC_large (:= GenVC_small dims) streaming, 1 s chunks, top_k=1, {streams} concurrent streams on one GPU stepped "
                       f"together, weights/KV {weights}, fp32 accumulation (BASELINE configs[3])",
           "weights": weights,
           "utts_per_s": streams / dt, "rtf_per_stream": dt / SRC_SECONDS, "first_chunk_latency_ms": first_ms,
           "decode_step_us": step_us, "decode_variant": variant,
           "roofline": {"bound": "hbm", "achieved": by / (step_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": by / (step_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                        "bytes_per_launch": by,
                        "kernel": f"{kname} (one decode step of {streams} streams, sampler + head launches included in the time)"}}
    out["stages_ms_per_utterance"] = stage_times(wl)
    if parity:
        out["parity"] = _guarded("streams8 parity", lambda: streams_parity(wl, weights))
    del wl
    torch.cuda.empty_cache()
    return out


C4_SRC_SECONDS, C4_REF_SECONDS, C4_SEG_SECONDS, C4_TOP_K = 30.0, 10.0, 6.0, 50


class Config4:
    """BASELINE configs[4]: 30 s source + 10 s reference, top_k=50.  The conversion is inference_utils.synthesize_utt's (mel +
    Perceiver per 6 s reference chunk and the mean; per 6 s source segment ContentVec -> DVAE -> generate -> latent re-pass; one
    vocoder call over all latents) with the five segments -- independent given the conditioning latents,
    /root/reference/inference/inference_utils.py:43-77 -- carried as ONE batch: 5 x 110-row prefill, 5 streams per decode step.
    Fixed token budget (141 per segment); sampling never finishes a row (eos = -1)."""

    def __init__(self, wl, rank):
        self.wl, m = wl, wl.model
        dev = wl.dev
        self.ref = synth.synth_audio(900 + rank, "ref", int(C4_REF_SECONDS * 24000)).to(dev)
        n_seg = int(C4_SRC_SECONDS / C4_SEG_SECONDS)
        self.src = synth.synth_audio(901 + rank, "src", int(C4_SRC_SECONDS * 16000)).view(n_seg, -1).contiguous().to(dev)
        self.B = n_seg
        self.n_new = int(round(C4_SEG_SECONDS * 23.4375)) + 0            # 141
        from genvc_amd.engine import sample_params
        self.sp = sample_params(dict(gcfg.DEFAULT_SAMPLING, top_k=C4_TOP_K), wl.dims["num_audio_tokens"], -1, 17)
        self.slots = torch.arange(self.B, device=dev, dtype=torch.int32)
        self.ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]
        self.dev = dev
        self.repass = False

    def utterance(self, record=False):
        wl, m, eng, B, n = self.wl, self.wl.model, self.wl.eng, self.B, self.n_new
        ev = self.ev
        if record:
            ev[0].record()
        # 6 s + 4 s chunks: mel + Perceiver on 563 + 376 frames, mean -- on a second stream beside ContentVec + DVAE (independent chains);
        # the recorded pass keeps them one after the other so that the per-stage times mean what they say
        cond_future = None if record else m.get_gpt_cond_latents_async(self.ref, 24000)
        cond = m.get_gpt_cond_latents(self.ref, 24000) if record else None
        if record:
            ev[1].record()
        feat = m.content_extractor.extract_content_features(self.src)     # [5, 299, 256]
        codes = m.content_dvae._engine.encode(feat, frames_major=True)    # [5, 75]
        if record:
            ev[2].record()
        if cond is None:
            cond = cond_future.result()
        condB = cond.expand(B, -1, -1).contiguous()
        prefix = eng.prefix_embeddings(condB, codes)
        P = prefix.shape[1]
        ids = torch.ones(B, P + 1 + n + 8, device=self.dev, dtype=torch.int32)
        ids[:, P] = wl.dims["start_audio_token"]
        ids_len = torch.full((B,), P + 1, device=self.dev, dtype=torch.int32)
        fin = torch.zeros(B, device=self.dev, dtype=torch.int32)
        toks = torch.zeros(B, n, device=self.dev, dtype=torch.int32)
        lats = torch.empty(B, n, wl.dims["d_model"], device=self.dev)
        eng.prefill(self.slots, prefix, want_outputs=False)               # 5 x 110 rows
        if record:
            ev[3].record()
        for g in range(0, n, 48):
            k = min(48, n - g)
            eng.generate(self.slots, ids, ids_len, fin, self.sp, g, k, toks, lats, max_keys=P + 1 + g + k)
        if record:
            ev[4].record()
        # the latents the vocoder takes: the decode loop's own (no stop token is ever emitted here: the re-pass would recompute exactly
        # these 141 rows per segment, gpt.py:375-508, 5 x (110 + 141 + 4) rows -- `repass` runs it, as the reference does)
        lat = eng.latents(self.slots, prefix, toks) if self.repass else lats
        if record:
            ev[5].record()
        self.wav = m.hifigan.forward_latents(lat.reshape(1, B * n, -1), 4)    # one vocoder call over all latents (inference_utils.py:79-87)
        if record:
            ev[6].record()
        self.P = P
        self.last = (cond, codes, toks, lats)
        return toks


def config4_parity(c4, n_cmp=48):
    """the first n_cmp sampled steps (top_k = 50, 5 streams, 110+ cached keys: the real configs[4] decode shape) of the leg's last utterance
    against the oracle's loop (reference layers/stream_generator.py:809-881 with HF's processors' semantics) drawing from the same counter
    RNG, fed the HIP path's own conditioning latents and content codes.  A draw within float rounding of a CDF boundary may differ; from
    there on the two runs are different token sequences, so the claim is equality up to the first divergence and where it happens."""
    from oracle import genvc_oracle as O
    wl = c4.wl
    torch.set_num_threads(min(32, os.cpu_count() or 8))
    cond, codes, toks, lats = c4.last
    w = {k[len("gpt."):]: v.detach().cpu() for k, v in wl.model.state_dict().items() if k.startswith("gpt.")}
    dims = dict(wl.dims, stop_audio_token=-1)                     # the leg samples with eos = -1: no row ever finishes
    samp = dict(gcfg.DEFAULT_SAMPLING, top_k=C4_TOP_K)
    t0 = time.time()
    ref_t, ref_l, _ = O.generate(w, dims, cond.cpu().expand(c4.B, -1, -1).contiguous(), codes.long().cpu(), samp, max_new=n_cmp, seed=17,
                                 stop_on_eos=False)
    got = toks[:, :n_cmp].long().cpu()
    agree = got == ref_t
    first = None
    common = n_cmp
    for b in range(c4.B):
        bad = (~agree[b]).nonzero()
        if len(bad):
            common = min(common, int(bad[0]))
            if first is None or int(bad[0]) < first["step"]:
                first = {"stream": b, "step": int(bad[0])}
    d = (lats[:, :max(common, 1)].cpu() - ref_l[:, :max(common, 1)]).abs()
    return {"streams": c4.B, "steps_compared": n_cmp, "tokens_equal_until_step": common, "first_divergence": first,
            "agreement": float(agree.float().mean()), "latents_abs_dev_max_before_divergence": float(d.max()), "oracle_seconds": time.time() - t0,
            "what": "sampled ids (top_k = 50) of the timed configs[4] decode vs the CPU oracle's loop on the same counter RNG; equality up to the "
                    "first divergence (a draw at a CDF boundary), unscreened input"}


def config4_leg(wl, rank, world, dist, device, steps=3, parity=True):
    c4 = Config4(wl, rank)
    c4.utterance()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        toks = c4.utterance()
    if dist is not None:                                    # the batched offline path's collective: token ids of every rank's utterance
        parts = [torch.empty_like(toks) for _ in range(world)]
        dist.all_gather(parts, toks)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    c4.utterance(record=True)
    torch.cuda.synchronize()
    st = [c4.ev[i].elapsed_time(c4.ev[i + 1]) for i in range(6)]
    # the reference's second forward pass (inference_utils.py:71-76), which the default path no longer runs: its cost and what it changes
    c4.repass = True
    c4.utterance(record=True)
    torch.cuda.synchronize()
    repass_ms = c4.ev[4].elapsed_time(c4.ev[5])
    c4.repass = False
    T = c4.P + 1
    fl_prefill = prefill_flops(wl.dims, c4.B, T)
    d = wl.dims["d_model"]
    # Perceiver MACs per SURVEY 8d, F frames: 4 layers x [(32+F) d 1024 + 32 d 512 + 2*8*32 (32+F) 64 + 32 512 d + 32 d 5460 + 32 2730 d] + F 80 d
    def perc(F):
        return 2.0 * (4 * ((32 + F) * d * 1024 + 32 * d * 512 + 2 * 8 * 32 * (32 + F) * 64 + 32 * 512 * d + 32 * d * 5460 + 32 * 2730 * d) + F * 80 * d)
    fl_perc = perc(563) + perc(376)
    return {"workload": f"GenVC_large (:= GenVC_small dims) non-streaming, {C4_SRC_SECONDS:.0f} s source + {C4_REF_SECONDS:.0f} s reference, "
                        f"top_k={C4_TOP_K} (BASELINE configs[4]): Perceiver on 563 + 376 mel frames; five 6 s segments as one batch "
                        f"(5 x {T}-row prefill, 5-stream sampled decode x {c4.n_new} steps, one vocoder call over the decode loop's latents -- the reference's latent "
                        f"re-pass recomputes the same vectors, SURVEY.md 8a row 12, and is kept behind repass_latents=True); one utterance per GPU at a time",
            "n_gpus": world, "utts_per_s": steps * world / dt, "rtf": dt / steps / C4_SRC_SECONDS, "ms_per_utterance": dt / steps * 1e3,
            "stages_ms": {"mel+perceiver(563+376 frames)": st[0], "contentvec+dvae(5 x 6 s)": st[1], f"prefix+prefill(5x{T})": st[2],
                          f"decode({c4.n_new} steps x 5 streams, top_k={C4_TOP_K})": st[3], "vocoder": st[5]},
            "latent_repass_ms_when_enabled": repass_ms,
            "prefill_mfma_frac": fl_prefill / (st[2] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "perceiver_mfma_frac_incl_mel": fl_perc / (st[0] * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
            "decode_step_us": st[3] / c4.n_new * 1e3, "decode_variant": wl.eng.decode_variant(),
            "parity": _guarded("config4 parity", lambda: config4_parity(c4)) if parity and rank == 0 else None}


def cold_probe(device, rank, weights, max_slots):
    """First-chunk latency of the FIRST utterance after model_init (host clock from the call to the first vocoder chunk, device
    synchronised; inputs resident): once on a fresh model as it comes out of model_init, once on a fresh model after
    GenVCModel.warmup() (include/genvc_hip.h: gvc_gpt_warmup).  `gpt_lazy_inits_*`: allocations / device syncs / graph captures the
    GPT context did inside data-path calls of that first utterance (0 after the warm-up: SURVEY.md 8(b)(iii)).  Returns the numbers
    and the warmed-up workload, which the rest of the bench goes on to use."""
    out = {}
    t0 = time.perf_counter()
    wl = Workload(device, rank, 1, weights, max_slots=8)
    torch.cuda.synchronize()
    out["model_init_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    wl.utterance(0, sync_first=True)
    out["cold_first_chunk_latency_ms_no_warmup"] = (wl.t_first - t0) * 1e3
    torch.cuda.synchronize()
    out["gpt_lazy_inits_no_warmup"] = wl.eng.lazy_inits()
    del wl
    torch.cuda.empty_cache()
    wl = Workload(device, rank, 1, weights, max_slots=max_slots)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    wl.model.warmup(seg_len=CHUNK_SECONDS, streams=1, ref_seconds=REF_SECONDS, stream_chunk_size=GROUP, top_k=1, max_new_tokens=STEPS_PER_CHUNK)
    out["warmup_call_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    wl.utterance(0, sync_first=True)
    out["cold_first_chunk_latency_ms"] = (wl.t_first - t0) * 1e3
    torch.cuda.synchronize()
    out["gpt_lazy_inits_after_warmup"] = wl.eng.lazy_inits()
    out["window"] = ("first utterance after model_init: host clock from the call (inputs resident in HBM) to the first 8-token vocoder "
                     "chunk, device synchronised; `cold_first_chunk_latency_ms` = after GenVCModel.warmup(), `..._no_warmup` = without")
    return out, wl


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher: become `torch.distributed.run` with one rank per GPU"""
    import socket
    n_dev = torch.cuda.device_count()
    if n_dev < args.gpus and not os.environ.get("GVC_BENCH_SAME_DEVICE"):
        sys.exit(f"bench.py --gpus {args.gpus}: only {n_dev} GPU(s) visible on this node (set GVC_BENCH_SAME_DEVICE=1 with "
                 "GVC_BENCH_BACKEND=gloo for a dry run of the multi-rank path on one GPU)")
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // args.gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-offline", action="store_true", help="skip the batched-offline leg (BASELINE configs[2])")
    ap.add_argument("--no-harness", action="store_true", help="skip the harness-level leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the configs[3] / configs[4] / prefill legs")
    ap.add_argument("--no-cold", action="store_true", help="skip the cold-start probe (first utterance after model_init, with / without warm-up)")
    ap.add_argument("--streams", type=int, default=1, help="concurrent streams per GPU (1 = headline configuration)")
    ap.add_argument("--weights", default="fp32", choices=["fp32", "bf16", "bf16_kv", "bf16_act"],
                    help="GPT weight / KV-cache storage (fp32 = headline configuration; bf16_kv with --streams 8 = BASELINE configs[3])")
    args = ap.parse_args()
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        self_launch(args)                         # does not return
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}: the JSON line must describe the job that ran"
    from genvc_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):       # a checkout without the in-tree build: compile it here (hipcc, gfx950)
        if rank == 0:
            from genvc_amd.build import build as build_lib
            build_lib(verbose=False)
        else:
            while not os.path.exists(_lib.LIB_PATH):
                time.sleep(1.0)
    # dry-run hooks for a box with fewer GPUs than ranks (tests of the multi-rank code path only): every rank on GPU 0,
    # gloo instead of RCCL (RCCL refuses two ranks on one device)
    if os.environ.get("GVC_BENCH_SAME_DEVICE"):
        local = 0
    torch.cuda.set_device(local)
    device = f"cuda:{local}"
    dist = None
    if world > 1:
        import torch.distributed as dist
        backend = os.environ.get("GVC_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(device))
        else:
            dist.init_process_group(backend)

    headline = args.streams == 1
    do_offline = headline and not args.no_offline and args.weights == "fp32"
    # KV slots: a micro-batch of OFFLINE_MICRO_BATCH utterances is OFFLINE_SEGMENTS segment-streams per utterance decoded JOINTLY
    # (layers/gpt.py generate_groups), so every rank needs that many slots whatever the world size -- the per-rank path at N > 1
    # must be the one N = 1 is measured on; N = 1 also runs the fully batched figure (one class of OFFLINE_UTTS streams at a time)
    slots_needed = max(16, OFFLINE_SEGMENTS * OFFLINE_MICRO_BATCH, OFFLINE_UTTS if (do_offline and world == 1) else 0)
    cold = None
    if world == 1 and headline and not args.no_cold:
        cold, wl = cold_probe(device, rank, args.weights, slots_needed)
    else:
        wl = Workload(device, rank, args.streams, args.weights, max_slots=slots_needed)
    for u in range(args.warmup):
        wl.utterance(u)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    gathered = None
    for u in range(args.steps):
        toks = wl.utterance(u)
        if dist is not None:                          # collect this wave's token ids on every rank
            gathered = [torch.empty_like(toks) for _ in range(world)]
            dist.all_gather(gathered, toks)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    offline = offline_leg(wl, rank, world, dist, device) if do_offline else None
    do_extra = headline and args.weights == "fp32" and not args.no_extra
    config4 = config4_leg(wl, rank, world, dist, device, parity=not args.no_cpu_baseline) if do_extra else None
    rccl_ranks = None
    if dist is not None:
        # ranks that really took part in a collective on the job's backend (nccl = RCCL): every rank contributes a one
        ones = [torch.zeros(1, device=device, dtype=torch.int32) for _ in range(world)]
        dist.all_gather(ones, torch.ones(1, device=device, dtype=torch.int32))
        rccl_ranks = int(torch.cat(ones).sum().item())
        assert rccl_ranks == world == args.gpus, f"{rccl_ranks} ranks answered the collective, --gpus {args.gpus}"
    # which physical device every rank ran on (an N > 1 line must show N distinct GPUs unless the dry-run hook put them on one)
    props = torch.cuda.get_device_properties(torch.cuda.current_device())
    my_uuid = str(getattr(props, "uuid", "")) or f"{props.name}#{local}"
    device_uuids = [my_uuid]
    if dist is not None:
        device_uuids = [None] * world
        dist.all_gather_object(device_uuids, my_uuid)

    if rank == 0:
        # latency / per-stage numbers from one recorded utterance (device events on the launch stream)
        wl.utterance(0, record=True)
        torch.cuda.synchronize()
        first_ms = wl.ev[0].elapsed_time(wl.ev[1])
        utt_ms = wl.ev[0].elapsed_time(wl.ev[2])
        wl.utterance(0, record=True, registered=True)
        torch.cuda.synchronize()
        first_registered_ms = wl.ev[0].elapsed_time(wl.ev[1])
        # ---- roofline of the dominant kernel, measured live with HIP events on the launch stream (gvc_gpt_time_kernel) ----
        S = wl.P + 1 + STEPS_PER_CHUNK // 2
        tok = torch.zeros(1, device=device, dtype=torch.int32)
        s1 = wl.slots[:1].contiguous()

        def fresh():
            wl.eng.prefill(s1, wl.eng.prefix_embeddings(wl.model.get_gpt_cond_latents(wl.ref[0], 24000),
                                                        torch.zeros(1, wl.Tc, device=device, dtype=torch.int32)), want_outputs=False)
        reps = 40
        wb = 4 if args.weights == "fp32" else 2
        kvb = 2 if args.weights in ("bf16_kv", "bf16_act") else 4
        fresh()
        whole_us, _ = wl.eng.time_kernel(6, s1, tok, reps)          # launch-per-phase step (the fallback path; the bf16 contexts' path)
        # which step the B = 1 generation loop really replays is the engine's to say (3 = the one-launch step): a device with fewer
        # than 256 CUs, a failed LDS opt-in, GVC_PERSIST=0 or bf16 storage all leave it on the launch-per-phase step
        probe_ids = torch.ones(1, wl.P + 1 + 16, device=device, dtype=torch.int32)
        probe_ids[:, wl.P] = wl.dims["start_audio_token"]
        fresh()
        wl.eng.generate(s1, probe_ids, torch.full((1,), wl.P + 1, device=device, dtype=torch.int32), torch.zeros(1, device=device, dtype=torch.int32),
                        wl.sp, 0, 2, torch.zeros(1, 2, device=device, dtype=torch.int32), torch.zeros(1, 2, wl.dims["d_model"], device=device),
                        max_keys=wl.P + 4)
        torch.cuda.synchronize()
        one_launch = wl.eng.decode_variant() == 3
        kern = []
        if one_launch:
            # the one-stream decode step is ONE launch (csrc/persist_kernel.h) = 80 % of the utterance: it IS the dominant kernel.
            # `reps` launches replayed from a graph between two events; the context grows from wl.P + 1 by one position per launch
            fresh()
            step_us, n = wl.eng.time_kernel(7, s1, tok, 24)
            s_mid = wl.P + 1 + 12
            kern.append({"kernel": f"k_decode_persist<{wl.dims['d_model'] // 256}> (whole decode step of one stream, one launch; instantiation "
                                   f"<{wl.dims['d_model'] // 256}, {int(wb == 2)}, {int(kvb == 2)}, XL>: the MLP's hidden units stay inside their XCD at d_model 1024)", "avg_us": step_us,
                         "launches_per_step": 1, "bytes": step_bytes(wl.dims, s_mid, wb, kvb)})
            dom = 0
        else:
            # whole step and the step without each class, replayed back to back: in-situ cost of a class = (whole - without) /
            # launches.  (Single-kernel event pairs are useless at 5 us; a class launched alone re-reads its own stale inputs
            # from L2 and looks 5-12 % faster than rocprofv3 sees it in the step.)
            for which in range(6):
                per_step = 1 if which == 5 else wl.dims["n_layer"]
                fresh()
                iso, n = wl.eng.time_kernel(which, s1, tok, 128 if which == 5 else 32)
                if n == 0:
                    continue
                fresh()
                without_us, _ = wl.eng.time_kernel(16 + which, s1, tok, reps)
                kern.append({"kernel": KERNEL_NAMES[which], "avg_us": (whole_us - without_us) / per_step, "avg_us_launched_alone": iso,
                             "launches_per_step": per_step, "bytes": kernel_bytes(wl.dims, which, S, wb, kvb)})
            # dominant = the weight-streaming GEMV with the largest share of the step
            cand = [i for i, k in enumerate(kern) if "gemv" in k["kernel"] and "head" not in k["kernel"]]
            dom = max(cand, key=lambda i: kern[i]["avg_us"] * kern[i]["launches_per_step"])
        achieved = kern[dom]["bytes"] / (kern[dom]["avg_us"] * 1e-6) / 1e9
        traffic, traffic_source = None, None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc) and args.weights == "fp32":      # the PMC passes were taken on the fp32 build
            tj = json.load(open(pmc))
            key = kern[dom]["kernel"].split(" ")[0]
            if key in tj:
                traffic = tj[key]
                traffic_source = ("imported from profiles/pmc_traffic.json (a separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE run of this kernel, "
                                  + tj.get("_source_" + key, "see the file's notes") + "); `achieved` / `frac` are this run's timing")
        n_utts = args.steps * world * args.streams
        ms_step = dt / args.steps * 1e3
        out = {
            "metric": "utterances/s (streaming, 1 s chunks; with RTF and first-chunk latency)",
            "value": n_utts / dt, "unit": "utterances/s", "n_gpus": world,
            "collective_backend": None if dist is None else dist.get_backend(), "rccl_ranks": rccl_ranks, "device_uuids": device_uuids,
            "distinct_devices": len(set(device_uuids)),
            "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.weights == "fp32" else "bf16 storage, f32 arithmetic", "data": "synthetic", "weights": args.weights,
            "rtf": (dt / args.steps) / SRC_SECONDS, "first_chunk_latency_ms": first_ms, "streams_per_gpu": args.streams,
            "first_chunk_latency_window": "device events, inputs resident in HBM (the harness-level window with the host->device "
                                          "copies inside is `harness.first_chunk_latency_ms`)",
            "ms_per_utterance_device": utt_ms,
            "first_chunk_latency_ms_registered_speaker": first_registered_ms,
            "first_chunk_latency_ms_registered_speaker_note": "a streaming session whose target speaker was registered before the source arrives "
                                                              "(conditioning latents + gvc_gpt_prefill_cond outside the window: StreamSessions.open); the reference's window, "
                                                              "`first_chunk_latency_ms`, starts before the conditioning latents (inference_utils.py:148-154)",
            "config": {"workload": ("GenVC_small streaming, 1 s chunks, top_k=1, batch 1 per GPU (BASELINE configs[1])" if args.streams == 1
                                    else f"GenVC_small streaming, 1 s chunks, top_k=1, {args.streams} concurrent streams per GPU stepped together "
                                         f"(BASELINE configs[3] shape, weights/KV: {args.weights}); one step = that many utterances"),
                       "arch": f"L=30 d=1024 H=4 V=1026 {args.weights}, synthetic weights (train_genVC.py dims; no checkpoint ships)",
                       "utterance": "10 s source @16 kHz (10 chunks x 16000 samples -> 49 ContentVec frames -> 13 codes), 3 s reference @24 kHz",
                       "per_chunk": f"ContentVec (HuBERT-base) + DVAE/VQ + prefill {wl.P + 1} rows (chunks after the first: {wl.P + 1 - 32} rows, the 32 conditioning rows stay cached) + {STEPS_PER_CHUNK} decode steps; HiFi-GAN vocoder every {GROUP} tokens",
                       "excluded_from_timed_path": [],
                       "parallelism": f"replicas x{world}, utterances sharded by rank, all_gather of token ids"},
            "roofline": {"bound": "hbm", "kernel": kern[dom]["kernel"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "bytes_per_launch": kern[dom]["bytes"], "avg_us": kern[dom]["avg_us"],
                         "decode_step_us": kern[dom]["avg_us"] if one_launch else whole_us,
                         "decode_step_us_launch_per_phase": whole_us},
            "kernels": kern,
        }
        out["stages_ms_per_utterance"] = _guarded("stage_times", lambda: stage_times(wl))
        if offline is not None:
            out["offline"] = offline
            out["offline_utts_per_s"] = offline["offline_utts_per_s"]
        if config4 is not None:
            out["config4"] = config4
        if do_extra and world == 1:
            out["prefill_5x110"] = _guarded("prefill_5x110", lambda: prefill_leg(wl))
        if headline and not args.no_harness:
            out["harness"] = _guarded("harness", lambda: harness_leg(wl))
        if cold is not None:
            out["cold_start"] = cold
            out["cold_first_chunk_latency_ms"] = cold["cold_first_chunk_latency_ms"]
        if world == 1 and not args.no_cpu_baseline:
            gpu = None
            if headline:                        # the ids of utterance 0 (src[0], ref[0]) from the timed path, for parity_in_bench
                wl.keep_codes = []
                g_toks = wl.utterance(0)[0].clone()
                torch.cuda.synchronize()
                gpu = (g_toks.cpu(), wl.keep_codes)
                wl.keep_codes = None
            cb = _guarded("cpu_baseline", lambda: cpu_baseline(wl, gpu=gpu))
            out["parity_in_bench"] = cb.pop("parity", None)
            out["cpu_baseline"] = cb
        if do_extra and world == 1:
            # BASELINE configs[3]: bf16 weights + KV cache + bf16 activations across the rows step's hand-offs (weight_dtype 3), with its parity
            # block; beside it the same leg with fp32 activations (weight_dtype 2: what this key measured up to round 5)
            def both():
                leg = streams_leg(device, rank, weights="bf16_act", parity=not args.no_cpu_baseline)
                ref2 = streams_leg(device, rank, weights="bf16_kv", parity=False)
                leg["fp32_activations_bf16_kv"] = {k: ref2[k] for k in ("utts_per_s", "decode_step_us", "first_chunk_latency_ms", "decode_variant")}
                leg["fp32_activations_bf16_kv"]["roofline_frac"] = ref2["roofline"]["frac"]
                return leg
            out["streams8_bf16_kv"] = _guarded("streams8_bf16_kv", both)
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
