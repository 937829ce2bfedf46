"""GPU: round-5 additions -- the warm-up entry point (no allocation / device sync in data-path calls afterwards), per-row retirement in
the rolling decode, the sampler's distribution, time-out recovery of the one-shot harnesses, and the parity RATE on unscreened
full-size inputs (every other full-size test runs on margin-screened seeds)."""
import numpy as np
import pytest
import torch

from genvc_amd import config as gcfg
from genvc_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
GREEDY = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
GREEDY_KW = dict(do_sample=True, top_k=1, top_p=0.85, temperature=0.85, repetition_penalty=2.0, num_beams=1, length_penalty=1.0)
WIDE2 = dict(gcfg.DEFAULT_MODEL_ARGS, gpt_layers=2)          # d = 1024, H = 4, L = 2: the width the one-launch steps serve


def _engine(margs, seed, max_slots, weight_dtype="fp32"):
    from genvc_amd.engine import GptEngine
    torch.cuda.empty_cache()
    dims = gcfg.gpt_dims(margs)
    w = synth.make_weights(seed, synth.gpt_weight_spec(dims), device=DEV)
    eng = GptEngine(dims, max_slots=max_slots, max_rows=4096, weight_dtype=weight_dtype)
    eng.bind(w)
    return dims, w, eng


def test_warmup_leaves_no_lazy_work_to_the_data_path():
    """"no hidden sync or allocation after create".  After gvc_gpt_warmup(B, max_keys, top_k) for the shapes a
    streaming deployment uses, prefill / cached chunk prefill / generate / decode_step do no hipMalloc, no hipDeviceSynchronize and no
    graph capture (gvc_gpt_lazy_inits stays at 0); without the warm-up the same calls do (the counter is live); and the warmed-up
    context generates the oracle's ids (reference loop: /root/reference/layers/stream_generator.py:809-881)."""
    from test_gpu_gpt import run_generate
    from oracle import genvc_oracle as O
    dims, w, eng = _engine(WIDE2, 3, 8)
    B, Tc, n = 1, 13, 24
    cond = synth.uniform(300, "cond_latents", (2, 32, 1024), 1.0)
    codes = synth.integers(300, "content_codes", (2, Tc), 256)
    P1 = 32 + Tc + 3
    assert eng.lazy_inits() == 0
    eng.warmup(1, P1 + 8, 1)
    eng.warmup(1, P1 + n, 1)
    eng.warmup(2, P1 + n, 1)
    eng.warmup(2, P1 + n, 15)
    assert eng.lazy_inits() == 0, "the warm-up's own work must not count"
    _, toks, lats = run_generate(eng, dims, cond[:1], codes[:1], n)                 # one stream: the one-launch step
    assert eng.decode_variant() == 3
    _, toks2, _ = run_generate(eng, dims, cond, codes, n)                           # two streams: the one-launch rows step
    assert eng.decode_variant() == 5
    # the <= 16 uncached rows of a later chunk (cached chunk prefill on the rows step) and a sampled run
    slots = torch.zeros(1, device=DEV, dtype=torch.int32)
    eng.prefill(slots, eng.prefix_embeddings(cond[:1].to(DEV), codes[:1].to(DEV).int()), want_outputs=False, n_cached=32)
    run_generate(eng, dims, cond, codes, 16, sampling=dict(gcfg.DEFAULT_SAMPLING, top_k=15), seed=5)
    torch.cuda.synchronize()
    eng.health()
    assert eng.lazy_inits() == 0, f"{eng.lazy_inits()} allocations / device syncs / graph captures inside warmed-up data-path calls"
    wc = {k: v.cpu() for k, v in w.items()}
    ref_t, ref_l, _ = O.generate(wc, dims, cond, codes, GREEDY, max_new=n, stop_on_eos=False)
    assert torch.equal(toks.long(), ref_t[:1]) and torch.equal(toks2.long(), ref_t)
    np.testing.assert_allclose(lats.numpy(), ref_l[:1].numpy(), atol=1e-4)
    eng.close()
    # a context that was NOT warmed up does its first-use work inside the calls -- and says so
    dims, w, eng = _engine(WIDE2, 3, 8)
    run_generate(eng, dims, cond[:1], codes[:1], 8)
    assert eng.lazy_inits() >= 2                                                    # persist_prepare + at least one graph capture
    eng.close()


def _gpt_module(margs, seed, max_slots, stop_bias=None):
    from genvc_amd.layers.gpt import GPT
    a = margs
    dims = gcfg.gpt_dims(a)
    g = GPT(layers=a["gpt_layers"], model_dim=a["gpt_n_model_channels"], heads=a["gpt_n_heads"],
            max_text_tokens=a["gpt_max_text_tokens"], max_mel_tokens=a["gpt_max_audio_tokens"],
            max_prompt_tokens=a["gpt_max_prompt_tokens"], number_text_tokens=a["gpt_number_text_tokens"],
            start_text_token=a["gpt_start_text_token"], stop_text_token=a["gpt_stop_text_token"],
            num_audio_tokens=a["gpt_num_audio_tokens"], start_audio_token=a["gpt_start_audio_token"],
            stop_audio_token=a["gpt_stop_audio_token"], code_stride_len=a["gpt_code_stride_len"])
    w = synth.make_weights(seed, synth.gpt_weight_spec(dims))
    if stop_bias is not None:
        w["mel_head.bias"][dims["stop_audio_token"]] = stop_bias
    missing, unexpected = g.load_state_dict(w, strict=False)
    assert not unexpected
    g = g.to(DEV)
    g.init_gpt_for_inference(max_slots=max_slots, max_rows=4096)
    return g, dims, w


ROLL_SEED, ROLL_BIAS = 16, 4.0          # screened on the CPU (oracle): smallest live top-1 / top-2 margin 5.8e-3, 13 distinct stop steps in 0..20
ROLL_JOBS = [(5, 9), (3, 13), (8, 9), (2, 20), (6, 13), (4, 9), (7, 5), (3, 9)]      # (rows, content codes) per job
ROLL_BUDGET = 40


def rolling_inputs():
    jobs = []
    for i, (b, tc) in enumerate(ROLL_JOBS):
        jobs.append((synth.uniform(ROLL_SEED + i, "cond_latents", (b, 32, 1024), 1.0), synth.integers(ROLL_SEED + i, "content_codes", (b, tc), 256)))
    return jobs


def rolling_oracle(jobs, w, dims):
    """per job: the reference loop (oracle) with EOS, and the smallest top-1 / top-2 margin over its live decisions"""
    from oracle import genvc_oracle as O
    outs, margin = [], float("inf")
    for cond, codes in jobs:
        t, _, logits = O.generate(w, dims, cond, codes, GREEDY, max_new=ROLL_BUDGET, stop_on_eos=True)
        _, ids0 = O.compute_embeddings(w, dims, cond, codes)
        alive = torch.ones(t.shape[0], dtype=torch.bool)
        for i in range(t.shape[1]):
            sc = O.process_logits(logits[i], torch.cat([ids0, t[:, :i]], 1), 2.0, 1.0, 0, 1.0)
            t2 = sc.topk(2, -1)[0]
            if alive.any():
                margin = min(margin, float((t2[:, 0] - t2[:, 1])[alive].min()))
            alive &= t[:, i] != dims["stop_audio_token"]
        outs.append(t)
    return outs, margin


def test_rolling_decode_retires_rows_at_ragged_eos_and_keeps_the_step_full():
    """VERDICT round 4, item 4 (/root/reference/layers/stream_generator.py:861-874: `unfinished_sequences` per row): eight jobs of
    2..8 streams with different prefix lengths on a stop-biased model (the gpt_eos trick: the stop logit's bias raised until greedy
    decoding ends within a few dozen steps, at a different step per stream) through GPT.generate_rolling over 16 KV slots with
    repetition_penalty 2.  A stream that has stopped frees its slot at the next host look and the next job's rows take it.  Every
    job must come out exactly as the oracle's loop gives it alone (ids equal, finished rows padded with the stop token up to the
    job's last EOS), and at least 90 % of the row-steps the decode calls ran must have been live rows."""
    g, dims, w = _gpt_module(WIDE2, ROLL_SEED, 16, ROLL_BIAS)
    jobs = rolling_inputs()
    ref, margin = rolling_oracle(jobs, w, dims)
    ends = sorted({int((t[b] == 1025).nonzero()[0]) if (t[b] == 1025).any() else ROLL_BUDGET for t in ref for b in range(t.shape[0])})
    assert len(ends) >= 6, f"the streams should stop at ragged steps: {ends}"
    assert margin >= 2e-3, f"input seed no longer margin-screened ({margin:.2e})"
    gj = [(c.to(DEV), t.to(DEV)) for c, t in jobs]
    g.rolling_stats = {}
    out = g.generate_rolling(gj, group=2, max_new_tokens=ROLL_BUDGET, **GREEDY_KW)
    assert g.engine.decode_variant() in (3, 5)        # (the one-launch steps: rows step, or the one-stream step when a single row is left)
    for i, (o, r) in enumerate(zip(out, ref)):
        assert o.shape == r.shape and torch.equal(o.cpu(), r), f"job {i}: ids differ from the oracle's loop"
    st = g.rolling_stats
    occ = st["row_steps_live"] / st["row_steps_issued"]
    print(f"rolling decode: {st['calls']} calls, {st['row_steps_issued']} row-steps issued, {st['row_steps_live']} live -> occupancy {occ:.3f}; "
          f"streams stop at steps {ends}; oracle min margin {margin:.2e}")
    assert occ >= 0.90
    # the same jobs, job-wise retirement emulated by one generate() per job: same ids (what round 4 shipped)
    for i, (c, t) in enumerate(gj[:3]):
        solo = g.generate(c, t, max_new_tokens=ROLL_BUDGET, **GREEDY_KW)
        assert torch.equal(solo.cpu(), ref[i])
    g.engine.close()


@pytest.mark.parametrize("k,p", [(15, 0.85), (50, 0.85), (1026, 1.0)])
def test_sampler_draws_follow_softmax_of_the_processed_scores(k, p):
    """the reference draws with torch.multinomial(softmax(scores)) (stream_generator.py:856-857); the kernel
    replaces it by a counter RNG + inverse CDF.  20 480 draws of gvc_sample per (top_k, top_p) on fixed logits -- the CLI default
    (15, 0.85), configs[4]'s (50, 0.85), and no truncation at all -- against softmax of the HF-processed scores (the oracle's
    processors, pinned to HuggingFace's by tests/golden/sampler.npz): chi-square over the tokens with an expected count >= 5 (the
    rest pooled), p-value > 1e-4; no draw outside the surviving set."""
    from scipy import stats
    from genvc_amd.engine import sample_params
    from oracle import genvc_oracle as O
    dims, w, eng = _engine(gcfg.TINY_MODEL_ARGS, 3, 1)
    V, R, calls = 1026, 64, 320
    logits1 = synth.uniform(77, "logits", (1, V), 3.0)
    ids0 = synth.integers(78, "ids", (1, 40), 1024)
    scores = O.process_logits(logits1, ids0.long(), 2.0, 0.85, k, p)
    probs = torch.softmax(scores.double(), -1)[0].numpy()
    logits = logits1.expand(R, V).contiguous().to(DEV)
    ids = torch.zeros(R, 64, dtype=torch.int32, device=DEV)
    ids[:, :40] = ids0.to(DEV).int()
    fin = torch.zeros(R, dtype=torch.int32, device=DEV)
    sp = sample_params(dict(gcfg.DEFAULT_SAMPLING, top_k=k, top_p=p), V, 1025, seed=4242)
    counts = np.zeros(V, dtype=np.int64)
    for step in range(calls):
        ids_len = torch.full((R,), 40, dtype=torch.int32, device=DEV)
        tok = eng.sample(logits, ids, ids_len, fin, sp, step)
        counts += np.bincount(tok.cpu().numpy(), minlength=V)
    n = R * calls
    assert counts.sum() == n
    assert counts[probs == 0].sum() == 0, "a draw outside the top-k / top-p set"
    exp = probs * n
    big = exp >= 5
    obs_b = np.append(counts[big], counts[~big].sum())
    exp_b = np.append(exp[big], exp[~big].sum())
    keep = exp_b > 0
    chi2 = float(((obs_b[keep] - exp_b[keep]) ** 2 / exp_b[keep]).sum())
    dof = int(keep.sum()) - 1
    pv = float(stats.chi2.sf(chi2, dof)) if dof > 0 else 1.0
    print(f"top_k={k} top_p={p}: {int((probs > 0).sum())} surviving tokens, {n} draws, chi2 {chi2:.1f} / dof {dof}, p-value {pv:.3g}")
    assert pv > 1e-4
    eng.close()


def test_unscreened_full_size_inputs_parity_rate():
    """VERDICT round 4 ("bit-exact holds on margin-screened seeds only ... no figure for unscreened inputs"): 32 RANDOM full-size inputs
    (L = 30, d = 1024, the first 1 s chunk's shape: 48-row prefill + 24 greedy steps), no screening.  The one-launch decode step (each
    input alone) and the one-launch rows step (the same inputs, 16 streams per call) against the oracle's loop
    (/root/reference/layers/stream_generator.py:809-881): the RATE of inputs whose 24 ids are all equal is printed, and every first
    divergence must sit at a step where the oracle's own top-1 / top-2 margin is below 1e-3 (a near-tie the fp32 summation order
    decides), never at a clear decision."""
    from test_gpu_gpt import run_generate
    from oracle import genvc_oracle as O
    dims, w, eng = _engine(gcfg.DEFAULT_MODEL_ARGS, 1, 16)
    N, Tc, n = 32, 13, 24
    cond = synth.uniform(9001, "cond_latents", (N, 32, 1024), 1.0)
    codes = synth.integers(9001, "content_codes", (N, Tc), 256)
    wc = {k: v.cpu() for k, v in w.items()}
    ref_t, _, logits = O.generate(wc, dims, cond, codes, GREEDY, max_new=n, stop_on_eos=False)
    _, ids0 = O.compute_embeddings(wc, dims, cond, codes)
    margins = torch.empty(N, n)
    for i in range(n):
        sc = O.process_logits(logits[i], torch.cat([ids0, ref_t[:, :i]], 1), 2.0, 1.0, 0, 1.0)
        t2 = sc.topk(2, -1)[0]
        margins[:, i] = t2[:, 0] - t2[:, 1]
    got = {}
    one = []
    for b in range(N):
        _, t, _ = run_generate(eng, dims, cond[b:b + 1], codes[b:b + 1], n)
        one.append(t)
    assert eng.decode_variant() == 3
    got["one-launch step (1 stream)"] = torch.cat(one, 0).long()
    rows = []
    for b in range(0, N, 16):
        _, t, _ = run_generate(eng, dims, cond[b:b + 16], codes[b:b + 16], n)
        rows.append(t)
    assert eng.decode_variant() == 5
    got["one-launch rows step (16 streams)"] = torch.cat(rows, 0).long()
    worst = 0.0
    for name, t in got.items():
        eq = (t == ref_t)
        ok = int(eq.all(1).sum())
        div = []
        for b in range(N):
            if not eq[b].all():
                i = int((~eq[b]).nonzero()[0])
                div.append((b, i, float(margins[b, i])))
        worst = max([worst] + [m for _, _, m in div])
        print(f"unscreened full-size inputs, {name}: {ok} / {N} inputs reproduce all {n} oracle ids ({N * n} decisions, oracle margins: min "
              f"{float(margins.min()):.2e}, {int((margins < 1e-3).sum())} below 1e-3); first divergences (input, step, oracle margin): {div}")
        for b, i, m in div:
            assert m < 1e-3, f"{name}: input {b} diverges at step {i} where the oracle's margin is {m:.2e} -- not a near-tie"
    eng.close()


def _wide_tiny_model(seed=3):
    from genvc_amd.inference.model_init import model_init_synthetic
    torch.cuda.empty_cache()
    cfg = gcfg.default_config(tiny=True)
    cfg.model_args.gpt_n_model_channels = 1024            # the width the one-launch steps serve (d = 1024, 4 heads), two layers
    cfg.vocoder_config.input_feat_dim = 1024
    m = model_init_synthetic(cfg, seed=seed, device=DEV, max_slots=16)[0]
    m.config.top_k = 1
    m.gpt.max_gen_mel_tokens = 24
    return m


def test_one_shot_harnesses_recover_from_a_hand_off_timeout(monkeypatch):
    """synthesize_utt / synthesize_utt_streaming / convert_offline used to RAISE when a hand-off of a
    one-launch step timed out (another context held CUs; simulated by launching the grid one workgroup short).  Now the work that
    produced garbage is repeated on the launch-per-phase paths the library has switched to: the call returns the tokens (and
    waveform) of a clean run, or fails loudly -- never the garbage (/root/reference/inference/inference_utils.py:135-217)."""
    from genvc_amd._lib import GenvcHipError
    from genvc_amd.inference.inference_utils import synthesize_utt, synthesize_utt_streaming
    from genvc_amd.parallel_offline import convert_offline
    src = synth.synth_audio(80, "src", 32000)
    ref = synth.synth_audio(60, "ref", 72000)
    srcs = [synth.synth_audio(90 + i, "src", 32000 if i % 2 else 16000) for i in range(5)]
    clean = _wide_tiny_model()
    want_s = synthesize_utt_streaming(clean, src, ref, seg_len=1.0, stream_chunk_size=8, verbose=False, return_details=True)
    want_n = synthesize_utt(clean, src, ref, seg_len=1.0, return_details=True)
    want_o = convert_offline(clean, srcs, ref, seg_len=1.0, micro_batch=4, max_new_tokens=24)
    want_r = convert_offline(clean, srcs, ref, seg_len=1.0, micro_batch=4, max_new_tokens=24, rolling=True)
    assert torch.equal(want_o, want_r)
    del clean
    # 1. streaming: the first decode call of a fresh model is launched short -> time-out -> the segment is generated again
    monkeypatch.setenv("GVC_PERSIST_TEST_GRID", "255")
    m = _wide_tiny_model()
    got = synthesize_utt_streaming(m, src, ref, seg_len=1.0, stream_chunk_size=8, verbose=False, return_details=True)
    assert m.gpt.recoveries == 1
    assert torch.equal(torch.cat(got["tokens"], 1), torch.cat(want_s["tokens"], 1))
    np.testing.assert_allclose(got["wav"].cpu().numpy(), want_s["wav"].cpu().numpy(), atol=2e-4)
    del m
    # 2. non-streaming
    m = _wide_tiny_model()
    got = synthesize_utt(m, src, ref, seg_len=1.0, return_details=True)
    assert m.gpt.recoveries == 1
    assert all(torch.equal(a, b) for a, b in zip(got["codes"], want_n["codes"]))
    np.testing.assert_allclose(got["wav"].cpu().numpy(), want_n["wav"].cpu().numpy(), atol=2e-4)
    del m
    # 3. the offline driver, wave by wave and rolling (the joint decode on the rows step times out)
    for rolling in (False, True):
        m = _wide_tiny_model()
        got = convert_offline(m, srcs, ref, seg_len=1.0, micro_batch=4, max_new_tokens=24, rolling=rolling)
        assert m.gpt.recoveries == 1
        assert torch.equal(got, want_o), f"convert_offline(rolling={rolling}) after a recovery differs from the clean run"
        del m
    monkeypatch.delenv("GVC_PERSIST_TEST_GRID")
    # 4. a time-out in the MIDDLE of a segment, after groups of it were already handed to the vocoder (injected: the health check of the
    #    second decode call reports one): the segment is re-generated from its start, the groups already emitted are skipped -- same
    #    tokens, nothing vocoded twice; a second time-out in the same segment propagates
    m = _wide_tiny_model()
    real = m.gpt.engine.health
    state = {"n": 0, "fail_at": (2,)}

    def flaky():
        state["n"] += 1
        if state["n"] in state["fail_at"]:
            raise GenvcHipError("health: injected -- an in-kernel hand-off of a one-launch decode step timed out", -5)
        real()
    m.gpt.engine.health = flaky
    got = synthesize_utt_streaming(m, src, ref, seg_len=1.0, stream_chunk_size=8, verbose=False, return_details=True)
    assert m.gpt.recoveries == 1
    assert torch.equal(torch.cat(got["tokens"], 1), torch.cat(want_s["tokens"], 1))
    np.testing.assert_allclose(got["wav"].cpu().numpy(), want_s["wav"].cpu().numpy(), atol=2e-4)
    state.update(n=0, fail_at=(2, 3, 4, 5))
    with pytest.raises(GenvcHipError):
        synthesize_utt_streaming(m, src, ref, seg_len=1.0, stream_chunk_size=8, verbose=False, return_details=True)
    del m
    torch.cuda.empty_cache()


def test_xcd_probe_gates_the_xcd_local_hand_off():
    """the XCD-local hand-off of the one-stream step assumed 8 XCDs x 32 resident workgroups without
    checking.  persist_prepare now probes the deal of a 256-workgroup grid of the step's own shape once per context; on an MI355X in
    SPX mode the probe passes and the XL instantiation runs (ids equal to the oracle); the per-XCD rank is per launch, so a launch
    that came up short (time-out test hook) does not poison the ranks of the next context."""
    from test_gpu_gpt import run_generate
    from oracle import genvc_oracle as O
    dims, w, eng = _engine(WIDE2, 3, 2)
    cond = synth.uniform(300, "cond_latents", (1, 32, 1024), 1.0)
    codes = synth.integers(300, "content_codes", (1, 13), 256)
    wc = {k: v.cpu() for k, v in w.items()}
    ref_t, _, _ = O.generate(wc, dims, cond, codes, GREEDY, max_new=40, stop_on_eos=False)
    for rep in range(3):                    # replays of differently sized graphs: ranks must be per launch, not accumulated
        _, toks, _ = run_generate(eng, dims, cond, codes, 40 if rep != 1 else 13, group=8 if rep != 1 else 5)
        assert eng.decode_variant() == 3
        assert torch.equal(toks.long(), ref_t[:, :toks.shape[1]])
    eng.close()


def test_perceiver_batches_beyond_one_skinny_group_and_short_contexts_vs_oracle():
    """The round-5 Perceiver (csrc/perceiver.hip: latents fragment-major on the skinny GEMM, context keys / values of all layers in one
    GEMM, graph replay per (B, F)) outside the fixture's two shapes: six batch elements (the latent path takes them in groups of four:
    128 rows per skinny GEMM), different contexts per element, a context shorter than one key tile and one that is not a multiple of
    16, a repeated call (graph replay) and a rebind -- against the oracle's PerceiverResampler restatement
    (/root/reference/layers/perceiver_encoder.py:265-319, pinned by tests/golden/perceiver.npz)."""
    from genvc_amd.engine import PerceiverEngine
    from oracle import genvc_oracle as O
    d = 256
    pre = "conditioning_perceiver."
    w = synth.make_weights(3, synth.perceiver_weight_spec(d, prefix=pre), device=DEV)
    eng = PerceiverEngine(dim=d, depth=4, dim_context=80, num_latents=32, dim_head=64, heads=8, ff_mult=4, max_batch=8, max_frames=600)
    eng.bind(w, prefix=pre)
    wc = {k: v.cpu() for k, v in w.items()}
    for B, Fr in ((6, 100), (1, 7), (3, 45), (6, 100)):
        x = synth.uniform(5, f"ctx_{B}_{Fr}", (B, Fr, 80), 1.0)
        y = eng.forward(x.to(DEV).contiguous()).cpu()
        ref = O.perceiver_forward(wc, x, prefix=pre)
        np.testing.assert_allclose(y.numpy(), ref.numpy(), atol=5e-5, err_msg=f"B={B} F={Fr}")
    # rebind other weights: the fragment-major / interleaved copies and the captured graphs must follow
    w2 = synth.make_weights(4, synth.perceiver_weight_spec(d, prefix=pre), device=DEV)
    eng.bind(w2, prefix=pre)
    x = synth.uniform(6, "ctx_rebind", (2, 100, 80), 1.0)
    y = eng.forward(x.to(DEV).contiguous()).cpu()
    ref = O.perceiver_forward({k: v.cpu() for k, v in w2.items()}, x, prefix=pre)
    np.testing.assert_allclose(y.numpy(), ref.numpy(), atol=5e-5)
    eng.close()


def test_stream_sessions_recover_when_the_time_out_surfaces_at_a_later_call(monkeypatch):
    """a hand-off time-out does not only surface at the health check behind the decode call -- the entry check of
    the NEXT library call (this step's prefill of a new segment, the next step's generate) reports what the previous call left behind.
    Injected: the prefill of a stream's second segment and, later, a generate call raise the library's time-out error once each.
    StreamSessions.step() must put every affected stream back to the start of its segment (the one that was popped but not yet
    decoding too), emit nothing twice, and every stream must still end with its solo conversion's tokens and waveform."""
    from genvc_amd._lib import GenvcHipError
    from genvc_amd.inference.inference_utils import segments, synthesize_utt_streaming
    from genvc_amd.streaming import StreamSessions
    m = _wide_tiny_model()
    refs = [synth.synth_audio(60 + i, "ref", 72000) for i in range(2)]
    srcs = [synth.synth_audio(80 + i, "src", 48000) for i in range(2)]
    solo = [synthesize_utt_streaming(m, srcs[i], refs[i], seg_len=1.0, stream_chunk_size=8, verbose=False, return_details=True) for i in range(2)]
    ss = StreamSessions(m, max_sessions=4, group=8)
    eng = ss.eng
    real_prefill, real_generate = eng.prefill, eng.generate
    calls = {"prefill": 0, "generate": 0}

    def prefill(*a, **k):
        calls["prefill"] += 1
        if calls["prefill"] == 3:
            raise GenvcHipError("prefill: injected -- an in-kernel hand-off of a one-launch decode step timed out", -5)
        return real_prefill(*a, **k)

    def generate(*a, **k):
        calls["generate"] += 1
        if calls["generate"] == 9:
            raise GenvcHipError("generate: injected -- an in-kernel hand-off of a one-launch decode step timed out", -5)
        return real_generate(*a, **k)
    eng.prefill, eng.generate = prefill, generate
    sids = [ss.open(r) for r in refs]
    for i, sid in enumerate(sids):
        for sg in segments(srcs[i], 16000, 5120):
            ss.push(sid, sg)
    wavs, steps = {}, 0
    while not ss.idle():
        for sid, chunks in ss.step().items():
            wavs.setdefault(sid, []).extend(chunks)
        steps += 1
        assert steps < 200
    assert ss.recoveries == 2
    for i, sid in enumerate(sids):
        mine = torch.cat(ss.close(sid), 1)[0].cpu()
        assert torch.equal(mine, torch.cat(solo[i]["tokens"], 1)[0].cpu()), f"stream {i}: tokens differ after the recoveries"
        w = torch.cat(wavs[sid], -1)
        assert w.shape == solo[i]["wav"].shape
        np.testing.assert_allclose(w.cpu().numpy(), solo[i]["wav"].cpu().numpy(), atol=2e-4)
    del m
    torch.cuda.empty_cache()


def test_rearm_returns_to_the_one_launch_steps_after_a_time_out_fallback(monkeypatch):
    """VERDICT round 4 (design / robustness): one transient hand-off time-out used to leave a context on the launch-per-phase paths for good.
    gvc_gpt_rearm puts it back on the one-launch steps once the caller knows the GPU is its own again: time-out (grid launched one workgroup
    short) -> error reported once -> fallback path generates the oracle's ids -> rearm -> the one-launch step again, same ids."""
    from genvc_amd._lib import GenvcHipError
    from test_gpu_gpt import run_generate
    from oracle import genvc_oracle as O
    dims, w, eng = _engine(WIDE2, 3, 4)
    cond = synth.uniform(300, "cond_latents", (2, 32, 1024), 1.0)
    codes = synth.integers(300, "content_codes", (2, 13), 256)
    n = 16
    ref_t, _, _ = O.generate({k: v.cpu() for k, v in w.items()}, dims, cond, codes, GREEDY, max_new=n, stop_on_eos=False)
    monkeypatch.setenv("GVC_PERSIST_TEST_GRID", "255")
    run_generate(eng, dims, cond[:1], codes[:1], 8)                 # garbage: every hand-off times out
    torch.cuda.synchronize()
    with pytest.raises(GenvcHipError):
        eng.health()
    monkeypatch.delenv("GVC_PERSIST_TEST_GRID")
    eng.reset(torch.arange(2, device=DEV, dtype=torch.int32))
    for B in (1, 2):
        _, toks, _ = run_generate(eng, dims, cond[:B], codes[:B], n)
        assert eng.decode_variant() in (1, 2, 4), eng.decode_variant()          # launch-per-phase paths
        assert torch.equal(toks.long(), ref_t[:B])
    eng.rearm()
    for B, variant in ((1, 3), (2, 5)):
        _, toks, _ = run_generate(eng, dims, cond[:B], codes[:B], n)
        assert eng.decode_variant() == variant                                  # the one-launch steps again
        assert torch.equal(toks.long(), ref_t[:B])
    eng.rearm()                                                                 # nothing to do: harmless
    eng.close()
