"""GPU: round-6 additions -- the HIP path against fixtures produced by the reference's OWN generation loop and harness functions
(oracle/make_golden.py:make_stream_loop / make_harness: `NewGenerationMixin.sample_stream`, `synthesize_utt_streaming`, `synthesize_utt`
executed, not restated), BASELINE configs[3] at full depth on the one-launch rows step, and the bf16-activation mode of that step."""
import numpy as np
import pytest
import torch

from genvc_amd import config as gcfg
from genvc_amd import synth

pytestmark = pytest.mark.gpu
DEV = "cuda"
GREEDY = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
GREEDY_KW = dict(do_sample=True, top_k=1, top_p=0.85, temperature=0.85, repetition_penalty=2.0, num_beams=1, length_penalty=1.0)


def _model(tiny, seed, stop_bias=None, max_new=None):
    from genvc_amd.inference.model_init import model_init_synthetic
    torch.cuda.empty_cache()
    m = model_init_synthetic(gcfg.default_config(tiny=tiny), seed=seed, device=DEV)[0]
    m.config.top_k = 1
    if stop_bias is not None:
        with torch.inference_mode():
            m.gpt.mel_head.bias[1025] = float(stop_bias)
        m.gpt.init_gpt_for_inference()
    if max_new is not None:
        m.gpt.max_gen_mel_tokens = int(max_new)
    return m


def test_generation_loop_vs_the_references_own_sample_stream(gold):
    """reference layers/stream_generator.py:645-881 run through the reference's GPT.get_generator / GPT.generate (gpt.py:594-621):
    three rows ending at three different steps (pads after a row's EOS, the loop ends with the last row, the EOS-step pair is yielded) and
    a run that ends on max_length -- GPT.generate and GPT.get_generator of the HIP path must yield the same ids and latents."""
    g = gold("stream_loop")
    m = _model(True, int(g["seed"]), stop_bias=float(g["eos_bias"]))
    s = int(g["eos_in_seed"])
    cond = synth.uniform(s, "cond_latents", (3, 32, 256), 1.0).to(DEV)
    codes = synth.integers(s, "content_codes", (3, 11), 256).to(DEV)
    toks = m.gpt.generate(cond, codes, group=4, **GREEDY_KW)
    assert np.array_equal(toks.cpu().numpy(), g["eos_tokens"])
    pairs = list(m.gpt.get_generator(m.gpt.compute_embeddings(cond, codes), **GREEDY_KW))
    assert len(pairs) == g["eos_tokens"].shape[1]                           # one pair per step up to and including the last row's EOS step
    assert np.array_equal(torch.stack([p[0] for p in pairs], 1).cpu().numpy(), g["eos_tokens"])
    lat = torch.stack([p[1] for p in pairs], 1).cpu().numpy()
    for b, e in enumerate(g["eos_ends"]):
        np.testing.assert_allclose(lat[b, :e + 1, :32], g["eos_latents_slice"][b, :e + 1], atol=1e-4)
    del m
    m = _model(True, int(g["seed"]), max_new=int(g["max_new"]))
    s = int(g["max_in_seed"])
    cond = synth.uniform(s, "cond_latents", (2, 32, 256), 1.0).to(DEV)
    codes = synth.integers(s, "content_codes", (2, 13), 256).to(DEV)
    toks = m.gpt.generate(cond, codes, **GREEDY_KW)                          # the cap comes from max_gen_mel_tokens, as in gpt.py:606
    assert toks.shape[1] == int(g["max_new"]) and np.array_equal(toks.cpu().numpy(), g["generate_tokens"])
    pairs = list(m.gpt.get_generator(m.gpt.compute_embeddings(cond, codes), **GREEDY_KW))
    assert np.array_equal(torch.stack([p[0] for p in pairs], 1).cpu().numpy(), g["max_tokens"])
    np.testing.assert_allclose(torch.stack([p[1] for p in pairs], 1).cpu().numpy()[:, :, :32], g["max_latents_slice"], atol=1e-4)
    del m
    torch.cuda.empty_cache()


@pytest.mark.parametrize("tag,case", [("tiny", "max"), ("tiny", "eos"), ("full", "max")])
def test_harness_vs_the_references_own_functions(gold, tag, case):
    """the reference's UNCHANGED synthesize_utt_streaming(seg_len=1.0, stream_chunk_size=8) and synthesize_utt(seg_len=1.0)
    (inference/inference_utils.py:135-217, :23-89) were run on the reference's own classes (tests/golden/harness_*.npz); the HIP harness on
    the same weights and the same 2.2 s source (a 0.2 s tail zero-padded to 0.32 s) must stream the same tokens in the same groups (short
    tails, EOS-step pairs), the same latents, and give both waveforms."""
    from genvc_amd.inference.inference_utils import synthesize_utt, synthesize_utt_streaming
    from test_oracle import check_harness_result
    g = gold("harness_" + tag)
    bias = float(g[case + "_stop_bias"])
    m = _model(tag == "tiny", int(g["seed"]), stop_bias=bias if bias >= 0 else None, max_new=int(g["max_new"]))
    src = synth.synth_audio(int(g[case + "_src_seed"]), "src", 35200)
    ref = synth.synth_audio(100, "ref", 72000)
    st = synthesize_utt_streaming(m, src, ref, seg_len=1.0, stream_chunk_size=8, verbose=False, return_details=True)
    ns = synthesize_utt(m, src, ref, seg_len=1.0, return_details=True)
    check_harness_result(g, case, st, ns, 2e-4, 1e-3)
    del m
    torch.cuda.empty_cache()


def _greedy_margins(O, ref_t, ref_logits, B, Tc, n):
    pen = [O.process_logits(ref_logits[i], torch.cat([torch.ones(B, 32 + Tc + 2, dtype=torch.long), torch.full((B, 1), 1024), ref_t[:, :i]], 1),
                            2.0, 1.0, 0, 1.0) for i in range(n)]
    return torch.stack([p.topk(2, -1)[0][:, 0] - p.topk(2, -1)[0][:, 1] for p in pen], 1)


@pytest.mark.parametrize("B,variant", [(8, 5), (1, 3)], ids=["8_streams_rows_step", "one_stream_step"])
def test_config3_bf16_weights_and_cache_at_full_depth_vs_oracle(B, variant):
    """BASELINE configs[3] at GenVC's full depth (L = 30, d = 1024, 4 heads; bf16 weight storage + bf16 KV cache, fp32 arithmetic):
    8 streams x 24 greedy steps on `k_rows_persist<8, 1, 1, 256>` (and one stream on the one-launch step) against `oracle.generate` on
    bf16-rounded weights with k / v rounded as they enter its cache (reference loop: layers/stream_generator.py:809-881, block math
    gpt_inference.py:92-112).  Input seed margin-screened on the CPU (smallest greedy gap of the oracle over all 8 x 24 decisions
    3.7e-3), re-asserted here: ids EQUAL, latents <= 2e-3 (thirty layers of bf16-rounded k / v: a value within 1e-7 of a rounding boundary
    lands one bf16 ulp apart)."""
    from genvc_amd.engine import GptEngine
    from oracle import genvc_oracle as O
    from test_gpu_gpt import run_generate, _round_bf16
    torch.cuda.empty_cache()
    dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
    w = synth.make_weights(5, synth.gpt_weight_spec(dims), device=DEV)
    eng = GptEngine(dims, max_slots=8, max_rows=2048, weight_dtype="bf16_kv")
    eng.bind(w)
    wr = _round_bf16({k: v.cpu() for k, v in w.items()})
    dims_o = dict(dims, kv_bf16=True)
    Tc, n = 13, 24
    cond = synth.uniform(100, "cond_latents", (8, 32, 1024), 1.0)[:B]
    codes = synth.integers(100, "content_codes", (8, Tc), 256)[:B]
    before = eng.rows_step_launches()
    _, toks, lats = run_generate(eng, dims, cond, codes, n)
    assert eng.decode_variant() == variant
    if B > 1:
        assert eng.rows_step_launches() > before, "the decode steps did not run on the one-launch rows step"
    ref_t, ref_l, ref_logits = O.generate(wr, dims_o, cond, codes, GREEDY, max_new=n, stop_on_eos=False)
    margins = _greedy_margins(O, ref_t, ref_logits, B, Tc, n)
    assert float(margins.min()) >= 3e-3, f"input seed 100 is not margin-screened any more: {float(margins.min()):.2e}"
    assert torch.equal(toks.long(), ref_t), "ids differ from the oracle on a margin-screened input"
    np.testing.assert_allclose(lats.numpy(), ref_l.numpy(), atol=2e-3)
    eng.close()


def _act_bf16_case(L, B, Tc, n, H=4, in_seed=100):
    """HIP weight_dtype 3 run + the oracle's act_bf16 run + the oracle re-run on an input perturbed by 2e-7 relative (its own
    reproducibility: rounding activations to bf16 makes the map discontinuous, a deviation of 1e-5 becomes 1e-3 within a layer)"""
    from genvc_amd.engine import GptEngine
    from oracle import genvc_oracle as O
    from test_gpu_gpt import run_generate, _round_bf16
    torch.cuda.empty_cache()
    dims = gcfg.gpt_dims(dict(gcfg.DEFAULT_MODEL_ARGS, gpt_layers=L, gpt_n_heads=H))
    w = synth.make_weights(5, synth.gpt_weight_spec(dims), device=DEV)
    eng = GptEngine(dims, max_slots=max(B, 8), max_rows=8192, weight_dtype="bf16_act")
    eng.bind(w)
    wr = _round_bf16({k: v.cpu() for k, v in w.items()})
    dims_o = dict(dims, kv_bf16=True, act_bf16=True)
    cond = synth.uniform(in_seed, "cond_latents", (B, 32, 1024), 1.0)
    codes = synth.integers(in_seed, "content_codes", (B, Tc), 256)
    _, toks, lats = run_generate(eng, dims, cond, codes, n)
    assert eng.decode_variant() == 5, "the decode steps did not run on the one-launch rows step"
    torch.cuda.synchronize()
    eng.health()
    eng.close()
    ref_t, ref_l, ref_logits = O.generate(wr, dims_o, cond, codes, GREEDY, max_new=n, stop_on_eos=False)
    g = torch.Generator().manual_seed(0)
    cond2 = cond * (1 + 2e-7 * torch.randn(cond.shape, generator=g))
    # (teacher-forced on the first run's tokens: the yardstick is the latents' reproducibility, not the loop's)
    _, cache = None, None
    prefix, ids = O.compute_embeddings(wr, dims_o, cond2, codes)
    z, _, cache = O.gpt_prefill(wr, dims_o, prefix)
    pert = [z]
    for j in range(1, n):
        z, _, cache = O.gpt_decode_step(wr, dims_o, cache, ref_t[:, j - 1], j)
        pert.append(z)
    pert_l = torch.stack(pert, 1)
    margins = _greedy_margins(O, ref_t, ref_logits, B, Tc, n)
    return toks.long(), lats, ref_t, ref_l, pert_l, margins


@pytest.mark.parametrize("L,B,Tc,n,H", [(2, 8, 13, 24, 4), (30, 8, 13, 24, 4), (2, 5, 120, 16, 4), (2, 12, 13, 12, 4), (2, 8, 13, 16, 16), (2, 3, 300, 10, 8),
                                        (2, 12, 150, 10, 4), (2, 16, 300, 8, 16)],
                         ids=["8_streams", "8_streams_full_depth", "key_chunks", "16_rows", "16_heads", "8_heads_4_chunks", "16_rows_2_chunks",
                              "16_rows_4_chunks_16_heads"])
def test_rows_step_bf16_activations_vs_oracle(L, B, Tc, n, H):
    """weight_dtype 3 (csrc/persist_rows_b16.h): the one-launch rows step with bf16 activations across its hand-offs and bf16 MFMAs, against
    the oracle with the same rounding points (`dims["act_bf16"]`; reference block math gpt_inference.py:92-112, loop stream_generator.py:809-881).
    bf16 cannot be bit-exact; the claim, per SURVEY: agreement rate + tolerance --
      * step 0 (the prefill's row: GEMM path, fp32 activations) agrees to 2e-3 like mode 2;
      * the latents of the rows steps deviate from the oracle by no more than the ORACLE ITSELF deviates when its input moves by 2e-7
        relative (median and 99.9 % quantile within 3x of that yardstick -- the deviation grows like the square root of the pre-rounding
        difference, and the HIP path's prefill differs from the oracle's by more than 2e-7 on long prefixes: measured ratios 1.0 - 2.2):
        the kernel is inside the oracle's reproducibility ball;
      * >= 85 % of the greedy ids equal the oracle's, and a first divergence only where the oracle's own top-1 / top-2 gap is < 2e-2 (the latents'
        reproducibility noise, ~2e-3 median / 2e-2 max, is ~1e-2 in the logits)."""
    toks, lats, ref_t, ref_l, pert_l, margins = _act_bf16_case(L, B, Tc, n, H)
    np.testing.assert_allclose(lats[:, 0].numpy(), ref_l[:, 0].numpy(), atol=2e-3)
    agree = toks == ref_t
    first = min(int((~agree[b]).nonzero()[0]) if (~agree[b]).any() else n for b in range(B))
    assert first >= 2
    d_hip = (lats[:, 1:first] - ref_l[:, 1:first]).abs().flatten()
    d_ref = (pert_l[:, 1:first] - ref_l[:, 1:first]).abs().flatten()
    q = lambda t, p: float(torch.quantile(t[::max(1, t.numel() // 200000)].double(), p))
    assert q(d_hip, 0.5) <= 3.0 * q(d_ref, 0.5) + 1e-4, (q(d_hip, 0.5), q(d_ref, 0.5))
    assert q(d_hip, 0.999) <= 3.0 * q(d_ref, 0.999) + 1e-3, (q(d_hip, 0.999), q(d_ref, 0.999))
    assert float(agree.float().mean()) >= 0.85, float(agree.float().mean())
    for b in range(B):
        bad = (~agree[b]).nonzero()
        if len(bad):
            assert float(margins[b, int(bad[0])]) < 2e-2, (b, int(bad[0]), float(margins[b, int(bad[0])]))


@pytest.mark.parametrize("tiny", [True, False], ids=["tiny", "full_size_6s_segment"])
def test_non_streaming_paths_reuse_the_decode_latents(tiny):
    """reference inference/inference_utils.py:68-76 recomputes a segment's latents with a second forward pass (gpt.py:375-508, trimmed with
    sub = -5, :491, :508); row i of that pass is the hidden state that predicted token i, i.e. the vector the decode loop produced at step i
    (stream_generator.py:865).  The harnesses reuse those by default (the EOS-step latent dropped); `repass_latents=True` runs the reference's
    pass: same tokens, latents and waveform <= 1e-4 apart -- synthesize_utt, synthesize_utt_chunked and GenVCModel.inference, with a
    max_length ending (no stop token: n rows) and a 6 s segment at full size (141 steps, contexts 110 -> 251)."""
    from genvc_amd.inference.inference_utils import synthesize_utt, synthesize_utt_chunked
    m = _model(tiny, 3 if tiny else 1, max_new=37 if tiny else 141)
    seg = 1.0 if tiny else 6.0
    src = synth.synth_audio(7, "src", 40000 if tiny else 96000)            # tiny: 1 s + 1 s + 0.5 s; full: one 6 s segment
    ref = synth.synth_audio(8, "ref", 72000)
    a = synthesize_utt(m, src, ref, seg_len=seg, return_details=True)
    b = synthesize_utt(m, src, ref, seg_len=seg, return_details=True, repass_latents=True)
    assert all(torch.equal(x, y) for x, y in zip(a["codes"], b["codes"]))
    assert a["latents"].shape == b["latents"].shape
    np.testing.assert_allclose(a["latents"].cpu().numpy(), b["latents"].cpu().numpy(), atol=1e-4)
    np.testing.assert_allclose(a["wav"].cpu().numpy(), b["wav"].cpu().numpy(), atol=1e-4)
    if tiny:
        ca = synthesize_utt_chunked(m, src, ref, seg_len=seg)
        cb = synthesize_utt_chunked(m, src, ref, seg_len=seg, repass_latents=True)
        np.testing.assert_allclose(ca.cpu().numpy(), cb.cpu().numpy(), atol=1e-4)
        # an EOS ending: the stop token is stripped, the EOS-step latent is not used
        with torch.inference_mode():
            m.gpt.mel_head.bias[1025] = 1.8
        m.gpt.init_gpt_for_inference()
        m.gpt.max_gen_mel_tokens = 37
        a = synthesize_utt(m, src, ref, seg_len=seg, return_details=True)
        b = synthesize_utt(m, src, ref, seg_len=seg, return_details=True, repass_latents=True)
        assert a["latents"].shape == b["latents"].shape and a["latents"].shape[1] == sum(int(c.numel()) for c in a["codes"])
        np.testing.assert_allclose(a["latents"].cpu().numpy(), b["latents"].cpu().numpy(), atol=1e-4)
    del m
    torch.cuda.empty_cache()


def test_rearm_after_a_single_failed_step_never_accepts_stale_granules(monkeypatch):
    """advisor finding (round 5): the one-stream step accepts a granule when its tag is (epoch + 1, layer, phase) and relies on a MONOTONIC
    epoch because nothing zeroes the granules.  The fallback / re-arm used to reset the epoch: if the timed-out call was a single decode step
    at epoch 0 (right after create), its granules kept the very tags the first re-armed step expects.  Now the epoch is never reset: a
    single failed decode_step, the fallback, a re-arm, then the one-launch step must give the oracle's ids (GVC_ERR_TIMEOUT is its own code)."""
    from genvc_amd._lib import GenvcHipError, GVC_ERR_TIMEOUT
    from genvc_amd.engine import GptEngine
    from test_gpu_gpt import run_generate
    from oracle import genvc_oracle as O
    torch.cuda.empty_cache()
    dims = gcfg.gpt_dims(dict(gcfg.DEFAULT_MODEL_ARGS, gpt_layers=2))
    w = synth.make_weights(3, synth.gpt_weight_spec(dims), device=DEV)
    eng = GptEngine(dims, max_slots=4, max_rows=2048)
    eng.bind(w)
    cond = synth.uniform(300, "cond_latents", (1, 32, 1024), 1.0)
    codes = synth.integers(300, "content_codes", (1, 13), 256)
    n = 16
    ref_t, _, _ = O.generate({k: v.cpu() for k, v in w.items()}, dims, cond, codes, GREEDY, max_new=n, stop_on_eos=False)
    slots = torch.zeros(1, device=DEV, dtype=torch.int32)
    eng.prefill(slots, eng.prefix_embeddings(cond.to(DEV), codes.to(DEV).int()), want_outputs=False)
    monkeypatch.setenv("GVC_PERSIST_TEST_GRID", "255")
    eng.decode_step(slots, torch.tensor([5], device=DEV, dtype=torch.int32))          # ONE step, at epoch 0: every hand-off times out
    torch.cuda.synchronize()
    with pytest.raises(GenvcHipError) as ei:
        eng.health()
    assert ei.value.code == GVC_ERR_TIMEOUT and ei.value.is_handoff_timeout
    monkeypatch.delenv("GVC_PERSIST_TEST_GRID")
    eng.reset(slots)
    eng.rearm()
    _, toks, _ = run_generate(eng, dims, cond, codes, n)
    assert eng.decode_variant() == 3, "not back on the one-launch step"
    assert torch.equal(toks.long(), ref_t), "the first re-armed steps accepted stale hand-off values"
    eng.close()


def test_conditioning_side_stream_never_overlaps_a_one_launch_step():
    """VERDICT round 5, item 8 (residency before issue): the one-launch steps need all 256 workgroups resident; the reference speaker's
    mel + Perceiver chain on the side stream (GenVCModel.get_gpt_cond_latents_async) would take CUs away -- a ~0.2 s bounded spin and a
    fallback.  A GPT call issued while that chain is in flight now waits for it first (GptEngine.watch_stream): 200 rounds of "start the
    conditioning chain, generate at once without joining": no hand-off time-out, the same ids every time, the one-launch step throughout."""
    m = _model(False, 1, max_new=12)
    eng = m.gpt.engine
    ref = synth.synth_audio(100, "ref", 72000).to(DEV)
    cond = m.get_gpt_cond_latents(ref, 24000)
    codes = synth.integers(5, "content_codes", (1, 13), 256).to(DEV)
    want = m.gpt.generate(cond, codes, **GREEDY_KW)
    assert eng.decode_variant() == 3
    torch.cuda.synchronize()
    before = eng.side_joins
    for i in range(200):
        fut = m.get_gpt_cond_latents_async(ref, 24000)
        got = m.gpt.generate(cond, codes, **GREEDY_KW)              # no fut.result(): the engine itself has to keep the two apart
        assert torch.equal(got, want), f"round {i}"
        c2 = fut.result()
    torch.cuda.synchronize()
    eng.health()                                                    # raises after a hand-off time-out
    assert eng.decode_variant() == 3 and m.gpt.recoveries == 0
    assert eng.side_joins - before >= 100, eng.side_joins - before   # the chain was really in flight when the GPT calls arrived
    np.testing.assert_allclose(c2.cpu().numpy(), cond.cpu().numpy(), atol=1e-6)
    del m
    torch.cuda.empty_cache()


def test_config4_decode_shape_topk50_vs_oracle():
    """BASELINE configs[4]'s real decode shape: 5 streams (the five 6 s segments of a 30 s source), Tc = 75 (110-row prefill, contexts 110 ->
    251 keys: the rows step's 2 / 4 key-chunk splits), top_k = 50, 141 sampled steps at full depth -- the graph-replayed loop (sampler kernel:
    bitonic sort, top-k threshold, top-p scan, inverse-CDF draw from the shared counter RNG) against the oracle's loop (reference
    layers/stream_generator.py:809-881 with HF's processors' semantics).  A draw within float rounding of a CDF boundary may differ and
    the sequences part there; everything before must agree, and at least 100 steps of every stream do."""
    from genvc_amd.engine import GptEngine
    from oracle import genvc_oracle as O
    from test_gpu_gpt import run_generate
    torch.cuda.empty_cache()
    dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
    w = synth.make_weights(1, synth.gpt_weight_spec(dims), device=DEV)
    eng = GptEngine(dims, max_slots=8, max_rows=2048)
    eng.bind(w)
    wc = {k: v.cpu() for k, v in w.items()}
    samp = dict(gcfg.DEFAULT_SAMPLING, top_k=50)
    B, Tc, n = 5, 75, 141
    cond = synth.uniform(77, "cond_latents", (1, 32, 1024), 1.0).expand(B, -1, -1).contiguous()
    codes = synth.integers(77, "content_codes", (B, Tc), 256)
    _, toks, lats = run_generate(eng, dims, cond, codes, n, sampling=samp, seed=17, group=48)
    assert eng.decode_variant() == 5
    ref_t, ref_l, _ = O.generate(wc, dims, cond, codes, samp, max_new=n, seed=17, stop_on_eos=False)
    m = min(toks.shape[1], ref_t.shape[1])
    agree = toks[:, :m].long() == ref_t[:, :m]
    first = [int((~agree[b]).nonzero()[0]) if (~agree[b]).any() else m for b in range(B)]
    assert min(first) >= 100, first
    k = min(first)
    np.testing.assert_allclose(lats[:, :k].numpy(), ref_l[:, :k].numpy(), atol=3e-4)
    eng.close()


@pytest.mark.parametrize("L,B", [(2, 1), (30, 1), (2, 3)], ids=["two_layers", "full_depth", "three_streams"])
def test_prefill_cond_then_cached_prefill_equals_the_full_prefill(L, B):
    """gvc_gpt_prefill_cond (a streaming session registers its speaker before the source arrives: the 32 conditioning rows go into the KV
    cache alone; the reference rebuilds them inside every segment's prefill, inference/inference_utils.py:43-66, gpt_inference.py:81-91):
    the first segment's prefill then computes 16 rows instead of 48.  Logits / latent of that cached prefill and of the decode steps behind
    it against the full prefill on fresh slots (<= 5e-5: the same rows through the same kernels) and against the oracle."""
    from genvc_amd.engine import GptEngine
    from oracle import genvc_oracle as O
    torch.cuda.empty_cache()
    dims = gcfg.gpt_dims(dict(gcfg.DEFAULT_MODEL_ARGS, gpt_layers=L))
    w = synth.make_weights(3, synth.gpt_weight_spec(dims), device=DEV)
    eng = GptEngine(dims, max_slots=8, max_rows=2048)
    eng.bind(w)
    wc = {k: v.cpu() for k, v in w.items()}
    cond = synth.uniform(91, "cond_latents", (B, 32, 1024), 1.0)
    codes = synth.integers(91, "codes", (B, 13), 256)
    s_sess = torch.arange(B, device=DEV, dtype=torch.int32)
    s_full = s_sess + 4
    eng.prefill_cond(s_sess, cond.to(DEV))
    prefix = eng.prefix_embeddings(cond.to(DEV), codes.to(DEV).int())
    lg_c, lat_c = eng.prefill(s_sess, prefix, n_cached=32)
    lg_f, lat_f = eng.prefill(s_full, prefix)
    z, logits, cache = O.gpt_prefill(wc, dims, O.compute_embeddings(wc, dims, cond, codes)[0])
    np.testing.assert_allclose(lg_c.cpu().numpy(), lg_f.cpu().numpy(), atol=5e-5)
    np.testing.assert_allclose(lg_c.cpu().numpy(), logits.numpy(), atol=1e-4)
    np.testing.assert_allclose(lat_c.cpu().numpy(), z.numpy(), atol=1e-4)
    tok = torch.tensor([5, 900, 77][:B], device=DEV, dtype=torch.int32)
    for j in range(1, 4):
        a = eng.decode_step(s_sess, tok)
        b = eng.decode_step(s_full, tok)
        z, logits, cache = O.gpt_decode_step(wc, dims, cache, tok.cpu().long(), j)
        np.testing.assert_allclose(a[0].cpu().numpy(), b[0].cpu().numpy(), atol=5e-5)
        np.testing.assert_allclose(a[0].cpu().numpy(), logits.numpy(), atol=1e-4)
    eng.close()


@pytest.mark.parametrize("NL", [16, 64])
def test_perceiver_other_latent_counts_vs_oracle(NL):
    """advisor finding (round 5): the rewritten Perceiver refused every latent count but GenVC's 32 (reference
    layers/perceiver_encoder.py:225-319 takes num_latents from its config).  16 and 64 latents, two batch elements, against the oracle."""
    from genvc_amd.engine import PerceiverEngine
    from oracle import genvc_oracle as O
    d, pre = 256, "conditioning_perceiver."
    w = synth.make_weights(3, synth.perceiver_weight_spec(d, num_latents=NL, prefix=pre), device=DEV)
    eng = PerceiverEngine(dim=d, depth=4, dim_context=80, num_latents=NL, dim_head=64, heads=8, ff_mult=4, max_batch=4, max_frames=600)
    eng.bind(w, prefix=pre)
    wc = {k: v.cpu() for k, v in w.items()}
    for B, Fr in ((1, 282), (2, 100), (3, 45)):
        x = synth.uniform(5, f"ctx_{B}_{Fr}", (B, Fr, 80), 1.0)
        y = eng.forward(x.to(DEV).contiguous()).cpu()
        ref = O.perceiver_forward(wc, x, prefix=pre)
        assert y.shape == ref.shape == (B, NL, d)
        np.testing.assert_allclose(y.numpy(), ref.numpy(), atol=5e-5, err_msg=f"NL={NL} B={B} F={Fr}")
    eng.close()
