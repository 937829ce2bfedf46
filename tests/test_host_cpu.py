"""CPU: C-ABI surface, host-side logic and the world_size-2 data-parallel path (gloo)."""
import os
import re
import subprocess

import numpy as np
import pytest
import torch

from genvc_amd import config as gcfg
from genvc_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from genvc_amd import _lib
    from genvc_amd.build import build
    lib = build(verbose=False)
    header = open(os.path.join(ROOT, "include", "genvc_hip.h")).read()
    declared = set(re.findall(r"\b(gvc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(_lib.exported_symbols()), declared ^ set(_lib.exported_symbols())
    out = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    have = {line.split()[-1] for line in out.splitlines() if line.strip()}
    assert declared <= have, declared - have
    # the library loads and answers without a GPU (no compute calls here)
    L = _lib.lib()
    assert L.gvc_version() >= 100
    assert L.gvc_last_error() is not None


def test_no_cpu_fallback_without_gpu():
    """the product path must fail loudly, never route through the oracle or torch arithmetic"""
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from genvc_amd._lib import GenvcHipError
    from genvc_amd.engine import GptEngine
    with pytest.raises(GenvcHipError):
        GptEngine(gcfg.gpt_dims(gcfg.TINY_MODEL_ARGS), max_slots=1, max_rows=64)
    src = "".join(open(os.path.join(dp, f)).read() for dp, _, fs in os.walk(os.path.join(ROOT, "genvc_amd"))
                  for f in fs if f.endswith(".py"))
    assert "import oracle" not in src and "from oracle" not in src


def test_synth_is_deterministic_and_device_independent_layout():
    a = synth.uniform(7, "x", (3, 5), 0.02)
    b = synth.uniform(7, "x", (3, 5), 0.02)
    assert torch.equal(a, b) and not torch.equal(a, synth.uniform(8, "x", (3, 5), 0.02))
    big = synth.uniform(1, "y", (1 << 16,), 1.0)
    assert abs(float(big.std()) - 1.0) < 0.02 and abs(float(big.mean())) < 0.02
    dims = gcfg.gpt_dims(gcfg.DEFAULT_MODEL_ARGS)
    assert dims["max_gen_mel_tokens"] == 602 and dims["max_mel_pos"] == 608 and dims["max_text_pos"] == 404
    assert dims["max_seq"] == 1083
    n = sum(int(np.prod(s)) for s, _ in synth.gpt_weight_spec(dims).values())
    assert n == 423635540                                            # SURVEY.md section 0 parameter count


def test_shell_state_dict_names_match_reference():
    from genvc_amd.layers.dvae import DiscreteVAE
    from genvc_amd.layers.gpt import GPT
    a = gcfg.TINY_MODEL_ARGS
    g = GPT(layers=a["gpt_layers"], model_dim=a["gpt_n_model_channels"], heads=a["gpt_n_heads"],
            max_text_tokens=a["gpt_max_text_tokens"], max_mel_tokens=a["gpt_max_audio_tokens"],
            max_prompt_tokens=a["gpt_max_prompt_tokens"])
    spec = synth.gpt_weight_spec(gcfg.gpt_dims(a))
    sd = g.state_dict()
    assert set(sd) == set(spec)
    assert all(tuple(sd[k].shape) == tuple(spec[k][0]) for k in spec)
    c = gcfg.TINY_CONTENT_DVAE
    dv = DiscreteVAE(channels=c["num_channels"], positional_dims=1, num_tokens=c["num_tokens"],
                     codebook_dim=c["codebook_dim"], hidden_dim=c["hidden_dim"],
                     num_resnet_blocks=c["num_resnet_blocks"], kernel_size=c["kernel_size"], num_layers=c["num_layers"],
                     use_transposed_convs=False)
    dspec = synth.dvae_weight_spec(c)
    assert set(dv.state_dict()) == set(dspec)
    assert all(tuple(dv.state_dict()[k].shape) == tuple(dspec[k][0]) for k in dspec)
    # ContentVec: fairseq HubertModel parameter names under `.model` (content_processor.py:14, model_init.py:29-30)
    from genvc_amd.layers.content_processor import ContentvecExtractor
    cv = ContentvecExtractor(gcfg.TINY_HUBERT)
    hspec = synth.hubert_weight_spec(gcfg.TINY_HUBERT)
    hsd = cv.model.state_dict()
    assert set(hsd) == set(hspec) and all(tuple(hsd[k].shape) == tuple(hspec[k][0]) for k in hspec)
    assert "model.encoder.layers.1.self_attn.q_proj.weight" in cv.state_dict()
    full = synth.hubert_weight_spec(gcfg.DEFAULT_HUBERT)
    n = sum(int(np.prod(v[0])) for v in full.values())
    assert 94_000_000 < n < 96_000_000                               # HuBERT-base


def test_harness_segmentation_and_chunks_match_oracle():
    from genvc_amd.inference.inference_utils import handle_chunks, segments
    from genvc_amd.layers.content_processor import contentvec_frames
    from oracle import genvc_oracle as O
    for n, sl in ((160000, 6.0), (160000, 1.0), (97000, 6.0), (24581, 6.0), (100, 6.0)):
        wav = torch.arange(n, dtype=torch.float32).unsqueeze(0) + 1
        got = [s.shape[-1] for s in segments(wav, int(sl * 16000), 5120)]
        assert got == [p for _, _, p in O.segment_source(n, sl)]
    assert [contentvec_frames(t) for t in (16000, 96000, 64000, 24581)] == [49, 299, 199, 76]
    w1 = synth.uniform(1, "w1", (8192,), 0.1)
    w2 = synth.uniform(1, "w2", (8192,), 0.1)
    c1, prev, ov = handle_chunks(w1.clone(), None, None)
    e1, eov = O.handle_chunks(w1, None)
    assert torch.equal(c1, e1) and torch.equal(ov, eov)
    c2, prev, ov2 = handle_chunks(w2.clone(), prev, ov)
    e2, _ = O.handle_chunks(w2, eov)
    assert torch.allclose(c2, e2)
    c3, _, ov3 = handle_chunks(torch.ones(1024), prev, ov2)           # short tail: raw tail returned (quirk 8)
    assert c3.shape[0] == 1024 and ov3 is None


def test_handle_chunks_matches_reference_fixture(gold):
    """genvc_amd's handle_chunks against outputs of the reference's own function (tests/golden/handle_chunks.npz)"""
    from genvc_amd.inference.inference_utils import handle_chunks
    g = gold("handle_chunks")
    prev, ov = None, None
    for i, n in enumerate(g["lens"]):
        wav = synth.uniform(int(g["seed"]), f"chunk{i}", (int(n),), 0.5)
        chunk, prev, ov = handle_chunks(wav.clone(), prev, ov, 1024)
        np.testing.assert_allclose(chunk.numpy(), g[f"chunk{i}"], rtol=0, atol=1e-7)
        assert (ov is not None) == bool(g[f"has_overlap{i}"])
        if ov is not None:
            np.testing.assert_array_equal(ov.numpy(), g[f"overlap{i}"])


def test_model_init_reads_the_reference_config_shape_and_reports_lost_tensors():
    """a config dict shaped like the reference's Coqpit dump (configs/vocoder_configs.py:19 `upsample_kernal_sizes`,
    hifigan_trainer.py:146 `content_dvae_config.audio.dvae_sample_rate`) and a checkpoint without a tensor of the path"""
    from genvc_amd.inference.model_init import GenVCModel, _load_checked, _merge
    cfg = gcfg.default_config(tiny=True)
    ck = {"vocoder_config": {"upsample_kernal_sizes": [4, 4, 8]},
          "content_dvae_config": {"audio": {"dvae_sample_rate": 22050}}}
    _merge(cfg, ck)
    cfg.vocoder_config.upsample_rates = [2, 2, 4]
    del cfg.vocoder_config["upsample_kernel_sizes"]
    m = GenVCModel(cfg)
    assert m.hifigan.cfg["upsample_kernel_sizes"] == [4, 4, 8] and m.content_sample_rate == 22050
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    # newer-torch weight-norm names are accepted
    g = sd.pop("hifigan.conv_pre.weight_g"); v = sd.pop("hifigan.conv_pre.weight_v")
    sd["hifigan.conv_pre.parametrizations.weight.original0"], sd["hifigan.conv_pre.parametrizations.weight.original1"] = g, v
    sd["discriminator.something"] = torch.zeros(3)                      # training-only tensors are ignored (strict=False)
    missing, unexpected = _load_checked(m, sd)
    assert not [k for k in missing if k.startswith("hifigan.")] and unexpected == ["discriminator.something"]
    del sd["gpt.gpt.h.1.mlp.c_fc.weight"]
    with pytest.raises(RuntimeError, match="c_fc"):
        _load_checked(m, sd)


def test_stop_len_rule():
    from genvc_amd.layers.gpt import GPT
    g = GPT(layers=1, model_dim=256, heads=4)
    t = torch.tensor([[5, 6, 1025, 1025, 1025], [7, 8, 9, 1025, 1025]])
    assert g._stop_len(t) == 4                       # loop ends at the step where the last row emits 1025
    assert g._stop_len(torch.tensor([[5, 6, 7], [8, 1025, 1025]])) == 3


def test_audio_loader_on_reference_like_wav(tmp_path, capsys):
    """WAV reader / writer round trip and the loader's checks (reference utils.py:49-75); resampling is the HIP kernel's job: without a GPU
    a file at another rate fails like the reference's loader fails -- message + None -- instead of taking a CPU path"""
    from genvc_amd.audio import load_audio, read_wav, save_wav
    x = synth.synth_audio(3, "a", 48000)[0]
    save_wav(str(tmp_path / "a.wav"), x, 48000)
    y, sr = read_wav(str(tmp_path / "a.wav"))
    assert sr == 48000 and y.shape == (1, 48000) and float((y[0] - x).abs().max()) < 1e-4
    z = load_audio(str(tmp_path / "a.wav"), 48000)                   # same rate: no resampling, checks + clip only
    assert z.shape == (1, 48000) and float(z.abs().max()) <= 1.0
    if not torch.cuda.is_available():
        assert load_audio(str(tmp_path / "a.wav"), 16000) is None
        assert "no CPU fallback" in capsys.readouterr().out
    save_wav(str(tmp_path / "pos.wav"), x.abs() + 0.01, 48000)      # no negative sample: rejected like the reference does (utils.py:66-69)
    assert load_audio(str(tmp_path / "pos.wav"), 48000) is None


def _dp_worker(rank, world, port, n_total, q):
    import torch.distributed as dist
    from genvc_amd.parallel_offline import gather_token_ids, shard
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    mine = shard(n_total, rank, world)
    local = torch.stack([torch.full((2, 6), 100 * j, dtype=torch.int32) + torch.arange(6, dtype=torch.int32)
                         for j in mine]) if mine else torch.zeros(0, 2, 6, dtype=torch.int32)
    out = gather_token_ids(local, n_total, 1025, rank, world)
    q.put((rank, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_total", [5, 8])
def test_data_parallel_gather_world2(n_total):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + n_total
    ps = [ctx.Process(target=_dp_worker, args=(r, 2, port, n_total, q)) for r in range(2)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(2))
    [p.join(timeout=60) for p in ps]
    exp = np.stack([np.full((2, 6), 100 * j, np.int32) + np.arange(6, dtype=np.int32) for j in range(n_total)])
    for r in range(2):
        assert np.array_equal(res[r], exp)            # every rank holds all utterances in order


class _StubModel:
    """duck-typed GenVCModel for the CPU test of the offline driver: every stage is a cheap deterministic function of its input
    (so any mix-up of utterances, segments or ranks changes the token ids), with the shapes of the real stages"""
    device = torch.device("cpu")
    content_sample_rate = 16000

    class _Cfg:
        top_p, top_k, temperature, length_penalty, repetition_penalty = 0.85, 1, 0.85, 1.0, 2.0

        class audio:
            sample_rate = 24000
    config = _Cfg()

    class _Extractor:
        @staticmethod
        def extract_content_features(wav):                               # [B,T] -> [B, T50, 4]
            t50 = (wav.shape[-1] - 400) // 320 + 1
            frames = wav[:, :t50 * 320].reshape(wav.shape[0], t50, 320)
            return torch.stack([frames.mean(-1), frames.amax(-1), frames.amin(-1), frames[..., 0]], -1)

    class _Dvae:
        @staticmethod
        def get_codebook_indices(x):                                     # [B,4,T50] -> int64 [B, ceil(T50 / 4)]
            tc = -(-x.shape[-1] // 4)
            v = torch.nn.functional.pad(x, (0, tc * 4 - x.shape[-1])).reshape(x.shape[0], 4, tc, 4).sum((1, 3))
            return (v * 1e4).round().long().abs() % 256

    class _Gpt:
        stop_audio_token, max_gen_mel_tokens = 1025, 12
        max_slots = 16                                                        # KV slots: bounds the streams of one joint decode
        calls = []
        seen_kwargs = []
        rolling_calls = []
        group_calls = []                                                      # streams per class of every generate_groups call

        def generate(self, cond, codes, **kw):
            self.calls.append(tuple(codes.shape))
            n = 5 + codes.shape[1] % 6
            base = (codes.sum(1, keepdim=True) + (cond[0].sum() * 1000).long()) % 1000
            return (base + torch.arange(n)[None, :]) % 1024

        def generate_rolling(self, jobs, **kw):                              # the real one keeps a rolling set of streams decoding
            self.rolling_calls.append(([int(t.shape[0]) for _, t in jobs], kw.get("max_rows"), kw.get("max_new_tokens")))
            return [self.generate(c, t) for c, t in jobs]

        def generate_groups(self, groups, **kw):                             # the real one decodes the groups together
            self.seen_kwargs.append(dict(kw))
            self.group_calls.append([int(t.shape[0]) for _, t in groups])
            assert sum(self.group_calls[-1]) <= self.max_slots or len(groups) == 1
            return [self.generate(c, t, **kw) for c, t in groups]

    content_extractor, content_dvae = _Extractor(), _Dvae()

    def __init__(self):
        self.gpt = self._Gpt()
        self.gpt.calls = []
        self.gpt.group_calls = []
        self.gpt.seen_kwargs = []
        self.gpt.rolling_calls = []

    def get_gpt_cond_latents(self, audio, sr):
        return audio[:, :64].reshape(1, 32, 2)


def _offline_job():
    lens = [160000, 96000, 230000, 160000, 40000, 96000, 160000]     # 10 s, 6 s, 14.4 s (3 segments), ..., 2.5 s
    srcs = [synth.synth_audio(900 + i, "src", n) for i, n in enumerate(lens)]
    return srcs, synth.synth_audio(7, "ref", 72000)


def _offline_worker(rank, world, port, q):
    import torch.distributed as dist
    from genvc_amd.parallel_offline import convert_offline
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    srcs, ref = _offline_job()
    m = _StubModel()
    # (`group` is generate_groups' "decode steps per host look", exactly as bench.py's offline leg passes it: it must reach the
    # model and must not be mistaken for the process group of the all_gather)
    out = convert_offline(m, srcs, ref, seg_len=6.0, micro_batch=2, rank=rank, world=world, group=48)
    assert m.gpt.seen_kwargs and all(k.get("group") == 48 for k in m.gpt.seen_kwargs)
    q.put((rank, out.numpy(), len(m.gpt.calls), m.gpt.group_calls))
    dist.barrier()
    dist.destroy_process_group()


def test_convert_offline_world2_unequal_lengths():
    """the offline driver itself at world size 2 (gloo): length-sorted round-robin sharding of utterances of different lengths,
    micro-batches that share a call per (segment, length), ONE all_gather -- every rank ends with the world-1 result"""
    import torch.multiprocessing as mp
    from genvc_amd.parallel_offline import convert_offline, plan
    srcs, ref = _offline_job()
    lens = [int(s.shape[-1]) for s in srcs]
    assert plan(lens, 0, 2) == [2, 3, 1, 4] and plan(lens, 1, 2) == [0, 6, 5]        # longest first, dealt round-robin
    m1 = _StubModel()
    one = convert_offline(m1, srcs, ref, seg_len=6.0, micro_batch=2, rank=0, world=1)
    assert one.shape == (7, 3, 12) and one.dtype == torch.int32
    # per utterance == converting it alone (no batch-mate leaks into a row); absent segments are all stop tokens
    for i, s in enumerate(srcs):
        alone = convert_offline(_StubModel(), [s], ref, seg_len=6.0, micro_batch=1)
        assert torch.equal(alone[0], one[i, :alone.shape[1]])
        assert bool((one[i, alone.shape[1]:] == 1025).all())
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    ps = [ctx.Process(target=_offline_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in ps]
    res = {r: (o, n, g) for r, o, n, g in (q.get(timeout=180) for _ in range(2))}
    [p.join(timeout=60) for p in ps]
    for r in range(2):
        assert np.array_equal(res[r][0], one.numpy())
    assert res[0][1] + res[1][1] <= len(m1.gpt.calls) + 3       # sharding does not multiply the generate calls
    # every wave (micro-batch of 2 utterances) is ONE joint decode over all of its (segment, length) classes -- the per-rank path
    # is the same at world size 1 and 2 (bench.py sized the KV slots so that N > 1 fell back to one decode per class)
    for r in range(2):
        n_mine = len(plan(lens, r, 2))
        assert len(res[r][2]) == -(-n_mine // 2), res[r][2]
        assert all(len(classes) >= 2 for classes in res[r][2][:1])          # the first wave holds utterances of >= 2 segments
    assert len(m1.gpt.group_calls) == -(-len(srcs) // 2)


def test_convert_batch_packs_classes_by_kv_slots():
    """convert_batch packs as many (segment, length) classes into one joint decode as the context has KV slots; with too few
    slots it degrades to more calls, never to a wrong result"""
    from genvc_amd.parallel_offline import convert_offline
    srcs, ref = _offline_job()
    big, small = _StubModel(), _StubModel()
    small.gpt.max_slots = 2
    a = convert_offline(big, srcs, ref, seg_len=6.0, micro_batch=4)
    b = convert_offline(small, srcs, ref, seg_len=6.0, micro_batch=4)
    assert torch.equal(a, b)
    assert len(big.gpt.group_calls) == 2 and len(small.gpt.group_calls) > 2
    assert all(sum(c) <= 2 or len(c) == 1 for c in small.gpt.group_calls)


def test_bench_self_launches_n_ranks(monkeypatch):
    """`python bench.py --gpus N` (N > 1) with no launcher in the environment becomes `torch.distributed.run --nproc-per-node N` on
    127.0.0.1 with the caller's flags; with fewer than N GPUs visible it refuses instead of running one process and printing
    `n_gpus: 1` (missing item 1)."""
    import importlib.util
    import argparse
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    seen = {}

    def fake_exec(exe, cmd, env):
        seen["exe"], seen["cmd"], seen["env"] = exe, cmd, env
        raise SystemExit(0)

    monkeypatch.setattr(bench.os, "execvpe", fake_exec)
    monkeypatch.setattr(bench.sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    monkeypatch.delenv("GVC_BENCH_SAME_DEVICE", raising=False)
    with pytest.raises(SystemExit):
        bench.self_launch(argparse.Namespace(gpus=4))
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert cmd[-6:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    # too few GPUs: a loud refusal, not a silent 1-process run
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 1)
    with pytest.raises(SystemExit) as e:
        bench.self_launch(argparse.Namespace(gpus=4))
    assert "only 1 GPU" in str(e.value)


def test_convert_offline_rolling_matches_waves():
    """the rolling offline driver (one generate_rolling call over all of a rank's classes, waves only share their front-end calls)
    returns what the wave-by-wave driver returns; the streams in flight are capped at what one wave holds"""
    from genvc_amd.parallel_offline import convert_offline
    srcs, ref = _offline_job()
    a, b = _StubModel(), _StubModel()
    waves = convert_offline(a, srcs, ref, seg_len=6.0, micro_batch=2)
    roll = convert_offline(b, srcs, ref, seg_len=6.0, micro_batch=2, rolling=True, tokens_per_second=2.0, max_new_tokens=12)
    assert torch.equal(waves, roll)
    assert len(b.gpt.rolling_calls) == 1 and not b.gpt.group_calls
    rows, max_rows, budgets = b.gpt.rolling_calls[0]
    assert sum(rows) == sum(-(-n // 96000) for n in [160000, 96000, 230000, 160000, 40000, 96000, 160000])     # one row per segment
    assert max_rows == 3 * 2                                   # 3 segments per utterance at most x micro-batch of 2
    assert len(budgets) == len(rows) and max(budgets) == 12 and min(budgets) >= 1
    # sampling runs keep the wave-by-wave driver (per-class random streams)
    c = _StubModel()
    c.config.top_k = 15
    convert_offline(c, srcs, ref, seg_len=6.0, micro_batch=2, rolling=True)
    assert not c.gpt.rolling_calls and c.gpt.group_calls
    _StubModel.config.top_k = 1


class _ToyEngine:
    """CPU stand-in for GptEngine behind the REAL GPT.generate / GPT.generate_rolling host loops: a deterministic toy language model
    whose next token depends on the slot's prefix AND on the whole id history of the row (what repetition_penalty reads), with
    per-slot state as the library keeps it.  Any mix-up of slots, histories, budgets or retirement order changes the ids."""
    STOP = 1025

    def __init__(self, stop_every=7):
        self.sig = {}                      # slot -> prefix signature
        self.stop_every = stop_every
        self.calls = []                    # rows per generate call

    def prefix_embeddings(self, cond, codes):
        B, Tc = codes.shape
        p = torch.zeros(B, cond.shape[1] + Tc + 2, 1)
        p[:, 0, 0] = (codes.long().sum(1) * 31 + (cond.sum((1, 2)) * 100).long() + Tc).float()
        return p

    def prefill(self, slots, prefix, want_outputs=False, n_cached=0):
        for b, s in enumerate(slots.tolist()):
            self.sig[s] = int(prefix[b, 0, 0])

    def generate(self, slots, ids, ids_len, fin, params, i0, n, toks, lats, max_keys=0):
        self.calls.append(int(slots.shape[0]))
        for b, s in enumerate(slots.tolist()):
            for i in range(n):
                L = int(ids_len[b])
                if int(fin[b]):
                    t = self.STOP
                else:
                    h = (self.sig[s] * 131 + int((ids[b, :L].long() * torch.arange(1, L + 1)).sum())) % 100003
                    t = self.STOP if h % self.stop_every == 0 else h % 1024
                    if t == self.STOP:
                        fin[b] = 1
                ids[b, L] = t
                ids_len[b] = L + 1
                toks[b, i0 + i] = t

    def health(self):
        pass


def _toy_gpt(max_slots, stop_every=7):
    from genvc_amd.layers.gpt import GPT
    g = GPT(layers=1, model_dim=64, heads=1)
    g.engine = _ToyEngine(stop_every)
    g.max_slots = max_slots
    return g


def test_generate_rolling_retires_rows_one_by_one_and_matches_generate():
    """GPT.generate_rolling (the offline leg's decode driver) over a fake engine: jobs of unequal size and prefix length, rows that stop
    at ragged steps (per-row retirement, /root/reference/layers/stream_generator.py:861-874), a job that runs out of budget, slots
    reused out of order with repetition_penalty > 1 (the id history of a reused slot must start clean), max_rows below the slot
    count -- every job must get exactly what generate() gives it alone."""
    torch.manual_seed(0)
    sizes = [(5, 9), (3, 4), (8, 13), (1, 6), (6, 9), (2, 20), (7, 5), (4, 11)]
    jobs = [(torch.rand(b, 32, 2), torch.randint(0, 256, (b, tc))) for b, tc in sizes]
    budgets = [40, 40, 12, 40, 25, 40, 40, 9]
    kw = dict(top_k=1, top_p=0.85, temperature=0.85, repetition_penalty=2.0, do_sample=True, num_beams=1)
    solo = []
    for (c, t), bud in zip(jobs, budgets):
        g = _toy_gpt(8)
        solo.append(g.generate(c, t, group=4, max_new_tokens=bud, **kw))
    ends = [int(((o == 1025).long().argmax(1) + (o != 1025).all(1).long() * o.shape[1]).max()) for o in solo]
    assert len(set(ends)) >= 4, ends                         # the jobs really end ragged
    for max_slots, max_rows, group in ((16, None, 4), (16, 9, 3), (8, None, 16), (12, None, 1)):
        g = _toy_gpt(max_slots)
        g.rolling_stats = {}
        out = g.generate_rolling(jobs, group=group, max_rows=max_rows, max_new_tokens=budgets, **kw)
        for i, (o, ref) in enumerate(zip(out, solo)):
            assert torch.equal(o, ref), f"max_slots {max_slots} max_rows {max_rows} group {group}: job {i} differs from generate()"
        cap = min(max_slots, max_rows or max_slots)
        assert max(g.engine.calls) <= cap
        st = g.rolling_stats
        assert st["row_steps_live"] <= st["row_steps_issued"]
        if group == 1:                                       # a host look after every step: a stopped row never takes another step
            assert st["row_steps_live"] == st["row_steps_issued"]
    # a job wider than the streams in flight is refused loudly
    with pytest.raises(ValueError):
        _toy_gpt(4).generate_rolling(jobs, max_new_tokens=8, **kw)


def _gather8_worker(rank, world, port, q):
    import torch.distributed as dist
    from genvc_amd.parallel_offline import convert_offline
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    torch.manual_seed(5)
    srcs = [torch.rand(1, 160000) for _ in range(64)]
    ref = torch.rand(1, 72000)
    toks = convert_offline(_StubModel(), srcs, ref, seg_len=6.0, micro_batch=8, rank=rank, world=world, rolling=True,
                           tokens_per_second=2.0, max_new_tokens=12)
    if rank == 0:
        q.put(toks.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_convert_offline_world8_equals_world1_for_64_utterances():
    """BASELINE configs[2]'s sharding at the size the driver runs it: 64 utterances, micro-batch 8, EIGHT ranks (gloo on CPU) -- the
    all_gather result on rank 0 must equal the single-process result, utterance by utterance
    (partitioning: /root/reference/inference/inference_utils.py:43-77; plan: genvc_amd/parallel_offline.py)."""
    import torch.multiprocessing as mp
    from genvc_amd.parallel_offline import convert_offline
    torch.manual_seed(5)
    srcs = [torch.rand(1, 160000) for _ in range(64)]
    ref = torch.rand(1, 72000)
    one = convert_offline(_StubModel(), srcs, ref, seg_len=6.0, micro_batch=8, rolling=True, tokens_per_second=2.0, max_new_tokens=12).numpy()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_gather8_worker, args=(r, 8, port, q)) for r in range(8)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got.shape == one.shape and np.array_equal(got, one)


def test_stream_sessions_rearm_policy_backs_off_and_gives_up():
    """advisor finding (round 5): StreamSessions used to re-arm the one-launch steps at EVERY idle moment after a time-out recovery -- under
    persistent contention a loop of time-out, recovery, idle, re-arm.  Now idle() is a pure predicate and maybe_rearm() re-arms only at an
    idle moment, only after `rearm_after_s` quiet seconds, doubles the wait after every failed try and stops after `rearm_max_tries`."""
    from genvc_amd.streaming import StreamSessions

    class Eng:
        def __init__(self):
            self.rearmed = 0

        def rearm(self):
            self.rearmed += 1

    import torch
    real_sync = torch.cuda.synchronize
    torch.cuda.synchronize = lambda *a, **k: None
    try:
        ss = object.__new__(StreamSessions)
        ss.sessions, ss.eng = {}, Eng()
        ss._rearm, ss.rearm_gave_up, ss.rearm_after_s, ss.rearm_max_tries = False, False, 5.0, 3
        ss._rearm_wait, ss._rearm_tries, ss._last_timeout, ss.rearms, ss.recoveries = 5.0, 0, 0.0, 0, 0
        assert ss.idle() and not ss.maybe_rearm(now=100.0)          # nothing pending
        ss._note_timeout(now=100.0)                                 # first time-out
        assert not ss.maybe_rearm(now=102.0)                        # too early
        assert ss.maybe_rearm(now=105.5) and ss.eng.rearmed == 1    # after 5 quiet seconds
        ss._note_timeout(now=106.0)                                 # ... which failed: the wait doubles
        assert ss._rearm_wait == 10.0 and not ss.maybe_rearm(now=112.0)
        assert ss.maybe_rearm(now=116.5) and ss.eng.rearmed == 2
        ss._note_timeout(now=117.0)
        assert ss._rearm_wait == 20.0 and ss.maybe_rearm(now=140.0) and ss.eng.rearmed == 3
        ss._note_timeout(now=141.0)                                 # the third failed try: stay on the fallback for good
        assert ss.rearm_gave_up and not ss.maybe_rearm(now=10000.0) and ss.eng.rearmed == 3
        # a busy scheduler never re-arms
        ss.rearm_gave_up, ss._rearm = False, True

        class S:
            decoding, queue = True, []
        ss.sessions = {0: S()}
        assert not ss.idle() and not ss.maybe_rearm(now=20000.0)
    finally:
        torch.cuda.synchronize = real_sync
