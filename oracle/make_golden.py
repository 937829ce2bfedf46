"""Generate tests/golden/*.npz by running the REFERENCE's own classes on synthetic weights.

Runs only in the build container (needs /root/reference); the outputs are small data
fixtures (inputs are re-derived from seeds by genvc_amd.synth, expected outputs are
stored).  Nothing from /root/reference is copied: the reference modules are imported,
loaded with deterministic weights through load_state_dict, and called.

Import recipe per SURVEY.md section 8(c): import transformers first, then register stub
modules for torchmetrics / torchaudio / librosa (only used by training code paths).

    python oracle/make_golden.py [--only tiny|full|perceiver|dvae|sampler]
"""
import argparse
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genvc_amd import config as gcfg      # noqa: E402
from genvc_amd import synth               # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def import_reference():
    import transformers  # noqa: F401  (must precede the torchaudio stub)
    sys.dont_write_bytecode = True

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Acc(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    stub("torchmetrics")
    stub("torchmetrics.classification", MulticlassAccuracy=_Acc)
    ta = stub("torchaudio")
    ta.transforms = stub("torchaudio.transforms")
    ta.functional = stub("torchaudio.functional")
    stub("librosa")
    sys.path.insert(0, "/root/reference")
    from layers.gpt import GPT
    from layers.dvae import DiscreteVAE
    return GPT, DiscreteVAE


def build_ref_gpt(GPT, model_args, weights):
    a = model_args
    g = GPT(layers=a["gpt_layers"], model_dim=a["gpt_n_model_channels"], heads=a["gpt_n_heads"],
            max_text_tokens=a["gpt_max_text_tokens"], max_mel_tokens=a["gpt_max_audio_tokens"],
            max_prompt_tokens=a["gpt_max_prompt_tokens"], number_text_tokens=a["gpt_number_text_tokens"],
            start_text_token=a["gpt_start_text_token"], stop_text_token=a["gpt_stop_text_token"],
            num_audio_tokens=a["gpt_num_audio_tokens"], start_audio_token=a["gpt_start_audio_token"],
            stop_audio_token=a["gpt_stop_audio_token"], code_stride_len=a["gpt_code_stride_len"])
    missing, unexpected = g.load_state_dict(weights, strict=False)
    missing = [m for m in missing if not m.endswith(".attn.bias") and not m.endswith("masked_bias")]
    assert not missing and not unexpected, (missing, unexpected)
    g.eval()
    g.init_gpt_for_inference()
    return g


@torch.inference_mode()
def ref_greedy(g, cond, codes, n_steps, sampling):
    """Drive the reference GPT2InferenceModel exactly as sample_stream does
    (layers/stream_generator.py:809-881), with HF's own logits processors."""
    from transformers.generation.logits_process import (
        RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)
    procs = [RepetitionPenaltyLogitsProcessor(sampling["repetition_penalty"]),
             TemperatureLogitsWarper(sampling["temperature"]),
             TopKLogitsWarper(sampling["top_k"], min_tokens_to_keep=1),
             TopPLogitsWarper(sampling["top_p"], min_tokens_to_keep=1)]
    gi = g.gpt_inference
    ids = g.compute_embeddings(cond, codes)
    prefix = gi.cached_prefix_emb.clone()
    stop = g.stop_audio_token
    B = ids.shape[0]
    unfinished = torch.ones(B, dtype=torch.long)
    pkv = None
    toks, lats, logit_rows, margins = [], [], [], []
    for step in range(n_steps):
        mask = torch.ones_like(ids)
        inputs = gi.prepare_inputs_for_generation(ids, past_key_values=pkv, attention_mask=mask, use_cache=True)
        out = gi(**inputs, return_dict=True, output_hidden_states=True)
        pkv = out.past_key_values
        logits = out.logits[:, -1, :]
        scores = logits
        for p in procs:
            scores = p(ids, scores)
        probs = torch.softmax(scores, dim=-1)
        assert int((probs > 0).sum(-1).max()) == 1, "top_k=1 must leave one candidate"
        nxt = probs.argmax(-1)
        nxt = nxt * unfinished + stop * (1 - unfinished)
        # margin of the decision on the repetition-penalised logits (what argmax sees)
        pen = procs[0](ids, logits)
        top2 = torch.topk(pen, 2, dim=-1)[0]
        margins.append((top2[:, 0] - top2[:, 1]).numpy())
        lat = gi.final_norm(out.hidden_states[-1][:, -1])          # stream_generator.py:865
        toks.append(nxt.numpy()); lats.append(lat.numpy()); logit_rows.append(logits.numpy())
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        unfinished = unfinished * (nxt != stop).long()
        if unfinished.max() == 0:
            break
    return dict(prefix=prefix.numpy(), fake_ids=g.compute_embeddings(cond, codes).numpy(),
                tokens=np.stack(toks, 1), latents=np.stack(lats, 1), logits=np.stack(logit_rows, 0),
                margins=np.stack(margins, 1))


def gpt_inputs(seed, dims, B, Tc):
    d = dims["d_model"]
    cond = synth.uniform(seed, "cond_latents", (B, 32, d), 1.0)
    codes = synth.integers(seed, "content_codes", (B, Tc), 256)
    return cond, codes


def make_gpt(GPT, tag, model_args, seed, B, Tc, n_steps, keep_rows, min_margin=2e-3):
    """Seeds are screened (SURVEY.md section 7 'hard parts'): a fixture is kept only if every
    greedy decision has a top-1/top-2 margin >= min_margin, so fp32 reassociation on another
    device cannot flip a token.  The weights depend on `seed`, the inputs on the screened seed."""
    dims = gcfg.gpt_dims(model_args)
    w = synth.make_weights(seed, synth.gpt_weight_spec(dims))
    g = build_ref_gpt(GPT, model_args, w)
    samp = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
    for in_seed in range(seed * 100, seed * 100 + 50):
        cond, codes = gpt_inputs(in_seed, dims, B, Tc)
        r = ref_greedy(g, cond, codes, n_steps, samp)
        if r["margins"].min() >= min_margin:
            break
        print(f"  gpt_{tag}: input seed {in_seed} rejected (min margin {r['margins'].min():.2e})")
    else:
        raise RuntimeError("no input seed passed the margin screen")
    toks = torch.from_numpy(r["tokens"])
    # latent re-pass (inference_utils.py:68-76) on the generated, stop-stripped codes of row 0
    gen = toks[0][toks[0] != g.stop_audio_token].unsqueeze(0)
    with torch.inference_mode():
        lat2 = g(codes[:1], torch.tensor([codes.shape[1]]), gen, torch.tensor([gen.shape[1] * 1024]),
                 cond_latents=cond[:1], return_latent=True)
    n = r["tokens"].shape[1]
    rows = sorted(set([i for i in keep_rows if i < n] + [n - 1]))
    out = dict(seed=seed, in_seed=in_seed, B=B, Tc=Tc, fake_ids=r["fake_ids"],
               prefix_sum=r["prefix"].astype(np.float64).sum(), prefix_slice=r["prefix"][:, -3:, :16],
               tokens=r["tokens"], margins=r["margins"], logit_rows=np.array(rows),
               logits=r["logits"][rows], latents_slice=r["latents"][:, :, :32],
               latents_rows=r["latents"][:, rows], relatents=lat2.numpy()[:, :, :32],
               relatents_full_rows=lat2.numpy()[:, rows[:2]])
    np.savez_compressed(os.path.join(GOLD, f"gpt_{tag}.npz"), **out)
    print(f"gpt_{tag}: {n} steps, min margin {r['margins'].min():.3e}, tokens[0,:12]={r['tokens'][0,:12]}")
    return g, w, dims


def make_eos(GPT, seed=11):
    """Tiny model whose stop-token bias is raised until greedy decoding ends within a few steps:
    pins the EOS step (token 1025 + its latent are yielded, finished rows emit the pad)."""
    model_args = gcfg.TINY_MODEL_ARGS
    dims = gcfg.gpt_dims(model_args)
    w = synth.make_weights(seed, synth.gpt_weight_spec(dims))
    cond, codes = gpt_inputs(seed, dims, 2, 9)
    samp = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
    for bias in np.arange(0.2, 3.0, 0.05):
        w["mel_head.bias"][1025] = float(bias)
        g = build_ref_gpt(GPT, model_args, w)
        r = ref_greedy(g, cond, codes, 40, samp)
        n = r["tokens"].shape[1]
        ends = [(r["tokens"][b] == 1025).argmax() if (r["tokens"][b] == 1025).any() else -1 for b in range(2)]
        if n < 40 and min(ends) >= 4 and ends[0] != ends[1] and r["margins"].min() > 2e-3:
            np.savez_compressed(os.path.join(GOLD, "gpt_eos.npz"), seed=seed, stop_bias=float(bias),
                                tokens=r["tokens"], margins=r["margins"], latents_slice=r["latents"][:, :, :32])
            print(f"gpt_eos: bias {bias:.2f}, ends {ends}, steps {n}, min margin {r['margins'].min():.2e}")
            return
    raise RuntimeError("no stop bias produced a ragged EOS fixture")


@torch.inference_mode()
def make_perceiver(g_tiny, g_full, seed=5):
    out = {}
    for tag, g in (("tiny", g_tiny), ("full", g_full)):
        for B, Fr in ((1, 282), (2, 563)):
            mel = synth.uniform(seed, f"mel_{B}_{Fr}", (B, 80, Fr), 1.0)
            y = g.get_style_emb(mel, None)                      # [B,d,32]
            out[f"{tag}_{B}_{Fr}"] = y.numpy()
    np.savez_compressed(os.path.join(GOLD, "perceiver.npz"), seed=seed, **out)
    print("perceiver:", {k: v.shape for k, v in out.items()})


@torch.inference_mode()
def make_dvae(DiscreteVAE, seed=7):
    out = {}
    for tag, c in (("tiny", gcfg.TINY_CONTENT_DVAE), ("full", gcfg.DEFAULT_CONTENT_DVAE)):
        m = DiscreteVAE(channels=c["num_channels"], normalization=None, positional_dims=1,
                        num_tokens=c["num_tokens"], codebook_dim=c["codebook_dim"], hidden_dim=c["hidden_dim"],
                        num_resnet_blocks=c["num_resnet_blocks"], kernel_size=c["kernel_size"],
                        num_layers=c["num_layers"], use_transposed_convs=False)
        w = synth.make_weights(seed, synth.dvae_weight_spec(c))
        missing, unexpected = m.load_state_dict(w, strict=False)
        assert not unexpected and all(("decoder" in k) or k.startswith("codebook.") or "discrete_loss" in k
                                      for k in missing), (missing, unexpected)
        m.eval()
        for B, T in ((1, 49), (2, 299), (1, 199), (1, 16)):
            feat = synth.uniform(seed, f"feat_{B}_{T}", (B, c["num_channels"], T), 1.0)
            codes = m.get_codebook_indices(feat)
            logits = m.encoder(feat).permute(0, 2, 1)
            # decision margin of every code (distance gap between best and second best)
            e = m.codebook.embed
            flat = logits.reshape(-1, logits.shape[-1])
            dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ e + e.pow(2).sum(0, keepdim=True)
            top2 = torch.topk(-dist, 2, dim=1)[0]
            out[f"{tag}_codes_{B}_{T}"] = codes.numpy()
            out[f"{tag}_margin_{B}_{T}"] = (top2[:, 0] - top2[:, 1]).reshape(codes.shape).numpy()
            out[f"{tag}_enc_{B}_{T}"] = logits.numpy()[:, :, :16]
    np.savez_compressed(os.path.join(GOLD, "dvae.npz"), seed=seed, **out)
    print("dvae:", {k: v.shape for k, v in out.items() if "codes" in k})


@torch.inference_mode()
def make_sampler(seed=9):
    """HF processors on synthetic logits: pins oracle.process_logits for top_k in {1, 15, 50}."""
    from transformers.generation.logits_process import (
        RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)
    V = 1026
    logits = synth.uniform(seed, "logits", (4, V), 2.0)
    ids = torch.cat([torch.ones(4, 48, dtype=torch.long), torch.full((4, 1), 1024),
                     synth.integers(seed, "hist", (4, 20), 1024)], 1)
    out = dict(seed=seed, ids=ids.numpy())
    for k, p in ((1, 0.85), (15, 0.85), (50, 0.85), (15, 1.0), (1026, 0.5)):
        s = logits
        for proc in (RepetitionPenaltyLogitsProcessor(2.0), TemperatureLogitsWarper(0.85),
                     TopKLogitsWarper(k, min_tokens_to_keep=1)) + \
                ((TopPLogitsWarper(p, min_tokens_to_keep=1),) if p < 1.0 else ()):
            s = proc(ids, s)
        out[f"scores_k{k}_p{int(p * 100)}"] = s.numpy()
    np.savez_compressed(os.path.join(GOLD, "sampler.npz"), **out)
    print("sampler:", [k for k in out if k.startswith("scores")])


@torch.inference_mode()
def make_hifigan(seed=13):
    """reference HiFiGAN generator (layers/hifigan.py imports nnAudio at file level: stubbed, only the
    discriminators use it) on synthetic weight-normed weights."""
    stub = types.ModuleType("nnAudio"); stub.features = types.ModuleType("nnAudio.features")
    sys.modules["nnAudio"], sys.modules["nnAudio.features"] = stub, stub.features
    from layers.hifigan import HiFiGAN
    out = dict(seed=seed)
    for tag, c in (("tiny", gcfg.TINY_VOCODER), ("full", gcfg.DEFAULT_VOCODER)):
        g = HiFiGAN(c["input_feat_dim"], c["upsample_initial_channel"], c["resblock_kernel_sizes"],
                    c["resblock_dilation_sizes"], c["upsample_rates"], c["upsample_kernel_sizes"], resblock_type="2")
        w = synth.make_weights(seed, synth.hifigan_weight_spec(c))
        missing, unexpected = g.load_state_dict(w, strict=True)
        g.eval()
        for B, n in ((1, 8), (2, 3), (1, 1)):
            lat = synth.uniform(seed, f"lat_{B}_{n}", (B, n, c["input_feat_dim"]), 1.0)
            mel = torch.nn.functional.interpolate(lat.transpose(1, 2), scale_factor=[4.0], mode="linear")
            wav = g(mel)
            out[f"{tag}_wav_{B}_{n}"] = wav.numpy()
            out[f"{tag}_mel_{B}_{n}"] = mel.numpy()[:, :8, :]
    np.savez_compressed(os.path.join(GOLD, "hifigan.npz"), **out)
    print("hifigan:", {k: v.shape for k, v in out.items() if "wav" in str(k)},
          "rms", float(np.sqrt((out["full_wav_1_8"] ** 2).mean())))


@torch.inference_mode()
def make_hubert(seed=17):
    """fairseq is absent: pin the HuBERT restatement to HuggingFace's HubertModel (same architecture, the class
    fairseq HuBERT/ContentVec checkpoints are converted to) loaded with the synthetic fairseq-named weights."""
    from transformers import HubertConfig, HubertModel
    out = dict(seed=seed)
    for tag, c in (("tiny", gcfg.TINY_HUBERT), ("full", gcfg.DEFAULT_HUBERT)):
        hc = HubertConfig(hidden_size=c["embed_dim"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                          intermediate_size=c["ffn_dim"], conv_dim=tuple(x[0] for x in c["conv_layers"]),
                          conv_kernel=tuple(x[1] for x in c["conv_layers"]), conv_stride=tuple(x[2] for x in c["conv_layers"]),
                          num_conv_pos_embeddings=c["pos_conv_kernel"], num_conv_pos_embedding_groups=c["pos_conv_groups"],
                          feat_extract_norm="group", do_stable_layer_norm=False, hidden_act="gelu", conv_bias=False,
                          apply_spec_augment=False, layer_norm_eps=1e-5)
        m = HubertModel(hc).eval()
        w = synth.make_weights(seed, synth.hubert_weight_spec(c))
        sd = {}
        for k, v in w.items():
            k2 = k
            if k.startswith("feature_extractor.conv_layers."):
                k2 = k.replace(".0.weight", ".conv.weight").replace(".2.weight", ".layer_norm.weight").replace(".2.bias", ".layer_norm.bias")
            elif k.startswith("layer_norm."):
                k2 = "feature_projection." + k
            elif k.startswith("post_extract_proj."):
                k2 = k.replace("post_extract_proj", "feature_projection.projection")
            elif k.startswith("encoder.pos_conv.0."):
                k2 = {"weight_g": "encoder.pos_conv_embed.conv.parametrizations.weight.original0",
                      "weight_v": "encoder.pos_conv_embed.conv.parametrizations.weight.original1",
                      "bias": "encoder.pos_conv_embed.conv.bias"}[k.rsplit(".", 1)[1]]
            elif k.startswith("encoder.layers."):
                k2 = (k.replace("self_attn_layer_norm", "layer_norm").replace("self_attn.", "attention.")
                       .replace("fc1.", "feed_forward.intermediate_dense.").replace("fc2.", "feed_forward.output_dense."))
            elif k.startswith("final_proj."):
                continue
            sd[k2] = v
        missing, unexpected = m.load_state_dict(sd, strict=False)
        assert not unexpected and all("masked_spec_embed" in x for x in missing), (missing, unexpected)
        for B, T in ((1, 16000), (2, 5120), (1, 24581)):
            wav = torch.cat([synth.synth_audio(seed + b, f"wav{T}", T) for b in range(B)], 0)
            h = m(wav).last_hidden_state
            feat = torch.nn.functional.linear(h, w["final_proj.weight"], w["final_proj.bias"])
            out[f"{tag}_feat_{B}_{T}"] = feat.numpy() if tag == "tiny" or T != 24581 else feat.numpy()[:, :, :64]
    np.savez_compressed(os.path.join(GOLD, "hubert.npz"), **out)
    print("hubert:", {k: v.shape for k, v in out.items() if "feat" in str(k)})


@torch.inference_mode()
def make_handle_chunks(seed=19):
    """reference inference/inference_utils.py:4-21 (imports with torch only): a streamed sequence of vocoder outputs through
    the reference's own handle_chunks -- first chunk (no overlap yet), regular cross-fades, a short last group (3 tokens),
    the `overlap_len > len(wav_chunk)` branch (a 1-token group: 1024 samples), and a chunk after that branch (overlap None)."""
    sys.path.insert(0, "/root/reference")
    from inference.inference_utils import handle_chunks as ref_handle_chunks
    lens = [8192, 8192, 3072, 8192, 1024, 8192, 2048]
    out = dict(seed=seed, lens=np.array(lens))
    prev, ov = None, None
    for i, n in enumerate(lens):
        wav = synth.uniform(seed, f"chunk{i}", (n,), 0.5)
        chunk, prev, ov = ref_handle_chunks(wav.clone(), prev, ov, 1024)
        out[f"chunk{i}"] = chunk.numpy().copy()
        out[f"has_overlap{i}"] = np.array(ov is not None)
        if ov is not None:
            out[f"overlap{i}"] = ov.numpy().copy()
    np.savez_compressed(os.path.join(GOLD, "handle_chunks.npz"), **out)
    print("handle_chunks:", [out[f"chunk{i}"].shape[0] for i in range(len(lens))])


def build_hf_hubert(c, w):
    """HuggingFace HubertModel (the class fairseq HuBERT / ContentVec checkpoints convert to) loaded with fairseq-named weights"""
    from transformers import HubertConfig, HubertModel
    hc = HubertConfig(hidden_size=c["embed_dim"], num_hidden_layers=c["layers"], num_attention_heads=c["heads"],
                      intermediate_size=c["ffn_dim"], conv_dim=tuple(x[0] for x in c["conv_layers"]),
                      conv_kernel=tuple(x[1] for x in c["conv_layers"]), conv_stride=tuple(x[2] for x in c["conv_layers"]),
                      num_conv_pos_embeddings=c["pos_conv_kernel"], num_conv_pos_embedding_groups=c["pos_conv_groups"],
                      feat_extract_norm="group", do_stable_layer_norm=False, hidden_act="gelu", conv_bias=False,
                      apply_spec_augment=False, layer_norm_eps=1e-5)
    m = HubertModel(hc).eval()
    sd = {}
    for k, v in w.items():
        k2 = k
        if k.startswith("feature_extractor.conv_layers."):
            k2 = k.replace(".0.weight", ".conv.weight").replace(".2.weight", ".layer_norm.weight").replace(".2.bias", ".layer_norm.bias")
        elif k.startswith("layer_norm."):
            k2 = "feature_projection." + k
        elif k.startswith("post_extract_proj."):
            k2 = k.replace("post_extract_proj", "feature_projection.projection")
        elif k.startswith("encoder.pos_conv.0."):
            k2 = {"weight_g": "encoder.pos_conv_embed.conv.parametrizations.weight.original0",
                  "weight_v": "encoder.pos_conv_embed.conv.parametrizations.weight.original1",
                  "bias": "encoder.pos_conv_embed.conv.bias"}[k.rsplit(".", 1)[1]]
        elif k.startswith("encoder.layers."):
            k2 = (k.replace("self_attn_layer_norm", "layer_norm").replace("self_attn.", "attention.")
                   .replace("fc1.", "feed_forward.intermediate_dense.").replace("fc2.", "feed_forward.output_dense."))
        elif k.startswith("final_proj."):
            continue
        sd[k2] = v
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert not unexpected and all("masked_spec_embed" in x for x in missing), (missing, unexpected)
    return m


@torch.inference_mode()
def make_chain(GPT, DiscreteVAE, seed=1, src_seed=402, ref_seed=100, n_chunks=3, n_steps=24, group=8):
    """The headline chain of bench.py -- inference_utils.synthesize_utt_streaming(seg_len=1.0, stream_chunk_size=8), reference
    inference/inference_utils.py:135-217 -- composed from the REFERENCE's own classes at full size (GenVC_small dims, the synthetic weights
    model_init_synthetic(seed=1) loads): GPT.get_style_emb (Perceiver) on the mel of the 3 s reference, per 1 s chunk HubertModel (HF: fairseq is
    absent) + final_proj -> DiscreteVAE.get_codebook_indices -> GPT.compute_embeddings + GPT2InferenceModel driven step by step with HF's logits
    processors (stream_generator.py cannot be imported: see ref_greedy) for a 24-token budget -> groups of 8 latents -> F.interpolate(x4, linear)
    + HiFiGAN -> the reference's handle_chunks.  The mel is the oracle's (torchaudio absent: parity unpinned for that stage).
    Source seed margin-screened (tests/chain_oracle.py); the screen is stored."""
    from oracle import genvc_oracle as O
    from genvc_amd.utils import DEFAULT_MEL_NORM_FILE, load_mel_norms
    stub = types.ModuleType("nnAudio"); stub.features = types.ModuleType("nnAudio.features")
    sys.modules["nnAudio"], sys.modules["nnAudio.features"] = stub, stub.features
    from layers.hifigan import HiFiGAN
    from inference.inference_utils import handle_chunks as ref_handle_chunks
    cfg = gcfg.default_config()
    dims = gcfg.gpt_dims(cfg.model_args)
    g = build_ref_gpt(GPT, cfg.model_args, synth.make_weights(seed, synth.gpt_weight_spec(dims)))
    c = cfg.content_dvae_config
    dv = DiscreteVAE(channels=c["num_channels"], normalization=None, positional_dims=1, num_tokens=c["num_tokens"],
                     codebook_dim=c["codebook_dim"], hidden_dim=c["hidden_dim"], num_resnet_blocks=c["num_resnet_blocks"],
                     kernel_size=c["kernel_size"], num_layers=c["num_layers"], use_transposed_convs=False)
    missing, unexpected = dv.load_state_dict(synth.make_weights(seed, synth.dvae_weight_spec(c)), strict=False)
    assert not unexpected
    dv.eval()
    hcfg = dict(cfg.hubert_config)
    hw = synth.make_weights(seed, synth.hubert_weight_spec(hcfg))
    hub = build_hf_hubert(hcfg, hw)
    v = cfg.vocoder_config
    voc = HiFiGAN(v["input_feat_dim"], v["upsample_initial_channel"], v["resblock_kernel_sizes"], v["resblock_dilation_sizes"],
                  v["upsample_rates"], v["upsample_kernel_sizes"], resblock_type="2")
    voc.load_state_dict(synth.make_weights(seed, synth.hifigan_weight_spec(v)), strict=True)
    voc.eval()
    samp = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
    src = synth.synth_audio(src_seed, "src", n_chunks * 16000)
    ref = synth.synth_audio(ref_seed, "ref", 72000)
    norms = torch.from_numpy(load_mel_norms(DEFAULT_MEL_NORM_FILE))
    cond = g.get_style_emb(O.mel_spectrogram(ref, norms), None).transpose(1, 2)          # one 3 s chunk: the mean over chunks is itself
    toks, lats, codes_all, margins, pred = [], [], [], [], []
    prev = ov = None
    for ci in range(n_chunks):
        seg = src[:, ci * 16000:(ci + 1) * 16000]
        feat = torch.nn.functional.linear(hub(seg).last_hidden_state, hw["final_proj.weight"], hw["final_proj.bias"])
        codes = dv.get_codebook_indices(feat.transpose(1, 2))
        r = ref_greedy(g, cond, codes, n_steps, samp)
        assert r["tokens"].shape[1] == n_steps, "the reference stopped early: pick another seed"
        codes_all.append(codes.numpy()); toks.append(r["tokens"]); lats.append(r["latents"]); margins.append(r["margins"])
        for g0 in range(0, n_steps, group):
            lat = torch.from_numpy(r["latents"][:, g0:g0 + group])
            mel_in = torch.nn.functional.interpolate(lat.transpose(1, 2), scale_factor=[4.0], mode="linear")
            wav = voc(mel_in).squeeze()
            chunk, prev, ov = ref_handle_chunks(wav.clone(), prev, ov, 1024)
            pred.append(chunk.numpy().copy())
    wav = np.concatenate(pred)
    tok = np.concatenate(toks, 1)
    lat = np.concatenate(lats, 1)
    mg = np.concatenate(margins, 1)
    np.savez_compressed(os.path.join(GOLD, "chain_full.npz"), seed=seed, src_seed=src_seed, ref_seed=ref_seed, n_steps=n_steps, group=group,
                        codes=np.concatenate(codes_all, 0), tokens=tok, margins=mg, latents_slice=lat[:, :, :32],
                        cond_slice=cond.numpy()[:, :, :16], wav_len=wav.shape[0], wav_head=wav[:4096], wav_stride16=wav[::16])
    print(f"chain_full: {tok.shape[1]} tokens in {n_chunks} chunks, min margin {mg.min():.3e}, wav {wav.shape[0]} samples, rms {np.sqrt((wav ** 2).mean()):.4f}")


# ---------------------------------------------------------------------------
# round 6: the reference's OWN generation loop and harness functions, executed (not restated)
# ---------------------------------------------------------------------------

def import_stream_generator():
    """/root/reference/layers/stream_generator.py imports four beam-search names that transformers 5 removed (:13-23) and
    `SampleOutput` (:24).  None of them is reached with num_beams = 1: empty stand-in classes on the `transformers` module let the
    file import, and `NewGenerationMixin.sample_stream` (:645-881) -- the loop GPT.get_generator / GPT.generate end in -- runs as is."""
    import transformers
    import transformers.generation.utils as gu
    for n in ("BeamSearchScorer", "ConstrainedBeamSearchScorer", "DisjunctiveConstraint", "PhrasalConstraint"):
        if not hasattr(transformers, n):
            setattr(transformers, n, type(n, (), {}))
    if not hasattr(gu, "SampleOutput"):
        gu.SampleOutput = object
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from layers import stream_generator as SG
    return SG


class _StoppingCriteria433(list):
    """transformers 4.33 StoppingCriteriaList.__call__: `any(criteria(input_ids, scores) for criteria in self)`, ONE bool for the
    batch (transformers 5 returns a per-row tensor, which `if ... or stopping_criteria(...)` at stream_generator.py:877 cannot take)."""
    def __call__(self, input_ids, scores, **kw):
        return any(bool(torch.as_tensor(c(input_ids, scores)).all()) for c in self)


class _MarginTap:
    """a pass-through logits processor behind the repetition penalty: records the top-1 / top-2 gap of the scores the draw sees"""
    def __init__(self):
        self.gaps = []

    def __call__(self, input_ids, scores):
        t2 = torch.topk(scores, 2, dim=-1)[0]
        self.gaps.append((t2[:, 0] - t2[:, 1]).numpy().copy())
        return scores


def arm_reference_generation(g, SG):
    """Give the reference GPT2InferenceModel what `PreTrainedModel.generate_stream = NewGenerationMixin.generate`
    (stream_generator.py:884-886) gave it under transformers 4.33.  The dispatcher `NewGenerationMixin.generate` (:46-640) is 600 lines
    against 4.33's private GenerationMixin API (`_validate_model_class`, `_get_logits_warper`, ...: gone in 5.x), so the part of it the
    reference's call reaches (do_sample=True, num_beams=1: :186-193 attention mask of ones since pad == eos, :181-182 use_cache,
    :337-347 `_get_logits_processor` -> RepetitionPenalty, :437 `_get_logits_warper` -> Temperature, TopK, TopP in that order with
    min_tokens_to_keep = 1, :345 MaxLengthCriteria(max_length), :449-461 the call) is this shim; the LOOP it calls is the reference's
    own `sample_stream`, unmodified, bound to the reference model.  `generate` (non-streaming; 4.33's `GenerationMixin.sample`, the
    function sample_stream was adapted from: same body with `yield` replaced by accumulation) drains the same loop and returns
    cat(input_ids, tokens) like `sample` does."""
    import types as _t
    from transformers import GenerationConfig, GenerationMixin, LogitsProcessorList
    from transformers.generation.logits_process import (
        RepetitionPenaltyLogitsProcessor, TemperatureLogitsWarper, TopKLogitsWarper, TopPLogitsWarper)
    from transformers.generation.stopping_criteria import MaxLengthCriteria
    gi = g.gpt_inference
    gi.generation_config = GenerationConfig()
    gi._update_model_kwargs_for_generation = _t.MethodType(GenerationMixin._update_model_kwargs_for_generation, gi)
    tap = _MarginTap()

    def stream(inputs, bos_token_id=None, pad_token_id=None, eos_token_id=None, max_length=None, do_stream=False,
               do_sample=True, top_p=1.0, top_k=50, temperature=1.0, length_penalty=1.0, repetition_penalty=1.0, num_beams=1,
               num_return_sequences=1, output_attentions=False, output_hidden_states=False):
        assert do_sample and num_beams == 1 and num_return_sequences == 1
        procs = LogitsProcessorList(([RepetitionPenaltyLogitsProcessor(repetition_penalty)] if repetition_penalty != 1.0 else []) + [tap])
        warp = LogitsProcessorList(
            ([TemperatureLogitsWarper(temperature)] if temperature != 1.0 else [])
            + ([TopKLogitsWarper(top_k, min_tokens_to_keep=1)] if top_k else [])
            + ([TopPLogitsWarper(top_p, min_tokens_to_keep=1)] if top_p < 1.0 else []))
        return SG.NewGenerationMixin.sample_stream(
            gi, inputs, logits_processor=procs, logits_warper=warp,
            stopping_criteria=_StoppingCriteria433([MaxLengthCriteria(max_length=max_length)]),
            pad_token_id=pad_token_id, eos_token_id=eos_token_id, output_attentions=output_attentions,
            output_hidden_states=True, attention_mask=torch.ones_like(inputs), use_cache=True)

    def generate(inputs, **kw):
        toks = [t for t, _ in stream(inputs, **kw)]
        return torch.cat([inputs, torch.stack(toks, 1)], dim=-1)

    gi.generate_stream = stream
    gi.generate = generate
    return tap


@torch.inference_mode()
def make_stream_loop(GPT, seed=29):
    """tests/golden/stream_loop.npz: `NewGenerationMixin.sample_stream` itself (stream_generator.py:645-881) driven through the reference's
    `GPT.get_generator` (gpt.py:610-621) on the reference GPT2InferenceModel: (a) B = 3 with a raised stop-token bias -- rows end at
    different steps, finished rows yield the pad, the loop ends with the last row, the EOS-step pair is yielded; (b) a run that ends on
    `max_length` (`max_gen_mel_tokens` lowered on the reference object); (c) `GPT.generate` (gpt.py:594-608) for the same inputs as (b)."""
    SG = import_stream_generator()
    model_args = gcfg.TINY_MODEL_ARGS
    dims = gcfg.gpt_dims(model_args)
    w = synth.make_weights(seed, synth.gpt_weight_spec(dims))
    kw = dict(top_p=0.85, top_k=1, temperature=0.85, length_penalty=1.0, repetition_penalty=2.0, do_sample=True, num_beams=1,
              num_return_sequences=1, output_attentions=False, output_hidden_states=True)

    def run(g, tap, cond, codes):
        tap.gaps.clear()
        fake = g.compute_embeddings(cond, codes)
        pairs = list(g.get_generator(fake_inputs=fake, **kw))
        toks = torch.stack([p[0] for p in pairs], 1)
        lats = torch.stack([p[1] for p in pairs], 1)
        return toks.numpy(), lats.numpy(), np.stack(tap.gaps, 1)

    out = dict(seed=seed)
    # (a) ragged EOS at B = 3
    found = False
    for bias in np.arange(0.2, 3.0, 0.05):
        w["mel_head.bias"][1025] = float(bias)
        g = build_ref_gpt(GPT, model_args, w)
        tap = arm_reference_generation(g, SG)
        for in_seed in range(seed * 100, seed * 100 + 6):
            cond, codes = gpt_inputs(in_seed, dims, 3, 11)
            toks, lats, gaps = run(g, tap, cond, codes)
            ends = [int((toks[b] == 1025).argmax()) if (toks[b] == 1025).any() else -1 for b in range(3)]
            live = np.concatenate([gaps[b, :ends[b] + 1] for b in range(3)]) if min(ends) >= 0 else np.zeros(1)
            if min(ends) >= 3 and len(set(ends)) == 3 and toks.shape[1] < 60 and live.min() > 2e-3:
                found = True
                break
        if found:
            break
    assert found, "no ragged-EOS case found"
    out.update(eos_bias=float(bias), eos_in_seed=in_seed, eos_tokens=toks, eos_latents_slice=lats[:, :, :32], eos_margins=gaps,
               eos_ends=np.array(ends))
    print(f"stream_loop (a): bias {bias:.2f}, in_seed {in_seed}, ends {ends}, {toks.shape[1]} yields, min live margin {live.min():.2e}")
    # (b) + (c): ends on max_length
    w["mel_head.bias"][1025] = 0.0
    g = build_ref_gpt(GPT, model_args, w)
    tap = arm_reference_generation(g, SG)
    g.max_gen_mel_tokens = 21
    for in_seed in range(seed * 100 + 50, seed * 100 + 80):
        cond, codes = gpt_inputs(in_seed, dims, 2, 13)
        toks, lats, gaps = run(g, tap, cond, codes)
        if gaps.min() > 2e-3 and not (toks == 1025).any():
            break
    else:
        raise RuntimeError("no max_length case passed the margin screen")
    gen = g.generate(cond, codes, **{k: v for k, v in kw.items() if k not in ("num_return_sequences", "output_hidden_states")})
    assert toks.shape[1] == 21 and torch.equal(gen, torch.from_numpy(toks))
    out.update(max_in_seed=in_seed, max_new=21, max_tokens=toks, max_latents_slice=lats[:, :, :32], max_margins=gaps,
               generate_tokens=gen.numpy())
    print(f"stream_loop (b): in_seed {in_seed}, {toks.shape[1]} yields (max_length), min margin {gaps.min():.2e}; generate() equal")
    np.savez_compressed(os.path.join(GOLD, "stream_loop.npz"), **out)


class _Duck:
    """attribute bag standing in for the reference's HiFiGANTrainer object"""
    def __init__(self, **kw):
        self.__dict__.update(kw)


def build_reference_model(GPT, DiscreteVAE, cfg, seed, max_new, stop_bias=None):
    """A model object with the attributes the reference's harness functions touch (inference/inference_utils.py:23-217), every one of
    them the REFERENCE's own class on the synthetic weights model_init_synthetic(cfg, seed) loads: `.gpt` = reference GPT (with the
    reference's sample_stream behind get_generator / generate), `.content_dvae` = reference DiscreteVAE, `.hifigan` = reference HiFiGAN,
    `.get_gpt_cond_latents` = trainers/hifigan_trainer.py:438-455 restated on the reference's `GPT.get_style_emb` (the trainer class
    pulls the whole training stack in) over the oracle's mel (torchaudio absent), `.content_extractor` = HuggingFace HubertModel +
    final_proj (fairseq absent); a segment with all-zero frames -- the zero-padded tail, content_processor.py:24 -- goes through the
    oracle's masked forward, the one stage nothing here can pin."""
    from oracle import genvc_oracle as O
    from genvc_amd.utils import DEFAULT_MEL_NORM_FILE, load_mel_norms
    stub = types.ModuleType("nnAudio"); stub.features = types.ModuleType("nnAudio.features")
    sys.modules["nnAudio"], sys.modules["nnAudio.features"] = stub, stub.features
    from layers.hifigan import HiFiGAN
    SG = import_stream_generator()
    dims = gcfg.gpt_dims(cfg.model_args)
    gw = synth.make_weights(seed, synth.gpt_weight_spec(dims))
    if stop_bias is not None:
        gw["mel_head.bias"][1025] = float(stop_bias)
    g = build_ref_gpt(GPT, cfg.model_args, gw)
    tap = arm_reference_generation(g, SG)
    g.max_gen_mel_tokens = max_new
    c = cfg.content_dvae_config
    dv = DiscreteVAE(channels=c["num_channels"], normalization=None, positional_dims=1, num_tokens=c["num_tokens"],
                     codebook_dim=c["codebook_dim"], hidden_dim=c["hidden_dim"], num_resnet_blocks=c["num_resnet_blocks"],
                     kernel_size=c["kernel_size"], num_layers=c["num_layers"], use_transposed_convs=False)
    missing, unexpected = dv.load_state_dict(synth.make_weights(seed, synth.dvae_weight_spec(c)), strict=False)
    assert not unexpected
    dv.eval()
    hcfg = dict(cfg.hubert_config)
    hw = synth.make_weights(seed, synth.hubert_weight_spec(hcfg))
    hub = build_hf_hubert(hcfg, hw)
    v = cfg.vocoder_config
    voc = HiFiGAN(v["input_feat_dim"], v["upsample_initial_channel"], v["resblock_kernel_sizes"], v["resblock_dilation_sizes"],
                  v["upsample_rates"], v["upsample_kernel_sizes"], resblock_type="2")
    voc.load_state_dict(synth.make_weights(seed, synth.hifigan_weight_spec(v)), strict=True)
    voc.eval()
    norms = torch.from_numpy(load_mel_norms(DEFAULT_MEL_NORM_FILE))
    log = dict(codes=[], masked_segments=0)

    def extract_content_features(wav):
        x = hub(wav).last_hidden_state
        n_frames = x.shape[1]
        if bool(O.hubert_frame_padding_mask(wav, n_frames).any()):
            log["masked_segments"] += 1
            return O.hubert_extract_features(hw, hcfg, wav)
        return torch.nn.functional.linear(x, hw["final_proj.weight"], hw["final_proj.bias"])

    def get_codebook_indices(feat):
        codes = dv.get_codebook_indices(feat)
        log["codes"].append(codes.numpy().copy())
        return codes

    def get_gpt_cond_latents(audio, sr, length=30, chunk_length=6):
        if sr != 24000:
            raise NotImplementedError
        audio = audio[:, :24000 * length]
        embs = []
        for i in range(0, audio.shape[1], 24000 * chunk_length):
            chunk = audio[:, i:i + 24000 * chunk_length]
            if chunk.size(-1) < 24000 * 0.33:
                continue
            embs.append(g.get_style_emb(O.mel_spectrogram(chunk, norms), None))
        return torch.stack(embs).mean(dim=0).transpose(1, 2)

    sampling = dict(gcfg.DEFAULT_SAMPLING, top_k=1)
    m = _Duck(gpt=g, content_dvae=_Duck(get_codebook_indices=get_codebook_indices), hifigan=voc,
              content_extractor=_Duck(extract_content_features=extract_content_features),
              get_gpt_cond_latents=get_gpt_cond_latents, content_sample_rate=16000, device=torch.device("cpu"),
              hifigan_scale_factor=4.0,
              config=_Duck(audio=_Duck(sample_rate=24000), model_args=_Duck(gpt_code_stride_len=1024), **sampling))
    return m, tap, log


@torch.inference_mode()
def make_harness(GPT, DiscreteVAE, only=("tiny", "full")):
    """tests/golden/harness_{tiny,full}.npz: the reference's UNCHANGED `synthesize_utt_streaming(seg_len=1.0, stream_chunk_size=8)`
    (inference/inference_utils.py:135-217) and `synthesize_utt(seg_len=1.0)` (:23-89) called on `build_reference_model`.  Source 2.2 s
    = 1 s + 1 s + a 0.2 s tail the harness zero-pads to 0.32 s (:43-50).  Two endings: `max_length` (26 yields: groups 8, 8, 8, 2) and a
    raised stop bias (the EOS-step pair is streamed, :189-196; `synthesize_utt` strips the stop token, :68).  The streamed (token, latent)
    pairs are taken from a tap on the generator, the group boundaries from the lengths the reference's `hifigan.forward` is called with."""
    import contextlib
    import io
    sys.path.insert(0, "/root/reference")
    from inference import inference_utils as RIU
    for tag in only:
        tiny = tag == "tiny"
        cfg = gcfg.default_config(tiny=tiny)
        seed = 3 if tiny else 1
        out = dict(seed=seed, max_new=26)
        ref = synth.synth_audio(100, "ref", 72000)
        cases = [("max", None)] + ([("eos", "search")] if tiny else [])
        for case, bias in cases:
            biases = [None] if bias is None else list(np.arange(0.3, 3.0, 0.05))
            done = False
            for b in biases:
                m, tap, log = build_reference_model(GPT, DiscreteVAE, cfg, seed, 26, stop_bias=b)
                vo_lens, pairs = [], []
                voc_forward = m.hifigan.forward

                def forward(x, _f=voc_forward):
                    vo_lens.append(int(x.shape[-1]) // 4)
                    return _f(x)
                m.hifigan.forward = forward
                get_gen = m.gpt.get_generator

                def tapped(*a, _g=get_gen, **k):
                    for t, lat in _g(*a, **k):
                        pairs.append((t.numpy().copy(), lat.numpy().copy()))
                        yield t, lat
                m.gpt.get_generator = tapped
                for src_seed in range(500, 500 + (12 if tiny else 4)):
                    src = synth.synth_audio(src_seed, "src", 35200)
                    tap.gaps.clear(); vo_lens.clear(); pairs.clear(); log["codes"].clear(); log["masked_segments"] = 0
                    try:
                        with contextlib.redirect_stdout(io.StringIO()):
                            wav_s = RIU.synthesize_utt_streaming(m, src, ref, seg_len=1.0, stream_chunk_size=8)
                    except (RuntimeError, ValueError) as e:   # a segment whose yields are a multiple of 8: torch.cat([]) at :196 raises
                        print(f"  harness_{tag}/{case}: src seed {src_seed} bias {b}: reference raised {str(e)[:60]!r}")
                        continue
                    toks = np.concatenate([p[0] for p in pairs])
                    gaps = np.concatenate(tap.gaps)
                    n_stream = len(pairs)
                    ok = gaps[:n_stream].min() > 2e-3
                    if case == "eos":
                        per_seg = np.split(toks, np.nonzero(toks == 1025)[0] + 1)[:-1]
                        ok = ok and len(per_seg) == 3 and all(3 <= len(s) < 26 for s in per_seg) and int((toks == 1025).sum()) == 3
                    if ok:
                        done = True
                        break
                if done:
                    break
            assert done, f"harness_{tag}/{case}: nothing passed the screen"
            lats = np.stack([p[1][0] for p in pairs])
            codes_stream = [c.copy() for c in log["codes"]]
            masked = log["masked_segments"]
            # non-streaming on the same model and source
            tap.gaps.clear(); log["codes"].clear()
            m.gpt.get_generator = get_gen
            lat_calls = []

            def forward2(x, _f=voc_forward):
                lat_calls.append(int(x.shape[-1]) // 4)
                return _f(x)
            m.hifigan.forward = forward2
            wav_n = RIU.synthesize_utt(m, src, ref, seg_len=1.0)
            gaps_n = np.concatenate(tap.gaps)
            p = f"{case}_"
            out.update({p + "src_seed": src_seed, p + "stop_bias": -1.0 if b is None else float(b), p + "tokens": toks.reshape(-1),
                        p + "groups": np.array(vo_lens), p + "latents_slice": lats[:, :32], p + "margin": float(gaps[:n_stream].min()),
                        p + "codes": np.concatenate([c.reshape(-1) for c in codes_stream]),
                        p + "code_lens": np.array([c.size for c in codes_stream]), p + "masked_segments": masked,
                        p + "wav_len": wav_s.shape[0], p + "wav_head": wav_s.numpy()[:4096], p + "wav_stride8": wav_s.numpy()[::8],
                        p + "ns_latent_rows": np.array(lat_calls), p + "ns_wav_len": wav_n.shape[0], p + "ns_margin": float(gaps_n.min()),
                        p + "ns_wav_head": wav_n.numpy()[:4096], p + "ns_wav_stride8": wav_n.numpy()[::8]})
            print(f"harness_{tag}/{case}: src seed {src_seed} bias {b}: {n_stream} streamed pairs, groups {vo_lens}, margin "
                  f"{gaps[:n_stream].min():.2e}, masked segments {masked}, wav {wav_s.shape[0]}; non-streaming rows {lat_calls} wav {wav_n.shape[0]} "
                  f"margin {gaps_n.min():.2e}")
        np.savez_compressed(os.path.join(GOLD, f"harness_{tag}.npz"), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    os.makedirs(GOLD, exist_ok=True)
    torch.manual_seed(0)
    GPT, DiscreteVAE = import_reference()
    want = lambda k: not args.only or k in args.only.split(",")
    g_tiny = g_full = None
    if want("tiny") or want("perceiver"):
        g_tiny, _, _ = make_gpt(GPT, "tiny", gcfg.TINY_MODEL_ARGS, seed=3, B=2, Tc=13, n_steps=48,
                                keep_rows=[0, 1, 2, 3, 10, 24])
    if want("tiny"):
        make_gpt(GPT, "tiny_b1", gcfg.TINY_MODEL_ARGS, seed=4, B=1, Tc=75, n_steps=64, keep_rows=[0, 1, 2, 40])
        make_eos(GPT)
    if want("full") or want("perceiver"):
        g_full, _, _ = make_gpt(GPT, "full", gcfg.DEFAULT_MODEL_ARGS, seed=1, B=1, Tc=13, n_steps=48,
                                keep_rows=[0, 1, 2, 23])
    if want("full"):
        make_gpt(GPT, "full_6s", gcfg.DEFAULT_MODEL_ARGS, seed=2, B=2, Tc=75, n_steps=32, keep_rows=[0, 1, 16])
    if want("full141"):
        # the CLI default (seg_len 6 s: Tc = 75, P = 109, 141 steps at 23.4375 tokens/s): one stream whose context grows from
        # 110 to 251 cached positions, across the 128-key boundary where the launch-per-phase step changes its attention variant
        make_gpt(GPT, "full_6s_b1", gcfg.DEFAULT_MODEL_ARGS, seed=2, B=1, Tc=75, n_steps=141, keep_rows=[0, 1, 2, 17, 18, 19, 64, 128, 140])
    if want("full94"):
        # the OTHER segment class of a 10 s utterance at seg_len 6 s (BASELINE configs[2]): the 4 s tail, Tc = 50, P = 84, 94 steps;
        # same weights as full_6s_b1, so one context decodes both classes together (parallel_offline / GPT.generate_groups)
        make_gpt(GPT, "full_4s_b1", gcfg.DEFAULT_MODEL_ARGS, seed=2, B=1, Tc=50, n_steps=94, keep_rows=[0, 1, 2, 46, 93])
    if want("perceiver"):
        make_perceiver(g_tiny, g_full)
    if want("dvae"):
        make_dvae(DiscreteVAE)
    if want("sampler"):
        make_sampler()
    if want("hifigan"):
        make_hifigan()
    if want("hubert"):
        make_hubert()
    if want("chunks"):
        make_handle_chunks()
    if want("chain"):
        make_chain(GPT, DiscreteVAE)
    if want("loop"):
        make_stream_loop(GPT)
    if want("harness"):
        make_harness(GPT, DiscreteVAE)
    if want("harness_tiny"):
        make_harness(GPT, DiscreteVAE, only=("tiny",))


if __name__ == "__main__":
    main()
