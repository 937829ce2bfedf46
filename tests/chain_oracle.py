"""Test infrastructure: the ORACLE driven through the streaming harness (reference inference/inference_utils.py:135-217) at any
model size, returning next to tokens / latents / waveform the decision margins of the run -- the smallest top-1 / top-2 gap of the
penalised greedy scores and the smallest nearest / second-nearest codebook gap of the content tokeniser -- so that a parity test can
tell a real divergence from a near-tie flipped by float rounding.

    python tests/chain_oracle.py [n_seeds]        screens source seeds on the CPU (full-size synthetic GenVC_small weights) and prints
                                                  their margins: the full-size chain test uses a seed whose margins are comfortable
                                                  (the same screen oracle/make_golden.py applies to the reference fixtures)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from genvc_amd import config as gcfg          # noqa: E402
from genvc_amd import synth                   # noqa: E402
from oracle import genvc_oracle as O          # noqa: E402


def synthetic_bundle(config, seed, max_new, device="cpu"):
    """the weights model_init_synthetic(config, seed) loads, as the oracle's state dicts (synth is a counter hash: the same on every
    device)"""
    from genvc_amd.utils import DEFAULT_MEL_NORM_FILE, load_mel_norms
    dims = gcfg.gpt_dims(config.model_args)
    hcfg = dict(config.get("hubert_config") or gcfg.DEFAULT_HUBERT)
    return dict(gpt=synth.make_weights(seed, synth.gpt_weight_spec(dims), device=device),
                dvae=synth.make_weights(seed, synth.dvae_weight_spec(config.content_dvae_config), device=device),
                hubert=synth.make_weights(seed, synth.hubert_weight_spec(hcfg), device=device), hubert_cfg=hcfg,
                hifigan=synth.make_weights(seed, synth.hifigan_weight_spec(config.vocoder_config), device=device),
                vocoder_cfg=dict(config.vocoder_config), mel_norms=torch.from_numpy(load_mel_norms(DEFAULT_MEL_NORM_FILE)),
                dims=dims, sampling=dict(gcfg.DEFAULT_SAMPLING, top_k=1), max_new=max_new)


def vq_margin(W, feat):
    """codes of the content tokeniser and the gap between the nearest and the second-nearest codebook entry per frame
    (reference layers/dvae.py:87-93)"""
    x = O.dvae_encode(W["dvae"], feat.transpose(1, 2))
    embed = W["dvae"]["codebook.embed"]
    flat = x.reshape(-1, x.shape[-1])
    dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ embed + embed.pow(2).sum(0, keepdim=True)
    top = (-dist).topk(2, dim=1)
    return top.indices[:, 0].view(*x.shape[:-1]), (top.values[:, 0] - top.values[:, 1]).view(*x.shape[:-1])


def streaming_chain(W, src_wav, tgt_audio, seg_len=1.0, stream_chunk_size=8):
    """O.synthesize_utt_streaming with the per-stage intermediates and margins kept: dict(cond, feats, codes, tokens (per group),
    latents (per group), wav, token_margin, vq_margin)"""
    cond = O.get_gpt_cond_latents(W["gpt"], tgt_audio, W["mel_norms"])
    dims = W["dims"]
    overlap = None
    out = dict(cond=cond, feats=[], codes=[], tokens=[], latents=[], token_margins=[], vq_margins=[])
    pred = []
    for seg in O._segments(src_wav, seg_len):
        feat = O.hubert_extract_features(W["hubert"], W["hubert_cfg"], seg)
        codes, vm = vq_margin(W, feat)
        assert torch.equal(codes, O.dvae_get_codebook_indices(W["dvae"], feat.transpose(1, 2)))
        toks, lats, logits = O.generate(W["gpt"], dims, cond, codes, W["sampling"], max_new=W.get("max_new"))
        n = toks.shape[1]
        P = 32 + codes.shape[1] + 2
        fake = torch.cat([torch.ones(1, P, dtype=torch.long), torch.full((1, 1), dims["start_audio_token"])], 1)
        s = W["sampling"]
        for i in range(n):
            pen = O.process_logits(logits[i], torch.cat([fake, toks[:, :i]], 1), s["repetition_penalty"], 1.0, 0, 1.0)
            t2 = pen.topk(2, -1)[0]
            out["token_margins"].append(float(t2[0, 0] - t2[0, 1]))
        out["feats"].append(feat); out["codes"].append(codes); out["vq_margins"].append(vm)
        for g0 in range(0, n, stream_chunk_size):
            lat = lats[:, g0:g0 + stream_chunk_size]
            out["tokens"].append(toks[:, g0:g0 + stream_chunk_size]); out["latents"].append(lat)
            wav = O.vocode_latents(W["hifigan"], W["vocoder_cfg"], lat).squeeze()
            chunk, overlap = O.handle_chunks(wav, overlap)
            pred.append(chunk)
    out["wav"] = torch.cat(pred, -1)
    out["token_margin"] = min(out["token_margins"])
    out["vq_margin"] = float(torch.cat([v.reshape(-1) for v in out["vq_margins"]]).min())
    return out


if __name__ == "__main__":
    import time
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    torch.set_num_threads(8)
    cfg = gcfg.default_config()
    W = synthetic_bundle(cfg, 1, 24)
    ref = synth.synth_audio(100, "ref", 72000)
    for seed in range(400, 400 + n):
        t0 = time.time()
        src = synth.synth_audio(seed, "src", 48000)
        r = streaming_chain(W, src, ref, 1.0, 8)
        print(f"src seed {seed}: token margin {r['token_margin']:.2e}  vq margin {r['vq_margin']:.2e}  "
              f"tokens {[int(t.shape[1]) for t in r['tokens']]}  {time.time() - t0:.1f} s", flush=True)
