"""Deterministic synthetic weights and inputs for the GenVC hot path.

No checkpoint ships with the reference, so every test,
golden fixture and benchmark derives its tensors from a counter-based integer
hash.  The hash runs in torch int64 arithmetic, which wraps identically on the
CPU and on the GPU, and the int -> float conversion is exact, so the container
that generates the golden fixtures and the MI355X box that checks them hold
bit-identical weights without any file travelling between them.

Tensor names and shapes follow the reference state dict
(/root/reference/layers/gpt.py:141-188, perceiver_encoder.py:241-263,
dvae.py:252-295); see SURVEY.md section 8(b)(ii).
"""
import math
import zlib

import torch

_M1 = -7046029254386353131          # 0x9E3779B97F4A7C15 as int64
_M2 = -4658895280553007687          # 0xBF58476D1CE4E5B9
_M3 = -7723592293110705685          # 0x94D049BB133111EB
_CHUNK = 1 << 22


def _lsr(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _mix(x):
    # splitmix64 finaliser; int64 multiply wraps modulo 2^64 on CPU and GPU.
    x = (x ^ _lsr(x, 30)) * _M2
    x = (x ^ _lsr(x, 27)) * _M3
    return x ^ _lsr(x, 31)


def name_key(seed, name):
    k = (zlib.crc32(name.encode()) << 20) ^ (seed * 0x9E3779B1)
    return k & 0x7FFFFFFFFFFFFFFF


def uniform(seed, name, shape, scale=1.0, mean=0.0, device="cpu"):
    """float32 tensor, i.i.d. uniform with standard deviation `scale` around `mean`."""
    n = 1
    for s in shape:
        n *= int(s)
    out = torch.empty(n, dtype=torch.float32, device=device)
    key = name_key(seed, name)
    amp = float(scale) * math.sqrt(12.0)
    for lo in range(0, n, _CHUNK):
        hi = min(n, lo + _CHUNK)
        i = torch.arange(lo, hi, dtype=torch.int64, device=device)
        h = _mix(i * _M1 + key)
        u = _lsr(h, 41).to(torch.float32)          # 23 bits, exact in fp32
        out[lo:hi] = ((u + 0.5) * (2.0 ** -23) - 0.5) * amp
    out = out.reshape(tuple(shape))
    if mean != 0.0:
        out = out + mean
    return out


def integers(seed, name, shape, high, device="cpu"):
    """int64 tensor of i.i.d. integers in [0, high)."""
    n = 1
    for s in shape:
        n *= int(s)
    i = torch.arange(n, dtype=torch.int64, device=device)
    h = _lsr(_mix(i * _M1 + name_key(seed, name)), 33)
    return (h % high).reshape(tuple(shape))


# --------------------------------------------------------------------------
# weight specifications (name -> shape, kind)
# --------------------------------------------------------------------------

def gpt_weight_spec(dims):
    """GPT + conditioning Perceiver tensors, named as in the reference `GPT` state dict."""
    d, L = dims["d_model"], dims["n_layer"]
    V, Vt = dims["num_audio_tokens"], dims["number_text_tokens"]
    spec = {
        "text_embedding.weight": ((Vt, d), "emb"),
        "mel_embedding.weight": ((V, d), "emb"),
        "mel_pos_embedding.emb.weight": ((dims["max_mel_pos"], d), "emb"),
        "text_pos_embedding.emb.weight": ((dims["max_text_pos"], d), "emb"),
        "gpt.ln_f.weight": ((d,), "ln_w"), "gpt.ln_f.bias": ((d,), "ln_b"),
        "final_norm.weight": ((d,), "ln_w"), "final_norm.bias": ((d,), "ln_b"),
        "mel_head.weight": ((V, d), "head"), "mel_head.bias": ((V,), "bias"),
        "text_head.weight": ((Vt, d), "mat"), "text_head.bias": ((Vt,), "bias"),
    }
    for i in range(L):
        p = f"gpt.h.{i}."
        spec[p + "ln_1.weight"] = ((d,), "ln_w"); spec[p + "ln_1.bias"] = ((d,), "ln_b")
        spec[p + "attn.c_attn.weight"] = ((d, 3 * d), "mat"); spec[p + "attn.c_attn.bias"] = ((3 * d,), "bias")
        spec[p + "attn.c_proj.weight"] = ((d, d), "mat"); spec[p + "attn.c_proj.bias"] = ((d,), "bias")
        spec[p + "ln_2.weight"] = ((d,), "ln_w"); spec[p + "ln_2.bias"] = ((d,), "ln_b")
        spec[p + "mlp.c_fc.weight"] = ((d, 4 * d), "mat"); spec[p + "mlp.c_fc.bias"] = ((4 * d,), "bias")
        spec[p + "mlp.c_proj.weight"] = ((4 * d, d), "mat"); spec[p + "mlp.c_proj.bias"] = ((d,), "bias")
    spec.update(perceiver_weight_spec(d, prefix="conditioning_perceiver."))
    return spec


def perceiver_weight_spec(dim, depth=4, dim_context=80, num_latents=32, dim_head=64, heads=8,
                          ff_mult=4, prefix=""):
    inner = dim_head * heads
    ffi = int(dim * ff_mult * 2 / 3)
    spec = {prefix + "latents": ((num_latents, dim), "emb"),
            prefix + "norm.gamma": ((dim,), "ln_w")}
    if dim_context != dim:
        spec[prefix + "proj_context.weight"] = ((dim, dim_context), "proj")
        spec[prefix + "proj_context.bias"] = ((dim,), "bias")
    for l in range(depth):
        p = f"{prefix}layers.{l}."
        spec[p + "0.to_q.weight"] = ((inner, dim), "mat")
        spec[p + "0.to_kv.weight"] = ((2 * inner, dim), "mat")
        spec[p + "0.to_out.weight"] = ((dim, inner), "mat")
        spec[p + "1.0.weight"] = ((2 * ffi, dim), "mat"); spec[p + "1.0.bias"] = ((2 * ffi,), "bias")
        spec[p + "1.2.weight"] = ((dim, ffi), "mat"); spec[p + "1.2.bias"] = ((dim,), "bias")
    return spec


def dvae_weight_spec(cfg, prefix=""):
    """Content DiscreteVAE encoder + codebook (reference layers/dvae.py:252-295)."""
    ch, hid, nl = cfg["num_channels"], cfg["hidden_dim"], cfg["num_layers"]
    k, cd, nt = cfg["kernel_size"], cfg["codebook_dim"], cfg["num_tokens"]
    chans = [ch] + [hid * 2 ** i for i in range(nl)]
    spec, idx = {}, 0
    for cin, cout in zip(chans[:-1], chans[1:]):
        spec[f"{prefix}encoder.{idx}.0.weight"] = ((cout, cin, k), "conv")
        spec[f"{prefix}encoder.{idx}.0.bias"] = ((cout,), "bias")
        idx += 1
    inner = chans[-1]
    for _ in range(cfg["num_resnet_blocks"]):
        for j, kk in ((0, 3), (2, 3), (4, 1)):
            spec[f"{prefix}encoder.{idx}.net.{j}.weight"] = ((inner, inner, kk), "conv")
            spec[f"{prefix}encoder.{idx}.net.{j}.bias"] = ((inner,), "bias")
        idx += 1
    spec[f"{prefix}encoder.{idx}.weight"] = ((cd, inner, 1), "conv")
    spec[f"{prefix}encoder.{idx}.bias"] = ((cd,), "bias")
    spec[f"{prefix}codebook.embed"] = ((cd, nt), "codebook")
    return spec


def hifigan_weight_spec(cfg, prefix=""):
    """HiFi-GAN generator with weight-norm parametrisation, named as the reference state dict
    (reference layers/hifigan.py:160-216): conv_pre, ups.{i}, resblocks.{i*nk+j}.convs.{0,1}, conv_post."""
    spec = {}
    ch = cfg["upsample_initial_channel"]

    def wn(name, shape, g_mean):
        spec[prefix + name + ".weight_g"] = ((shape[0], 1, 1), ("wn_g", g_mean))
        spec[prefix + name + ".weight_v"] = (shape, "conv")
        spec[prefix + name + ".bias"] = ((shape[1] if name.startswith("ups.") else shape[0],), "bias")

    wn("conv_pre", (ch, cfg["input_feat_dim"], 7), 1.0)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (r, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        wn(f"ups.{i}", (ch, ch // 2, k), 2.0)                      # ConvTranspose1d weight [Cin, Cout, k]
        ch //= 2
        for j, kk in enumerate(cfg["resblock_kernel_sizes"]):
            for q in range(2):
                wn(f"resblocks.{i * nk + j}.convs.{q}", (ch, ch, kk), 0.5)
    wn("conv_post", (1, ch, 7), 1.0)
    return spec


def hubert_weight_spec(cfg, prefix=""):
    """fairseq HubertModel / ContentVec tensors used by extract_features + final_proj, named as in the checkpoint
    (keys under `content_extractor.model.` in a GenVC checkpoint; reference layers/content_processor.py:10-31)."""
    spec = {}
    cin = 1
    for i, (c, k, s) in enumerate(cfg["conv_layers"]):
        spec[f"{prefix}feature_extractor.conv_layers.{i}.0.weight"] = ((c, cin, k), "conv")
        if i == 0:
            spec[f"{prefix}feature_extractor.conv_layers.0.2.weight"] = ((c,), "ln_w")
            spec[f"{prefix}feature_extractor.conv_layers.0.2.bias"] = ((c,), "ln_b")
        cin = c
    e, f = cfg["embed_dim"], cfg["ffn_dim"]
    spec[prefix + "layer_norm.weight"] = ((cin,), "ln_w"); spec[prefix + "layer_norm.bias"] = ((cin,), "ln_b")
    spec[prefix + "post_extract_proj.weight"] = ((e, cin), "lin"); spec[prefix + "post_extract_proj.bias"] = ((e,), "bias")
    kp, g = cfg["pos_conv_kernel"], cfg["pos_conv_groups"]
    spec[prefix + "encoder.pos_conv.0.weight_g"] = ((1, 1, kp), ("wn_g", 1.0))
    spec[prefix + "encoder.pos_conv.0.weight_v"] = ((e, e // g, kp), "conv")
    spec[prefix + "encoder.pos_conv.0.bias"] = ((e,), "bias")
    spec[prefix + "encoder.layer_norm.weight"] = ((e,), "ln_w"); spec[prefix + "encoder.layer_norm.bias"] = ((e,), "ln_b")
    for l in range(cfg["layers"]):
        p = f"{prefix}encoder.layers.{l}."
        for n in ("q_proj", "k_proj", "v_proj", "out_proj"):
            spec[p + f"self_attn.{n}.weight"] = ((e, e), "lin"); spec[p + f"self_attn.{n}.bias"] = ((e,), "bias")
        spec[p + "self_attn_layer_norm.weight"] = ((e,), "ln_w"); spec[p + "self_attn_layer_norm.bias"] = ((e,), "ln_b")
        spec[p + "fc1.weight"] = ((f, e), "lin"); spec[p + "fc1.bias"] = ((f,), "bias")
        spec[p + "fc2.weight"] = ((e, f), "lin"); spec[p + "fc2.bias"] = ((e,), "bias")
        spec[p + "final_layer_norm.weight"] = ((e,), "ln_w"); spec[p + "final_layer_norm.bias"] = ((e,), "ln_b")
    spec[prefix + "final_proj.weight"] = ((cfg["final_dim"], e), "lin"); spec[prefix + "final_proj.bias"] = ((cfg["final_dim"],), "bias")
    return spec


def make_weights(seed, spec, device="cpu", head_scale=0.05):
    """Materialise a spec.  Scales: matrices N(0,0.02)-like, LayerNorm gains near 1."""
    out = {}
    for name, (shape, kind) in spec.items():
        if isinstance(kind, tuple) and kind[0] == "wn_g":
            out[name] = uniform(seed, name, shape, 0.1 * kind[1], kind[1], device)
        elif kind == "ln_w":
            out[name] = uniform(seed, name, shape, 0.1, 1.0, device)
        elif kind in ("ln_b", "bias"):
            out[name] = uniform(seed, name, shape, 0.02, 0.0, device)
        elif kind == "head":
            out[name] = uniform(seed, name, shape, head_scale, 0.0, device)
        elif kind == "proj":
            out[name] = uniform(seed, name, shape, 0.1, 0.0, device)
        elif kind == "lin":
            out[name] = uniform(seed, name, shape, 1.0 / math.sqrt(shape[1]), 0.0, device)
        elif kind == "conv":
            fan_in = shape[1] * shape[2]
            out[name] = uniform(seed, name, shape, 1.0 / math.sqrt(fan_in), 0.0, device)
        elif kind == "codebook":
            out[name] = uniform(seed, name, shape, 1.0, 0.0, device)
        else:  # "mat", "emb"
            out[name] = uniform(seed, name, shape, 0.02, 0.0, device)
    return out


def synth_audio(seed, name, n_samples, amplitude=0.1, device="cpu"):
    """Noise plus a few tones; never exactly zero (ContentVec treats wav==0 as padding,
    reference layers/content_processor.py:24)."""
    # always built on the CPU (sin differs in the last bit between devices), then moved
    x = uniform(seed, name, (n_samples,), amplitude / 2.0, 0.0, "cpu")
    t = torch.arange(n_samples, dtype=torch.float32)
    for f, a in ((0.011, 0.5), (0.037, 0.3), (0.093, 0.2)):
        x = x + amplitude * a * torch.sin(2.0 * math.pi * f * t)
    x = torch.where(x == 0, torch.full_like(x, 1e-4), x)
    return x.clamp_(-1.0, 1.0).unsqueeze(0).to(device)
