"""CPU oracle for the GenVC codec-token generation hot path.

TEST INFRASTRUCTURE ONLY.  This file is a plain torch-CPU / numpy restatement of the
reference's algorithm for the path named in BASELINE.json `north_star` (SURVEY.md
section 8a).  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it; the product (`genvc_amd/`) never does and fails loudly when
its HIP library is missing.

Pinning status -- the reference ships no tests, so pins are
outputs of the reference's own classes run in the build container by
`oracle/make_golden.py` and committed under `tests/golden/`:
  * GPT prefill / decode / latent re-pass / compute_embeddings, Perceiver, content
    DVAE + VQ, HF logits processors: PINNED against the imported reference classes.
  * mel spectrogram: the arithmetic lives in torchaudio==2.3.0 (absent here):
    PARITY UNPINNED; restated from torchaudio's documented algorithm and cross-checked
    against a float64 direct DFT.
  * sample_stream loop (stream_generator.py cannot import under transformers 5):
    restated from stream_generator.py:809-881, the per-step processors are pinned.

Every function cites the reference file:line it follows (paths relative to
/root/reference).  Weights are passed as a dict keyed by the reference state-dict names.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

# ---------------------------------------------------------------------------
# row 1: mel front-end  (utils.py:97-162; torchaudio.transforms.MelSpectrogram)
# ---------------------------------------------------------------------------

MEL = dict(n_fft=2048, hop=256, win=1024, sr=24000, f_min=0.0, f_max=8000.0, n_mels=80)


def mel_filterbank(n_freqs=1025, f_min=0.0, f_max=8000.0, n_mels=80, sr=24000, dtype=torch.float32):
    """torchaudio.functional.melscale_fbanks(norm="slaney", mel_scale="htk") -> [n_freqs, n_mels]."""
    all_freqs = torch.linspace(0, sr // 2, n_freqs, dtype=dtype)
    m_min = 2595.0 * math.log10(1.0 + f_min / 700.0)
    m_max = 2595.0 * math.log10(1.0 + f_max / 700.0)
    m_pts = torch.linspace(m_min, m_max, n_mels + 2, dtype=dtype)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts.unsqueeze(0) - all_freqs.unsqueeze(1)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = torch.clamp(torch.min(down, up), min=0.0)
    enorm = 2.0 / (f_pts[2:n_mels + 2] - f_pts[:n_mels])
    return fb * enorm.unsqueeze(0)


def mel_spectrogram(wav, mel_norms, dtype=torch.float32):
    """utils.py:150-162: log(clamp(melfb . |STFT|^2, 1e-5)) / mel_norms.  wav [B,T] -> [B,80,1+T//256]."""
    c = MEL
    x = wav.to(dtype)
    window = torch.hann_window(c["win"], periodic=True, dtype=dtype)
    spec = torch.stft(x, c["n_fft"], c["hop"], c["win"], window=window, center=True,
                      pad_mode="reflect", normalized=False, onesided=True, return_complex=True)
    power = spec.real ** 2 + spec.imag ** 2                       # [B,1025,F]
    fb = mel_filterbank(c["n_fft"] // 2 + 1, c["f_min"], c["f_max"], c["n_mels"], c["sr"], dtype)
    mel = torch.matmul(power.transpose(1, 2), fb).transpose(1, 2)
    mel = torch.log(torch.clamp(mel, min=1e-5))
    return mel / mel_norms.to(dtype).view(1, -1, 1)


def mel_spectrogram_dft64(wav, mel_norms):
    """Independent float64 check of `mel_spectrogram`: explicit framing + direct DFT (numpy)."""
    c = MEL
    x = np.asarray(wav, dtype=np.float64)
    out = []
    n = np.arange(c["win"])
    win = 0.5 - 0.5 * np.cos(2.0 * np.pi * n / c["win"])
    k = np.arange(c["n_fft"] // 2 + 1)
    off = (c["n_fft"] - c["win"]) // 2
    # only the centre `win` taps of each n_fft frame are non-zero
    ang = -2.0 * np.pi * np.outer(k, n + off) / c["n_fft"]
    cosm, sinm = np.cos(ang), np.sin(ang)
    fb = mel_filterbank(dtype=torch.float64).numpy()
    for b in range(x.shape[0]):
        xp = np.pad(x[b], c["n_fft"] // 2, mode="reflect")
        nfr = 1 + x.shape[1] // c["hop"]
        fr = np.stack([xp[i * c["hop"] + off:i * c["hop"] + off + c["win"]] * win for i in range(nfr)], 1)
        power = (cosm @ fr) ** 2 + (sinm @ fr) ** 2
        mel = fb.T @ power
        out.append(np.log(np.maximum(mel, 1e-5)) / np.asarray(mel_norms, dtype=np.float64)[:, None])
    return np.stack(out)


def resample(wav, orig_sr, new_sr, lowpass_filter_width=6, rolloff=0.99):
    """Row f2: `utils.load_audio` -> `torchaudio.functional.resample(audio, lsr, sampling_rate)` (/root/reference/utils.py:53-62).
    torchaudio (pinned 2.3.0, README.md:43) is absent from this image: PARITY UNPINNED against torchaudio itself.  This restates
    its published algorithm (`_get_sinc_resample_kernel` + `_apply_sinc_resample_kernel`, default `sinc_interp_hann`,
    lowpass_filter_width 6, rolloff 0.99; recipe in SURVEY.md 8c) as an explicit polyphase sum in numpy float64:
        o, n = orig / gcd, new / gcd;  base = min(o, n) * rolloff;  width = ceil(lpw * o / base)
        output sample m = i * n + j (phase j of input block i) = sum_k x[i * o + k - width] * h_j[k],  k in [0, 2 width + o)
        h_j[k] = sinc(pi t) * cos^2(pi t / (2 lpw)) * base / o,  t = clamp((k - width) / o - j / n) * base, +-lpw)
    The kernel is rounded to float32 as torchaudio does (dtype=None branch) and the sums run in float64; zero padding on both
    sides; output length ceil(n * T / o).  tests/test_oracle.py cross-checks it on band-limited signals against the analytic
    resampled waveform and scipy.signal.resample_poly (another filter: agreement to the filters' ripple)."""
    import math
    import numpy as np
    x = wav.detach().cpu().numpy().astype(np.float64) if torch.is_tensor(wav) else np.asarray(wav, np.float64)
    g = math.gcd(int(orig_sr), int(new_sr))
    o, n = int(orig_sr) // g, int(new_sr) // g
    if o == n:
        return torch.from_numpy(x.astype(np.float32))
    base = min(o, n) * rolloff
    width = int(math.ceil(lowpass_filter_width * o / base))
    k = np.arange(-width, width + o, dtype=np.float64) / o                        # [2 width + o]
    t = (-np.arange(n, dtype=np.float64)[:, None] / n + k[None, :]) * base         # [n, taps]
    t = np.clip(t, -lowpass_filter_width, lowpass_filter_width)
    window = np.cos(t * math.pi / lowpass_filter_width / 2) ** 2
    tp = t * math.pi
    with np.errstate(invalid="ignore", divide="ignore"):
        sinc = np.where(tp == 0, 1.0, np.sin(tp) / tp)
    h = (sinc * window * (base / o)).astype(np.float32).astype(np.float64)         # torchaudio keeps the kernel in float32
    C, T = x.shape
    taps = h.shape[1]
    xp = np.concatenate([np.zeros((C, width)), x, np.zeros((C, width + o))], 1)
    n_blocks = (xp.shape[1] - taps) // o + 1
    idx = np.arange(n_blocks)[:, None] * o + np.arange(taps)[None, :]              # [blocks, taps]
    frames = xp[:, idx]                                                            # [C, blocks, taps]
    y = np.einsum("cbt,jt->cbj", frames, h).reshape(C, -1)                         # phase j of block i -> sample i n + j
    return torch.from_numpy(y[:, :int(math.ceil(n * T / o))].astype(np.float32))


# ---------------------------------------------------------------------------
# row 3: Perceiver resampler  (perceiver_encoder.py:225-319, gpt.py:351-373)
# ---------------------------------------------------------------------------

def _rmsnorm(x, gamma):
    # perceiver_encoder.py:177-179: F.normalize(x, dim=-1) * sqrt(dim) * gamma
    return F.normalize(x, dim=-1) * (x.shape[-1] ** 0.5) * gamma


def perceiver_forward(w, x, prefix="conditioning_perceiver.", heads=8, dim_head=64):
    """x [B,F,dim_context] -> [B,num_latents,dim]  (PerceiverResampler.forward :265-276)."""
    g = lambda n: w[prefix + n]
    B = x.shape[0]
    if prefix + "proj_context.weight" in w:
        x = F.linear(x, g("proj_context.weight"), g("proj_context.bias"))
    lat = g("latents").unsqueeze(0).expand(B, -1, -1)
    depth = 0
    while f"{prefix}layers.{depth}.0.to_q.weight" in w:
        depth += 1
    scale = dim_head ** -0.5
    for l in range(depth):
        p = f"layers.{l}."
        # Attention.forward :305-319 with cross_attn_include_queries -> keys = [latents; context]
        ctx = torch.cat((lat, x), dim=-2)
        q = F.linear(lat, g(p + "0.to_q.weight"))
        kv = F.linear(ctx, g(p + "0.to_kv.weight"))
        k, v = kv.chunk(2, dim=-1)
        sh = lambda t: t.reshape(B, t.shape[1], heads, dim_head).transpose(1, 2)
        q, k, v = sh(q), sh(k), sh(v)
        sim = torch.matmul(q, k.transpose(-1, -2)) * scale          # Attend.forward :128
        attn = sim.softmax(dim=-1)
        o = torch.matmul(attn, v).transpose(1, 2).reshape(B, lat.shape[1], heads * dim_head)
        lat = F.linear(o, g(p + "0.to_out.weight")) + lat
        # FeedForward :211-222 = Linear -> GEGLU (x, gate = chunk; gelu(gate) * x, exact erf) -> Linear
        h = F.linear(lat, g(p + "1.0.weight"), g(p + "1.0.bias"))
        a, gate = h.chunk(2, dim=-1)
        h = F.gelu(gate) * a
        lat = F.linear(h, g(p + "1.2.weight"), g(p + "1.2.bias")) + lat
    return _rmsnorm(lat, g("norm.gamma"))


def get_style_emb(w, mel):
    """gpt.py:351-373 (mask=None at inference): mel [B,80,F] -> [B,d,32]."""
    return perceiver_forward(w, mel.permute(0, 2, 1)).transpose(1, 2)


def get_gpt_cond_latents(w, audio, mel_norms, sr=24000, length=30, chunk_length=6):
    """trainers/hifigan_trainer.py:438-455: <=30 s, 6 s chunks, skip <0.33 s, mean over chunks -> [1,32,d]."""
    embs = []
    if audio.shape[1] > sr * length:
        audio = audio[:, :sr * length]
    for i in range(0, audio.shape[1], sr * chunk_length):
        chunk = audio[:, i:i + sr * chunk_length]
        if chunk.shape[-1] < sr * 0.33:
            continue
        embs.append(get_style_emb(w, mel_spectrogram(chunk, mel_norms)))
    return torch.stack(embs).mean(dim=0).transpose(1, 2)


# ---------------------------------------------------------------------------
# row 5: content DVAE encoder + VQ  (dvae.py:252-291, 324-331, 87-93)
# ---------------------------------------------------------------------------

def dvae_encode(w, feat, prefix=""):
    """feat [B,C,T] -> encoder output [B,T',codebook_dim] (channels last, as fed to Quantize)."""
    x = feat
    idx = 0
    while f"{prefix}encoder.{idx}.0.weight" in w:                 # strided conv + ReLU stages
        x = F.relu(F.conv1d(x, w[f"{prefix}encoder.{idx}.0.weight"], w[f"{prefix}encoder.{idx}.0.bias"],
                            stride=2, padding=(w[f"{prefix}encoder.{idx}.0.weight"].shape[-1] - 1) // 2))
        idx += 1
    while f"{prefix}encoder.{idx}.net.0.weight" in w:             # ResBlock :172-184
        p = f"{prefix}encoder.{idx}.net."
        h = F.relu(F.conv1d(x, w[p + "0.weight"], w[p + "0.bias"], padding=1))
        h = F.relu(F.conv1d(h, w[p + "2.weight"], w[p + "2.bias"], padding=1))
        x = F.conv1d(h, w[p + "4.weight"], w[p + "4.bias"]) + x
        idx += 1
    x = F.conv1d(x, w[f"{prefix}encoder.{idx}.weight"], w[f"{prefix}encoder.{idx}.bias"])
    return x.permute(0, 2, 1)


def vq_indices(x, embed):
    """Quantize.forward dvae.py:87-90: argmax(-(|x|^2 - 2 x.E + |E|^2)); first index wins ties."""
    flat = x.reshape(-1, x.shape[-1])
    dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ embed + embed.pow(2).sum(0, keepdim=True)
    _, ind = (-dist).max(1)
    return ind.view(*x.shape[:-1])


def dvae_get_codebook_indices(w, feat, prefix=""):
    return vq_indices(dvae_encode(w, feat, prefix), w[prefix + "codebook.embed"])


# ---------------------------------------------------------------------------
# rows 6-9, 12: GPT  (gpt.py, gpt_inference.py, HF GPT2Model)
# ---------------------------------------------------------------------------

def gelu_new(x):
    return 0.5 * x * (1.0 + torch.tanh(math.sqrt(2.0 / math.pi) * (x + 0.044715 * x ** 3)))


def _ln(x, w, name):
    return F.layer_norm(x, (x.shape[-1],), w[name + ".weight"], w[name + ".bias"], 1e-5)


def compute_embeddings(w, dims, cond_latents, codes):
    """gpt.py:572-592 -> (prefix_emb [B,P,d], fake ids int64 [B,P+1])."""
    ids = F.pad(codes, (0, 1), value=dims["stop_text_token"])
    ids = F.pad(ids, (1, 0), value=dims["start_text_token"])
    emb = w["text_embedding.weight"][ids] + w["text_pos_embedding.emb.weight"][:ids.shape[1]]
    emb = torch.cat([cond_latents, emb], dim=1)
    fake = torch.full((emb.shape[0], emb.shape[1] + 1), 1, dtype=torch.long)
    fake[:, -1] = dims["start_audio_token"]
    return emb, fake


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def _ln_fold_bf16(x, w, ln, proj):
    """The build's bf16-activation mode (include/genvc_hip.h: weight_dtype 3; csrc/persist_rows_b16.h) of `LN -> Conv1D`: the residual
    stream stays fp32, the row that crosses the hand-off is x~ = bf16(x), the LayerNorm gain is folded into the bf16 weights at pack time
    (W' = bf16(W g), W the bf16-rounded weight the context holds) and the LayerNorm bias into a per-output constant (C = b_ln W + b):
    y = ((x~ - mean(x~)) rstd(x~)) W' + C, fp32 accumulation.  What a bf16-autocast run of the reference would round is LN(x) instead of x
    -- the same 2^-9 relative step one normalisation earlier."""
    xt = _bf16(x)
    mean = xt.mean(-1, keepdim=True)
    var = (xt - mean).pow(2).mean(-1, keepdim=True)
    a = (xt - mean) * torch.rsqrt(var + 1e-5)
    W = w[proj + ".weight"]
    Wf = _bf16(W * w[ln + ".weight"][:, None])
    return a @ Wf + (w[ln + ".bias"] @ W + w[proj + ".bias"])


def gpt_blocks(w, dims, x, cache=None):
    """HF GPT2Model block stack on rows x [B,T,d]; `cache` = list of (K,V) [B,H,S,hd] or None.

    Conv1D: y = x @ W[in,out] + b.  Causal within the new rows, full view of the cache.
    Returns (ln_f(h), new_cache).
    dims["act_bf16"] (build-only mode, see _ln_fold_bf16): the four activations that cross a hand-off of the one-launch rows step --
    x into LN1, the attention output, x' into LN2, the gelu output -- are rounded to bf16 where they are published; weights (mode 1) and
    k / v (mode 2) are rounded as before; residual sums, softmax, LayerNorm statistics and every accumulation stay fp32.
    """
    act = bool(dims.get("act_bf16"))
    B, T, d = x.shape
    H = dims["n_head"]
    hd = d // H
    new_cache = []
    for l in range(dims["n_layer"]):
        p = f"gpt.h.{l}."
        if act:
            qkv = _ln_fold_bf16(x, w, p + "ln_1", p + "attn.c_attn")
        else:
            a = _ln(x, w, p + "ln_1")
            qkv = a @ w[p + "attn.c_attn.weight"] + w[p + "attn.c_attn.bias"]
        q, k, v = qkv.split(d, dim=-1)
        sh = lambda t: t.reshape(B, T, H, hd).transpose(1, 2)
        q, k, v = sh(q), sh(k), sh(v)
        if dims.get("kv_bf16"):
            # the build's bf16 KV cache (include/genvc_hip.h: weight_dtype 2): every k and v is rounded to nearest even when
            # it enters the cache, and attention reads the cache for the new rows too
            k, v = k.to(torch.bfloat16).to(torch.float32), v.to(torch.bfloat16).to(torch.float32)
        if cache is not None:
            k = torch.cat([cache[l][0], k], dim=2)
            v = torch.cat([cache[l][1], v], dim=2)
        new_cache.append((k, v))
        S = k.shape[2]
        s = torch.matmul(q, k.transpose(-1, -2)) * (hd ** -0.5)
        if T > 1:
            i = torch.arange(T).view(T, 1) + (S - T)
            j = torch.arange(S).view(1, S)
            s = s.masked_fill(j > i, torch.finfo(s.dtype).min)
        pr = torch.softmax(s, dim=-1)
        o = torch.matmul(pr, v).transpose(1, 2).reshape(B, T, d)
        if act:
            o = _bf16(o)
        x = x + (o @ w[p + "attn.c_proj.weight"] + w[p + "attn.c_proj.bias"])
        if act:
            h = _bf16(gelu_new(_ln_fold_bf16(x, w, p + "ln_2", p + "mlp.c_fc")))
        else:
            m = _ln(x, w, p + "ln_2")
            h = gelu_new(m @ w[p + "mlp.c_fc.weight"] + w[p + "mlp.c_fc.bias"])
        x = x + (h @ w[p + "mlp.c_proj.weight"] + w[p + "mlp.c_proj.bias"])
    return _ln(x, w, "gpt.ln_f"), new_cache


def head(w, h):
    """gpt_inference.py:18,111-112: lm_head = Sequential(final_norm, mel_head) on ln_f output.
    Returns (latent = final_norm(h), logits); the latent is what stream_generator.py:865 yields."""
    z = _ln(h, w, "final_norm")
    return z, F.linear(z, w["mel_head.weight"], w["mel_head.bias"])


def gpt_prefill(w, dims, prefix_emb):
    """gpt_inference.py:81-91: rows = [prefix | mel_embedding[start] + mel_pos[0]]; last row only."""
    B = prefix_emb.shape[0]
    row = w["mel_embedding.weight"][dims["start_audio_token"]] + w["mel_pos_embedding.emb.weight"][0]
    emb = torch.cat([prefix_emb, row.view(1, 1, -1).expand(B, 1, -1)], dim=1)
    # a weight_dtype-3 context rounds activations on the one-launch rows step only (decode steps, cached chunk prefills); a full
    # prefill runs on the GEMM path with fp32 activations: dims["act_bf16"] does not apply here unless the caller says so
    h, cache = gpt_blocks(w, dict(dims, act_bf16=bool(dims.get("act_bf16_prefill"))), emb)
    z, logits = head(w, h[:, -1])
    return z, logits, cache


def gpt_decode_step(w, dims, cache, tok, j):
    """gpt_inference.py:92-96: emb = mel_embedding[tok] + mel_pos[j], j = #acoustic positions cached."""
    emb = (w["mel_embedding.weight"][tok] + w["mel_pos_embedding.emb.weight"][j]).unsqueeze(1)
    h, cache = gpt_blocks(w, dims, emb, cache)
    z, logits = head(w, h[:, -1])
    return z, logits, cache


def gpt_latents(w, dims, cond_latents, codes, gen_codes):
    """GPT.forward(..., cond_latents=, return_latent=True) gpt.py:375-508 for one utterance batch
    with full-length rows: input [cond | <s>codes</s> | start, gen, stop x4], drop last 5 -> n latents."""
    stop_a, start_a = dims["stop_audio_token"], dims["start_audio_token"]
    text = F.pad(codes, (0, 1), value=dims["stop_text_token"])
    text = F.pad(text, (1, 0), value=dims["start_text_token"])
    n = gen_codes.shape[1]
    # code_lengths = ceil(n*1024/1024)+3 -> pad to n+3 with zeros, then stop, set_mel_padding -> stop
    audio = torch.cat([gen_codes, torch.full((gen_codes.shape[0], 4), stop_a, dtype=gen_codes.dtype)], 1)
    audio = F.pad(audio, (1, 0), value=start_a)
    temb = w["text_embedding.weight"][text] + w["text_pos_embedding.emb.weight"][:text.shape[1]]
    memb = w["mel_embedding.weight"][audio] + w["mel_pos_embedding.emb.weight"][:audio.shape[1]]
    emb = torch.cat([cond_latents, temb, memb], dim=1)
    h, _ = gpt_blocks(w, dims, emb)
    enc = _ln(h[:, cond_latents.shape[1]:], w, "final_norm")
    return enc[:, -audio.shape[1]:][:, :-5]


# ---------------------------------------------------------------------------
# row 10: sampling  (stream_generator.py:809-881; HF logits processors 4.33)
# ---------------------------------------------------------------------------

def rng_uniform(seed, step, row):
    """Counter-based uniform in [0,1): the reference's torch.multinomial stream is not
    reproducible, so the build (kernel and oracle alike) draws u from this hash."""
    m64 = (1 << 64) - 1
    x = (seed * 0x9E3779B97F4A7C15 + step * 0xBF58476D1CE4E5B9 + row * 0x94D049BB133111EB + 0x2545F4914F6CDD1D) & m64
    x ^= x >> 30; x = (x * 0xBF58476D1CE4E5B9) & m64
    x ^= x >> 27; x = (x * 0x94D049BB133111EB) & m64
    x ^= x >> 31
    return float(np.float32((x >> 40) * (2.0 ** -24)))


def process_logits(logits, ids, rep_penalty=2.0, temperature=0.85, top_k=15, top_p=0.85):
    """RepetitionPenalty -> Temperature -> TopK(min_keep 1) -> TopP(min_keep 1); logits [B,V], ids [B,S]."""
    s = logits.clone()
    g = torch.gather(s, 1, ids)
    g = torch.where(g < 0, g * rep_penalty, g / rep_penalty)
    s.scatter_(1, ids, g)
    s = s / temperature
    if top_k and top_k > 0:
        k = min(top_k, s.shape[-1])
        kth = torch.topk(s, k)[0][..., -1, None]
        s = s.masked_fill(s < kth, -float("inf"))
    if top_p is not None and top_p < 1.0:
        sl, si = torch.sort(s, descending=False)
        cp = sl.softmax(dim=-1).cumsum(dim=-1)
        rem = cp <= (1 - top_p)
        rem[..., -1:] = False
        s = s.masked_fill(rem.scatter(1, si, rem), -float("inf"))
    return s


def sample_from_scores(scores, seed, step):
    """softmax + inverse-CDF draw in vocabulary order: with e = exp(s - max) (fp32) and a running
    float64 mass, the token is the first surviving index whose mass reaches u * total."""
    out = torch.empty(scores.shape[0], dtype=torch.long)
    for b in range(scores.shape[0]):
        s = scores[b]
        kept = torch.isfinite(s)
        e = torch.where(kept, torch.exp(s - s[kept].max()), torch.zeros_like(s))
        cdf = torch.cumsum(e.double(), 0)
        target = float(np.float32(rng_uniform(seed, step, b))) * float(cdf[-1])
        hit = torch.nonzero((cdf >= target) & kept)
        out[b] = int(hit[0]) if len(hit) else int(torch.nonzero(kept)[-1])
    return out


def generate(w, dims, cond_latents, codes, sampling, max_new=None, seed=0, stop_on_eos=True):
    """gpt.generate / get_generator loop (gpt.py:594-621 + stream_generator.py:809-881).

    Returns (tokens int64 [B,n] incl. the EOS step, latents [B,n,d], logits of every step [n,B,V]).
    Finished rows emit the pad (=stop) token; the EOS step's latent is yielded too (:865)."""
    stop = dims["stop_audio_token"]
    max_new = dims["max_gen_mel_tokens"] if max_new is None else max_new
    prefix, ids = compute_embeddings(w, dims, cond_latents, codes)
    B = ids.shape[0]
    unfinished = torch.ones(B, dtype=torch.long)
    z, logits, cache = gpt_prefill(w, dims, prefix)
    toks, lats, all_logits = [], [], []
    for step in range(max_new):
        all_logits.append(logits)
        scores = process_logits(logits, ids, sampling["repetition_penalty"], sampling["temperature"],
                                sampling["top_k"], sampling["top_p"])
        nxt = sample_from_scores(scores, seed, step)
        nxt = nxt * unfinished + stop * (1 - unfinished)
        toks.append(nxt); lats.append(z)
        ids = torch.cat([ids, nxt[:, None]], dim=-1)
        unfinished = unfinished * (nxt != stop).long()
        if (stop_on_eos and unfinished.max() == 0) or step == max_new - 1:
            break
        z, logits, cache = gpt_decode_step(w, dims, cache, nxt, step + 1)
    return torch.stack(toks, 1), torch.stack(lats, 1), torch.stack(all_logits, 0)


# ---------------------------------------------------------------------------
# row 4 / f3: ContentVec = fairseq HubertModel.extract_features(output_layer=12) + final_proj
# (layers/content_processor.py:17-31).  fairseq is neither vendored nor installed: the arithmetic below restates the
# HuBERT-base forward (conv extractor with GroupNorm on layer 0, post-LN encoder, weight-normed grouped positional
# conv) and is pinned against HuggingFace's HubertModel -- the architecture fairseq checkpoints convert to -- by
# oracle/make_golden.py.  The reference's `padding_mask = (wav == 0)` (content_processor.py:24) is followed through
# fairseq's semantics restated from the library's published code (HubertModel.forward_padding_mask: drop the trailing
# T % F samples, view [F][T // F], a frame is padding when ALL its samples are zero; TransformerEncoder.extract_features:
# x[padding] = 0 ahead of pos_conv; MultiheadAttention key_padding_mask: scores of padding keys -> -inf).  That part has
# no pin at all (HF's HubertModel masks by length, not by zeros): parity unpinned.  With no all-zero chunk in the input
# the mask is empty and the function is the one the HF goldens pin.
# ---------------------------------------------------------------------------

def hubert_frame_padding_mask(wav, n_frames):
    """(wav == 0) [B,T] -> bool [B,F]; fairseq HubertModel.forward_padding_mask"""
    pm = wav == 0
    extra = pm.shape[1] % n_frames
    if extra > 0:
        pm = pm[:, :-extra]
    return pm.reshape(pm.shape[0], n_frames, -1).all(-1)


def hubert_extract_features(w, cfg, wav, prefix="", padding_mask=True):
    """wav [B,T] -> [B,T50,final_dim]; padding_mask=False: the mask-free forward (what HF's HubertModel computes)"""
    g = lambda n: w[prefix + n]
    x = wav[:, None, :]
    for i, (c, k, s) in enumerate(cfg["conv_layers"]):
        x = F.conv1d(x, g(f"feature_extractor.conv_layers.{i}.0.weight"), stride=s)
        if i == 0:
            x = F.group_norm(x, c, g("feature_extractor.conv_layers.0.2.weight"), g("feature_extractor.conv_layers.0.2.bias"), 1e-5)
        x = F.gelu(x)
    x = x.transpose(1, 2)
    x = F.layer_norm(x, (x.shape[-1],), g("layer_norm.weight"), g("layer_norm.bias"), 1e-5)
    x = F.linear(x, g("post_extract_proj.weight"), g("post_extract_proj.bias"))
    pm = hubert_frame_padding_mask(wav, x.shape[1]) if padding_mask else None
    if pm is not None and not bool(pm.any()):
        pm = None
    if pm is not None:
        x = x.masked_fill(pm[:, :, None], 0.0)
    v, gg = g("encoder.pos_conv.0.weight_v"), g("encoder.pos_conv.0.weight_g")
    wpc = gg * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()                  # weight_norm(dim=2)
    kp = cfg["pos_conv_kernel"]
    pc = F.conv1d(x.transpose(1, 2), wpc, g("encoder.pos_conv.0.bias"), padding=kp // 2, groups=cfg["pos_conv_groups"])
    if kp % 2 == 0:
        pc = pc[:, :, :-1]                                                        # SamePad
    x = x + F.gelu(pc).transpose(1, 2)
    e = x.shape[-1]
    x = F.layer_norm(x, (e,), g("encoder.layer_norm.weight"), g("encoder.layer_norm.bias"), 1e-5)
    H = cfg["heads"]
    hd = e // H
    B, T, _ = x.shape
    for l in range(cfg["layers"]):
        p = f"encoder.layers.{l}."
        sh = lambda t: t.reshape(B, T, H, hd).transpose(1, 2)
        q = sh(F.linear(x, g(p + "self_attn.q_proj.weight"), g(p + "self_attn.q_proj.bias")))
        k = sh(F.linear(x, g(p + "self_attn.k_proj.weight"), g(p + "self_attn.k_proj.bias")))
        vv = sh(F.linear(x, g(p + "self_attn.v_proj.weight"), g(p + "self_attn.v_proj.bias")))
        sc = torch.matmul(q, k.transpose(-1, -2)) * hd ** -0.5
        if pm is not None:
            sc = sc.masked_fill(pm[:, None, None, :], float("-inf"))
        a = torch.softmax(sc, dim=-1)
        o = torch.matmul(a, vv).transpose(1, 2).reshape(B, T, e)
        x = x + F.linear(o, g(p + "self_attn.out_proj.weight"), g(p + "self_attn.out_proj.bias"))
        x = F.layer_norm(x, (e,), g(p + "self_attn_layer_norm.weight"), g(p + "self_attn_layer_norm.bias"), 1e-5)
        h = F.gelu(F.linear(x, g(p + "fc1.weight"), g(p + "fc1.bias")))
        x = x + F.linear(h, g(p + "fc2.weight"), g(p + "fc2.bias"))
        x = F.layer_norm(x, (e,), g(p + "final_layer_norm.weight"), g(p + "final_layer_norm.bias"), 1e-5)
    return F.linear(x, g("final_proj.weight"), g("final_proj.bias"))


# ---------------------------------------------------------------------------
# row f1: HiFi-GAN generator  (layers/hifigan.py:119-157, 160-233)
# ---------------------------------------------------------------------------

def _wn(w, name):
    """fold torch.nn.utils.weight_norm (dim 0): w = g * v / |v|"""
    v, g = w[name + ".weight_v"], w[name + ".weight_g"]
    return g * v / v.pow(2).sum(dim=tuple(range(1, v.dim())), keepdim=True).sqrt()


def hifigan_forward(w, cfg, x, prefix=""):
    """x [B,input_feat_dim,T] -> [B,1,T*prod(rates)]"""
    p = prefix
    x = F.conv1d(x, _wn(w, p + "conv_pre"), w[p + "conv_pre.bias"], padding=3)
    nk = len(cfg["resblock_kernel_sizes"])
    for i, (r, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
        x = F.leaky_relu(x, 0.1)
        x = F.conv_transpose1d(x, _wn(w, f"{p}ups.{i}"), w[f"{p}ups.{i}.bias"], stride=r, padding=(k - r) // 2)
        xs = None
        for j, (kk, dil) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            r_ = x
            for q in range(2):                                     # ResBlock2.forward :147-152
                n = f"{p}resblocks.{i * nk + j}.convs.{q}"
                xt = F.conv1d(F.leaky_relu(r_, 0.1), _wn(w, n), w[n + ".bias"], dilation=dil[q],
                              padding=(kk * dil[q] - dil[q]) // 2)
                r_ = xt + r_
            xs = r_ if xs is None else xs + r_
        x = xs / nk
    x = F.leaky_relu(x)                                            # default slope 0.01 (:230)
    x = F.conv1d(x, _wn(w, p + "conv_post"), w[p + "conv_post.bias"], padding=3)
    return torch.tanh(x)


def vocode_latents(w, cfg, latents, scale=4.0):
    """inference_utils.py:81-87: latents [B,n,d] -> interpolate x4 (linear) -> HiFi-GAN"""
    mel = F.interpolate(latents.transpose(1, 2), scale_factor=[scale], mode="linear")
    return hifigan_forward(w, cfg, mel)


# ---------------------------------------------------------------------------
# row 13: harness pieces  (inference/inference_utils.py)
# ---------------------------------------------------------------------------

def segment_source(n_samples, seg_len_s, sr=16000):
    """inference_utils.py:34-50: [(start, end, padded_len)] per segment; last padded to >= 0.32 s."""
    seg = int(seg_len_s * sr)
    min_len = int(0.32 * sr)
    out = []
    for i in range(0, n_samples, seg):
        end = min(i + seg, n_samples)
        out.append((i, end, max(end - i, min_len) if end == n_samples else end - i))
    return out


def handle_chunks(wav_gen, wav_overlap, overlap_len=1024):
    """inference_utils.py:4-21 (cross-fade of streamed vocoder chunks)."""
    chunk = wav_gen[:-overlap_len].clone()
    if wav_overlap is not None:
        if overlap_len > len(chunk):
            return wav_gen[-overlap_len:], None
        fade_in = chunk[:overlap_len] * torch.linspace(0.0, 1.0, overlap_len)
        chunk[:overlap_len] = wav_overlap * torch.linspace(1.0, 0.0, overlap_len) + fade_in
    return chunk, wav_gen[-overlap_len:]


# ---------------------------------------------------------------------------
# row 13 (callers): the three conversion harnesses of inference/inference_utils.py driven on the oracle's own stages.
# `W` bundles the state dicts: dict(gpt=..., dvae=..., hubert=..., hubert_cfg=..., hifigan=..., vocoder_cfg=..., mel_norms=...,
# dims=..., sampling=..., max_new=...).  Not pinned as a whole (the reference harness needs a checkpoint): every stage is
# pinned on its own and the glue follows the cited lines.
# ---------------------------------------------------------------------------

def _segments(src_wav, seg_len_s, sr=16000):
    """inference_utils.py:43-50 (and :109-116, :158-165): segment tensors, the last one zero-padded to >= 0.32 s."""
    out = []
    for i, end, padded in segment_source(src_wav.shape[-1], seg_len_s, sr):
        seg = src_wav[:, i:end]
        if padded > end - i:
            seg = F.pad(seg, (0, padded - (end - i)), "constant", 0)
        out.append(seg)
    return out


def _segment_codes(W, seg):
    """inference_utils.py:52-53: ContentVec features -> content DVAE codes."""
    feat = hubert_extract_features(W["hubert"], W["hubert_cfg"], seg)
    return dvae_get_codebook_indices(W["dvae"], feat.transpose(1, 2))


def synthesize_utt(W, src_wav, tgt_audio, seg_len=6.0):
    """inference_utils.py:23-89: per segment generate -> strip stop tokens (:68) -> latent re-pass (:71-76); latents
    concatenated (:79) -> x4 interpolation + HiFi-GAN (:81-87).  Returns dict(codes=[...], latents, wav)."""
    cond = get_gpt_cond_latents(W["gpt"], tgt_audio, W["mel_norms"])
    dims, stop = W["dims"], W["dims"]["stop_audio_token"]
    lat_all, codes_all = [], []
    for seg in _segments(src_wav, seg_len):
        codes = _segment_codes(W, seg)
        toks, _, _ = generate(W["gpt"], dims, cond, codes, W["sampling"], max_new=W.get("max_new"))
        gen = toks[0][toks[0] != stop]
        if gen.numel() == 0:
            continue
        lat_all.append(gpt_latents(W["gpt"], dims, cond, codes, gen.unsqueeze(0)))
        codes_all.append(gen)
    latents = torch.cat(lat_all, dim=1)
    return dict(codes=codes_all, latents=latents, wav=vocode_latents(W["hifigan"], W["vocoder_cfg"], latents)[0].squeeze())


def synthesize_utt_streaming(W, src_wav, tgt_audio, seg_len=6.0, stream_chunk_size=8):
    """inference_utils.py:135-217: per segment the (token, latent) stream is cut into groups of stream_chunk_size (the
    EOS-step pair included, :189-196); each group -> x4 interpolation + HiFi-GAN (:196-202) -> handle_chunks (:203-205).
    Returns dict(tokens=[[1,n] per group], latents=[[1,n,d] per group], wav)."""
    cond = get_gpt_cond_latents(W["gpt"], tgt_audio, W["mel_norms"])
    overlap = None
    toks_g, lats_g, pred = [], [], []
    for seg in _segments(src_wav, seg_len):
        codes = _segment_codes(W, seg)
        toks, lats, _ = generate(W["gpt"], W["dims"], cond, codes, W["sampling"], max_new=W.get("max_new"))
        n = toks.shape[1]
        for g0 in range(0, n, stream_chunk_size):
            lat = lats[:, g0:g0 + stream_chunk_size]
            toks_g.append(toks[:, g0:g0 + stream_chunk_size]); lats_g.append(lat)
            wav = vocode_latents(W["hifigan"], W["vocoder_cfg"], lat).squeeze()
            chunk, overlap = handle_chunks(wav, overlap)
            pred.append(chunk)
    return dict(tokens=toks_g, latents=lats_g, wav=torch.cat(pred, -1))


def inference(W, src_seg, cond):
    """HiFiGANTrainer.inference, trainers/hifigan_trainer.py:457-500: one segment -> waveform [1,1,T]."""
    codes = _segment_codes(W, src_seg)
    toks, _, _ = generate(W["gpt"], W["dims"], cond, codes, W["sampling"], max_new=W.get("max_new"))
    gen = toks[0][toks[0] != W["dims"]["stop_audio_token"]]
    lat = gpt_latents(W["gpt"], W["dims"], cond, codes, gen.unsqueeze(0))
    return vocode_latents(W["hifigan"], W["vocoder_cfg"], lat)


def synthesize_utt_chunked(W, src_wav, tgt_audio, seg_len=6.0):
    """inference_utils.py:92-133: per segment `inference` -> waveform-level concatenation through handle_chunks (:127-129)."""
    cond = get_gpt_cond_latents(W["gpt"], tgt_audio, W["mel_norms"])
    overlap, pred = None, []
    for seg in _segments(src_wav, seg_len):
        chunk, overlap = handle_chunks(inference(W, seg, cond).squeeze(), overlap)
        pred.append(chunk)
    return torch.cat(pred, dim=-1)
