"""Thin Python owners of the libgenvc_hip contexts.  torch supplies device memory and the stream;
all arithmetic runs in the HIP library."""
import ctypes as C

import torch

from . import _lib
from ._lib import check, lib, ptr, stream


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "need contiguous fp32 CUDA tensor"
    return t


def _i32(t):
    assert t.is_cuda and t.dtype == torch.int32 and t.is_contiguous(), "need contiguous int32 CUDA tensor"
    return t


def sample_params(sampling, vocab, eos, seed=0):
    return _lib.SampleParams(float(sampling["repetition_penalty"]), float(sampling["temperature"]),
                             float(sampling["top_p"]), int(sampling["top_k"]), int(eos), int(vocab), int(seed))


class GptEngine:
    """KV-cached GPT-2 stack of GenVC (reference layers/gpt.py + layers/gpt_inference.py)."""

    def __init__(self, dims, max_slots=8, max_rows=2048, max_seq=None, weight_dtype="fp32"):
        self.dims = dict(dims)
        self.d = dims["d_model"]
        self.V = dims["num_audio_tokens"]
        self.max_slots = max_slots
        max_seq = max_seq or ((dims["max_seq"] + 63) // 64) * 64
        cd = _lib.GptDims(dims["n_layer"], dims["d_model"], dims["n_head"], dims["num_audio_tokens"],
                          dims["max_mel_pos"], dims["max_text_pos"], dims["number_text_tokens"], max_seq,
                          max_slots, max_rows, {"fp32": 0, "bf16": 1, "bf16_kv": 2, "bf16_act": 3}[weight_dtype])
        self._h = C.c_void_p()
        self._pending_side = None     # end-of-work event of another stream of this process (watch_stream)
        self.side_joins = 0
        check(lib().gvc_gpt_create(C.byref(cd), C.byref(self._h)), "gvc_gpt_create")

    def close(self):
        if self._h:
            lib().gvc_gpt_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, weights, prefix=""):
        """weights: name -> CUDA fp32 tensor, named as the reference GPT state dict."""
        for name, t in weights.items():
            if not name.startswith(prefix) or not torch.is_tensor(t) or not t.is_floating_point():
                continue
            t = _f32(t.detach().to(torch.float32).contiguous())
            check(lib().gvc_gpt_bind_weight(self._h, name[len(prefix):].encode(), ptr(t), t.numel(), stream()),
                  f"bind {name}")
        torch.cuda.current_stream().synchronize()      # sources may be temporaries
        missing = lib().gvc_gpt_missing_weights(self._h)
        if missing:
            raise _lib.GenvcHipError(f"{missing} GPT weight tensors missing after bind")

    def prefix_embeddings(self, cond_latents, codes):
        B, n_cond, _ = cond_latents.shape
        Tc = codes.shape[1]
        out = torch.empty(B, n_cond + Tc + 2, self.d, device=cond_latents.device, dtype=torch.float32)
        check(lib().gvc_gpt_prefix_embeddings(self._h, ptr(_f32(cond_latents)), n_cond, ptr(_i32(codes)), B, Tc,
                                              self.dims["start_text_token"], self.dims["stop_text_token"],
                                              ptr(out), stream()), "prefix_embeddings")
        return out

    def watch_stream(self, event):
        """Residency before issue: the one-launch steps need all 256 workgroups co-resident, so work of this process that is still in
        flight on ANOTHER stream (the conditioning side stream of model_init._CondFuture: mel + Perceiver, ~25 short launches) must not
        overlap them -- a missing workgroup costs a ~0.2 s bounded spin and a fallback (include/genvc_hip.h: gvc_gpt_health).  The owner of
        such a stream registers the event that marks the end of its work here; the next prefill / decode_step / generate / latents call
        makes the caller's stream wait for it first (`side_joins` counts the calls that really had to wait)."""
        self._pending_side = event

    def _join_side(self):
        ev = self._pending_side
        if ev is not None:
            self._pending_side = None
            if not ev.query():
                torch.cuda.current_stream().wait_event(ev)
                self.side_joins += 1

    def prefill(self, slots, prefix_emb, want_outputs=True, n_cached=0):
        """n_cached > 0: the first n_cached rows of the prefix (the conditioning latents) are already in the slots' KV
        cache from an earlier prefill with the same leading rows -- only the rest is computed (bit-identical results)"""
        self._join_side()
        B, P, _ = prefix_emb.shape
        logits = latent = None
        if want_outputs:
            logits = torch.empty(B, self.V, device=prefix_emb.device, dtype=torch.float32)
            latent = torch.empty(B, self.d, device=prefix_emb.device, dtype=torch.float32)
        check(lib().gvc_gpt_prefill_cached(self._h, ptr(_i32(slots)), B, ptr(_f32(prefix_emb)), P, int(n_cached),
                                           self.dims["start_audio_token"], ptr(logits), ptr(latent), stream()), "prefill")
        return logits, latent

    def prefill_cond(self, slots, cond_latents):
        """the conditioning rows alone into the slots' KV caches (include/genvc_hip.h: gvc_gpt_prefill_cond): later prefills of these slots
        pass n_cached = cond_latents.shape[1]"""
        self._join_side()
        B, n, _ = cond_latents.shape
        check(lib().gvc_gpt_prefill_cond(self._h, ptr(_i32(slots)), B, ptr(_f32(cond_latents)), n, stream()), "prefill_cond")

    def decode_step(self, slots, tok, logits=None, latent=None):
        self._join_side()
        B = slots.shape[0]
        if logits is None:
            logits = torch.empty(B, self.V, device=slots.device, dtype=torch.float32)
            latent = torch.empty(B, self.d, device=slots.device, dtype=torch.float32)
        check(lib().gvc_gpt_decode_step(self._h, ptr(_i32(slots)), B, ptr(_i32(tok)), ptr(logits), ptr(latent),
                                        stream()), "decode_step")
        return logits, latent

    def reset(self, slots):
        check(lib().gvc_gpt_reset_slots(self._h, ptr(_i32(slots)), slots.shape[0], stream()), "reset_slots")

    def latents(self, slots, prefix_emb, gen_codes):
        self._join_side()
        B, P, _ = prefix_emb.shape
        n = gen_codes.shape[1]
        out = torch.empty(B, n, self.d, device=prefix_emb.device, dtype=torch.float32)
        check(lib().gvc_gpt_latents(self._h, ptr(_i32(slots)), B, ptr(_f32(prefix_emb)), P, ptr(_i32(gen_codes)), n,
                                    self.dims["start_audio_token"], self.dims["stop_audio_token"], ptr(out),
                                    stream()), "latents")
        return out

    def sample(self, logits, ids, ids_len, finished, params, step):
        B = logits.shape[0]
        tok = torch.empty(B, device=logits.device, dtype=torch.int32)
        check(lib().gvc_sample(ptr(_f32(logits)), B, ptr(_i32(ids)), ids.shape[1], ptr(_i32(ids_len)),
                               ptr(_i32(finished)), C.byref(params), step, ptr(tok), stream()), "sample")
        return tok

    def generate(self, slots, ids, ids_len, finished, params, i0, n_steps, tokens_out, latents_out, max_keys=0):
        """tokens_out [B, >= i0+n_steps] int32 and latents_out [B, >= i0+n_steps, d] may be column slices of larger
        buffers (row strides are passed on); step i of this call lands in column i0 + i.  max_keys: cached positions of the
        longest stream after the call (0: unknown, the width of `ids` is taken)."""
        self._join_side()
        B = slots.shape[0]
        assert tokens_out.is_cuda and tokens_out.dtype == torch.int32 and tokens_out.stride(1) == 1
        lat_stride = 0
        if latents_out is not None:
            assert latents_out.is_cuda and latents_out.dtype == torch.float32 and latents_out.stride(2) == 1
            assert latents_out.stride(1) == self.d and latents_out.stride(0) % self.d == 0
            lat_stride = latents_out.stride(0) // self.d
        check(lib().gvc_gpt_generate(self._h, ptr(_i32(slots)), B, ptr(_i32(ids)), ids.shape[1], ptr(_i32(ids_len)),
                                     ptr(_i32(finished)), C.byref(params), i0, n_steps, int(max_keys), ptr(tokens_out),
                                     tokens_out.stride(0), ptr(latents_out), lat_stride, stream()), "generate")

    def decode_variant(self):
        """which decode step the last generate() call replayed (include/genvc_hip.h: gvc_gpt_decode_variant)"""
        return int(lib().gvc_gpt_decode_variant(self._h))

    def health(self):
        """after a synchronisation: raises if the work just finished hit a hand-off timeout (the context then continues on the
        launch-per-phase paths) or a full KV cache (include/genvc_hip.h: gvc_gpt_health)"""
        check(lib().gvc_gpt_health(self._h), "health")

    def warmup(self, B=1, max_keys=0, top_k=1):
        """everything the first generate / decode_step / cached prefill of this shape would do on first use (buffers, weight pack,
        topology probe, step-graph capture); afterwards such calls neither allocate nor synchronise (include/genvc_hip.h: gvc_gpt_warmup)"""
        check(lib().gvc_gpt_warmup(self._h, int(B), int(max_keys), int(top_k)), "warmup")

    def lazy_inits(self):
        """allocations / device syncs / graph captures done inside data-path calls so far (include/genvc_hip.h: gvc_gpt_lazy_inits)"""
        return int(lib().gvc_gpt_lazy_inits(self._h))

    def warmup_range(self, B, min_keys, max_keys, top_k=1):
        """warm-up of every context class a generation passes through while its longest stream grows from min_keys to max_keys cached
        positions (include/genvc_hip.h: gvc_gpt_warmup_range -- the library enumerates its own classes)"""
        check(lib().gvc_gpt_warmup_range(self._h, B, min_keys, max_keys, top_k), "warmup_range")

    def rearm(self):
        """back to the one-launch steps after a time-out fallback, when the GPU is the caller's own again (include/genvc_hip.h: gvc_gpt_rearm)"""
        check(lib().gvc_gpt_rearm(self._h), "rearm")

    def rows_step_launches(self):
        """one-launch rows steps issued so far (include/genvc_hip.h: gvc_gpt_rows_step_launches)"""
        return int(lib().gvc_gpt_rows_step_launches(self._h))

    def time_kernel(self, which, slots, tok, n_steps):
        """(mean us per launch, launches) of one kernel class of the decode step, launched back to back"""
        avg, n = C.c_float(), C.c_int32()
        check(lib().gvc_gpt_time_kernel(self._h, which, ptr(_i32(slots)), slots.shape[0], ptr(_i32(tok)), n_steps,
                                        C.byref(avg), C.byref(n), stream()), "time_kernel")
        return avg.value, n.value


class PerceiverEngine:
    """PerceiverResampler.forward (reference layers/perceiver_encoder.py:265-276)."""

    def __init__(self, dim, depth=2, dim_context=None, num_latents=32, dim_head=64, heads=8, ff_mult=4,
                 max_batch=8, max_frames=2816):
        dim_context = dim if dim_context is None else dim_context
        self.dim, self.num_latents, self.dim_context = dim, num_latents, dim_context
        cd = _lib.PerceiverDims(dim, depth, dim_context, num_latents, dim_head, heads, ff_mult, max_batch, max_frames)
        self._h = C.c_void_p()
        check(lib().gvc_perceiver_create(C.byref(cd), C.byref(self._h)), "gvc_perceiver_create")

    def close(self):
        if self._h:
            lib().gvc_perceiver_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, weights, prefix=""):
        for name, t in weights.items():
            if not name.startswith(prefix) or not torch.is_tensor(t):
                continue
            t = _f32(t.detach().to(torch.float32).contiguous())
            check(lib().gvc_perceiver_bind_weight(self._h, name[len(prefix):].encode(), ptr(t), t.numel(), stream()),
                  f"bind {name}")
        torch.cuda.current_stream().synchronize()
        missing = lib().gvc_perceiver_missing_weights(self._h)
        if missing:
            raise _lib.GenvcHipError(f"{missing} Perceiver weight tensors missing after bind")

    def forward(self, x):
        """x [B,F,dim_context] -> [B,num_latents,dim]"""
        B, F, _ = x.shape
        out = torch.empty(B, self.num_latents, self.dim, device=x.device, dtype=torch.float32)
        check(lib().gvc_perceiver_forward(self._h, ptr(_f32(x)), B, F, ptr(out), stream()), "perceiver_forward")
        return out


class MelEngine:
    """TorchMelSpectrogram.forward (reference utils.py:150-162)."""

    def __init__(self, mel_norms, n_fft=2048, hop=256, win=1024, sample_rate=24000, f_min=0.0, f_max=8000.0, n_mels=80):
        import numpy as np
        norms = np.ascontiguousarray(np.asarray(mel_norms, dtype=np.float32))
        assert norms.shape == (n_mels,)
        self.hop, self.n_mels = hop, n_mels
        self._h = C.c_void_p()
        check(lib().gvc_mel_create(n_fft, hop, win, sample_rate, f_min, f_max, n_mels,
                                   norms.ctypes.data_as(_lib.c_f32p), C.byref(self._h)), "gvc_mel_create")

    def close(self):
        if self._h:
            lib().gvc_mel_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, wav, frames_major=False):
        """wav [B,T] -> [B,n_mels,1+T//hop]  (and [B,1+T//hop,n_mels] when frames_major)"""
        B, T = wav.shape
        nf = 1 + T // self.hop
        out = torch.empty(B, self.n_mels, nf, device=wav.device, dtype=torch.float32)
        fm = torch.empty(B, nf, self.n_mels, device=wav.device, dtype=torch.float32) if frames_major else None
        check(lib().gvc_mel_forward(self._h, ptr(_f32(wav)), B, T, ptr(out), ptr(fm), stream()), "mel_forward")
        return (out, fm) if frames_major else out


class DvaeEngine:
    """DiscreteVAE.get_codebook_indices (reference layers/dvae.py:324-331)."""

    def __init__(self, cfg, max_batch=8, max_frames=1504):
        self.cfg = dict(cfg)
        cd = _lib.DvaeDims(cfg["num_channels"], cfg["hidden_dim"], cfg["num_layers"], cfg["num_resnet_blocks"],
                           cfg["kernel_size"], cfg["codebook_dim"], cfg["num_tokens"], max_batch, max_frames)
        self._h = C.c_void_p()
        check(lib().gvc_dvae_create(C.byref(cd), C.byref(self._h)), "gvc_dvae_create")

    def close(self):
        if self._h:
            lib().gvc_dvae_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, weights, prefix=""):
        for name, t in weights.items():
            if not name.startswith(prefix) or not torch.is_tensor(t) or not t.is_floating_point():
                continue
            t = _f32(t.detach().to(torch.float32).contiguous())
            check(lib().gvc_dvae_bind_weight(self._h, name[len(prefix):].encode(), ptr(t), t.numel(), stream()),
                  f"bind {name}")
        torch.cuda.current_stream().synchronize()
        missing = lib().gvc_dvae_missing_weights(self._h)
        if missing:
            raise _lib.GenvcHipError(f"{missing} DVAE weight tensors missing after bind")

    def out_frames(self, T):
        k = self.cfg["kernel_size"]
        for _ in range(self.cfg["num_layers"]):
            T = (T + 2 * ((k - 1) // 2) - k) // 2 + 1
        return T

    def encode(self, feat, return_enc=False, frames_major=False):
        """feat [B,C,T] (or [B,T,C] with frames_major) -> int32 codes [B,Tc] (and the encoder output [B,Tc,codebook_dim])"""
        if frames_major:
            B, T, _ = feat.shape
        else:
            B, _, T = feat.shape
        Tc = self.out_frames(T)
        codes = torch.empty(B, Tc, device=feat.device, dtype=torch.int32)
        enc = torch.empty(B, Tc, self.cfg["codebook_dim"], device=feat.device, dtype=torch.float32) if return_enc else None
        fn = lib().gvc_dvae_encode_frames if frames_major else lib().gvc_dvae_encode
        check(fn(self._h, ptr(_f32(feat)), B, T, ptr(codes), ptr(enc), stream()), "dvae_encode")
        return (codes, enc) if return_enc else codes


def vq_argmin(x, embed):
    """x [N,dim], embed [dim,n_embed] -> int32 [N] (Quantize.forward, reference layers/dvae.py:87-90)."""
    N, dim = x.shape
    idx = torch.empty(N, device=x.device, dtype=torch.int32)
    work = torch.empty(N * embed.shape[1], device=x.device, dtype=torch.float32)
    check(lib().gvc_vq_argmin(ptr(_f32(x)), ptr(_f32(embed)), N, dim, embed.shape[1], ptr(idx), ptr(work), stream()),
          "vq_argmin")
    return idx


class HifiganEngine:
    """HiFiGAN.forward (reference layers/hifigan.py:218-233) and the latent entry point with the x4 interpolation."""

    def __init__(self, cfg, max_batch=2, max_frames=2560):
        self.cfg = dict(cfg)
        d = _lib.HifiganDims()
        d.in_dim, d.up_init_ch = cfg["input_feat_dim"], cfg["upsample_initial_channel"]
        d.n_ups, d.n_kernels = len(cfg["upsample_rates"]), len(cfg["resblock_kernel_sizes"])
        for i, (r, k) in enumerate(zip(cfg["upsample_rates"], cfg["upsample_kernel_sizes"])):
            d.up_rates[i], d.up_kernels[i] = r, k
        for j, (k, dl) in enumerate(zip(cfg["resblock_kernel_sizes"], cfg["resblock_dilation_sizes"])):
            d.res_kernels[j] = k
            d.res_dilations[j][0], d.res_dilations[j][1] = dl
        d.max_batch, d.max_frames = max_batch, max_frames
        self.max_frames = max_frames
        self.total_up = 1
        for r in cfg["upsample_rates"]:
            self.total_up *= r
        self.in_dim = cfg["input_feat_dim"]
        self._h = C.c_void_p()
        check(lib().gvc_hifigan_create(C.byref(d), C.byref(self._h)), "gvc_hifigan_create")

    def close(self):
        if self._h:
            lib().gvc_hifigan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, weights, prefix=""):
        """weights: reference state dict; weight-norm pairs (weight_g, weight_v) are folded here (loader plumbing)."""
        sd = {k[len(prefix):]: v for k, v in weights.items() if k.startswith(prefix) and torch.is_tensor(v)}
        folded = {}
        for k, v in sd.items():
            if k.endswith(".weight_v"):
                g = sd[k[:-2] + "_g"]
                norm = v.float().pow(2).sum(dim=tuple(range(1, v.dim())), keepdim=True).sqrt()
                folded[k[:-9] + ".weight"] = (g.float() * v.float() / norm)
            elif k.endswith(".weight_g"):
                continue
            else:
                folded[k] = v
        for name, t in folded.items():
            t = _f32(t.detach().to(torch.float32).contiguous())
            check(lib().gvc_hifigan_bind_weight(self._h, name.encode(), ptr(t), t.numel(), stream()), f"bind {name}")
        torch.cuda.current_stream().synchronize()
        missing = lib().gvc_hifigan_missing_weights(self._h)
        if missing:
            raise _lib.GenvcHipError(f"{missing} HiFi-GAN weight tensors missing after bind")

    def forward(self, x):
        """x [B,in_dim,T] -> [B,1,T*256]"""
        B, _, T = x.shape
        wav = torch.empty(B, 1, T * self.total_up, device=x.device, dtype=torch.float32)
        check(lib().gvc_hifigan_forward(self._h, ptr(_f32(x)), B, T, ptr(wav), stream()), "hifigan_forward")
        return wav

    def forward_latents(self, latents, scale=4):
        """latents [B,n,in_dim] -> [B,1,n*scale*256]  (F.interpolate(scale, 'linear') fused in front)"""
        B, n, _ = latents.shape
        wav = torch.empty(B, 1, n * scale * self.total_up, device=latents.device, dtype=torch.float32)
        check(lib().gvc_hifigan_forward_latents(self._h, ptr(_f32(latents)), B, n, scale, ptr(wav), stream()),
              "hifigan_forward_latents")
        return wav


class HubertEngine:
    """ContentVecExtractor.extract_content_features (reference layers/content_processor.py:17-31): HuBERT-base
    extract_features(output_layer=n_layers) + final_proj, weights under fairseq's names."""

    def __init__(self, cfg, max_batch=2, max_samples=16000 * 30):
        self.cfg = dict(cfg)
        d = _lib.HubertDims()
        d.n_conv = len(cfg["conv_layers"])
        for i, (c, k, s) in enumerate(cfg["conv_layers"]):
            d.conv_dim[i], d.conv_kernel[i], d.conv_stride[i] = c, k, s
        d.embed_dim, d.n_layers, d.n_heads, d.ffn_dim = cfg["embed_dim"], cfg["layers"], cfg["heads"], cfg["ffn_dim"]
        d.pos_conv_kernel, d.pos_conv_groups, d.final_dim = cfg["pos_conv_kernel"], cfg["pos_conv_groups"], cfg["final_dim"]
        d.max_batch, d.max_samples = max_batch, max_samples
        self.final_dim = cfg["final_dim"]
        self._h = C.c_void_p()
        check(lib().gvc_hubert_create(C.byref(d), C.byref(self._h)), "gvc_hubert_create")

    def close(self):
        if self._h:
            lib().gvc_hubert_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def bind(self, weights, prefix=""):
        """weights: fairseq-named state dict; the weight-normed positional conv (weight_g, weight_v; dim=2) is
        folded here (loader plumbing, as torch's remove_weight_norm would)."""
        sd = {k[len(prefix):]: v for k, v in weights.items() if k.startswith(prefix) and torch.is_tensor(v)}
        pc = "encoder.pos_conv.0."
        if pc + "weight_v" in sd:
            v, g = sd.pop(pc + "weight_v").float(), sd.pop(pc + "weight_g").float()
            sd[pc + "weight"] = g * v / v.pow(2).sum(dim=(0, 1), keepdim=True).sqrt()
        for name, t in sd.items():
            if not t.is_floating_point():
                continue
            t = _f32(t.detach().to(torch.float32).contiguous())
            check(lib().gvc_hubert_bind_weight(self._h, name.encode(), ptr(t), t.numel(), stream()), f"bind {name}")
        torch.cuda.current_stream().synchronize()
        missing = lib().gvc_hubert_missing_weights(self._h)
        if missing:
            raise _lib.GenvcHipError(f"{missing} HuBERT weight tensors missing after bind")

    def frames(self, n_samples):
        return lib().gvc_hubert_frames(self._h, int(n_samples))

    def forward(self, wav):
        """wav [B,T] 16 kHz -> [B,T50,final_dim]"""
        B, T = wav.shape
        n = self.frames(T)
        if n < 1:
            raise ValueError(f"{T} samples are too short for the HuBERT conv stack")
        out = torch.empty(B, n, self.final_dim, device=wav.device, dtype=torch.float32)
        check(lib().gvc_hubert_forward(self._h, ptr(_f32(wav)), B, T, ptr(out), stream()), "hubert_forward")
        return out


def resample(wav, orig_sr, new_sr):
    """wav [B,T] (CUDA) -> [B, ceil(T*new/orig)]: torchaudio.functional.resample defaults (reference utils.py:58-62)"""
    B, T = wav.shape
    n = lib().gvc_resample_length(T, int(orig_sr), int(new_sr))
    out = torch.empty(B, n, device=wav.device, dtype=torch.float32)
    check(lib().gvc_resample(ptr(_f32(wav)), B, T, int(orig_sr), int(new_sr), ptr(out), stream()), "resample")
    return out
