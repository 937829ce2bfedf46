"""Inference harness with the reference's function names and semantics
(/root/reference/inference/inference_utils.py): segmentation (:34-50), per-segment independence,
stop-token stripping (:68), streaming groups of `stream_chunk_size` tokens with the EOS-step latent
included (:187-207), cross-fade of vocoder chunks (:4-21), latency / RTF bookkeeping (:148, 208-216).

The vocoder call (x4 linear interpolation + HiFi-GAN) runs on libgenvc_hip; when
`model.hifigan` is None the functions return the acoustic latents instead of a waveform.
"""
import time

import torch
import torch.nn.functional as F


@torch.inference_mode()
def handle_chunks(wav_gen, wav_gen_prev, wav_overlap, overlap_len=1024):
    """reference :4-21"""
    wav_chunk = wav_gen[:-overlap_len]
    if wav_overlap is not None:
        if overlap_len > len(wav_chunk):
            return wav_gen[-overlap_len:], wav_gen, None
        fade_in = wav_chunk[:overlap_len] * torch.linspace(0.0, 1.0, overlap_len, device=wav_chunk.device)
        wav_chunk[:overlap_len] = wav_overlap * torch.linspace(1.0, 0.0, overlap_len, device=wav_overlap.device)
        wav_chunk[:overlap_len] += fade_in
    return wav_chunk, wav_gen, wav_gen[-overlap_len:]


def segments(src_wav, seg_len, min_len):
    """reference :43-50: fixed-length segments, the last one zero-padded to at least 0.32 s"""
    total = src_wav.shape[-1]
    for i in range(0, total, seg_len):
        if i + seg_len >= total:
            seg = src_wav[:, i:]
            if seg.shape[-1] < min_len:
                seg = F.pad(seg, (0, min_len - seg.shape[-1]), "constant", 0)
        else:
            seg = src_wav[:, i:i + seg_len]
        yield seg


def _sampling_kwargs(model):
    c = model.config
    return dict(top_p=c.top_p, top_k=c.top_k, temperature=c.temperature, length_penalty=c.length_penalty,
                repetition_penalty=c.repetition_penalty, do_sample=True, num_beams=1)


def _vocode(model, latents):
    """latents [1,n,d] -> waveform: F.interpolate(scale_factor=hifigan_scale_factor, mode='linear') + HiFi-GAN
    (reference :81-87, :196-202); both run inside one library call.  None without a vocoder."""
    if model.hifigan is None:
        return None
    scale = int(model.hifigan_scale_factor)
    if hasattr(model.hifigan, "forward_latents") and scale == model.hifigan_scale_factor:
        return model.hifigan.forward_latents(latents, scale)
    mel_input = F.interpolate(latents.transpose(1, 2), scale_factor=[model.hifigan_scale_factor], mode="linear").squeeze(1)
    return model.hifigan(mel_input)


@torch.inference_mode()
def _segment_latents(m, cond_latent, codes, gen, repass_latents):
    """the acoustic latents of one segment's n non-stop tokens.  The reference recomputes them with a second, teacher-forced forward pass
    over [cond | codes | start, gen, stop x 4] and trims it with `sub = -5` (inference_utils.py:71-76, gpt.py:375-508, :491, :508); row i of
    that pass is the hidden state that predicted token i -- the very vector the decode loop already produced at step i
    (stream_generator.py:865; SURVEY.md 8a row 12: equal to <= 4.8e-7).  Default: reuse the decode-time latents of `GPT.generate`
    (`last_latents`, the EOS-step latent dropped: the re-pass has n rows); `repass_latents=True` runs the reference's second pass."""
    n = int(gen.shape[-1])
    if not repass_latents and getattr(m.gpt, "last_latents", None) is not None and m.gpt.last_latents.shape[1] >= n:
        return m.gpt.last_latents[:1, :n]
    out_len = torch.tensor([n * m.config.model_args.gpt_code_stride_len], device=m.device)
    clen = torch.tensor([codes.shape[-1]], device=m.device)
    return m.gpt(codes, clen, gen.unsqueeze(0), out_len, cond_latents=cond_latent, return_latent=True)


@torch.inference_mode()
def synthesize_utt(genVC_mdl, src_wav, tgt_audio, seg_len=6.0, return_details=False, repass_latents=False):
    """non-streaming conversion, latent-level concatenation (reference :23-89)"""
    m = genVC_mdl
    min_len = int(0.32 * m.content_sample_rate)
    src_wav = src_wav.to(m.device)
    seg = int(seg_len * m.content_sample_rate)
    cond_latent = m.get_gpt_cond_latents(tgt_audio.to(m.device), m.config.audio.sample_rate)
    final_latents, all_codes = [], []
    for src_seg in segments(src_wav, seg, min_len):
        feat = m.content_extractor.extract_content_features(src_seg)
        codes = m.content_dvae.get_codebook_indices(feat.transpose(1, 2))
        gen = m.gpt.generate(cond_latent, codes, output_attentions=False, **_sampling_kwargs(m))[0]
        gen = gen[gen != m.gpt.stop_audio_token]                        # reference :68 (0-d collapse guarded)
        if gen.numel() == 0:
            continue
        final_latents.append(_segment_latents(m, cond_latent, codes, gen, repass_latents))
        all_codes.append(gen)
    if not final_latents:                    # every segment ended on its first token: nothing to vocode
        empty = torch.zeros(0, device=m.device)
        return dict(latents=None, codes=[], wav=empty) if return_details else empty
    latents = torch.cat(final_latents, dim=1)
    wav = _vocode(m, latents)
    if return_details or wav is None:
        return dict(latents=latents, codes=all_codes, wav=None if wav is None else wav[0].squeeze())
    return wav[0].squeeze()


@torch.inference_mode()
def synthesize_utt_chunked(genVC_mdl, src_wav, tgt_audio, seg_len=6.0, repass_latents=False):
    """non-streaming conversion with waveform-level concatenation (reference :92-133): every segment goes through
    `genVC_mdl.inference` (trainers/hifigan_trainer.py:457-500) and the segment waveforms are joined by `handle_chunks`
    (1024 samples dropped from each, cross-fade over the previous tail)."""
    m = genVC_mdl
    wav_gen_prev, wav_overlap = None, None
    pred_audios = []
    min_len = int(0.32 * m.content_sample_rate)
    src_wav = src_wav.to(m.device)
    seg = int(seg_len * m.content_sample_rate)
    cond_latent = m.get_gpt_cond_latents(tgt_audio.to(m.device), m.config.audio.sample_rate)
    c = m.config
    for src_seg in segments(src_wav, seg, min_len):
        audio_pred = m.inference(src_seg, cond_latent, top_p=c.top_p, top_k=c.top_k, temperature=c.temperature,
                                 length_penalty=c.length_penalty, repetition_penalty=c.repetition_penalty, repass_latents=repass_latents)
        wav_chunk, wav_gen_prev, wav_overlap = handle_chunks(audio_pred.squeeze(), wav_gen_prev, wav_overlap, 1024)
        pred_audios.append(wav_chunk)
    return torch.cat(pred_audios, dim=-1)


@torch.inference_mode()
def synthesize_utt_streaming(genVC_mdl, src_wav, tgt_audio, seg_len=6.0, stream_chunk_size=8, verbose=True,
                             return_details=False):
    """streaming conversion (reference :135-217); the clock starts before the host->device copies (:148)"""
    m = genVC_mdl
    wav_gen_prev, wav_overlap = None, None
    total = src_wav.shape[-1]
    pred, chunks_lat, tokens = [], [], []
    min_len = int(0.32 * m.content_sample_rate)
    begin = time.time()
    latency = None
    src_wav = src_wav.to(m.device)
    seg = int(seg_len * m.content_sample_rate)
    # the reference computes the conditioning latents first (:152-153); they do not depend on the source, so here their mel + Perceiver
    # chain runs on a second stream beside the first segment's ContentVec + DVAE chain (same arithmetic, shorter first-chunk latency)
    # -- and its ~25 launches are enqueued AFTER the first segment's ContentVec + DVAE launches (the side stream only waits for the uploads):
    # the GPU starts on the source while the host is still busy with the conditioning chain
    tgt_dev = tgt_audio.to(m.device)
    can_async = hasattr(m, "get_gpt_cond_latents_async")
    uploaded = torch.cuda.Event() if can_async else None
    if uploaded is not None:
        uploaded.record()
    cond_future = None
    cond_latent = None if can_async else m.get_gpt_cond_latents(tgt_dev, m.config.audio.sample_rate)
    cached = 0          # prefix caching: after the first segment the conditioning rows are already in the KV cache
    for src_seg in segments(src_wav, seg, min_len):
        feat = m.content_extractor.extract_content_features(src_seg)
        codes = m.content_dvae.get_codebook_indices(feat.transpose(1, 2))
        if cond_latent is None:
            if cond_future is None:
                cond_future = m.get_gpt_cond_latents_async(tgt_dev, m.config.audio.sample_rate, after=uploaded)
            cond_latent = cond_future.result()
        fake = m.gpt.compute_embeddings(cond_latent, codes)
        gen = m.gpt.get_generator(fake_inputs=fake, num_return_sequences=1, output_attentions=False,
                                  output_hidden_states=True, stream_group=max(stream_chunk_size, 1),
                                  cached_cond_rows=cached, **_sampling_kwargs(m))
        cached = cond_latent.shape[1]
        last_tokens, all_latents = [], []
        is_end = False
        while not is_end:
            try:
                x, latent = next(gen)
                last_tokens.append(x)
                all_latents.append(latent)
            except StopIteration:
                is_end = True
            # DIVERGENCE from the reference, on purpose: inference_utils.py:192-196 runs `torch.cat(all_latents)` whenever `is_end` is
            # set, so a segment whose token count is a multiple of stream_chunk_size (the last group was flushed by the count rule,
            # then StopIteration arrives with nothing pending) crashes there on torch.cat([]).  Here an empty tail is skipped.
            if (is_end and all_latents) or (stream_chunk_size > 0 and len(last_tokens) >= stream_chunk_size):
                acoustic = torch.cat(all_latents, dim=0)[None, :]       # EOS-step latent included (:189-196)
                chunks_lat.append(acoustic)
                tokens.append(torch.stack(last_tokens, 1))
                audio = _vocode(m, acoustic)
                if audio is not None:
                    wav_chunk, wav_gen_prev, wav_overlap = handle_chunks(audio.squeeze(), wav_gen_prev, wav_overlap, 1024)
                    pred.append(wav_chunk)
                last_tokens, all_latents = [], []
                if latency is None:
                    torch.cuda.synchronize()
                    latency = time.time() - begin
                    if verbose:
                        print(f"Latency: {latency:.3f}s")
    if cond_latent is None and cond_future is not None:
        cond_future.result()          # no segment consumed the latents: still join the side stream (its scratch buffers are per context)
    torch.cuda.synchronize()
    rtf = (time.time() - begin) / (total / m.content_sample_rate)
    if verbose:
        print(f"Real-time factor: {rtf:.3f}")
    if return_details or not pred:
        return dict(wav=torch.cat(pred, -1) if pred else None, latents=chunks_lat, tokens=tokens, latency=latency, rtf=rtf)
    return torch.cat(pred, dim=-1)


@torch.inference_mode()
def synthesize_streams_streaming(genVC_mdl, src_wavs, tgt_audios, seg_len=1.0, stream_chunk_size=8, return_details=True):
    """BASELINE configs[3]: B concurrent streams stepped together.  Each stream is converted exactly as
    `synthesize_utt_streaming` converts it on its own (same segments, same per-stream EOS rule, same vocoder grouping and
    cross-fade); the streams only SHARE the launches: ContentVec / DVAE / prefill / vocoder run on the batch and one
    decode step serves every stream (the MFMA rows path from 5 streams up).

    src_wavs [B,T] (equal lengths), tgt_audios [B,Tr] or [1,Tr] (one reference for all).
    Returns per-stream lists: wav [B] tensors, tokens, plus first-chunk latency and the batch RTF."""
    m = genVC_mdl
    B, total = src_wavs.shape
    min_len = int(0.32 * m.content_sample_rate)
    begin = time.time()
    latency = None
    src_wavs = src_wavs.to(m.device)
    seg = int(seg_len * m.content_sample_rate)
    sr = m.config.audio.sample_rate
    tgt = tgt_audios.to(m.device)
    conds = [m.get_gpt_cond_latents(tgt[i:i + 1], sr) for i in range(tgt.shape[0])]
    cond_latent = torch.cat(conds if len(conds) == B else conds * B, 0)
    stop = m.gpt.stop_audio_token
    prev, overlap = [None] * B, [None] * B
    pred, tokens = [[] for _ in range(B)], [[] for _ in range(B)]
    cached = 0
    for src_seg in segments(src_wavs, seg, min_len):
        feat = m.content_extractor.extract_content_features(src_seg)
        codes = m.content_dvae.get_codebook_indices(feat.transpose(1, 2))
        fake = m.gpt.compute_embeddings(cond_latent, codes)
        gen = m.gpt.get_generator(fake_inputs=fake, num_return_sequences=1, output_attentions=False,
                                  output_hidden_states=True, stream_group=max(stream_chunk_size, 1),
                                  cached_cond_rows=cached, **_sampling_kwargs(m))
        cached = cond_latent.shape[1]
        alive = [True] * B                        # a stream stops with its own EOS step (latent included, :189-196)
        g_tok, g_lat = [], []
        is_end = False
        while not is_end:
            try:
                x, latent = next(gen)
                g_tok.append(x)
                g_lat.append(latent)
            except StopIteration:
                is_end = True
            if (is_end and g_tok) or (stream_chunk_size > 0 and len(g_tok) >= stream_chunk_size):
                toks = torch.stack(g_tok, 1)                                  # [B,n]
                lats = torch.stack(g_lat, 1)                                  # [B,n,d]
                toks_h = toks.cpu()
                keep = []
                for b in range(B):
                    nb = 0
                    if alive[b]:
                        row = toks_h[b]
                        hit = (row == stop).nonzero()
                        nb = int(hit[0]) + 1 if hit.numel() else row.shape[0]
                        if hit.numel():
                            alive[b] = False
                        tokens[b].append(toks[b:b + 1, :nb])
                    keep.append(nb)
                # streams with the same group length share one vocoder call
                for nb in sorted(set(k for k in keep if k > 0)):
                    idx = [b for b in range(B) if keep[b] == nb]
                    audio = _vocode(m, lats[idx, :nb].contiguous())
                    if audio is None:
                        continue
                    for j, b in enumerate(idx):
                        chunk, prev[b], overlap[b] = handle_chunks(audio[j].squeeze(), prev[b], overlap[b], 1024)
                        pred[b].append(chunk)
                g_tok, g_lat = [], []
                if latency is None:
                    torch.cuda.synchronize()
                    latency = time.time() - begin
    torch.cuda.synchronize()
    rtf = (time.time() - begin) / (total / m.content_sample_rate)
    return dict(wav=[torch.cat(p, -1) if p else None for p in pred], tokens=tokens, latency=latency, rtf=rtf)
