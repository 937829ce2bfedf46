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
def synthesize_utt(genVC_mdl, src_wav, tgt_audio, seg_len=6.0, return_details=False):
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
        out_len = torch.tensor([gen.shape[-1] * m.config.model_args.gpt_code_stride_len], device=m.device)
        clen = torch.tensor([codes.shape[-1]], device=m.device)
        final_latents.append(m.gpt(codes, clen, gen.unsqueeze(0), out_len, cond_latents=cond_latent, return_latent=True))
        all_codes.append(gen)
    latents = torch.cat(final_latents, dim=1)
    wav = _vocode(m, latents)
    if return_details or wav is None:
        return dict(latents=latents, codes=all_codes, wav=None if wav is None else wav[0].squeeze())
    return wav[0].squeeze()


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
    cond_latent = m.get_gpt_cond_latents(tgt_audio.to(m.device), m.config.audio.sample_rate)
    for src_seg in segments(src_wav, seg, min_len):
        feat = m.content_extractor.extract_content_features(src_seg)
        codes = m.content_dvae.get_codebook_indices(feat.transpose(1, 2))
        fake = m.gpt.compute_embeddings(cond_latent, codes)
        gen = m.gpt.get_generator(fake_inputs=fake, num_return_sequences=1, output_attentions=False,
                                  output_hidden_states=True, stream_group=max(stream_chunk_size, 1),
                                  **_sampling_kwargs(m))
        last_tokens, all_latents = [], []
        is_end = False
        while not is_end:
            try:
                x, latent = next(gen)
                last_tokens.append(x)
                all_latents.append(latent)
            except StopIteration:
                is_end = True
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
    torch.cuda.synchronize()
    rtf = (time.time() - begin) / (total / m.content_sample_rate)
    if verbose:
        print(f"Real-time factor: {rtf:.3f}")
    if return_details or not pred:
        return dict(wav=torch.cat(pred, -1) if pred else None, latents=chunks_lat, tokens=tokens, latency=latency, rtf=rtf)
    return torch.cat(pred, dim=-1)
