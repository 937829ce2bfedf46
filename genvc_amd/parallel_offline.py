"""Batched offline conversion, data-parallel over the GPUs of one node (BASELINE configs[2]).

The reference has no multi-GPU inference; what makes sharding trivial is that utterances -- and the 6 s
segments inside one -- are independent given the reference speaker's conditioning latents
(/root/reference/inference/inference_utils.py:43-77).  One process per GPU (torch.distributed, backend
"nccl" = RCCL over xGMI), weights replicated, rank r converts utterances r, r+W, ...; the ONLY collective
is one all_gather of the padded int32 token ids at the end (~38 KB per rank, latency-bound).
"""
import torch
import torch.distributed as dist
import torch.nn.functional as F

from .inference.inference_utils import segments, _sampling_kwargs


def shard(n_items, rank, world):
    """utterance indices of this rank (round-robin: equal shapes -> balanced)"""
    return list(range(rank, n_items, world))


def gather_token_ids(local, n_total, pad_id, rank, world, group=None):
    """local: int32 [n_local, n_seg, max_len] of the utterances `shard(n_total, rank, world)`;
    returns int32 [n_total, n_seg, max_len] in utterance order on every rank (one all_gather)."""
    per = (n_total + world - 1) // world
    buf = torch.full((per,) + tuple(local.shape[1:]), pad_id, dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    if world == 1:
        return buf[:n_total]
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf, group=group)
    out = torch.full((n_total,) + tuple(local.shape[1:]), pad_id, dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = shard(n_total, r, world)
        out[idx] = parts[r][:len(idx)]
    return out


@torch.inference_mode()
def convert_batch(model, src_wavs, cond_latent, seg_len=6.0, max_len=None, **gen_kwargs):
    """Tokens of a micro-batch of equal-length utterances: segment s of every utterance forms one batched
    generate() call (same prefix length).  Returns int32 [B, n_seg, max_len] padded with the stop token."""
    m = model
    stop = m.gpt.stop_audio_token
    max_len = max_len or m.gpt.max_gen_mel_tokens
    seg = int(seg_len * m.content_sample_rate)
    min_len = int(0.32 * m.content_sample_rate)
    per_utt = [list(segments(w.to(m.device), seg, min_len)) for w in src_wavs]
    n_seg = len(per_utt[0])
    assert all(len(p) == n_seg for p in per_utt), "micro-batch needs equal-length utterances"
    B = len(src_wavs)
    out = torch.full((B, n_seg, max_len), stop, dtype=torch.int32, device=m.device)
    kw = dict(_sampling_kwargs(m))
    kw.update(gen_kwargs)
    for s in range(n_seg):
        wav = torch.cat([p[s] for p in per_utt], 0)
        feat = m.content_extractor.extract_content_features(wav)
        codes = m.content_dvae.get_codebook_indices(feat.transpose(1, 2))
        gen = m.gpt.generate(cond_latent.expand(B, -1, -1).contiguous(), codes, **kw)
        out[:, s, :gen.shape[1]] = gen.to(torch.int32)
    return out


@torch.inference_mode()
def convert_offline(model, src_wavs, ref_audio, seg_len=6.0, micro_batch=8, rank=0, world=1, **gen_kwargs):
    """All utterances of the job, sharded by rank, in waves of `micro_batch`; token ids gathered on every rank."""
    m = model
    cond = m.get_gpt_cond_latents(ref_audio.to(m.device), m.config.audio.sample_rate)
    mine = shard(len(src_wavs), rank, world)
    n_seg = len(list(segments(src_wavs[0], int(seg_len * m.content_sample_rate), int(0.32 * m.content_sample_rate))))
    max_len = m.gpt.max_gen_mel_tokens
    local = torch.full((len(mine), n_seg, max_len), m.gpt.stop_audio_token, dtype=torch.int32, device=m.device)
    for i in range(0, len(mine), micro_batch):
        wave = mine[i:i + micro_batch]
        local[i:i + len(wave)] = convert_batch(m, [src_wavs[j] for j in wave], cond, seg_len, max_len, **gen_kwargs)
    return gather_token_ids(local, len(src_wavs), m.gpt.stop_audio_token, rank, world)
