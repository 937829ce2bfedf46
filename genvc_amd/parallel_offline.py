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


def plan(lengths, rank, world):
    """Utterance indices of this rank.  Utterances are ordered by length (longest first; ties by index) and dealt round-robin:
    equal shapes stay balanced, unequal ones are balanced to within one utterance per length class, and
    neighbours in a rank's list have similar lengths, so a micro-batch shares as many batched calls as possible."""
    order = sorted(range(len(lengths)), key=lambda i: (-int(lengths[i]), i))
    return order[rank::world]


def shard(n_items, rank, world):
    """equal-length job: the plan is plain round-robin"""
    return plan([0] * n_items, rank, world)


def gather_token_ids(local, n_total, pad_id, rank, world, group=None, lengths=None):
    """local: int32 [n_local, n_seg, max_len] of the utterances `plan(lengths, rank, world)` (in that order);
    returns int32 [n_total, n_seg, max_len] in utterance order on every rank (ONE all_gather)."""
    lengths = [0] * n_total if lengths is None else lengths
    per = (n_total + world - 1) // world
    buf = torch.full((per,) + tuple(local.shape[1:]), pad_id, dtype=local.dtype, device=local.device)
    buf[:local.shape[0]] = local
    if world == 1:
        parts = [buf]
    else:
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf, group=group)
    out = torch.full((n_total,) + tuple(local.shape[1:]), pad_id, dtype=local.dtype, device=local.device)
    for r in range(world):
        idx = plan(lengths, r, world)
        if idx:
            out[idx] = parts[r][:len(idx)]
    return out


def _n_segments(n_samples, seg):
    return max(1, -(-int(n_samples) // seg))


@torch.inference_mode()
def convert_batch(model, src_wavs, cond_latent, seg_len=6.0, max_len=None, n_seg=None, tokens_per_second=None, utt_ids=None, **gen_kwargs):
    """Tokens of a micro-batch of utterances.  Segment s of the utterances that HAVE a segment s of the same length forms
    one class (same prefix length: all full segments of the batch, and equal-length tails), prefilled in one batched call;
    tails of other lengths are classes of their own -- never padded, which would change the reference's result
    (inference_utils.py:43-50).  With greedy decoding (top_k = 1, the configuration BASELINE configs[2] names) the classes are
    then decoded TOGETHER, as many per joint decode as the context has KV slots; a sampling run (top_k > 1) decodes one class
    after another, each with its own random stream (layers/gpt.py generate_groups).
    tokens_per_second (benchmark mode, SURVEY.md 8d: synthetic weights seldom emit the stop token): a class of segments of t seconds
    gets the fixed budget round(t * tokens_per_second) instead of max_new_tokens (23.4375: 141 tokens for 6 s, 94 for 4 s); a class
    whose budget is spent leaves the joint decode.
    utt_ids: the utterances' indices in the whole job (default 0..B-1).  A sampling run seeds the class of segment s whose first row is
    utterance u with seed + 7919 * (u * n_seg + s): distinct over all the generate_groups calls of a job (calls, waves and ranks),
    not restarting at every call.
    Returns int32 [B, n_seg, max_len] padded with the stop token (also the rows of segments an utterance does not have)."""
    m = model
    stop = m.gpt.stop_audio_token
    max_len = max_len or m.gpt.max_gen_mel_tokens
    seg = int(seg_len * m.content_sample_rate)
    min_len = int(0.32 * m.content_sample_rate)
    per_utt = [list(segments(w.to(m.device), seg, min_len)) for w in src_wavs]
    n_seg = n_seg or max(len(p) for p in per_utt)
    B = len(src_wavs)
    out = torch.full((B, n_seg, max_len), stop, dtype=torch.int32, device=m.device)
    kw = dict(_sampling_kwargs(m))
    kw.update(gen_kwargs)
    # every (segment index, segment length) class of the micro-batch: content codes per class, then ONE decode over all of them
    # (layers/gpt.py generate_groups: a decode step streams the weights once whatever the number of streams)
    classes = []
    for s in range(n_seg):
        groups = {}
        for b, p in enumerate(per_utt):
            if s < len(p):
                groups.setdefault(p[s].shape[-1], []).append(b)
        for n_samples, rows in sorted(groups.items(), reverse=True):
            wav = torch.cat([per_utt[b][s] for b in rows], 0)
            feat = m.content_extractor.extract_content_features(wav)
            codes = m.content_dvae.get_codebook_indices(feat.transpose(1, 2))
            classes.append((s, rows, codes, n_samples))
    max_slots = getattr(m.gpt, "max_slots", 8)
    i = 0
    while i < len(classes):                  # as many classes per joint decode as the KV slots hold
        j, n = i, 0
        while j < len(classes) and n + len(classes[j][1]) <= max_slots:
            n += len(classes[j][1])
            j += 1
        j = max(j, i + 1)
        kj = dict(kw)
        if tokens_per_second:
            cap = kw.get("max_new_tokens") or max_len
            kj["max_new_tokens"] = [max(1, min(cap, int(round(ns / m.content_sample_rate * tokens_per_second)))) for _, _, _, ns in classes[i:j]]
        ids = list(range(B)) if utt_ids is None else list(utt_ids)
        kj["class_seeds"] = [int(kw.get("seed", 0)) + 7919 * (ids[rows[0]] * n_seg + s) for s, rows, _, _ in classes[i:j]]
        gens = m.gpt.generate_groups([(cond_latent.expand(len(rows), -1, -1).contiguous(), codes) for _, rows, codes, _ in classes[i:j]], **kj)
        for (s, rows, _, _), gen in zip(classes[i:j], gens):
            out[rows, s, :gen.shape[1]] = gen.to(torch.int32)
        i = j
    return out


def _wave_classes(m, src_wavs, seg, min_len, n_seg):
    """the (segment index, segment length) classes of a wave of utterances, content codes computed: [(s, rows, codes, n_samples)]"""
    per_utt = [list(segments(w.to(m.device), seg, min_len)) for w in src_wavs]
    classes = []
    for s in range(n_seg):
        groups = {}
        for b, p in enumerate(per_utt):
            if s < len(p):
                groups.setdefault(p[s].shape[-1], []).append(b)
        for n_samples, rows in sorted(groups.items(), reverse=True):
            wav = torch.cat([per_utt[b][s] for b in rows], 0)
            feat = m.content_extractor.extract_content_features(wav)
            codes = m.content_dvae.get_codebook_indices(feat.transpose(1, 2))
            classes.append((s, rows, codes, n_samples))
    return classes


@torch.inference_mode()
def convert_rolling(model, src_wavs, cond_latent, seg_len=6.0, micro_batch=8, max_len=None, n_seg=None, tokens_per_second=None, **gen_kwargs):
    """convert_batch over ALL of a rank's utterances with a rolling decode (layers/gpt.py generate_rolling; greedy decoding): the
    utterances are taken in waves of `micro_batch` -- a wave shares its ContentVec / DVAE / prefill calls per (segment, length) class
    exactly as in convert_batch -- but a class joins the decode as soon as KV slots are free and leaves it when its budget is spent or
    its rows have stopped, so the step keeps 2 x micro_batch rows busy across wave boundaries instead of draining to the longest class
    of every wave.  Same tokens as convert_batch (streams are independent given their prefix).  Returns int32 [n, n_seg, max_len]."""
    m = model
    stop = m.gpt.stop_audio_token
    max_len = max_len or m.gpt.max_gen_mel_tokens
    seg = int(seg_len * m.content_sample_rate)
    min_len = int(0.32 * m.content_sample_rate)
    n_seg = n_seg or max(_n_segments(int(w.shape[-1]), seg) for w in src_wavs)
    out = torch.full((len(src_wavs), n_seg, max_len), stop, dtype=torch.int32, device=m.device)
    kw = dict(_sampling_kwargs(m))
    kw.update(gen_kwargs)
    cap = kw.get("max_new_tokens") or max_len
    jobs, where, budgets = [], [], []
    for i in range(0, len(src_wavs), micro_batch):
        for s, rows, codes, ns in _wave_classes(m, src_wavs[i:i + micro_batch], seg, min_len, n_seg):
            jobs.append((cond_latent.expand(len(rows), -1, -1).contiguous(), codes))
            where.append((s, [i + r for r in rows]))
            budgets.append(max(1, min(cap, int(round(ns / m.content_sample_rate * tokens_per_second)))) if tokens_per_second else cap)
    kw["max_new_tokens"] = budgets
    # streams in flight: what one wave holds (its classes decoded jointly, as in convert_batch) -- the rolling decode never runs a
    # larger step than the fixed micro-batch does, it only keeps that step full
    kw.setdefault("max_rows", max(max(len(r) for _, r in where), min(n_seg * micro_batch, getattr(m.gpt, "max_slots", 8))))
    gens = m.gpt.generate_rolling(jobs, **kw)
    for (s, rows), gen in zip(where, gens):
        out[rows, s, :gen.shape[1]] = gen.to(torch.int32)
    return out


@torch.inference_mode()
def convert_offline(model, src_wavs, ref_audio, seg_len=6.0, micro_batch=8, rank=0, world=1, process_group=None, **gen_kwargs):
    """All utterances of the job (any lengths), sharded by rank (see plan), in waves of `micro_batch`; no collective while
    converting, ONE all_gather of the padded token ids at the end.  Returns int32 [n_utts, n_seg, max_len] on every rank.
    rolling (gen_kwargs, default False): greedy runs only -- decode with a rolling set of streams (convert_rolling) instead of one joint
    decode per wave: same tokens, no drain at the end of every wave.
    process_group: the torch.distributed group of the all_gather (default: the world).  Everything in gen_kwargs goes to
    GPT.generate_groups -- including its `group` (decode steps per host look at the finished flags), which this function's
    process-group parameter used to shadow: with `group=48` in the kwargs every N > 1 run died in the all_gather."""
    m = model
    g = gen_kwargs.get("group")
    if g is not None and (isinstance(g, bool) or not isinstance(g, int)):
        raise TypeError(f"convert_offline: group={g!r} -- `group` is the number of decode steps per host look (an int, passed on to "
                        "GPT.generate_groups); the torch.distributed process group of the all_gather is `process_group=`")
    cond = m.get_gpt_cond_latents(ref_audio.to(m.device), m.config.audio.sample_rate)
    lengths = [int(w.shape[-1]) for w in src_wavs]
    mine = plan(lengths, rank, world)
    seg = int(seg_len * m.content_sample_rate)
    n_seg = max(_n_segments(n, seg) for n in lengths)
    max_len = m.gpt.max_gen_mel_tokens
    local = torch.full((len(mine), n_seg, max_len), m.gpt.stop_audio_token, dtype=torch.int32, device=m.device)
    rolling = bool(gen_kwargs.pop("rolling", False)) and gen_kwargs.get("top_k", m.config.top_k) == 1 and hasattr(m.gpt, "generate_rolling")
    if rolling and mine:
        local[:] = convert_rolling(m, [src_wavs[j] for j in mine], cond, seg_len, micro_batch, max_len, n_seg, **gen_kwargs)
        return gather_token_ids(local, len(src_wavs), m.gpt.stop_audio_token, rank, world, process_group, lengths)
    for i in range(0, len(mine), micro_batch):
        wave = mine[i:i + micro_batch]
        local[i:i + len(wave)] = convert_batch(m, [src_wavs[j] for j in wave], cond, seg_len, max_len, n_seg, utt_ids=wave, **gen_kwargs)
    return gather_token_ids(local, len(src_wavs), m.gpt.stop_audio_token, rank, world, process_group, lengths)
