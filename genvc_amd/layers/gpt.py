"""`GPT` with the reference's constructor, method names and argument meaning
(/root/reference/layers/gpt.py:87-621) for the inference path.  The module only holds the parameters
(same names and shapes as the reference state dict, so `load_state_dict(ckpt['model'])` works) and
drives libgenvc_hip through `GptEngine`; there is no PyTorch arithmetic fallback.

Mapped entry points: init_gpt_for_inference (:197), get_style_emb (:351), forward(return_latent=True)
(:375-508), compute_embeddings (:572), generate (:594), get_generator (:612), inference (:569).
Training-only paths (losses, masks, eval_sample) raise NotImplementedError: out of scope.
"""
import torch
from torch import nn

from .._lib import GenvcHipError
from ..engine import GptEngine, sample_params
from .perceiver_encoder import PerceiverResampler


class _Holder(nn.Module):
    pass


def _p(*shape, std=0.02, ones=False):
    t = torch.ones(*shape) if ones else torch.empty(*shape).normal_(std=std) if std else torch.zeros(*shape)
    return nn.Parameter(t, requires_grad=False)


class LearnedPositionEmbeddings(nn.Module):
    """holds `emb.weight` [seq_len, dim] (reference gpt.py:21-40)"""

    def __init__(self, seq_len, model_dim, init=0.02):
        super().__init__()
        self.emb = _Holder()
        self.emb.weight = _p(seq_len, model_dim, std=init)
        self.seq_len = seq_len


def _conv1d(nin, nout):
    m = _Holder()
    m.weight = _p(nin, nout)            # HF Conv1D layout [in, out]
    m.bias = _p(nout, std=0)
    return m


def _ln(d):
    m = _Holder()
    m.weight = _p(d, ones=True)
    m.bias = _p(d, std=0)
    return m


class GPT(nn.Module):
    def __init__(self, start_text_token=256, stop_text_token=257, layers=8, model_dim=512, heads=8,
                 max_text_tokens=120, max_mel_tokens=250, max_prompt_tokens=70, max_conditioning_inputs=1,
                 code_stride_len=1024, number_text_tokens=258, num_audio_tokens=1026, start_audio_token=1024,
                 stop_audio_token=1025, train_solo_embeddings=False, checkpointing=False,
                 average_conditioning_embeddings=False, fix_condition_embeddings=False, label_smoothing=0.0,
                 perceiver_cond_length_compression=256):
        super().__init__()
        self.number_text_tokens = number_text_tokens
        self.start_text_token, self.stop_text_token = start_text_token, stop_text_token
        self.num_audio_tokens = num_audio_tokens
        self.start_audio_token, self.stop_audio_token = start_audio_token, stop_audio_token
        self.layers, self.heads, self.model_dim = layers, heads, model_dim
        self.max_conditioning_inputs = max_conditioning_inputs
        self.max_gen_mel_tokens = max_mel_tokens - max_conditioning_inputs - 2             # gpt.py:131
        self.max_mel_tokens = max_mel_tokens + 2 + max_conditioning_inputs                 # :132
        self.max_text_tokens = max_text_tokens + 2                                         # :133
        self.max_prompt_tokens = max_prompt_tokens
        self.code_stride_len = code_stride_len

        d = model_dim
        self.text_embedding = _Holder(); self.text_embedding.weight = _p(number_text_tokens, d)
        self.mel_embedding = _Holder(); self.mel_embedding.weight = _p(num_audio_tokens, d)
        self.mel_pos_embedding = LearnedPositionEmbeddings(self.max_mel_tokens, d)
        self.text_pos_embedding = LearnedPositionEmbeddings(self.max_text_tokens, d)
        self.gpt = _Holder()
        self.gpt.h = nn.ModuleList()
        for _ in range(layers):
            blk = _Holder()
            blk.ln_1, blk.ln_2 = _ln(d), _ln(d)
            blk.attn = _Holder(); blk.attn.c_attn = _conv1d(d, 3 * d); blk.attn.c_proj = _conv1d(d, d)
            blk.mlp = _Holder(); blk.mlp.c_fc = _conv1d(d, 4 * d); blk.mlp.c_proj = _conv1d(4 * d, d)
            self.gpt.h.append(blk)
        self.gpt.ln_f = _ln(d)
        self.final_norm = _ln(d)
        self.text_head = _Holder(); self.text_head.weight = _p(number_text_tokens, d); self.text_head.bias = _p(number_text_tokens, std=0)
        self.mel_head = _Holder(); self.mel_head.weight = _p(num_audio_tokens, d); self.mel_head.bias = _p(num_audio_tokens, std=0)
        # hyper-parameters hard-coded in the reference (gpt.py:179-188)
        self.conditioning_perceiver = PerceiverResampler(dim=d, depth=4, dim_context=80, num_latents=32, dim_head=64,
                                                         heads=8, ff_mult=4, use_flash_attn=False)
        self.engine = None
        self._prefix = None
        self.max_slots = 8
        self.recoveries = 0          # generations repeated after a hand-off time-out of a one-launch step (_recovering)

    # ------------------------------------------------------------------------------------------
    def dims(self):
        return dict(n_layer=self.layers, d_model=self.model_dim, n_head=self.heads,
                    num_audio_tokens=self.num_audio_tokens, number_text_tokens=self.number_text_tokens,
                    start_text_token=self.start_text_token, stop_text_token=self.stop_text_token,
                    start_audio_token=self.start_audio_token, stop_audio_token=self.stop_audio_token,
                    max_gen_mel_tokens=self.max_gen_mel_tokens, max_mel_pos=self.max_mel_tokens,
                    max_text_pos=self.max_text_tokens, max_prompt_tokens=self.max_prompt_tokens,
                    code_stride_len=self.code_stride_len,
                    max_seq=self.max_prompt_tokens + self.max_mel_tokens + self.max_text_tokens + 1)   # gpt.py:198

    def init_gpt_for_inference(self, kv_cache=True, use_deepspeed=False, max_slots=8, max_rows=4096, weight_dtype="fp32"):
        """reference gpt.py:197-218: here = create the HIP context and repack the weights into it.
        weight_dtype: "fp32" (reference numerics), "bf16" (bf16 weight storage), "bf16_kv" (+ bf16 KV cache) or "bf16_act" (+ bf16
        activations across the hand-offs of the one-launch rows step: include/genvc_hip.h, weight_dtype 3)."""
        if not kv_cache:
            raise NotImplementedError("the HIP path always uses the KV cache")
        if use_deepspeed:
            raise NotImplementedError("DeepSpeed kernel injection is replaced by libgenvc_hip")
        if self.engine is not None:
            self.engine.close()
        self.max_slots = max_slots
        self.engine = GptEngine(self.dims(), max_slots=max_slots, max_rows=max_rows, weight_dtype=weight_dtype)
        sd = {k: v for k, v in self.state_dict().items() if not k.startswith("conditioning_perceiver.")}
        self.engine.bind(sd)
        self.conditioning_perceiver.bind()
        return self

    def _need_engine(self):
        if self.engine is None:
            raise RuntimeError("call init_gpt_for_inference() first (reference inference/model_init.py:31)")

    def _recovering(self, n_slots, fn):
        """fn() -- a whole generation that has not handed anything to its caller yet -- with ONE retry after a hand-off time-out of a
        one-launch step (another context held CUs: include/genvc_hip.h, gvc_gpt_health).  The failed attempt's tokens and K/V rows are
        garbage; the library has already switched this context to the launch-per-phase paths, so the slots are reset and the work is
        repeated there.  Either the caller gets the tokens of a clean run or the error propagates: never the garbage
        (reference semantics: a call returns its tokens or fails, /root/reference/inference/inference_utils.py:135-217).  Other
        state errors (a full KV cache) are not recoverable by repeating and propagate at once."""
        try:
            return fn()
        except GenvcHipError as e:
            if not e.is_handoff_timeout:
                raise
            self.recoveries = getattr(self, "recoveries", 0) + 1
            torch.cuda.synchronize()
            dev = next(self.parameters()).device
            self.engine.reset(torch.arange(n_slots, device=dev, dtype=torch.int32))
            return fn()

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def get_style_emb(self, cond_input, return_latent=False, seq_lens=None, frames_major=None):
        """cond_input (b,80,s) or (b,1,80,s) -> (b, d, 32)   (reference gpt.py:351-373).  frames_major (extension): the same mel as
        (b,s,80), which the mel kernel writes alongside -- saves the permute(0, 2, 1).contiguous() copy in front of the Perceiver"""
        if return_latent:
            return cond_input.unsqueeze(1)
        if cond_input.ndim == 4:
            cond_input = cond_input.squeeze(1)
        if seq_lens is not None:
            raise NotImplementedError("perceiver masks are a training-only path")
        x = frames_major if frames_major is not None else cond_input.permute(0, 2, 1).contiguous()
        return self.conditioning_perceiver(x).transpose(1, 2)

    @torch.inference_mode()
    def compute_embeddings(self, cond_latents, text_inputs):
        """reference gpt.py:572-592: stores the prefix embeddings, returns the fake ids (1 ... 1, start)."""
        self._need_engine()
        self._prefix = self.engine.prefix_embeddings(cond_latents.to(torch.float32).contiguous(),
                                                     text_inputs.to(torch.int32).contiguous())
        B, P, _ = self._prefix.shape
        ids = torch.full((B, P + 1), 1, dtype=torch.long, device=text_inputs.device)
        ids[:, -1] = self.start_audio_token
        return ids

    def _start(self, fake_inputs, kw):
        """prefill + device-side loop state for the stored prefix"""
        if kw.get("num_beams", 1) != 1:
            raise NotImplementedError("beam search is not on GenVC's inference path (num_beams=1)")
        B, n0 = fake_inputs.shape
        dev = fake_inputs.device
        max_new = kw.get("max_new_tokens") or self.max_gen_mel_tokens                     # gpt.py:606,618
        slots = torch.arange(B, device=dev, dtype=torch.int32)
        ids = torch.empty(B, n0 + max_new + 8, device=dev, dtype=torch.int32)
        ids[:, :n0] = fake_inputs.to(torch.int32)
        st = dict(B=B, slots=slots, ids=ids,
                  ids_len=torch.full((B,), n0, device=dev, dtype=torch.int32),
                  finished=torch.zeros(B, device=dev, dtype=torch.int32),
                  toks=torch.full((B, max_new), self.stop_audio_token, device=dev, dtype=torch.int32),
                  lats=torch.empty(B, max_new, self.model_dim, device=dev, dtype=torch.float32),
                  max_new=max_new, done=0, n0=n0)
        samp = dict(repetition_penalty=kw.get("repetition_penalty", 1.0), temperature=kw.get("temperature", 1.0),
                    top_p=kw.get("top_p", 1.0), top_k=kw.get("top_k", 0) if kw.get("do_sample", True) else 1)
        st["params"] = sample_params(samp, self.num_audio_tokens, self.stop_audio_token, kw.get("seed", 0))
        # `cached_cond_rows` (extension): the leading rows of the prefix -- the conditioning latents, identical for every
        # segment of an utterance -- are still in the KV cache from the previous segment's prefill of these slots
        self.engine.prefill(slots, self._prefix, want_outputs=False, n_cached=int(kw.get("cached_cond_rows", 0)))
        return st

    def _advance(self, st, n):
        """n graph-replayed (sample, decode) steps; returns True when every row has emitted the stop token"""
        n = min(n, st["max_new"] - st["done"])
        if n > 0:
            # the cache holds n0 positions after the prefill and one more per step: the library picks its decode kernels for the
            # context length this call reaches (not for the 602-token cap the ids rows are sized for)
            self.engine.generate(st["slots"], st["ids"], st["ids_len"], st["finished"], st["params"], st["done"], n,
                                 st["toks"], st["lats"], max_keys=st["n0"] + st["done"] + n)
            st["done"] += n
        end = bool(st["finished"].all().item()) or st["done"] >= st["max_new"]
        self.engine.health()          # (the .item() above synchronised: a hand-off timeout of these steps surfaces here, not a call later)
        return end

    @torch.inference_mode()
    def generate(self, cond_latents, text_inputs, **generate_kwargs):
        """reference gpt.py:594-609 -> int64 [B, n_generated]; finished rows are padded with the stop token.
        `group` (extra kwarg) = decode steps per host check of the finished flags."""
        fake = self.compute_embeddings(cond_latents, text_inputs)
        group = generate_kwargs.pop("group", 16)

        attempt = []

        def run():
            # (a retry after a hand-off time-out prefills in full: the reset slots have lost any cached conditioning rows)
            st = self._start(fake, dict(generate_kwargs, cached_cond_rows=0) if attempt else generate_kwargs)
            attempt.append(1)
            while not self._advance(st, group):
                pass
            return st
        st = self._recovering(int(fake.shape[0]), run)
        # the reference loop stops at the step where the last row emits 1025
        toks = st["toks"][:, :st["done"]].long()
        n = self._stop_len(toks)
        self.last_latents = st["lats"][:, :n]
        return toks[:, :n]

    @torch.inference_mode()
    def generate_groups(self, groups, **generate_kwargs):
        """Several generate() calls decoded TOGETHER: groups = [(cond_latents [B_i, 32, d], text_inputs [B_i, Tc_i]), ...] with
        different code lengths.  Each group is prefilled on its own (its rows share a prefix length) into its own KV slots; the
        decode steps then run over all streams at once, so the weights stream once per step for the whole set.  Streams are
        independent, so with deterministic decoding (top_k = 1) every group's result is what generate() returns for it (bit for
        bit when both land on the same decode kernels, i.e. the rows path from 5 streams up; within float rounding otherwise);
        with sampling the per-row random streams would be numbered differently, so that case runs the groups one after another.
        `max_new_tokens` may be a list with one budget per group (benchmark mode: synthetic weights seldom stop, SURVEY.md 8d fixes
        the tokens of a segment by its duration): a group whose budget is spent leaves the joint decode, and the steps that remain
        run over the live streams only (fewer rows per step: the 8-row instead of the 16-row one-launch step for configs[2]'s tail).
        Returns a list of int64 [B_i, n_i] (reference gpt.py:594-609 per group)."""
        self._need_engine()
        kw = dict(generate_kwargs)
        group = kw.pop("group", 16)          # decode steps per engine call (one host look at the finished flags per call)
        class_seeds = kw.pop("class_seeds", None)      # sampling runs: one seed per group (default: seed + 7919 * group index)
        budgets = kw.get("max_new_tokens")
        if isinstance(budgets, (list, tuple)):
            if len(budgets) != len(groups):
                raise ValueError(f"generate_groups: {len(budgets)} token budgets for {len(groups)} groups")
            budgets = [int(b) for b in budgets]
            kw["max_new_tokens"] = max(budgets)
        else:
            budgets = None
        greedy = kw.get("top_k", 0) == 1 or not kw.get("do_sample", True)
        total = sum(int(t.shape[0]) for _, t in groups)
        stats = getattr(self, "groups_stats", None)        # {"joint": n, "separate": n}: bench.py / tests count the two paths
        if not greedy or len(groups) == 1 or total > self.max_slots or kw.get("num_beams", 1) != 1:
            if stats is not None and len(groups) > 1:
                stats["separate"] += 1
            # one generate() per class; a sampling run gives every class its own random stream (the rows of one class keep the
            # per-row numbering of the counter RNG): with one shared seed all classes would draw identical per-row sequences.
            # Seed semantics: class gi of THIS call draws with class_seeds[gi] when the caller numbers its classes across calls
            # (parallel_offline.convert_batch does: several calls per job), else with seed + 7919 * gi
            outs = []
            for gi, (c, t) in enumerate(groups):
                kg = dict(kw)
                if not greedy:
                    kg["seed"] = int(class_seeds[gi]) if class_seeds is not None else int(kw.get("seed", 0)) + 7919 * gi
                if budgets is not None:
                    kg["max_new_tokens"] = budgets[gi]
                outs.append(self.generate(c, t, **kg))
            self.last_latents = None      # (same contract as the joint path: callers of generate_groups want tokens)
            return outs
        return self._recovering(total, lambda: self._generate_groups_joint(groups, kw, budgets, group, stats))

    def _generate_groups_joint(self, groups, kw, budgets, group, stats):
        total = sum(int(t.shape[0]) for _, t in groups)
        if stats is not None:
            stats["joint"] += 1
        dev = groups[0][1].device
        max_new = kw.get("max_new_tokens") or self.max_gen_mel_tokens
        # rows in order of falling budget: the live streams are always the first rows of every buffer
        order = sorted(range(len(groups)), key=lambda i: -(budgets[i] if budgets else max_new))
        groups = [groups[i] for i in order]
        gb = [budgets[i] if budgets else max_new for i in order]
        prefixes = [self.engine.prefix_embeddings(c.to(torch.float32).contiguous(), t.to(torch.int32).contiguous()) for c, t in groups]
        n0s = [int(p.shape[1]) + 1 for p in prefixes]
        width = max(n0s) + max_new + 8
        ids = torch.full((total, width), 1, device=dev, dtype=torch.int32)
        ids_len = torch.empty(total, device=dev, dtype=torch.int32)
        slots = torch.arange(total, device=dev, dtype=torch.int32)
        row = 0
        spans = []
        for p, n0 in zip(prefixes, n0s):
            b = int(p.shape[0])
            ids[row:row + b, n0 - 1] = self.start_audio_token
            ids_len[row:row + b] = n0
            self.engine.prefill(slots[row:row + b].contiguous(), p, want_outputs=False)
            spans.append((row, row + b))
            row += b
        finished = torch.zeros(total, device=dev, dtype=torch.int32)
        toks = torch.full((total, max_new), self.stop_audio_token, device=dev, dtype=torch.int32)
        lats = torch.empty(total, max_new, self.model_dim, device=dev, dtype=torch.float32)
        samp = dict(repetition_penalty=kw.get("repetition_penalty", 1.0), temperature=kw.get("temperature", 1.0),
                    top_p=kw.get("top_p", 1.0), top_k=1)
        params = sample_params(samp, self.num_audio_tokens, self.stop_audio_token, kw.get("seed", 0))
        done = 0
        while done < max_new:
            live_groups = [g for g in range(len(groups)) if gb[g] > done]
            live = spans[live_groups[-1]][1]                                   # rows [0, live) still have tokens to produce
            n = min(group, min(gb[g] for g in live_groups) - done)              # (a call never crosses the end of a budget)
            self.engine.generate(slots[:live], ids[:live], ids_len[:live], finished[:live], params, done, n, toks[:live], lats[:live],
                                 max_keys=max(n0s[g] for g in live_groups) + done + n)
            done += n
            stop = bool(finished[:live].all().item())
            self.engine.health()
            if stop:
                break
        out = [None] * len(groups)
        for g, (lo, hi) in enumerate(spans):
            t = toks[lo:hi, :min(done, gb[g])].long()
            out[order[g]] = t[:, :self._stop_len(t)]
        self.last_latents = None          # (per-group latents are not kept: the callers of this path want tokens)
        return out

    @torch.inference_mode()
    def generate_rolling(self, jobs, **generate_kwargs):
        """generate_groups with a ROLLING set of streams (greedy decoding only): jobs = [(cond_latents [B_i, 32, d], text_inputs
        [B_i, Tc_i]), ...] are admitted in order as KV slots become free.  Retirement is PER ROW, as the reference's loop tracks
        `unfinished_sequences` per row (stream_generator.py:861-874): a row that has emitted the stop token gives its KV slot back at
        the next host look (every `group` steps) and stops taking a row of the decode step; the rows of a job whose budget is spent
        leave together.  Freed slots go to the next job as soon as all ITS rows fit, so the decode step stays full instead of
        draining to the longest stream (configs[2]: the 47-step tail of a micro-batch's 6 s class runs beside the NEXT micro-batch's
        4 s class; a real checkpoint ends every class ragged).  Streams are independent given their prefix, so every job gets what
        generate() returns for it: finished rows padded with the stop token up to the step where the job's last row stops
        (reference gpt.py:594-609).  `max_new_tokens`: one budget, or a list with one per job.  Returns a list of int64 [B_i, n_i]
        in job order.  `self.rolling_stats` (if the attribute is a dict) accumulates row_steps_issued / row_steps_live: rows x steps
        the decode calls ran, and how many of them produced a token the reference's loop would have produced."""
        self._need_engine()
        kw = dict(generate_kwargs)
        group = kw.pop("group", 16)
        kw.pop("class_seeds", None)
        max_rows = kw.pop("max_rows", None)      # streams in flight at most (default: every KV slot of the context)
        if not (kw.get("top_k", 0) == 1 or not kw.get("do_sample", True)) or kw.get("num_beams", 1) != 1:
            raise NotImplementedError("generate_rolling serves greedy decoding (top_k = 1): with sampling use generate_groups")
        budgets = kw.get("max_new_tokens")
        if isinstance(budgets, (list, tuple)):
            if len(budgets) != len(jobs):
                raise ValueError(f"generate_rolling: {len(budgets)} token budgets for {len(jobs)} jobs")
            budgets = [int(b) for b in budgets]
        else:
            budgets = [int(budgets or self.max_gen_mel_tokens)] * len(jobs)
        if not jobs:
            return []
        n0s = [int(t.shape[1]) + int(c.shape[1]) + 3 for c, t in jobs]            # prefix rows (cond + text + 2) + the start token
        width = max(n0 + b for n0, b in zip(n0s, budgets)) + 8
        S = min(self.max_slots, int(max_rows)) if max_rows else self.max_slots
        if max(int(t.shape[0]) for _, t in jobs) > S:
            raise ValueError(f"generate_rolling: a job has more rows than streams may be in flight ({S}; KV slots {self.max_slots})")
        return self._recovering(S, lambda: self._rolling(jobs, kw, budgets, group, n0s, width, S))

    def _rolling(self, jobs, kw, budgets, group, n0s, width, S):
        dev = jobs[0][1].device
        eng = self.engine
        stop = self.stop_audio_token
        ids_all = torch.ones(S, width, device=dev, dtype=torch.int32)
        len_all = torch.zeros(S, device=dev, dtype=torch.int32)
        fin_all = torch.zeros(S, device=dev, dtype=torch.int32)
        samp = dict(repetition_penalty=kw.get("repetition_penalty", 1.0), temperature=kw.get("temperature", 1.0),
                    top_p=kw.get("top_p", 1.0), top_k=1)
        params = sample_params(samp, self.num_audio_tokens, stop, kw.get("seed", 0))
        stats = getattr(self, "groups_stats", None)
        rstats = getattr(self, "rolling_stats", None)
        free = list(range(S))
        live, out, nxt = [], [None] * len(jobs), 0
        while nxt < len(jobs) or live:
            # admit jobs in order while all their rows fit
            while nxt < len(jobs) and int(jobs[nxt][1].shape[0]) <= len(free):
                c, t = jobs[nxt]
                b = int(t.shape[0])
                mine = free[:b]
                del free[:b]
                sl = torch.tensor(mine, device=dev, dtype=torch.int32)
                prefix = eng.prefix_embeddings(c.to(torch.float32).contiguous(), t.to(torch.int32).contiguous())
                eng.prefill(sl, prefix, want_outputs=False)
                idx = sl.long()
                ids_all[idx] = 1                                    # (a reused slot starts with a clean history: repetition_penalty reads it)
                ids_all[idx, n0s[nxt] - 1] = self.start_audio_token
                len_all[idx] = n0s[nxt]
                fin_all[idx] = 0
                live.append(dict(job=nxt, slots=mine, alive=list(range(b)), n0=n0s[nxt], budget=budgets[nxt], done=0,
                                 toks=torch.full((b, budgets[nxt]), stop, device=dev, dtype=torch.int32)))
                nxt += 1
            n = min(group, min(j["budget"] - j["done"] for j in live))
            row_slots = [j["slots"][r] for j in live for r in j["alive"]]
            rows = torch.tensor(row_slots, device=dev, dtype=torch.int32)
            idx = rows.long()
            W = max(j["n0"] + j["done"] for j in live) + n + 8
            ids = ids_all[idx, :W].contiguous()
            ids_len = len_all[idx].contiguous()
            fin = fin_all[idx].contiguous()
            toks = torch.full((len(row_slots), n), stop, device=dev, dtype=torch.int32)
            eng.generate(rows, ids, ids_len, fin, params, 0, n, toks, None, max_keys=W - 8)
            ids_all[idx, :W] = ids
            len_all[idx] = ids_len
            fin_all[idx] = fin
            fin_h = fin.cpu()                                      # (synchronises)
            eng.health()
            if stats is not None:
                stats["joint"] += 1
            if rstats is not None:
                th = toks.cpu()
                hit = th == stop
                first = torch.where(hit.any(1), hit.int().argmax(1) + 1, torch.full((th.shape[0],), n))
                rstats["row_steps_issued"] = rstats.get("row_steps_issued", 0) + n * len(row_slots)
                rstats["row_steps_live"] = rstats.get("row_steps_live", 0) + int(first.sum())
                rstats["calls"] = rstats.get("calls", 0) + 1
            r = 0
            keep = []
            for j in live:
                k = len(j["alive"])
                a = torch.tensor(j["alive"], device=dev, dtype=torch.long)
                j["toks"][a, j["done"]:j["done"] + n] = toks[r:r + k]
                j["done"] += n
                still = [row for i, row in enumerate(j["alive"]) if not bool(fin_h[r + i])]
                gone = [row for i, row in enumerate(j["alive"]) if bool(fin_h[r + i])]
                r += k
                if j["done"] >= j["budget"] or not still:           # the job is over: its budget is spent or its last row has stopped
                    t = j["toks"][:, :j["done"]].long()
                    out[j["job"]] = t[:, :self._stop_len(t)]
                    gone = j["alive"]
                else:
                    j["alive"] = still
                    keep.append(j)
                free.extend(j["slots"][row] for row in gone)        # a stopped row's slot serves the next job from the next call on
                free.sort()
            live = keep
        self.last_latents = None
        return out

    def _stop_len(self, toks):
        """steps the reference loop runs: up to and including the step where the last row emits the stop token"""
        is_stop = toks == self.stop_audio_token
        if not bool(is_stop.any(1).all()):
            return toks.shape[1]
        return int(is_stop.long().argmax(1).max().item()) + 1

    @torch.inference_mode()
    def get_generator(self, fake_inputs, **generate_kwargs):
        """reference gpt.py:612-621 + stream_generator.py:865: yields (tokens int64[B], latent float[B,d]) per step,
        the EOS step included.  Steps run in groups of `stream_group` (default 8, the vocoder chunk of
        inference_utils.py:195) with one host check of the finished flags per group."""
        self._need_engine()
        group = generate_kwargs.pop("stream_group", 8)
        B = int(fake_inputs.shape[0])
        emitted = 0
        st = None
        retried = False

        def restart():
            # a hand-off time-out (see _recovering): the segment is generated again from its start on the launch-per-phase paths --
            # a full prefill (the reset slots have lost their cached conditioning rows) -- and the steps already yielded are skipped:
            # greedy decoding repeats them (with top_k > 1 what follows comes from a different token sequence)
            self.recoveries = getattr(self, "recoveries", 0) + 1
            torch.cuda.synchronize()
            self.engine.reset(torch.arange(B, device=fake_inputs.device, dtype=torch.int32))
            s2 = self._start(fake_inputs, dict(generate_kwargs, cached_cond_rows=0))
            while s2["done"] < emitted and not self._advance(s2, min(group, emitted - s2["done"])):
                pass
            return s2
        while True:
            try:
                if st is None:
                    st = self._start(fake_inputs, generate_kwargs)
                end = self._advance(st, group)
            except GenvcHipError as e:
                if not e.is_handoff_timeout or retried:
                    raise
                retried = True
                st = restart()
                continue
            toks = st["toks"][:, emitted:st["done"]].long()
            n = toks.shape[1]
            if end and n:
                n = min(n, self._stop_len(st["toks"][:, :st["done"]].long()) - emitted)
            for i in range(n):
                yield toks[:, i], st["lats"][:, emitted + i]
            emitted += n
            if end:
                return

    def inference(self, cond_latents, text_inputs, **generate_kwargs):
        return self.generate(cond_latents, text_inputs, **generate_kwargs)

    @torch.inference_mode()
    def forward(self, text_inputs, text_lengths, audio_codes, wav_lengths, cond_mels=None, cond_lens=None,
                cond_latents=None, return_attentions=False, return_latent=False):
        """Inference use of reference gpt.py:375-508: `return_latent=True` with `cond_latents` given
        (inference_utils.py:71-76) -> latents [B,n,d] of the n = ceil(wav_lengths/1024) codes."""
        self._need_engine()
        if not return_latent or cond_latents is None or return_attentions:
            raise NotImplementedError("only forward(..., cond_latents=..., return_latent=True) is on the inference path")
        B, n = audio_codes.shape
        if int(text_lengths.min()) != text_inputs.shape[1] or int(torch.ceil(wav_lengths / self.code_stride_len).min()) != n:
            raise NotImplementedError("ragged text/code lengths are a training-only path")
        prefix = self.engine.prefix_embeddings(cond_latents.to(torch.float32).contiguous(),
                                               text_inputs.to(torch.int32).contiguous())
        slots = torch.arange(B, device=audio_codes.device, dtype=torch.int32)
        return self.engine.latents(slots, prefix, audio_codes.to(torch.int32).contiguous())
