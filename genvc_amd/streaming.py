"""Streaming sessions: persistent per-stream KV slots and a step scheduler.

The reference converts one utterance per call and re-runs everything per segment (inference_utils.py:135-217).  A
deployment has several live streams whose segments arrive at different times.  `StreamSessions` keeps one KV-cache slot
per open stream and, at every `step()`, batches whatever the open streams need next:

  * streams with a queued source segment that are not decoding: ContentVec -> DVAE/VQ -> prefix embeddings -> prefill,
    batched over the streams whose segment has the same length (the conditioning rows of a stream stay in its slot after
    its first segment: `gvc_gpt_prefill_cached`);
  * streams that are decoding: ONE `gvc_gpt_generate` call of `group` steps over all of them, whatever their positions
    (per-slot cache lengths live on the device; the MFMA rows path from 5 streams up);
  * the vocoder for the token groups that came out, batched over equal group lengths, cross-faded per stream exactly as
    `handle_chunks` does inside `synthesize_utt_streaming`.

Each stream produces the tokens and the waveform of `synthesize_utt_streaming(model, its_source, its_reference,
seg_len)` at top_k = 1.

`left_context_s` (extension beyond the reference, SURVEY.md 8f row f4: "chunked ContentVec with left context"): the reference feeds
ContentVec one segment at a time, so the first frames of every segment see no past (its positional conv spans +-64 frames =
1.3 s, its attention the whole input, its layer-0 GroupNorm statistics the whole input).  With `left_context_s > 0` a stream keeps
that many seconds of its already fed source audio (a multiple of 320 samples, ContentVec's hop) and extracts the features of
[kept past | new segment], passing only the segment's frames on to the content tokeniser; frame i of the window starts at sample
320 i, so the segment's frames are exactly the last ones.  The default (0) is the reference's behaviour.

With top_k > 1 the draws differ from a solo run: the counter RNG is keyed by the position in the call, not in the utterance.
"""
import time

import torch

from .inference.inference_utils import _sampling_kwargs, _vocode, handle_chunks
from .engine import sample_params
from ._lib import GenvcHipError


class _Session:
    def __init__(self, slot, cond):
        self.slot, self.cond = slot, cond
        self.queue = []            # pending source segments [1, n]
        self.decoding = False
        self.prefilled = False     # the conditioning rows are in the slot's cache
        self.done = 0              # tokens generated for the current segment
        self.prev = self.overlap = None
        self.tokens = []           # per segment: int64 [1, n]
        self.past = None           # left context: the tail of the source audio fed so far [1, <= ctx samples]
        self.current = None        # the segment being decoded and its ContentVec left context as they were when it started (recovery)
        self.skip = 0              # tokens of the current segment already emitted before a failed step (recovery re-run)


class StreamSessions:
    def __init__(self, model, max_sessions=8, group=8, left_context_s=0.0, rearm_after_s=5.0, rearm_max_tries=3, prefill_speaker=True):
        m = self.m = model
        self.ctx = int(round(left_context_s * model.content_sample_rate / 320.0)) * 320       # whole ContentVec hops
        g = m.gpt
        g._need_engine()
        self.eng = g.engine
        self.prefill_speaker = bool(prefill_speaker)
        self.group = group
        self.max_new = g.max_gen_mel_tokens
        self.free = list(range(max_sessions))
        self.sessions = {}
        self._next_id = 0
        dev = m.device
        kw = _sampling_kwargs(m)
        samp = dict(repetition_penalty=kw["repetition_penalty"], temperature=kw["temperature"], top_p=kw["top_p"], top_k=kw["top_k"])
        self.params = sample_params(samp, g.num_audio_tokens, g.stop_audio_token, 0)
        self.calls = 0
        self.recoveries = 0          # decode calls dropped and re-run after a hand-off time-out (gvc_gpt_health)
        self._rearm = False          # a recovery happened: the one-launch steps may be re-armed (policy: maybe_rearm)
        self.rearm_after_s = float(rearm_after_s)
        self.rearm_max_tries = int(rearm_max_tries)
        self._rearm_wait = self.rearm_after_s
        self._rearm_tries = 0        # re-arms since the last clean stretch
        self._last_timeout = 0.0
        self.rearms = 0
        self.rearm_gave_up = False
        self.stop = g.stop_audio_token
        # per-slot history of input ids (fake prefix + generated), as wide as the longest run
        self.width = 32 + g.max_text_tokens + 2 + 1 + self.max_new + 8
        self.ids = torch.ones(max_sessions, self.width, device=dev, dtype=torch.int32)
        self.ids_len = torch.zeros(max_sessions, device=dev, dtype=torch.int32)
        self.finished = torch.zeros(max_sessions, device=dev, dtype=torch.int32)

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def open(self, ref_audio):
        """ref_audio [1, n] at the model rate -> session id"""
        if not self.free:
            raise RuntimeError("no free stream slot")
        cond = self.m.get_gpt_cond_latents(ref_audio.to(self.m.device), self.m.config.audio.sample_rate)
        sid = self._next_id
        self._next_id += 1
        s = self.sessions[sid] = _Session(self.free.pop(0), cond)
        if self.prefill_speaker:
            # the speaker is known before the first source segment arrives: its 32 conditioning rows go into the slot's KV cache NOW
            # (gvc_gpt_prefill_cond), so the first segment too computes only its text rows + start token -- the first audio chunk of a
            # session no longer waits for them (the reference rebuilds them in every segment's prefill, inference_utils.py:43-66)
            self.eng.prefill_cond(torch.tensor([s.slot], device=self.m.device, dtype=torch.int32), cond.to(torch.float32).contiguous())
            s.prefilled = True
        return sid

    def push(self, sid, src_segment):
        """queue one source segment [1, n] at 16 kHz (what `segments()` yields for a whole utterance)"""
        self.sessions[sid].queue.append(src_segment.to(self.m.device))

    def close(self, sid):
        s = self.sessions.pop(sid)
        self.free.append(s.slot)
        return s.tokens

    def idle(self):
        """True when no stream is decoding and nothing is queued (a pure predicate: no device work)"""
        return all(not s.decoding and not s.queue for s in self.sessions.values())

    def maybe_rearm(self, now=None):
        """Re-arm policy after a time-out recovery (advisor finding, round 5).  A recovery leaves the context on the launch-per-phase paths,
        a stable state ~28 % slower.  Going back to the one-launch steps is only worth trying when the GPU is probably ours again, and a
        failed try is expensive (a ~0.2 s bounded spin, a device sync, graph re-capture, every decoding stream restarted -- with top_k > 1
        an audible splice), so: only at an idle moment, only after `rearm_after_s` seconds without a time-out, with the wait DOUBLED after
        every failed try (a try has failed when the next time-out follows it), and never again after `rearm_max_tries` consecutive failed
        tries (the context then stays on the fallback for good; `rearm_gave_up`).  Called by step() at idle moments; callers that know
        better (the other process is gone) call it or `eng.rearm()` themselves.  Returns True when it re-armed.  Errors other than a
        pending time-out propagate."""
        if not self._rearm or self.rearm_gave_up or not self.idle():
            return False
        now = time.monotonic() if now is None else now
        if now - self._last_timeout < self._rearm_wait:
            return False
        self._rearm = False
        self._rearm_tries += 1
        torch.cuda.synchronize()
        try:
            self.eng.rearm()
        except GenvcHipError as e:
            if not e.is_handoff_timeout:
                raise
            # a time-out was still pending: the library reported it and stays on the fallback; count it like a failed try
            self._note_timeout(now)
            return False
        self.rearms += 1
        return True

    def _note_timeout(self, now=None):
        now = time.monotonic() if now is None else now
        self._last_timeout = now
        self._rearm = True
        if self._rearm_tries > 0:                       # the time-out follows a re-arm: that try failed
            self._rearm_wait = min(self._rearm_wait * 2.0, 3600.0)
            if self._rearm_tries >= self.rearm_max_tries:
                self.rearm_gave_up = True
        return now

    @torch.inference_mode()
    def segment_features(self, ss, wav):
        """ContentVec features of the new segments `wav` [B, n] (one per session of `ss`).  With left context each stream's kept past is
        put in front of its segment and only the segment's frames are returned (streams go one by one: their pasts differ)."""
        m = self.m
        if self.ctx == 0:
            return m.content_extractor.extract_content_features(wav)
        n_frames = (wav.shape[1] - 400) // 320 + 1              # frames of the segment alone (HuBERT conv stack: receptive field 400, hop 320)
        outs = []
        for i, x in enumerate(ss):
            seg = wav[i:i + 1]
            win = seg if x.past is None else torch.cat([x.past, seg], 1)
            f = m.content_extractor.extract_content_features(win)
            outs.append(f[:, f.shape[1] - n_frames:])
            keep = win[:, max(0, win.shape[1] - self.ctx):]
            x.past = keep[:, keep.shape[1] % 320:].contiguous()  # a whole number of hops: the next segment's frames stay aligned
        return torch.cat(outs, 0)

    # ------------------------------------------------------------------------------------------
    @torch.inference_mode()
    def step(self):
        """advance every open stream by one scheduling step; returns {sid: [wav chunks emitted now]}.
        A hand-off time-out of a one-launch step (another context held CUs) can surface from ANY library call of the step -- the health
        check after the decode call, but also the entry check of this step's prefill / cached chunk prefill / generate, which reports
        what the PREVIOUS call left behind.  Wherever it surfaces, nothing of this step is kept: every stream that was decoding or was
        started in this step goes back to the start of its segment (slots reset), and the next step() re-runs them on the
        launch-per-phase paths the library has switched to.  Greedy decoding repeats the tokens, so the audio continues where it
        stopped; with top_k > 1 the re-run draws with new per-call seeds -- the groups already emitted are skipped, what follows
        comes from a different token sequence (an audible splice is possible)."""
        self._popped = []
        self.maybe_rearm()           # (a no-op unless a time-out recovery is pending, the scheduler is idle and the back-off has run out)
        try:
            return self._step()
        except GenvcHipError as e:
            if not e.is_handoff_timeout:           # (a full KV cache, ... : not recoverable by repeating the work)
                raise
            self._recover()
            return {}

    def _recover(self):
        self.recoveries += 1
        self._note_timeout()
        redo = []
        for s in self.sessions.values():
            if s.decoding:
                self._requeue(s)
                redo.append(s)
        for s in self._popped:                     # started in the failed step, not yet marked decoding: put the segment back
            if s not in redo and not s.decoding and s.current is not None and (not s.queue or s.queue[0] is not s.current[0]):
                seg, past = s.current
                s.queue.insert(0, seg)
                s.past = past
                s.prefilled = False
                redo.append(s)
        if redo:
            try:
                torch.cuda.synchronize()
            except Exception:
                pass
            self.eng.reset(torch.tensor([s.slot for s in redo], device=self.m.device, dtype=torch.int32))

    def _step(self):
        m, eng, dev = self.m, self.eng, self.m.device
        out = {}
        # 1. segments that can start: batch by (segment length, cached or not)
        starts = {}
        for sid, s in self.sessions.items():
            if not s.decoding and s.queue:
                starts.setdefault((s.queue[0].shape[-1], s.prefilled), []).append(sid)
        for (n, cached), sids in starts.items():
            ss = [self.sessions[i] for i in sids]
            for s in ss:
                s.current = (s.queue[0], s.past)
            self._popped.extend(ss)
            wav = torch.cat([s.queue.pop(0) for s in ss], 0)
            feat = self.segment_features(ss, wav)
            codes = m.content_dvae.get_codebook_indices(feat.transpose(1, 2))
            cond = torch.cat([s.cond for s in ss], 0)
            prefix = eng.prefix_embeddings(cond.to(torch.float32).contiguous(), codes.to(torch.int32).contiguous())
            P = prefix.shape[1]
            slots = torch.tensor([s.slot for s in ss], device=dev, dtype=torch.int32)
            eng.prefill(slots, prefix, want_outputs=False, n_cached=cond.shape[1] if cached else 0)
            idx = slots.long()
            self.ids[idx] = 1
            self.ids[idx, P] = m.gpt.start_audio_token
            self.ids_len[idx] = P + 1
            self.finished[idx] = 0
            for s in ss:
                s.decoding, s.prefilled, s.done, s.p1 = True, True, 0, P + 1
                s.tokens.append([])
        # 2. one decode call for every stream that is decoding
        act = [(sid, s) for sid, s in self.sessions.items() if s.decoding]
        if not act:
            return out
        n = self.group      # every stream keeps its own group boundaries; steps past a stream's token cap are discarded below
        slots = torch.tensor([s.slot for _, s in act], device=dev, dtype=torch.int32)
        idx = slots.long()
        W = max(s.p1 + s.done for _, s in act) + n + 8
        ids = self.ids[idx, :W].contiguous()
        ids_len = self.ids_len[idx].contiguous()
        fin = self.finished[idx].contiguous()
        B = len(act)
        toks = torch.full((B, n), self.stop, device=dev, dtype=torch.int32)
        lats = torch.empty(B, n, m.gpt.model_dim, device=dev, dtype=torch.float32)
        self.params.seed = self.calls          # a fresh counter-RNG stream per call (only matters for top_k > 1)
        self.calls += 1
        eng.generate(slots, ids, ids_len, fin, self.params, 0, n, toks, lats, max_keys=W - 8)
        th = toks.cpu()                                           # (synchronises: the steps above have run)
        # a hand-off of the one-launch step that timed out (not all workgroups resident, e.g. another context on the GPU) raises here:
        # these tokens and latents are garbage and so are the K/V rows the steps appended.  Nothing of this call is kept or vocoded
        # (step() catches it); the library has switched the context to the launch-per-phase paths, on which the affected segments are
        # decoded again from their start
        eng.health()
        self.ids[idx, :W] = ids
        self.ids_len[idx] = ids_len
        self.finished[idx] = fin
        keep, emit = [], []
        for b, (sid, s) in enumerate(act):
            hit = (th[b] == self.stop).nonzero()
            nb = int(hit[0]) + 1 if hit.numel() else n            # the EOS step is part of the group (reference :189-196)
            nb = min(nb, self.max_new - s.done)                   # max_length cap of the reference loop (gpt.py:606,618)
            s.done += nb
            s.tokens[-1].append(toks[b:b + 1, :nb].long())
            if (hit.numel() and int(hit[0]) < nb) or s.done >= self.max_new:
                s.decoding = False
                s.tokens[-1] = torch.cat(s.tokens[-1], 1)
            keep.append(nb)
            emit.append(s.done > s.skip)                          # (a recovery re-run: the groups emitted before the failure stay silent)
        # 3. vocoder, batched over equal group lengths
        for nb in sorted(set(keep)):
            rows = [b for b in range(B) if keep[b] == nb and emit[b]]
            if not rows:
                continue
            audio = _vocode(m, lats[rows, :nb].contiguous())
            if audio is None:
                continue
            for j, b in enumerate(rows):
                sid, s = act[b]
                chunk, s.prev, s.overlap = handle_chunks(audio[j].squeeze(), s.prev, s.overlap, 1024)
                out.setdefault(sid, []).append(chunk)
        for _, s in act:
            if not s.decoding:
                s.skip = 0
        return out

    def _requeue(self, s):
        """put the segment a session was decoding back at the head of its queue, as it was when the segment started; the groups it has
        already emitted (whole groups: a failed call emits nothing) are skipped by the re-run, so the stream's audio continues where
        it stopped (greedy decoding repeats the tokens; the cross-fade state was only ever advanced by emitted chunks)"""
        seg, past = s.current
        s.skip = max(s.skip, s.done)
        s.queue.insert(0, seg)
        s.past = past
        s.decoding, s.prefilled, s.done = False, False, 0
        s.tokens.pop()
