"""`model_init(checkpoint_path, device) -> (model, config)` with the reference's contract
(/root/reference/inference/model_init.py:9-34): the checkpoint is `{'config': dict, 'model': state_dict}`
with keys prefixed `gpt.`, `content_dvae.`, `hifigan.`, `content_extractor.model.`, ...; it is loaded
non-strictly.  The container mirrors the attributes of `HiFiGANTrainer` that the inference harness uses
(trainers/hifigan_trainer.py:25-167, 438-455).  No coqpit: the config is a plain dict (genvc_amd.config).
"""
import torch
from torch import nn

from .. import config as gcfg
from ..layers.content_processor import ContentvecExtractor
from ..layers.dvae import DiscreteVAE
from ..layers.gpt import GPT
from ..layers.hifigan import HiFiGAN
from ..utils import DEFAULT_MEL_NORM_FILE, TorchMelSpectrogram


class GenVCModel(nn.Module):
    """inference view of the reference's HiFiGANTrainer"""

    def __init__(self, config, content_extractor=None, hifigan=None):
        super().__init__()
        self.config = config
        a = config.model_args
        self.gpt = GPT(layers=a.gpt_layers, model_dim=a.gpt_n_model_channels, heads=a.gpt_n_heads,
                       max_text_tokens=a.gpt_max_text_tokens, max_mel_tokens=a.gpt_max_audio_tokens,
                       max_prompt_tokens=a.gpt_max_prompt_tokens, number_text_tokens=a.gpt_number_text_tokens,
                       start_text_token=a.gpt_start_text_token, stop_text_token=a.gpt_stop_text_token,
                       num_audio_tokens=a.gpt_num_audio_tokens, start_audio_token=a.gpt_start_audio_token,
                       stop_audio_token=a.gpt_stop_audio_token, code_stride_len=a.gpt_code_stride_len)
        c = config.content_dvae_config
        self.content_dvae = DiscreteVAE(channels=c.num_channels, normalization=None, positional_dims=1,
                                        num_tokens=c.num_tokens, codebook_dim=c.codebook_dim, hidden_dim=c.hidden_dim,
                                        num_resnet_blocks=c.num_resnet_blocks, kernel_size=c.kernel_size,
                                        num_layers=c.num_layers, use_transposed_convs=False)
        # hifigan_trainer.py:146 reads config.content_dvae_config.audio.dvae_sample_rate
        self.content_sample_rate = (c.get("audio") or {}).get("dvae_sample_rate", c.get("dvae_sample_rate", 16000))
        # ContentVec = HuBERT-base under fairseq's parameter names (hifigan_trainer.py:85); another extractor with
        # the same interface can be passed in
        self.content_extractor = content_extractor or ContentvecExtractor(config.get("hubert_config"))
        v = config.get("vocoder_config")
        if hifigan is None and v is not None:                             # hifigan_trainer.py:47-55
            # the reference's config spells the field `upsample_kernal_sizes` (configs/vocoder_configs.py:19, hifigan_trainer.py:53)
            up_kernels = v.get("upsample_kernal_sizes", v.get("upsample_kernel_sizes"))
            hifigan = HiFiGAN(v.input_feat_dim, v.upsample_initial_channel, v.resblock_kernel_sizes,
                              v.resblock_dilation_sizes, v.upsample_rates, up_kernels,
                              resblock_type=v.get("resblock_type", "2"))
        self.hifigan = hifigan
        self.hifigan_scale_factor = a.gpt_code_stride_len / (v.get("hop_length", 256) if v is not None else 256)  # :56
        self.torch_mel_spectrogram_style_encoder = TorchMelSpectrogram(
            filter_length=2048, hop_length=256, win_length=1024, normalize=False,
            sampling_rate=config.audio.sample_rate, mel_fmin=0, mel_fmax=8000, n_mel_channels=80,
            mel_norm_file=a.get("mel_norm_file") or DEFAULT_MEL_NORM_FILE)
        self._device = torch.device("cpu")

    @property
    def device(self):
        return self._device

    def to(self, device):
        self._device = torch.device(device)
        return super().to(device)

    @torch.inference_mode()
    def get_gpt_cond_latents(self, audio, sr, length=30, chunk_length=6):
        """reference trainers/hifigan_trainer.py:438-455 -> [1, 32, d]"""
        # a conditioning chain started by get_gpt_cond_latents_async whose result() was never called (no segment, an exception)
        # may still be running on the side stream with the per-context mel / Perceiver scratch buffers: wait for it first
        side = getattr(self, "_cond_stream", None)
        if side is not None and audio.is_cuda and torch.cuda.current_stream(audio.device) != side:
            torch.cuda.current_stream(audio.device).wait_stream(side)
        embs = []
        if audio.shape[1] > sr * length:
            audio = audio[:, :sr * length]
        for i in range(0, audio.shape[1], sr * chunk_length):
            chunk = audio[:, i:i + sr * chunk_length]
            if chunk.size(-1) < sr * 0.33:
                continue
            # (the mel kernel also writes the frames-major copy the Perceiver reads: get_style_emb's permute(0, 2, 1).contiguous() is free)
            mel, mel_fm = self.torch_mel_spectrogram_style_encoder(chunk.to(self.device).contiguous(), frames_major=True)
            embs.append(self.gpt.get_style_emb(mel, None, frames_major=mel_fm))
        return torch.stack(embs).mean(dim=0).transpose(1, 2).contiguous()

    def get_gpt_cond_latents_async(self, audio, sr, length=30, chunk_length=6, after=None):
        """get_gpt_cond_latents on a second HIP stream: the reference speaker's mel + Perceiver chain (~25 short launches) does not
        depend on the source audio, so the streaming harness runs it BESIDE the first segment's ContentVec + DVAE chain instead of
        in front of it (first-chunk latency; the arithmetic and its order inside each chain are unchanged).  Returns a handle whose
        .result() makes the caller's current stream wait for the latents and returns them.
        `after` (a torch.cuda.Event on the caller's stream at which `audio` is ready): the side chain waits for THAT instead of for everything
        the caller's stream holds -- so a caller can enqueue its own first kernels (ContentVec of the first segment) BEFORE spending the
        ~0.3 ms of host time this chain's ~25 launches take, and the GPU is not idle meanwhile."""
        return _CondFuture(self, audio, sr, length, chunk_length, after)

    @torch.inference_mode()
    def warmup(self, seg_len=1.0, streams=1, ref_seconds=3.0, stream_chunk_size=8, top_k=None, max_new_tokens=None):
        """Everything the FIRST conversion of this shape would otherwise pay inside its latency window (the reference leaves warm-up
        to the user: /root/reference/infer.py:27-30 runs a conversion first).  For `streams` concurrent streams of `seg_len`-second
        segments and a `ref_seconds` reference:
          * GPT context: gvc_gpt_warmup for every context class the generation calls of a segment reach (the one-launch steps' buffers,
            weight pack and topology probe; the captured step graphs) -- after it no GPT data-path call allocates or synchronises;
          * ContentVec / DVAE / HiFi-GAN / mel + Perceiver: one pass over zeros of the real shapes (their per-shape graphs and plans).
        No token is generated and no KV slot is left occupied."""
        dev = self.device
        g = self.gpt
        g._need_engine()
        eng = g.engine
        n_src = int(seg_len * self.content_sample_rate)
        wav = torch.zeros(streams, n_src, device=dev)
        wav[:, ::7] = 0.01                                   # (not digital silence: the `wav == 0` frame mask must not swallow the input)
        feat = self.content_extractor.extract_content_features(wav)
        codes = self.content_dvae.get_codebook_indices(feat.transpose(1, 2))
        ref = torch.zeros(1, int(ref_seconds * self.config.audio.sample_rate), device=dev)
        ref[:, ::5] = 0.01
        cond = self.get_gpt_cond_latents(ref, self.config.audio.sample_rate)
        n0 = cond.shape[1] + codes.shape[1] + 3
        top_k = self.config.top_k if top_k is None else top_k
        max_new = int(max_new_tokens or g.max_gen_mel_tokens)
        grp = max(stream_chunk_size, 1)
        # the calls get_generator / _advance will make reach max_keys = n0 + grp ... n0 + max_new: the library warms every context class in
        # that range (its thresholds are its own: gvc_gpt_warmup_range)
        hi = min(n0 + max_new, eng.dims["max_seq"] - 1)
        eng.warmup_range(streams, min(n0 + grp, hi), hi, top_k)
        if self.hifigan is not None:
            lat = torch.zeros(streams, grp, g.model_dim, device=dev)
            for n in {grp, max(1, max_new % grp)}:
                self.hifigan.forward_latents(lat[:, :n].contiguous(), int(self.hifigan_scale_factor))
        torch.cuda.synchronize()
        return self

    @torch.no_grad()
    def inference(self, src_audio, cond_latent, do_sample=True, top_p=0.85, top_k=15, temperature=0.75, num_beams=1,
                  length_penalty=1.0, repetition_penalty=10.0, output_attentions=False, repass_latents=False):
        """reference trainers/hifigan_trainer.py:457-500: one source segment [1,T] + conditioning latents -> waveform
        [1,1,1024 n]: ContentVec -> content codes -> generate -> strip stop tokens -> latent re-pass -> x4 linear
        interpolation -> HiFi-GAN.  (The reference's 0-d collapse at exactly one non-stop token, SURVEY appendix B.9, is
        guarded: boolean indexing keeps the dimension.)  The latents are the decode loop's own unless `repass_latents=True`
        (inference_utils._segment_latents)."""
        feat = self.content_extractor.extract_content_features(src_audio)
        codes = self.content_dvae.get_codebook_indices(feat.transpose(1, 2))
        gen = self.gpt.generate(cond_latent, codes, do_sample=do_sample, top_p=top_p, top_k=top_k, temperature=temperature,
                                num_beams=num_beams, length_penalty=length_penalty, repetition_penalty=repetition_penalty,
                                output_attentions=output_attentions)[0]
        gen = gen[gen != self.gpt.stop_audio_token]
        if gen.numel() == 0:
            return torch.zeros(1, 1, 0, device=self.device)
        from genvc_amd.inference.inference_utils import _segment_latents
        lat = _segment_latents(self, cond_latent, codes, gen, repass_latents)
        mel_input = torch.nn.functional.interpolate(lat.transpose(1, 2), scale_factor=[self.hifigan_scale_factor],
                                                    mode="linear").squeeze(1)
        return self.hifigan.forward(mel_input)


class _CondFuture:
    def __init__(self, model, audio, sr, length, chunk_length, after=None):
        if not audio.is_cuda:
            audio = audio.to(model.device)
        main = torch.cuda.current_stream(audio.device)
        side = getattr(model, "_cond_stream", None)
        if side is None:
            side = model._cond_stream = torch.cuda.Stream(device=audio.device)
        if after is not None:
            side.wait_event(after)                   # the audio is ready at `after`; what the caller enqueued behind it runs beside this chain
        else:
            side.wait_stream(main)                   # (the audio may still be in flight on the caller's stream)
        audio.record_stream(side)
        with torch.cuda.stream(side):
            self.cond = model.get_gpt_cond_latents(audio, sr, length, chunk_length)
        self.side = side
        # residency before issue: a GPT call on another stream (a one-launch step needs every CU) waits for this chain first
        done = torch.cuda.Event()
        done.record(side)
        eng = getattr(model.gpt, "engine", None)
        if eng is not None:
            eng.watch_stream(done)

    def result(self):
        main = torch.cuda.current_stream(self.cond.device)
        main.wait_stream(self.side)
        self.cond.record_stream(main)
        return self.cond


def build_model(config, device, content_extractor=None, hifigan=None, max_slots=8, weight_dtype="fp32"):
    model = GenVCModel(config, content_extractor, hifigan)
    return model, (lambda: _finish(model, device, max_slots, weight_dtype))


def _finish(model, device, max_slots, weight_dtype="fp32"):
    model.eval()
    model.to(device)
    # prefill capacity: every slot may bring a 6 s segment (110 rows) into one batched call
    model.gpt.init_gpt_for_inference(max_slots=max_slots, max_rows=max(4096, 128 * max_slots), weight_dtype=weight_dtype)
    mb = max(2, max_slots)
    model.content_dvae.bind(max_batch=max(8, max_slots))
    if hasattr(model.content_extractor, "bind"):
        if hasattr(model.content_extractor, "max_batch"):
            model.content_extractor.max_batch = mb
        model.content_extractor.bind()
    if model.hifigan is not None:
        model.hifigan.bind(max_batch=mb)
    return model


@torch.inference_mode()
def model_init(checkpoint_path, device, content_extractor=None, hifigan=None, weight_dtype="fp32"):
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    config = gcfg.default_config()
    _merge(config, ckpt["config"])
    model, finish = build_model(config, device, content_extractor, hifigan, weight_dtype=weight_dtype)
    _load_checked(model, ckpt["model"])                                  # model_init.py:22 (strict=False)
    finish()
    print("Model initialized")
    return model, config


@torch.inference_mode()
def model_init_synthetic(config=None, seed=1, device="cuda", max_slots=8, weight_dtype="fp32"):
    """No checkpoint ships with the reference: deterministic synthetic weights of the same architecture."""
    from .. import synth
    config = config or gcfg.default_config()
    model, finish = build_model(config, device, max_slots=max_slots, weight_dtype=weight_dtype)
    dims = gcfg.gpt_dims(config.model_args)
    w = {"gpt." + k: v for k, v in synth.make_weights(seed, synth.gpt_weight_spec(dims), device=device).items()}
    w.update({"content_dvae." + k: v for k, v in
              synth.make_weights(seed, synth.dvae_weight_spec(config.content_dvae_config), device=device).items()})
    if model.hifigan is not None:
        w.update({"hifigan." + k: v for k, v in
                  synth.make_weights(seed, synth.hifigan_weight_spec(config.vocoder_config), device=device).items()})
    if isinstance(model.content_extractor, ContentvecExtractor):
        w.update({"content_extractor.model." + k: v for k, v in
                  synth.make_weights(seed, synth.hubert_weight_spec(model.content_extractor.cfg), device=device).items()})
    model.to(device)
    missing, unexpected = model.load_state_dict(w, strict=False)
    assert not unexpected, unexpected
    finish()
    return model, config


_REQUIRED_PREFIXES = ("gpt.", "content_dvae.", "hifigan.", "content_extractor.model.")


def _load_checked(model, state):
    """load_state_dict(strict=False) as the reference does (model_init.py:22: the checkpoint also holds discriminators, the
    acoustic DVAE ...), but a checkpoint that LACKS a tensor of the inference path would leave a placeholder (zeros / random)
    in the engines and produce garbage audio silently: those keys are reported.  Weight-norm tensors saved by newer torch
    (`parametrizations.weight.original0/1`) are renamed to the `weight_g` / `weight_v` the reference's checkpoints carry."""
    state = dict(state)
    for k in list(state):
        if ".parametrizations.weight.original" in k:
            new = k.replace(".parametrizations.weight.original0", ".weight_g").replace(".parametrizations.weight.original1", ".weight_v")
            state[new] = state.pop(k)
    missing, unexpected = model.load_state_dict(state, strict=False)
    lost = [k for k in missing if k.startswith(_REQUIRED_PREFIXES)
            and not k.endswith((".attn.bias", ".attn.masked_bias"))]
    if lost:
        raise RuntimeError(f"checkpoint lacks {len(lost)} tensors of the inference path, e.g. {lost[:6]} "
                           f"(unexpected keys: {len(unexpected)})")
    return missing, unexpected


def _merge(dst, src):
    for k, v in dict(src).items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = gcfg.to_attr(v)
