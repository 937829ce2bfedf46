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
        self.content_sample_rate = c.get("dvae_sample_rate", 16000)
        # ContentVec = HuBERT-base under fairseq's parameter names (hifigan_trainer.py:85); another extractor with
        # the same interface can be passed in
        self.content_extractor = content_extractor or ContentvecExtractor(config.get("hubert_config"))
        v = config.get("vocoder_config")
        if hifigan is None and v is not None:                             # hifigan_trainer.py:47-55
            hifigan = HiFiGAN(v.input_feat_dim, v.upsample_initial_channel, v.resblock_kernel_sizes,
                              v.resblock_dilation_sizes, v.upsample_rates, v.upsample_kernel_sizes,
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
        embs = []
        if audio.shape[1] > sr * length:
            audio = audio[:, :sr * length]
        for i in range(0, audio.shape[1], sr * chunk_length):
            chunk = audio[:, i:i + sr * chunk_length]
            if chunk.size(-1) < sr * 0.33:
                continue
            mel = self.torch_mel_spectrogram_style_encoder(chunk.to(self.device).contiguous())
            embs.append(self.gpt.get_style_emb(mel, None))
        return torch.stack(embs).mean(dim=0).transpose(1, 2).contiguous()


def build_model(config, device, content_extractor=None, hifigan=None, max_slots=8, weight_dtype="fp32"):
    model = GenVCModel(config, content_extractor, hifigan)
    return model, (lambda: _finish(model, device, max_slots, weight_dtype))


def _finish(model, device, max_slots, weight_dtype="fp32"):
    model.eval()
    model.to(device)
    model.gpt.init_gpt_for_inference(max_slots=max_slots, weight_dtype=weight_dtype)
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
    model.load_state_dict(ckpt["model"], strict=False)                   # model_init.py:22
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


def _merge(dst, src):
    for k, v in dict(src).items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = gcfg.to_attr(v)
