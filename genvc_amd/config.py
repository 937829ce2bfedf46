"""Plain-dict configuration for the hot path.

The reference reads these from Coqpit dataclasses stored in the checkpoint's `config`
dict (/root/reference/inference/model_init.py:11-12, configs/genVC_configs.py:127-139,
train_genVC.py:28-55).  The build reads the same field names from a plain dict, so no
coqpit is needed.
"""
import copy

# configs/genVC_configs.py:127-139 defaults + train_genVC.py:41-55 (gpt_n_heads=4)
DEFAULT_MODEL_ARGS = dict(
    gpt_layers=30, gpt_n_model_channels=1024, gpt_n_heads=4,
    gpt_max_audio_tokens=605, gpt_max_text_tokens=402, gpt_max_prompt_tokens=70,
    gpt_number_text_tokens=258, gpt_start_text_token=256, gpt_stop_text_token=257,
    gpt_num_audio_tokens=1026, gpt_start_audio_token=1024, gpt_stop_audio_token=1025,
    gpt_code_stride_len=1024, mel_norm_file=None,
)
# train_genVC.py:28-39
DEFAULT_CONTENT_DVAE = dict(num_channels=256, num_tokens=256, codebook_dim=512, hidden_dim=512,
                            num_resnet_blocks=3, kernel_size=3, num_layers=2, dvae_sample_rate=16000)
# configs/genVC_train_configs.py:76-80
DEFAULT_SAMPLING = dict(temperature=0.85, length_penalty=1.0, repetition_penalty=2.0, top_k=15, top_p=0.85)

# configs/vocoder_configs.py:7-20 (HiFi-GAN generator, ResBlock2)
DEFAULT_VOCODER = dict(input_feat_dim=1024, upsample_initial_channel=256, resblock_kernel_sizes=[3, 5, 7],
                       resblock_dilation_sizes=[[1, 2], [2, 6], [3, 12]], upsample_rates=[8, 8, 4],
                       upsample_kernel_sizes=[16, 16, 8], resblock_type="2", hop_length=256)
TINY_VOCODER = dict(DEFAULT_VOCODER, input_feat_dim=256, upsample_initial_channel=64)

# HuBERT-base / ContentVec (fairseq HubertModel): conv extractor (dim, kernel, stride), post-LN encoder
DEFAULT_HUBERT = dict(conv_layers=[(512, 10, 5)] + [(512, 3, 2)] * 4 + [(512, 2, 2)] * 2, embed_dim=768, layers=12, heads=12,
                      ffn_dim=3072, pos_conv_kernel=128, pos_conv_groups=16, final_dim=256)
TINY_HUBERT = dict(conv_layers=[(64, 10, 5)] + [(64, 3, 2)] * 4 + [(64, 2, 2)] * 2, embed_dim=128, layers=2, heads=2,
                   ffn_dim=256, pos_conv_kernel=128, pos_conv_groups=16, final_dim=256)

TINY_MODEL_ARGS = dict(DEFAULT_MODEL_ARGS, gpt_layers=2, gpt_n_model_channels=256, gpt_n_heads=4)
TINY_CONTENT_DVAE = dict(DEFAULT_CONTENT_DVAE, codebook_dim=64, hidden_dim=32, num_resnet_blocks=1)


def gpt_dims(model_args):
    """Derived sizes, following GPT.__init__ (/root/reference/layers/gpt.py:126-135,197-198)."""
    a = model_args
    max_cond = 1
    return dict(
        n_layer=a["gpt_layers"], d_model=a["gpt_n_model_channels"], n_head=a["gpt_n_heads"],
        num_audio_tokens=a["gpt_num_audio_tokens"], number_text_tokens=a["gpt_number_text_tokens"],
        start_text_token=a["gpt_start_text_token"], stop_text_token=a["gpt_stop_text_token"],
        start_audio_token=a["gpt_start_audio_token"], stop_audio_token=a["gpt_stop_audio_token"],
        max_gen_mel_tokens=a["gpt_max_audio_tokens"] - max_cond - 2,          # 602
        max_mel_pos=a["gpt_max_audio_tokens"] + 2 + max_cond,                 # 608
        max_text_pos=a["gpt_max_text_tokens"] + 2,                            # 404
        max_prompt_tokens=a["gpt_max_prompt_tokens"],
        code_stride_len=a["gpt_code_stride_len"],
        # positions of the inference GPT2Config: prompt + mel + text + 1 (gpt.py:198)
        max_seq=a["gpt_max_prompt_tokens"] + a["gpt_max_audio_tokens"] + 2 + max_cond
        + a["gpt_max_text_tokens"] + 2 + 1,                                   # 1083
    )


class AttrDict(dict):
    """dict with attribute access, standing in for the reference's Coqpit objects."""
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def to_attr(d):
    if isinstance(d, dict):
        return AttrDict({k: to_attr(v) for k, v in d.items()})
    return d


def default_config(tiny=False):
    cfg = dict(
        model_args=copy.deepcopy(TINY_MODEL_ARGS if tiny else DEFAULT_MODEL_ARGS),
        content_dvae_config=copy.deepcopy(TINY_CONTENT_DVAE if tiny else DEFAULT_CONTENT_DVAE),
        vocoder_config=copy.deepcopy(TINY_VOCODER if tiny else DEFAULT_VOCODER),
        hubert_config=copy.deepcopy(TINY_HUBERT if tiny else DEFAULT_HUBERT),
        audio=dict(sample_rate=24000),
        **DEFAULT_SAMPLING,
    )
    return to_attr(cfg)
