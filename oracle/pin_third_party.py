"""One-command recipe that closes the three stages this repository can only pin to a published algorithm: the mel front end
and the resampler (torchaudio==2.3.0: /root/reference/utils.py:97-162, :58-62) and ContentVec (fairseq HuBERT: /root/reference/layers/
content_processor.py:11-31, including the `wav == 0` padding mask of :24).  TEST INFRASTRUCTURE, like everything under oracle/.

Run it where the wheels ARE importable (they are not in the build container: no network):

    python oracle/pin_third_party.py            # writes tests/golden/{mel,resample,hubert_fairseq}.npz and prints the oracle's deviation

  * torchaudio present: `torchaudio.transforms.MelSpectrogram` with the reference's arguments (and, when /root/reference is there, the
    reference's own TorchMelSpectrogram class) on the seeded 3 s input -> mel.npz; `torchaudio.functional.resample` for the four rate
    pairs the tests use -> resample.npz; both compared with oracle.mel_spectrogram / oracle.resample.
  * fairseq present: a fairseq `HubertModel` of HuBERT-base shape built from its config (no checkpoint ships with GenVC) loaded with the
    synthetic fairseq-named weights of genvc_amd.synth, run as content_processor.py does -- `extract_features(source, padding_mask =
    (wav == 0), output_layer = 12)` + final_proj -- on a plain input, a zero-padded tail (the harness's short last segment,
    inference_utils.py:47-48) and an input with interior digital silence -> hubert_fairseq.npz, compared with
    oracle.hubert_extract_features (whose mask semantics are restated from fairseq's published code and pinned by nothing else).
  * neither: prints "unpinned" for the stage and exits 0 -- the state of this repository's fixtures.

tests/test_oracle.py picks the files up when they exist (test_third_party_pins_when_present)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from genvc_amd import config as gcfg          # noqa: E402
from genvc_amd import synth                   # noqa: E402
from oracle import genvc_oracle as O          # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
RATES = [(96000, 16000), (96000, 24000), (22050, 16000), (16000, 24000)]


def pin_torchaudio():
    try:
        import torchaudio
    except Exception as e:                                   # noqa: BLE001
        print(f"mel / resampler: unpinned (torchaudio not importable: {type(e).__name__})")
        return False
    from genvc_amd.utils import DEFAULT_MEL_NORM_FILE, load_mel_norms
    norms = torch.from_numpy(load_mel_norms(DEFAULT_MEL_NORM_FILE))
    wav = synth.synth_audio(100, "ref", 72000)
    ms = torchaudio.transforms.MelSpectrogram(n_fft=2048, hop_length=256, win_length=1024, power=2, normalized=False, sample_rate=24000,
                                              f_min=0, f_max=8000, n_mels=80, norm="slaney")        # trainers/hifigan_trainer.py's arguments
    mel = torch.log(torch.clamp(ms(wav), min=1e-5)) / norms.view(1, -1, 1)                          # utils.py:150-162
    out = dict(torchaudio=str(torchaudio.__version__), mel=mel.numpy())
    if os.path.isdir("/root/reference"):
        sys.path.insert(0, "/root/reference")
        try:
            from utils import TorchMelSpectrogram     # the reference's own class, now importable
            ref = TorchMelSpectrogram(filter_length=2048, hop_length=256, win_length=1024, normalize=False, sampling_rate=24000,
                                      mel_fmin=0, mel_fmax=8000, n_mel_channels=80, mel_norm_file=None)
            out["mel_reference_class"] = (ref(wav) / norms.view(1, -1, 1)).numpy()
        except Exception as e:                               # noqa: BLE001
            print(f"  (reference TorchMelSpectrogram not importable here: {type(e).__name__}: {e})")
    d = float((O.mel_spectrogram(wav, norms) - mel).abs().max())
    print(f"mel: torchaudio {torchaudio.__version__}: oracle deviates by {d:.3e} (bar 1e-4)")
    np.savez_compressed(os.path.join(GOLD, "mel.npz"), **out)
    rs = dict(torchaudio=str(torchaudio.__version__))
    for o, n in RATES:
        x = synth.synth_audio(3, "rs", 6000 * o // 16000)
        y = torchaudio.functional.resample(x, o, n)
        rs[f"y_{o}_{n}"] = y.numpy()
        print(f"resample {o} -> {n}: oracle deviates by {float((O.resample(x, o, n) - y).abs().max()):.3e} (bar 2e-5)")
    np.savez_compressed(os.path.join(GOLD, "resample.npz"), **rs)
    return True


def pin_fairseq():
    try:
        from fairseq.models.hubert import HubertConfig, HubertModel
        from fairseq.tasks.hubert_pretraining import HubertPretrainingConfig
    except Exception as e:                                   # noqa: BLE001
        print(f"ContentVec: unpinned (fairseq not importable: {type(e).__name__})")
        return False
    c = gcfg.DEFAULT_HUBERT
    cfg = HubertConfig(encoder_layers=c["layers"], encoder_embed_dim=c["embed_dim"], encoder_ffn_embed_dim=c["ffn_dim"],
                       encoder_attention_heads=c["heads"], final_dim=c["final_dim"], extractor_mode="default", layer_norm_first=False,
                       conv_feature_layers=str([tuple(x) for x in c["conv_layers"]]), conv_pos=c["pos_conv_kernel"],
                       conv_pos_groups=c["pos_conv_groups"], dropout=0.0, attention_dropout=0.0, activation_dropout=0.0,
                       encoder_layerdrop=0.0, dropout_input=0.0, dropout_features=0.0, feature_grad_mult=0.0, untie_final_proj=False)
    m = HubertModel(cfg, HubertPretrainingConfig(), [list(range(100))]).eval()
    w = synth.make_weights(17, synth.hubert_weight_spec(c))
    missing, unexpected = m.load_state_dict(w, strict=False)
    print(f"  fairseq HubertModel: {len(missing)} keys left at init (label embeddings / mask embedding), {len(unexpected)} unexpected")
    assert not unexpected, unexpected
    plain = synth.synth_audio(17, "wav16000", 16000)
    tail = torch.zeros(1, 5120)
    tail[:, :2000] = synth.synth_audio(41, "tail", 2000)
    hole = synth.synth_audio(43, "hole", 16000).clone()
    hole[:, 6000:9000] = 0.0
    out = dict()
    with torch.no_grad():
        for name, wav in (("plain", plain), ("zero_tail", tail), ("interior_silence", hole)):
            feats = m.extract_features(source=wav, padding_mask=torch.eq(wav, torch.zeros_like(wav)), output_layer=12)[0]
            y = m.final_proj(feats)                                                        # content_processor.py:24-30
            out[name] = y.numpy()
            d = float((O.hubert_extract_features(w, c, wav) - y).abs().max())
            print(f"ContentVec {name}: oracle deviates by {d:.3e} (bar 2e-4)")
    np.savez_compressed(os.path.join(GOLD, "hubert_fairseq.npz"), **out)
    return True


if __name__ == "__main__":
    torch.manual_seed(0)
    a = pin_torchaudio()
    b = pin_fairseq()
    print("pinned:", ", ".join(n for n, ok in (("mel + resampler (torchaudio)", a), ("ContentVec (fairseq)", b)) if ok) or "nothing -- unpinned")
