"""CLI of the reference (/root/reference/infer.py:8-36) on the MI355X-native path.

Flags and defaults are the reference's; added: --seg_len (the reference hard-codes 6.0; the README's
"1-second chunk" numbers correspond to --seg_len 1.0), --stream_chunk_size, --synthetic (no checkpoint
ships with the reference: run the same pipeline on deterministic synthetic weights) and --save_tokens.
The waveform is written like the reference does (24 kHz PCM16); --save_tokens also stores the codec tokens/latents.
"""
import argparse

import torch

from genvc_amd.audio import load_audio, save_wav
from genvc_amd.inference.inference_utils import synthesize_utt, synthesize_utt_streaming
from genvc_amd.inference.model_init import model_init, model_init_synthetic

if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--model_path", type=str, default="pre_trained/GenVC_large.pth")
    parser.add_argument("--device", type=str, default="cuda")
    parser.add_argument("--src_wav", type=str, default="samples/EF4_ENG_0112_1.wav")
    parser.add_argument("--ref_audio", type=str, default="samples/EM1_ENG_0037_1.wav")
    parser.add_argument("--output_path", type=str, default="samples/converted.wav")
    parser.add_argument("--top_k", type=int, default=15)
    parser.add_argument("--streaming", action="store_true")
    parser.add_argument("--seg_len", type=float, default=6.0)
    parser.add_argument("--stream_chunk_size", type=int, default=8)
    parser.add_argument("--synthetic", action="store_true", help="deterministic synthetic weights instead of --model_path")
    parser.add_argument("--tiny", action="store_true", help="with --synthetic: the 2-layer test architecture")
    parser.add_argument("--save_tokens", type=str, default=None)
    parser.add_argument("--weights", type=str, default="fp32", choices=["fp32", "bf16", "bf16_kv", "bf16_act"],
                        help="GPT weight / KV-cache storage on the GPU (fp32 = the reference's numerics)")
    args = parser.parse_args()

    if args.synthetic:
        from genvc_amd import config as gcfg
        model, config = model_init_synthetic(gcfg.default_config(tiny=args.tiny), device=args.device, weight_dtype=args.weights)
    else:
        model, config = model_init(args.model_path, args.device, weight_dtype=args.weights)
    model.config.top_k = args.top_k
    src_wav = load_audio(args.src_wav, model.content_sample_rate, device=args.device)
    ref_audio = load_audio(args.ref_audio, model.config.audio.sample_rate, device=args.device)
    if src_wav is None or ref_audio is None:
        raise SystemExit("could not load the input audio")

    if args.streaming:
        out = synthesize_utt_streaming(model, src_wav, ref_audio, seg_len=args.seg_len,
                                       stream_chunk_size=args.stream_chunk_size, return_details=True)
        toks = torch.cat(out["tokens"], 1)
        lat = torch.cat(out["latents"], 1)
    else:
        out = synthesize_utt(model, src_wav, ref_audio, seg_len=args.seg_len, return_details=True)
        toks = torch.cat(out["codes"]).unsqueeze(0)
        lat = out["latents"]
    print(f"generated {toks.shape[-1]} codec tokens, latents {tuple(lat.shape)}")
    if out["wav"] is not None:
        save_wav(args.output_path, out["wav"], config.audio.sample_rate)
    else:
        print("no vocoder in the model: waveform not written")
    if args.save_tokens:
        torch.save(dict(tokens=toks.cpu(), latents=lat.cpu()), args.save_tokens)
