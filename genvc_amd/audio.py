"""Audio I/O of the CLI ("next"): `load_audio` with the reference's checks
(/root/reference/utils.py:49-75) without torchaudio.  WAV PCM 16/24/32-bit and float32 are read with the
standard library; resampling (torchaudio.functional.resample's default: sinc_interp_hann, lowpass_filter_width=6,
rolloff=0.99) runs on the HIP kernel gvc_resample -- there is no CPU resampler in the product tree (the float64 restatement
that checks the kernel lives in oracle/).  Not part of the timed hot path."""
import struct
import wave

import numpy as np
import torch


def read_wav(path):
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"WAVE":
        raise ValueError(f"{path}: not a RIFF/WAVE file")
    pos, fmt, pcm = 12, None, None
    while pos + 8 <= len(data):
        cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
        body = data[pos + 8:pos + 8 + size]
        if cid == b"fmt ":
            fmt = struct.unpack("<HHIIHH", body[:16])
            if fmt[0] == 0xFFFE and len(body) >= 26:                       # WAVE_FORMAT_EXTENSIBLE
                fmt = (struct.unpack("<H", body[24:26])[0],) + fmt[1:]
        elif cid == b"data":
            pcm = body
        pos += 8 + size + (size & 1)
    if fmt is None or pcm is None:
        raise ValueError(f"{path}: missing fmt/data chunk")
    tag, ch, sr, _, _, bits = fmt
    if tag == 3 and bits == 32:
        x = np.frombuffer(pcm, "<f4").astype(np.float32)
    elif tag == 1 and bits == 16:
        x = np.frombuffer(pcm, "<i2").astype(np.float32) / 32768.0
    elif tag == 1 and bits == 32:
        x = np.frombuffer(pcm, "<i4").astype(np.float32) / 2147483648.0
    elif tag == 1 and bits == 24:
        b = np.frombuffer(pcm[:len(pcm) // 3 * 3], np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        x = ((v ^ 0x800000) - 0x800000).astype(np.float32) / 8388608.0
    else:
        raise ValueError(f"{path}: unsupported WAV format tag={tag} bits={bits}")
    return torch.from_numpy(x.reshape(-1, ch).T.copy()), sr


def load_audio(audiopath, sampling_rate, device=None):
    """reference utils.py:49-75: mono mix, resample, range sanity checks, clip to [-1, 1]; None on failure (the reference
    prints and returns None).  Resampling runs on the HIP kernel (gvc_resample) on `device` (default: the current CUDA device);
    a file that needs resampling without a GPU is a failure like any other: no CPU fallback."""
    try:
        audio, lsr = read_wav(audiopath)
        if audio.size(0) != 1:
            audio = torch.mean(audio, dim=0, keepdim=True)
        assert audio.size(1) > 10
        if lsr != sampling_rate:
            if device is None or not str(device).startswith("cuda"):
                if not torch.cuda.is_available():
                    raise RuntimeError(f"resampling {lsr} -> {sampling_rate} Hz needs the HIP library and a GPU (no CPU fallback)")
                device = "cuda"
            from .engine import resample as hip_resample
            audio = hip_resample(audio.to(device).contiguous(), lsr, sampling_rate).cpu()
    except Exception as e:                                                   # noqa: BLE001 (mirrors the reference)
        print(f"Error with {audiopath}. {e}")
        return None
    if torch.any(audio > 10) or not torch.any(audio < 0):
        print(f"Error with {audiopath}. Max={audio.max()} min={audio.min()}")
        return None
    audio.clip_(-1, 1)
    return audio


def save_wav(path, wav, sample_rate):
    """16-bit PCM like the reference's output file (torchaudio.save, infer.py:36)"""
    x = (wav.detach().cpu().clamp(-1, 1).numpy().reshape(-1) * 32767.0).round().astype("<i2")
    with wave.open(path, "wb") as f:
        f.setnchannels(1)
        f.setsampwidth(2)
        f.setframerate(sample_rate)
        f.writeframes(x.tobytes())
