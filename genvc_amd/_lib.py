"""ctypes binding of libgenvc_hip.so (include/genvc_hip.h).

The product path has NO CPU fallback: if the library is missing or a call fails this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GENVC_HIP_LIB") or os.path.join(_HERE, "lib", "libgenvc_hip.so")     # override: A/B runs of two builds

c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)


class GptDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_layer", "d_model", "n_head", "vocab", "max_mel_pos", "max_text_pos",
                                         "n_text", "max_seq", "max_slots", "max_rows", "weight_dtype")]


class SampleParams(C.Structure):
    _fields_ = [("repetition_penalty", C.c_float), ("temperature", C.c_float), ("top_p", C.c_float),
                ("top_k", C.c_int32), ("eos_token", C.c_int32), ("vocab", C.c_int32), ("seed", C.c_uint64)]


class PerceiverDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dim", "depth", "dim_context", "num_latents", "dim_head", "heads",
                                         "ff_mult", "max_batch", "max_frames")]


class DvaeDims(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("channels", "hidden_dim", "num_layers", "num_resnet_blocks", "kernel_size",
                                         "codebook_dim", "num_tokens", "max_batch", "max_frames")]


class HifiganDims(C.Structure):
    _fields_ = [("in_dim", C.c_int32), ("up_init_ch", C.c_int32), ("n_ups", C.c_int32), ("up_rates", C.c_int32 * 4),
                ("up_kernels", C.c_int32 * 4), ("n_kernels", C.c_int32), ("res_kernels", C.c_int32 * 4),
                ("res_dilations", (C.c_int32 * 2) * 4), ("max_batch", C.c_int32), ("max_frames", C.c_int32)]


class HubertDims(C.Structure):
    _fields_ = [("n_conv", C.c_int32), ("conv_dim", C.c_int32 * 8), ("conv_kernel", C.c_int32 * 8),
                ("conv_stride", C.c_int32 * 8)] + [(n, C.c_int32) for n in (
                    "embed_dim", "n_layers", "n_heads", "ffn_dim", "pos_conv_kernel", "pos_conv_groups", "final_dim",
                    "max_batch", "max_samples")]


_P = C.c_void_p
_SIGNATURES = {
    "gvc_version": (C.c_int, []),
    "gvc_last_error": (C.c_char_p, []),
    "gvc_gpt_create": (C.c_int, [C.POINTER(GptDims), C.POINTER(_P)]),
    "gvc_gpt_destroy": (C.c_int, [_P]),
    "gvc_gpt_bind_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "gvc_gpt_missing_weights": (C.c_int, [_P]),
    "gvc_gpt_prefix_embeddings": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "gvc_gpt_prefill": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "gvc_gpt_prefill_cached": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
    "gvc_gpt_prefill_cond": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P]),
    "gvc_gpt_decode_step": (C.c_int, [_P, _P, C.c_int32, _P, _P, _P, _P]),
    "gvc_gpt_reset_slots": (C.c_int, [_P, _P, C.c_int32, _P]),
    "gvc_gpt_latents": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "gvc_sample": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, _P, C.POINTER(SampleParams), C.c_int32, _P, _P]),
    "gvc_gpt_generate": (C.c_int, [_P, _P, C.c_int32, _P, C.c_int32, _P, _P, C.POINTER(SampleParams), C.c_int32,
                                   C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int32, _P]),
    "gvc_gpt_decode_variant": (C.c_int, [_P]),
    "gvc_gpt_rows_step_launches": (C.c_longlong, [_P]),
    "gvc_gpt_health": (C.c_int, [_P]),
    "gvc_gpt_warmup": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32]),
    "gvc_gpt_warmup_range": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "gvc_gpt_lazy_inits": (C.c_longlong, [_P]),
    "gvc_gpt_rearm": (C.c_int, [_P]),
    "gvc_gpt_time_kernel": (C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, c_f32p, c_i32p, _P]),
    "gvc_gemm_probe": (C.c_int, [C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, c_f32p, _P]),
    "gvc_perceiver_create": (C.c_int, [C.POINTER(PerceiverDims), C.POINTER(_P)]),
    "gvc_perceiver_destroy": (C.c_int, [_P]),
    "gvc_perceiver_bind_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "gvc_perceiver_missing_weights": (C.c_int, [_P]),
    "gvc_perceiver_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "gvc_mel_create": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32,
                                 c_f32p, C.POINTER(_P)]),
    "gvc_mel_destroy": (C.c_int, [_P]),
    "gvc_mel_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "gvc_dvae_create": (C.c_int, [C.POINTER(DvaeDims), C.POINTER(_P)]),
    "gvc_dvae_destroy": (C.c_int, [_P]),
    "gvc_dvae_bind_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "gvc_dvae_missing_weights": (C.c_int, [_P]),
    "gvc_dvae_encode": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "gvc_dvae_encode_frames": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "gvc_hubert_create": (C.c_int, [C.POINTER(HubertDims), C.POINTER(_P)]),
    "gvc_hubert_destroy": (C.c_int, [_P]),
    "gvc_hubert_bind_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "gvc_hubert_missing_weights": (C.c_int, [_P]),
    "gvc_hubert_frames": (C.c_int, [_P, C.c_int32]),
    "gvc_hubert_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "gvc_hifigan_create": (C.c_int, [C.POINTER(HifiganDims), C.POINTER(_P)]),
    "gvc_hifigan_destroy": (C.c_int, [_P]),
    "gvc_hifigan_bind_weight": (C.c_int, [_P, C.c_char_p, _P, C.c_int64, _P]),
    "gvc_hifigan_missing_weights": (C.c_int, [_P]),
    "gvc_hifigan_forward": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P]),
    "gvc_hifigan_forward_latents": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "gvc_resample_length": (C.c_int, [C.c_int32, C.c_int32, C.c_int32]),
    "gvc_resample": (C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P]),
    "gvc_vq_argmin": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P]),
}

_lib = None


GVC_ERR_TIMEOUT = -5          # include/genvc_hip.h: a hand-off of a one-launch step timed out; repeat the call after resetting the slots


class GenvcHipError(RuntimeError):
    """a library call failed; `.code` is its GVC_ERR_* return value (None: raised on the Python side)"""

    def __init__(self, msg, code=None):
        super().__init__(msg)
        self.code = code

    @property
    def is_handoff_timeout(self):
        return self.code == GVC_ERR_TIMEOUT


def exported_symbols():
    """Names declared in include/genvc_hip.h (used by the CPU test that checks the .so exports them)."""
    return sorted(_SIGNATURES)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise GenvcHipError(f"{LIB_PATH} not found: build it with `python -m genvc_amd.build` "
                                "(there is no CPU fallback for the product path)")
        import torch  # noqa: F401  (first: the HIP runtime must be the one torch loads, never two copies)
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = lib().gvc_last_error().decode(errors="replace")
        raise GenvcHipError(f"{what} failed with code {rc}: {msg}", rc)


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else C.c_void_p(t.data_ptr())


def stream():
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
