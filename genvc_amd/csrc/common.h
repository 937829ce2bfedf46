// Shared device/host helpers for libgenvc_hip (gfx950 only: wave64, DPP, 160 KiB LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/genvc_hip.h"

namespace gvc {

// ---- error plumbing -------------------------------------------------------------------------
void set_error(const char* fmt, ...);

#define GVC_CHECK_HIP(expr)                                                                  \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            gvc::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                        \
            return GVC_ERR_HIP;                                                              \
        }                                                                                    \
    } while (0)

#define GVC_REQUIRE(cond, code, ...)      \
    do {                                  \
        if (!(cond)) {                    \
            gvc::set_error(__VA_ARGS__);  \
            return (code);                \
        }                                 \
    } while (0)

#define GVC_LAUNCH_CHECK()                                                           \
    do {                                                                             \
        hipError_t _e = hipGetLastError();                                           \
        if (_e != hipSuccess) {                                                      \
            gvc::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e), \
                           __FILE__, __LINE__);                                      \
            return GVC_ERR_HIP;                                                      \
        }                                                                            \
    } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---- wave64 reductions (DPP inside a 16-lane row, readlane across rows) ----------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(
        float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// sum over each 16-lane row, result in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x140>(v);  // row_mirror
    return v;
}

__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_mov<0xB1>(v));
    v = fmaxf(v, dpp_mov<0x4E>(v));
    v = fmaxf(v, dpp_mov<0x141>(v));
    v = fmaxf(v, dpp_mov<0x140>(v));
    return v;
}

// sum over the whole wave, result uniform (SGPR-broadcast) in every lane
__device__ __forceinline__ float wave_sum(float v) {
    v = row16_sum(v);
    int b = __builtin_bit_cast(int, v);
    float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    r += __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16));
    r += __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32));
    r += __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48));
    return r;
}

__device__ __forceinline__ float wave_max(float v) {
    v = row16_max(v);
    int b = __builtin_bit_cast(int, v);
    float r = __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 0));
    r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 16)));
    r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 32)));
    r = fmaxf(r, __builtin_bit_cast(float, __builtin_amdgcn_readlane(b, 48)));
    return r;
}

__device__ __forceinline__ float dot4(const float4& a, const float4& b) {
    return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w)));
}

// fp32 -> bf16, round to nearest even (finite inputs); the rule k_round_bf16 applies to the weights
__device__ __forceinline__ unsigned short f32_to_bf16(float v) {
    const unsigned u = __float_as_uint(v);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
}

// gelu_new of HF GPT-2 (tanh approximation), reference activation of GPT2MLP: 0.5 x (1 + tanh(u)), u = sqrt(2 / pi) (x + 0.044715 x^3).
// Evaluated as x / (1 + exp(-2 u)) -- the same function (1 + tanh(u) = 2 / (1 + exp(-2 u))) on the hardware exp / rcp: ~8 instructions
// instead of the library tanhf's ~40 (it sits on the critical path of every decode phase D, four times per final lane in the rows
// step), and without the cancellation of 1 + tanh(u) for negative x.  Within 3e-7 relative of the fp32 reference expression.
__device__ __forceinline__ float gelu_new(float x) {
    const float k = 0.7978845608028654f;  // sqrt(2/pi)
    const float u = k * (x + 0.044715f * x * x * x);
    return x * __frcp_rn(1.0f + __expf(-2.0f * u));
}

// exact (erf) GELU of the Perceiver's GEGLU (perceiver_encoder.py:205-208)
__device__ __forceinline__ float gelu_erf(float x) {
    return 0.5f * x * (1.0f + erff(x * 0.7071067811865476f));
}

// block-wide (256 threads = 4 waves) sum through LDS scratch of >= 4 floats
__device__ __forceinline__ float block4_sum(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

__device__ __forceinline__ float block4_max(float v, float* red) {
    v = wave_max(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

}  // namespace gvc
