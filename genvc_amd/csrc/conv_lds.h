// One-round-trip convolution kernel shared by the vocoder (hifigan.hip) and the ContentVec feature extractor (hubert.hip).
#pragma once
#include "gemm.h"

namespace gvc {

// ---------------------------------------------------------------------------------------------
// Convolutions over time-major activations whose K = taps * CI extent is small -- HiFi-GAN: ResBlock convs (CI = 32 / 64 / 128 channels
// in and out, <= 7 taps, dilation <= 12), the polyphase ConvTranspose1d layers (3 taps of CI = 256 / 128 / 64), conv_pre in 64-channel
// slices; ContentVec: the strided feature-extractor convs 1..6 (2-3 taps of CI = 512).  Instead of the tiled GEMM's k-loop (one
// global->LDS->sync round trip per 32 columns: 11 us for 0.05 GFLOP) a workgroup pays ONE memory round trip:
//   * it owns 32 (64 when CI <= 64) frames x 16 output columns; their (frames-1)*stride + (k-1)*dil + 1 input rows go to LDS once
//     (leaky-ReLU -- and, for the input of a HiFi-GAN stage, the sum of the three ResBlock outputs and the 1/3 -- applied on the way
//     in; rows of a strided conv are stored de-interleaved by row % stride, so the fragment reads stay bank-conflict-free);
//   * weights are stored in MFMA fragment order (FM16, gemm.h) and go straight from global memory into registers, 1 KiB per
//     wave-wide load, all of a wave's loads in flight before the input rows are staged;
//   * the NW waves split the (tap, 16-channel block) steps of the reduction (v_mfma_f32_16x16x4_f32, two M tiles share the B
//     operand) and combine through LDS; bias, residual and GELU in the epilogue.
// The three ResBlocks of a stage are independent given the stage input (hifigan.py:224-229: xs += resblocks[i*nk + j](x)), so
// ONE launch runs the first conv of all three (grid.y = job x column tile) and a second launch the second convs; each ResBlock
// writes its own output plane and the consumer of the stage adds the planes while staging, in the reference's order.
// ---------------------------------------------------------------------------------------------
// 16-frame M tiles per workgroup of k_conv_lds: 64 frames for the narrow late stages (thousands of frames: half the workgroups, every
// one resident at once), 32 otherwise
__host__ __device__ constexpr int conv_lds_mt(int ci, bool split) { return (ci <= 64 && !split) ? 4 : 2; }

__device__ __forceinline__ float conv_lds_act(float v, int act) {
    if (act == ACT_GELU_ERF) return gelu_erf(v);
    if (act == ACT_RELU) return v > 0.f ? v : 0.f;
    return v;
}

struct ConvLdsJob {
    const float4* wp;                    // FM16 copy of [N][k*CI]
    const float* b;                      // [N]
    int k, dil, row_off;                 // output frame t reads input rows x_row0 + t*stride + row_off + j*dil, j < k
};

struct ConvLdsArgs {
    // input: time-major [B][x_rows][ldx]; job j reads plane(s) at x + j*x_ps (x_ps = 0: every job reads the same input); row indices
    // are clamped to x_rows - 1 (rows past the end only feed frames >= T)
    const float* x; long long x_bs, x_ps, x_ss;      // x_ss: distance between the NSUM planes that are added while staging
    float x_scale, slope;                            // staged value = lrelu(x_scale * (p0 + p1 + p2)); x_scale only with NSUM > 1
    // output element (t, n) of job j, batch b: y[b*y_bs + j*y_ps + y_off + t*ldy + n]; resid (nullable) is indexed the same way
    float* y; long long y_bs, y_ps, y_off; int ldy;
    const float* resid; long long r_ps;
    int T, ntiles;                                   // output frames, N / 16
    int ldx;                                         // floats between input rows (CI, or more when the jobs are channel slices)
    int x_row0, x_rows, stride;                      // first input row of frame 0 (the padding of a padded buffer), rows per batch element, conv stride
    int act;                                         // GemmAct applied after bias and residual (ACT_NONE / ACT_RELU / ACT_GELU_ERF)
    // SPLIT (conv_pre: few outputs, K = 7 * 1024): job j < split is the channel slice [j*CI, (j+1)*CI) of every tap -- x_ps = CI, weights
    // job[0].wp + j*wp_js -- and writes RAW partial sums to plane j of y; the last workgroup to finish a tile (cnt) adds the
    // planes in order, adds the bias (and resid, indexed like yf), applies act and writes yf[b*yf_bs + yf_off + t*ldy + n]
    int split; long long wp_js; int* cnt;
    float* yf; long long yf_bs, yf_off;
    ConvLdsJob job[3];
};

typedef float hf_f32x4 __attribute__((ext_vector_type(4)));

template <int CI, int NW, int NSUM, bool SPLIT = false>
__global__ __launch_bounds__(NW * 64) void k_conv_lds(const ConvLdsArgs A) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int XS = CI + 4, NTH = NW * 64, C4 = CI / 4, CB = CI / 16;
    constexpr int MT = conv_lds_mt(CI, SPLIT), TF = 16 * MT;       // 16-frame M tiles, frames per workgroup
    constexpr int MAXS = CI <= 64 ? 4 : (CI >= 512 ? 12 : 8);       // steps per wave held in registers (7 taps: 3.5 / 3.5 / 7; ups: 6; 3 x 512: 12)
    float* Xs = lds;                       // [R][XS]
    float* red = lds;                      // [NW][4 * MT][64] after the MFMA loop (aliases Xs)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jb = blockIdx.y / A.ntiles, tile = blockIdx.y - jb * A.ntiles;
    const ConvLdsJob J = A.job[SPLIT ? 0 : jb];
    const int t0 = blockIdx.x * TF, b = blockIdx.z;
    const int S = A.stride;
    const int nsteps = J.k * CB, R = (TF - 1) * S + (J.k - 1) * J.dil + 1, RH = (R + S - 1) / S;      // input row r lives in LDS row (r % S)*RH + r / S
    // this wave's B operands: steps wave, wave + NW, ...; requested before anything else
    const float4* wp = J.wp + (SPLIT ? (size_t)jb * A.wp_js : 0) + (size_t)tile * nsteps * 64 + lane;
    float4 wv[MAXS];
#pragma unroll
    for (int u = 0; u < MAXS; ++u) wv[u] = wp[(size_t)min(wave + NW * u, nsteps - 1) * 64];
    const float bn = J.b ? J.b[tile * 16 + (lane & 15)] : 0.f;
    __builtin_amdgcn_sched_barrier(0);
    // input rows -> LDS: UL requests per thread go out before the first LDS store (rows past the buffer only feed frames >= T)
    const int last_row = A.x_rows - 1, row0 = A.x_row0 + t0 * S + J.row_off;
    const float* xb = A.x + (size_t)b * A.x_bs + (size_t)jb * A.x_ps;
    constexpr int UL = NSUM == 1 ? (CI >= 512 ? 9 : 8) : 4;
    for (int i0 = tid; i0 < R * C4; i0 += NTH * UL) {
        float4 xv[UL][NSUM];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int i = min(i0 + u * NTH, R * C4 - 1);
            const float* src = xb + (size_t)min(row0 + i / C4, last_row) * A.ldx + (i % C4) * 4;
#pragma unroll
            for (int p = 0; p < NSUM; ++p) xv[u][p] = *reinterpret_cast<const float4*>(src + (size_t)p * A.x_ss);
        }
        __builtin_amdgcn_sched_barrier(0);         // (the scheduler would sink every request to just above its store)
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int i = i0 + u * NTH;
            if (i < R * C4) {
                float4 v = xv[u][0];
                if constexpr (NSUM > 1) {
#pragma unroll
                    for (int p = 1; p < NSUM; ++p) { v.x += xv[u][p].x; v.y += xv[u][p].y; v.z += xv[u][p].z; v.w += xv[u][p].w; }
                    v.x *= A.x_scale; v.y *= A.x_scale; v.z *= A.x_scale; v.w *= A.x_scale;
                }
                v.x = v.x > 0.f ? v.x : v.x * A.slope; v.y = v.y > 0.f ? v.y : v.y * A.slope;
                v.z = v.z > 0.f ? v.z : v.z * A.slope; v.w = v.w > 0.f ? v.w : v.w * A.slope;
                const int r = i / C4;
                *reinterpret_cast<float4*>(&Xs[((r % S) * RH + r / S) * XS + (i % C4) * 4]) = v;
            }
        }
    }
    // thread (wave, lane) will finish accumulator registers r = wave, wave + NW, ... of the tile -- frame 16*(r/4) + 4*(lane/16) + r%4 --
    // and asks for its residuals now, so that the round trip hides under the matrix work
    constexpr int RPW = 4 * MT / NW;
    const int n = tile * 16 + (lane & 15);
    const size_t ob = (size_t)b * A.y_bs + A.y_off + n;
    float r1[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave + NW * rr, m = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
        r1[rr] = (A.resid && t0 + m < A.T) ? A.resid[ob + (size_t)jb * A.r_ps + (size_t)(t0 + m) * A.ldy] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    hf_f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
    const int fi = lane & 15, fg = lane >> 4;
    auto step = [&](int it, const float4& w4) {
        const int tap = it / CB, cb = it - tap * CB;
        const int ro = tap * J.dil;
        const float* xp = &Xs[((ro % S) * RH + ro / S + fi) * XS + 16 * cb + 4 * fg];
        float4 a[MT];
#pragma unroll
        for (int t = 0; t < MT; ++t) a[t] = *reinterpret_cast<const float4*>(xp + 16 * t * XS);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].x, w4.x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].y, w4.y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].z, w4.z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t].w, w4.w, acc[t], 0, 0, 0);
    };
#pragma unroll
    for (int u = 0; u < MAXS; ++u)
        if (wave + NW * u < nsteps) step(wave + NW * u, wv[u]);
    for (int it = wave + NW * MAXS; it < nsteps; it += NW) step(it, wp[(size_t)it * 64]);      // (more taps than any GenVC config has)
    __syncthreads();                      // everyone is done reading Xs: the region becomes the reduction buffer
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int v = 0; v < 4; ++v) red[(wave * 4 * MT + 4 * t + v) * 64 + lane] = acc[t][v];
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave + NW * rr, m = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * 4 * MT + r) * 64 + lane];
        if constexpr (!SPLIT) {
            v += bn;
            if (A.resid) v += r1[rr];
            v = conv_lds_act(v, A.act);
        }
        if (t0 + m < A.T) {
            float* dst = A.y + ob + (size_t)jb * A.y_ps + (size_t)(t0 + m) * A.ldy;
            // (partial sums are handed to another workgroup: write-through stores and, below, cache-bypassing loads -- a
            // __threadfence() here writes back the whole L2 of the XCD, 50 us for this kernel)
            if constexpr (SPLIT) __hip_atomic_store(dst, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *dst = v;
        }
    }
    if constexpr (SPLIT) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // this workgroup's partial sums have landed before its arrival is counted
        __syncthreads();
        if (tid == 0) {
            int* cn = A.cnt + ((size_t)b * gridDim.x + blockIdx.x) * A.ntiles + tile;
            const int last = __hip_atomic_fetch_add(cn, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == A.split - 1;
            if (last) __hip_atomic_store(cn, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // every slice has arrived: ready for the next call
            reinterpret_cast<int*>(lds)[0] = last;          // (the reduction buffer was consumed before the barrier above)
        }
        __syncthreads();
        if (!reinterpret_cast<const int*>(lds)[0]) return;
#pragma unroll
        for (int rr = 0; rr < RPW; ++rr) {
            const int r = wave + NW * rr, m = 16 * (r >> 2) + 4 * (lane >> 4) + (r & 3);
            if (t0 + m >= A.T) continue;
            const float* pp = A.y + ob + (size_t)(t0 + m) * A.ldy;
            float v = 0.f;
            for (int p = 0; p < A.split; ++p) v += __hip_atomic_load(pp + (size_t)p * A.y_ps, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const size_t o = (size_t)b * A.yf_bs + A.yf_off + (size_t)(t0 + m) * A.ldy + n;
            v += bn;
            if (A.resid) v += A.resid[o];              // (SPLIT: the residual is indexed like yf and may BE yf: one thread reads and writes an element)
            A.yf[o] = conv_lds_act(v, A.act);
        }
    }
}

// conv weights [Co][Ci][k] (the reference's layout) -> per channel slice s (CS channels) the FM16 copy of [Co][k*CS] (column tap*CS + c):
// [Ci/CS][FM16].  CS = Ci: the one FM16 matrix of an unsplit conv.
static __global__ void k_conv_pack_slices(const float* w, float* out, int Co, int Ci, int k, int CS) {
    const int K = k * CS;
    const size_t per = (size_t)Co * K, n4 = per * (Ci / CS) / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int sl = (int)(i * 4 / per);
        const size_t r = i * 4 - (size_t)sl * per;
        const int n = (int)(r / K), kc = (int)(r % K), tap = kc / CS, cl = kc % CS;
        const float* src = w + ((size_t)n * Ci + sl * CS + cl) * k + tap;
        *reinterpret_cast<float4*>(out + (size_t)sl * per + fm16_index(n, kc, K)) = make_float4(src[0], src[k], src[2 * k], src[3 * k]);
    }
}

// K-split launch (k_conv_lds<CS, 8, 1, true>): A.split channel slices of CS = 64 or 256 channels, partial sums in A.y, result in A.yf
static inline int launch_conv_lds_split(int cs, const ConvLdsArgs& A, int B, size_t lds, hipStream_t s) {
    const dim3 grid(cdiv(A.T, 32), A.split * A.ntiles, B);
    if (cs == 64) hipLaunchKernelGGL((k_conv_lds<64, 8, 1, true>), grid, dim3(512), lds, s, A);
    else if (cs == 256) hipLaunchKernelGGL((k_conv_lds<256, 8, 1, true>), grid, dim3(512), lds, s, A);
    else return GVC_ERR_UNSUPPORTED;
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

// LDS bytes of k_conv_lds: the input rows, later the reduction buffer
static inline size_t conv_lds_bytes(int ci, int k, int dil, bool split = false, int stride = 1) {
    const int mt = conv_lds_mt(ci, split), R = (16 * mt - 1) * stride + (k - 1) * dil + 1, RH = (R + stride - 1) / stride;
    const size_t stage = (size_t)RH * stride * (ci + 4), red = (size_t)8 * 4 * mt * 64;
    return (stage > red ? stage : red) * sizeof(float);
}
constexpr size_t kConvLdsMax = 150 * 1024;
static inline bool conv_lds_ci_ok(int ci) { return ci == 32 || ci == 64 || ci == 128 || ci == 256 || ci == 512; }

#define GVC_CONV_LDS_FOR_EACH(X) X(32, 4) X(64, 8) X(128, 8) X(256, 8) X(512, 8)

// up to ~150 KB of dynamic LDS: raise the per-kernel limits once, outside any capture
static inline void conv_lds_init_attributes() {
#define GVC_CL_ATTR(CI, NW)                                                                                                                   \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_lds<CI, NW, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_lds<CI, NW, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    GVC_CONV_LDS_FOR_EACH(GVC_CL_ATTR)
#undef GVC_CL_ATTR
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_lds<64, 8, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_lds<256, 8, 1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipGetLastError();
}

// grid (frame tiles, jobs x column tiles, batch)
static inline int launch_conv_lds(int ci, int nsum, const ConvLdsArgs& A, int njobs, int B, size_t lds, hipStream_t s) {
    const dim3 grid(cdiv(A.T, 16 * conv_lds_mt(ci, false)), njobs * A.ntiles, B);
#define GVC_CL_LAUNCH(CI, NW)                                                                            \
    if (ci == CI) {                                                                                      \
        if (nsum == 1) hipLaunchKernelGGL((k_conv_lds<CI, NW, 1>), grid, dim3(NW * 64), lds, s, A);     \
        else hipLaunchKernelGGL((k_conv_lds<CI, NW, 3>), grid, dim3(NW * 64), lds, s, A);               \
    }
    GVC_CONV_LDS_FOR_EACH(GVC_CL_LAUNCH)
#undef GVC_CL_LAUNCH
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

}  // namespace gvc
