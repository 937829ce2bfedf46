// HiFi-GAN generator: reference layers/hifigan.py:160-243 (HiFiGAN.forward :218-233,
// ResBlock2 :119-157) with the config of configs/vocoder_configs.py:7-20, fed by the x4 linear interpolation
// of the GPT latents (inference/inference_utils.py:81-85, 196-202).
//
// Everything is a GEMM on the fp32 MFMA kernel (gemm.h) over time-major activations [T + 2*PAD][C]:
//   * Conv1d(k, dilation)      A row t = taps at rows t - pad + j*dil (implicit im2col, no copy), W -> [Cout][k*Cin];
//   * ConvTranspose1d(k, s)    ONE GEMM with N = s*Cout: row q of the output holds the s frames s*q .. s*q+s-1, which
//                              IS the time-major layout of the upsampled signal; W -> [s*Cout][ntap*Cin] (polyphase);
//   * leaky_relu on the conv inputs is applied while the A tile is staged; bias, the residual add, the running sum over
//     the three ResBlocks and the final 1/3 are fused in the epilogue.
// conv_post (Cout = 1) + tanh is a small dedicated kernel.  Weight-norm (weight_g, weight_v) is folded by the loader.
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "gemm.h"

namespace gvc {

constexpr int kHfPad = 40;      // >= the largest conv padding (kernel 7, dilation 12 -> 36), multiple of 4

// latents [B][n][d] -> x0 [B][n*scale + 2*PAD][d] rows PAD..; F.interpolate(scale_factor=scale, mode="linear")
__global__ void k_interp_linear(const float* lat, float* x0, int n, int d, int scale, int T0) {
    const int b = blockIdx.y, t = blockIdx.x;
    const float* src = lat + (size_t)b * n * d;
    float* dst = x0 + ((size_t)b * (T0 + 2 * kHfPad) + kHfPad + t) * d;
    float pos = ((float)t + 0.5f) / (float)scale - 0.5f;        // align_corners=False
    if (pos < 0.f) pos = 0.f;
    int i0 = (int)pos;
    if (i0 > n - 1) i0 = n - 1;
    const int i1 = i0 + 1 < n ? i0 + 1 : n - 1;
    const float w1 = pos - (float)i0, w0 = 1.0f - w1;
    for (int k = threadIdx.x * 4; k < d; k += blockDim.x * 4) {
        const float4 a = *reinterpret_cast<const float4*>(src + (size_t)i0 * d + k);
        const float4 c = *reinterpret_cast<const float4*>(src + (size_t)i1 * d + k);
        *reinterpret_cast<float4*>(dst + k) = make_float4(w0 * a.x + w1 * c.x, w0 * a.y + w1 * c.y, w0 * a.z + w1 * c.z,
                                                          w0 * a.w + w1 * c.w);
    }
}

// channels-first input [B][d][T] -> x0 time-major padded
__global__ void k_cf_to_time_major(const float* x, float* x0, int d, int T) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const float* src = x + (size_t)b * d * T;
    float* dst = x0 + (size_t)b * (T + 2 * kHfPad) * d;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r, t = t0 + threadIdx.x;
        if (c < d && t < T) tile[r][threadIdx.x] = src[(size_t)c * T + t];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int t = t0 + r, c = c0 + threadIdx.x;
        if (c < d && t < T) dst[(size_t)(t + kHfPad) * d + c] = tile[threadIdx.x][r];
    }
}

// Conv1d weight [Co][Ci][k] -> [Co][k*Ci]
__global__ void k_hf_repack_conv(const float* w, float* out, int Co, int Ci, int k) {
    const size_t n = (size_t)Co * Ci * k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % k);
        const int ci = (int)((i / k) % Ci);
        const int co = (int)(i / ((size_t)k * Ci));
        out[((size_t)co * k + j) * Ci + ci] = w[i];
    }
}

// ConvTranspose1d weight [Ci][Co][k] (stride s, padding pad) -> polyphase [s*Co][ntap*Ci]:
// out[s*q + p][co] = sum_{dpos, ci} x[q + dmin + dpos][ci] * W[ci][co][p + pad - s*(dmin + dpos)]
__global__ void k_hf_repack_convT(const float* w, float* out, int Ci, int Co, int k, int s, int pad, int dmin, int ntap) {
    const size_t n = (size_t)s * Co * ntap * Ci;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % Ci);
        const int dpos = (int)((i / Ci) % ntap);
        const int row = (int)(i / ((size_t)Ci * ntap));
        const int co = row % Co, p = row / Co;
        const int kk = p + pad - s * (dmin + dpos);
        out[i] = (kk >= 0 && kk < k) ? w[((size_t)ci * Co + co) * k + kk] : 0.f;
    }
}

__global__ void k_hf_tile_bias(const float* b, float* out, int Co, int s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < Co * s) out[i] = b[i % Co];
}

// wav[b][t] = tanh(bias + sum_{j<k, ci} lrelu(x[t - pad + j][ci], slope) * w[j*C + ci])      (conv_post, Cout = 1)
__global__ void k_conv_post_tanh(const float* x, const float* w, const float* bias, float* wav, int T, int C, int k,
                                 float slope) {
    extern __shared__ float ws[];
    for (int i = threadIdx.x; i < k * C; i += blockDim.x) ws[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const int pad = (k - 1) / 2;
    const float* xr = x + ((size_t)b * (T + 2 * kHfPad) + kHfPad + t - pad) * C;
    float acc = bias[0];
    for (int i = 0; i < k * C; i += 4) {
        const float4 v = *reinterpret_cast<const float4*>(xr + i);
        acc = fmaf(v.x > 0.f ? v.x : v.x * slope, ws[i], acc);
        acc = fmaf(v.y > 0.f ? v.y : v.y * slope, ws[i + 1], acc);
        acc = fmaf(v.z > 0.f ? v.z : v.z * slope, ws[i + 2], acc);
        acc = fmaf(v.w > 0.f ? v.w : v.w * slope, ws[i + 3], acc);
    }
    wav[(size_t)b * T + t] = tanhf(acc);
}

// ---------------------------------------------------------------------------------------------
// ResBlock convs (C = 32, 64 or 128 channels in and out, k <= 7 taps, dilation <= 12): the whole
// K = k*C extent is small, so instead of the tiled GEMM's k-loop (one global->LDS->sync round trip per 32 columns:
// 11-22 us for 0.1 GFLOP) a workgroup stages EVERYTHING it needs in one round trip -- the 32 + (k-1)*dil input rows of
// its 32 output frames (leaky-ReLU applied while staging) and the full [C][k*C] weight matrix -- and then runs the
// implicit im2col straight out of LDS.  A workgroup owns 32 frames x 32 output channels; its 4 (C = 32) or 8 (C = 64)
// waves split the (tap, 8-channel group) steps of the reduction and combine through LDS; bias / residual(s) / scale in
// the epilogue.  grid (T/32, C/32, B).
// ---------------------------------------------------------------------------------------------
struct ConvSmallArgs {
    const float* x; float* y;            // padded time-major [B][T + 2*kHfPad][C]
    const float* w; const float* b;      // [C][k*C] (column tap*C + ci), [C]
    const float* resid; const float* resid2;
    int T, k, dil;
    int tap_chunk;                       // taps of the weight tile staged in LDS at a time
    float slope, out_scale;
};

typedef float hf_f32x16 __attribute__((ext_vector_type(16)));
typedef float hf_f32x4 __attribute__((ext_vector_type(4)));

template <int C, int NW>
__global__ __launch_bounds__(NW * 64) void k_conv_small(const ConvSmallArgs A) {
    // grid (T/32, C/32, B): a workgroup owns 32 frames x one 32-channel tile of the outputs; NW waves split the reduction
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int XS = C + 4, NTH = NW * 64;
    const int K = A.k * C;
    const int tc = A.tap_chunk;                 // taps of W staged at a time (all of them when the tile fits in LDS)
    const int WS = tc * C + 4;
    const int R = 32 + (A.k - 1) * A.dil;
    float* Xs = lds;                       // [R][XS]
    float* Ws = lds + (size_t)R * XS;      // [32][WS]: taps [tap0, tap0 + tc) of the 32-channel weight tile
    float* red = lds;                      // [NW][16][64] after the MFMA loop (aliases Xs/Ws)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int t0 = blockIdx.x * 32, n0 = blockIdx.y * 32, b = blockIdx.z;
    const int pad = A.dil * (A.k - 1) / 2;
    const size_t bs = (size_t)(A.T + 2 * kHfPad) * C;
    const float* xb = A.x + b * bs + (size_t)(kHfPad + t0 - pad) * C;
    // staging loops: UL requests per thread go out before the first LDS store (with a run-time trip count and one float4 per
    // iteration the compiler emitted load / s_waitcnt vmcnt(0) / ds_write, a memory round trip per float4)
    constexpr int UL = 8;
    for (int i0 = tid; i0 < R * (C / 4); i0 += NTH * UL) {
        float4 xv[UL];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int i = min(i0 + u * NTH, R * (C / 4) - 1);
            xv[u] = *reinterpret_cast<const float4*>(xb + (size_t)(i / (C / 4)) * C + (i % (C / 4)) * 4);
        }
        __builtin_amdgcn_sched_barrier(0);         // (the scheduler would sink every request to just above its store)
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int i = i0 + u * NTH;
            if (i < R * (C / 4)) {
                float4 v = xv[u];
                v.x = v.x > 0.f ? v.x : v.x * A.slope; v.y = v.y > 0.f ? v.y : v.y * A.slope;
                v.z = v.z > 0.f ? v.z : v.z * A.slope; v.w = v.w > 0.f ? v.w : v.w * A.slope;
                *reinterpret_cast<float4*>(&Xs[(i / (C / 4)) * XS + (i % (C / 4)) * 4]) = v;
            }
        }
    }
    hf_f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const int m = lane & 31, half = lane >> 5;
    for (int tap0 = 0; tap0 < A.k; tap0 += tc) {
        const int nt = min(tc, A.k - tap0), kc4 = nt * C / 4;
        if (tap0 > 0) __syncthreads();             // the previous chunk of W has been consumed
        // wave w stages rows w, w + NW, ... of the weight tile; the 32 / NW row requests of a column step go out together
        constexpr int WR = 32 / NW;
        for (int kk = lane; kk < kc4; kk += 64) {
            hf_f32x4 wv[WR];
#pragma unroll
            for (int j = 0; j < WR; ++j)
                wv[j] = *reinterpret_cast<const hf_f32x4*>(A.w + (size_t)(n0 + wave + NW * j) * K + tap0 * C + kk * 4);
            __builtin_amdgcn_sched_barrier(0);         // (the scheduler would sink every request to just above its store)
#pragma unroll
            for (int j = 0; j < WR; ++j) *reinterpret_cast<hf_f32x4*>(&Ws[(wave + NW * j) * WS + kk * 4]) = wv[j];
        }
        __syncthreads();
        const int steps = nt * (C / 8);
        for (int it = wave; it < steps; it += NW) {
            const int tl = it / (C / 8), q = it - tl * (C / 8);
            const float4 a4 = *reinterpret_cast<const float4*>(&Xs[(m + (tap0 + tl) * A.dil) * XS + 8 * q + 4 * half]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Ws[m * WS + tl * C + 8 * q + 4 * half]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
    }
    __syncthreads();                      // everyone is done reading Xs / Ws: the region becomes the reduction buffer
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    // thread (wave, lane) finishes accumulator registers wave*16/NW .. of the tile
    constexpr int RPW = 16 / NW;
    const size_t ob = b * bs + (size_t)(kHfPad + t0) * C;
    // bias and residuals of this thread's outputs are requested together, ahead of the LDS sums
    const int n = n0 + (lane & 31);
    const float bn = A.b[n];
    float r1[RPW], r2[RPW];
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        const size_t o = ob + (size_t)((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * C + n;
        r1[rr] = A.resid ? A.resid[o] : 0.f;
        r2[rr] = A.resid2 ? A.resid2[o] : 0.f;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rr = 0; rr < RPW; ++rr) {
        const int r = wave * RPW + rr;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * 16 + r) * 64 + lane];
        const int mo = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const size_t o = ob + (size_t)mo * C + n;
        v += bn;
        if (A.resid) v += r1[rr];
        if (A.resid2) v += r2[rr];
        if (A.out_scale != 0.f) v *= A.out_scale;
        A.y[o] = v;
    }
}

}  // namespace gvc

using namespace gvc;

struct HfConv { float *w = nullptr, *b = nullptr; int Co = 0, Ci = 0, k = 0, dil = 1; };
struct HfUp { float *w = nullptr, *b = nullptr, *braw = nullptr; int Ci = 0, Co = 0, k = 0, s = 0, pad = 0, dmin = 0, ntap = 0; };

struct gvc_hifigan {
    gvc_hifigan_dims dm;
    HfConv pre, post;
    std::vector<HfUp> ups;
    std::vector<HfConv> res;                 // [stage][kernel][2]
    std::map<std::string, int> bound;
    int n_expected = 0;
    float* x0 = nullptr;                     // interpolated input
    float* x1 = nullptr;                     // conv_pre output
    std::vector<float*> U, R, S0, S1;        // per stage
    float* work = nullptr;
    long long work_cap = 0;
    std::vector<void*> allocs;
    int cur_T0 = -1, cur_B = -1;
    // the GEMM chain between the input staging and conv_post only touches context buffers: it is captured once per
    // (B, frames) and replayed (about 30 launches of a few microseconds each are host-bound when launched eagerly)
    std::map<long long, hipGraphExec_t> graphs;
    hipStream_t cap_stream = nullptr;
    int use_graph = 1;
    int small_conv = 1;                      // GVC_VOCODER_SMALL_CONV=0: ResBlock convs of the last stages through the tiled GEMM
};

static int halloc(gvc_hifigan* c, float** p, size_t n) {
    GVC_CHECK_HIP(hipMalloc((void**)p, n * sizeof(float)));
    GVC_CHECK_HIP(hipMemset(*p, 0, n * sizeof(float)));
    c->allocs.push_back(*p);
    return GVC_OK;
}

extern "C" int gvc_hifigan_create(const gvc_hifigan_dims* dims, gvc_hifigan** out) {
    GVC_REQUIRE(dims && out, GVC_ERR_ARG, "gvc_hifigan_create: null argument");
    const gvc_hifigan_dims& D = *dims;
    GVC_REQUIRE(D.n_ups >= 1 && D.n_ups <= 4 && D.n_kernels >= 1 && D.n_kernels <= 4 && D.in_dim % 4 == 0, GVC_ERR_ARG,
                "hifigan: bad dims");
    auto* c = new gvc_hifigan();
    c->dm = D;
    int rc = GVC_OK;
    auto mkconv = [&](HfConv& w, int Co, int Ci, int k, int dil) {
        w.Co = Co; w.Ci = Ci; w.k = k; w.dil = dil;
        int r = halloc(c, &w.w, (size_t)Co * Ci * k);
        return r ? r : halloc(c, &w.b, Co);
    };
    rc = mkconv(c->pre, D.up_init_ch, D.in_dim, 7, 1);
    int ch = D.up_init_ch;
    size_t T = D.max_frames;
    const size_t B = D.max_batch;
    if (!rc) rc = halloc(c, &c->x0, B * (T + 2 * kHfPad) * D.in_dim);
    if (!rc) rc = halloc(c, &c->x1, B * (T + 2 * kHfPad) * ch);
    for (int i = 0; i < D.n_ups && !rc; ++i) {
        HfUp u;
        u.Ci = ch; u.Co = ch / 2; u.k = D.up_kernels[i]; u.s = D.up_rates[i]; u.pad = (u.k - u.s) / 2;
        GVC_REQUIRE(u.Co % 4 == 0 && (u.k - u.s) % 2 == 0, GVC_ERR_UNSUPPORTED, "hifigan: unsupported upsample layer");
        // input offsets d with some phase p in [0,s) such that 0 <= p + pad - s*d < k
        int dmin = 1 << 20, dmax = -(1 << 20);
        for (int d = -8; d <= 8; ++d)
            for (int p = 0; p < u.s; ++p) {
                const int kk = p + u.pad - u.s * d;
                if (kk >= 0 && kk < u.k) { if (d < dmin) dmin = d; if (d > dmax) dmax = d; }
            }
        u.dmin = dmin; u.ntap = dmax - dmin + 1;
        if ((rc = halloc(c, &u.w, (size_t)u.s * u.Co * u.ntap * u.Ci)) || (rc = halloc(c, &u.b, (size_t)u.s * u.Co)) ||
            (rc = halloc(c, &u.braw, u.Co)))
            break;
        c->ups.push_back(u);
        ch = u.Co;
        T *= u.s;
        for (int j = 0; j < D.n_kernels && !rc; ++j)
            for (int q = 0; q < 2 && !rc; ++q) {
                HfConv w;
                rc = mkconv(w, ch, ch, D.res_kernels[j], D.res_dilations[j][q]);
                GVC_REQUIRE(rc || w.dil * (w.k - 1) / 2 <= kHfPad, GVC_ERR_UNSUPPORTED, "hifigan: conv padding exceeds %d", kHfPad);
                c->res.push_back(w);
            }
        float *u_, *r_, *s0, *s1;
        const size_t n = B * (T + 2 * kHfPad) * ch;
        if (!rc && !(rc = halloc(c, &u_, n)) && !(rc = halloc(c, &r_, n)) && !(rc = halloc(c, &s0, n)) && !(rc = halloc(c, &s1, n))) {
            c->U.push_back(u_); c->R.push_back(r_); c->S0.push_back(s0); c->S1.push_back(s1);
        }
    }
    if (!rc) rc = mkconv(c->post, 1, ch, 7, 1);
    c->work_cap = 4ll << 20;
    if (!rc) rc = halloc(c, &c->work, (size_t)c->work_cap);
    c->n_expected = 2 * (2 + D.n_ups + 2 * D.n_ups * D.n_kernels);
    if (rc) { gvc_hifigan_destroy(c); return rc; }
    if (getenv("GVC_VOCODER_GRAPH")) c->use_graph = atoi(getenv("GVC_VOCODER_GRAPH"));
    if (getenv("GVC_VOCODER_SMALL_CONV")) c->small_conv = atoi(getenv("GVC_VOCODER_SMALL_CONV"));
    // up to ~150 KB of dynamic LDS (64 channels, 7 taps): raise the per-kernel limit once, outside any capture
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_small<32, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_small<64, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_small<128, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    GVC_CHECK_HIP(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    *out = c;
    return GVC_OK;
}

extern "C" int gvc_hifigan_destroy(gvc_hifigan* c) {
    if (!c) return GVC_OK;
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    if (c->cap_stream) hipStreamDestroy(c->cap_stream);
    for (void* p : c->allocs) hipFree(p);
    delete c;
    return GVC_OK;
}

static int hf_bind_conv(HfConv& w, bool is_bias, const float* src, int64_t numel, const char* name, hipStream_t s) {
    if (is_bias) {
        GVC_REQUIRE(numel == w.Co, GVC_ERR_ARG, "%s: expected %d elements, got %lld", name, w.Co, (long long)numel);
        GVC_CHECK_HIP(hipMemcpyAsync(w.b, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
        return GVC_OK;
    }
    GVC_REQUIRE(numel == (int64_t)w.Co * w.Ci * w.k, GVC_ERR_ARG, "%s: expected %lld elements, got %lld", name,
                (long long)w.Co * w.Ci * w.k, (long long)numel);
    hipLaunchKernelGGL(k_hf_repack_conv, dim3(512), dim3(256), 0, s, src, w.w, w.Co, w.Ci, w.k);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

// names as in the reference state dict AFTER weight-norm folding: "conv_pre.weight", "ups.0.bias",
// "resblocks.4.convs.1.weight", "conv_post.weight", ...
extern "C" int gvc_hifigan_bind_weight(gvc_hifigan* c, const char* name, const float* src, int64_t numel, gvc_stream sv) {
    GVC_REQUIRE(c && name && src, GVC_ERR_ARG, "gvc_hifigan_bind_weight: null argument");
    hipStream_t s = (hipStream_t)sv;
    std::string n(name);
    const bool is_bias = n.size() >= 5 && n.compare(n.size() - 5, 5, ".bias") == 0;
    const bool is_w = n.size() >= 7 && n.compare(n.size() - 7, 7, ".weight") == 0;
    int rc = GVC_OK;
    bool known = is_bias || is_w;
    if (!known) return GVC_OK;
    if (n.rfind("conv_pre.", 0) == 0) rc = hf_bind_conv(c->pre, is_bias, src, numel, name, s);
    else if (n.rfind("conv_post.", 0) == 0) rc = hf_bind_conv(c->post, is_bias, src, numel, name, s);
    else if (n.rfind("ups.", 0) == 0) {
        const int i = atoi(n.c_str() + 4);
        GVC_REQUIRE(i >= 0 && i < (int)c->ups.size(), GVC_ERR_ARG, "%s: layer out of range", name);
        HfUp& u = c->ups[i];
        if (is_bias) {
            GVC_REQUIRE(numel == u.Co, GVC_ERR_ARG, "%s: wrong size", name);
            GVC_CHECK_HIP(hipMemcpyAsync(u.braw, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
            hipLaunchKernelGGL(k_hf_tile_bias, dim3(cdiv(u.Co * u.s, 256)), dim3(256), 0, s, u.braw, u.b, u.Co, u.s);
            GVC_LAUNCH_CHECK();
        } else {
            GVC_REQUIRE(numel == (int64_t)u.Ci * u.Co * u.k, GVC_ERR_ARG, "%s: wrong size", name);
            hipLaunchKernelGGL(k_hf_repack_convT, dim3(512), dim3(256), 0, s, src, u.w, u.Ci, u.Co, u.k, u.s, u.pad, u.dmin,
                               u.ntap);
            GVC_LAUNCH_CHECK();
        }
    } else if (n.rfind("resblocks.", 0) == 0) {
        const int bi = atoi(n.c_str() + 10);
        const size_t pos = n.find(".convs.");
        GVC_REQUIRE(pos != std::string::npos, GVC_ERR_ARG, "malformed weight name %s", name);
        const int q = atoi(n.c_str() + pos + 7);
        const int idx = bi * 2 + q;
        GVC_REQUIRE(bi >= 0 && q >= 0 && q < 2 && idx < (int)c->res.size(), GVC_ERR_ARG, "%s: out of range", name);
        rc = hf_bind_conv(c->res[idx], is_bias, src, numel, name, s);
    } else {
        known = false;
    }
    if (rc == GVC_OK && known) c->bound[n] = 1;
    return rc;
}

extern "C" int gvc_hifigan_missing_weights(gvc_hifigan* c) { return c ? c->n_expected - (int)c->bound.size() : -1; }

// out rows [PAD, PAD+T) = epilogue(conv(lrelu?(src)))
// LDS bytes of k_conv_small with `tc` taps of the weight tile resident
static size_t conv_small_lds(const HfConv& w, int tc) {
    const size_t R = 32 + (size_t)(w.k - 1) * w.dil;
    const size_t stage = R * (w.Ci + 4) + (size_t)32 * ((size_t)tc * w.Ci + 4), red = (size_t)8 * 16 * 64;
    return (stage > red ? stage : red) * sizeof(float);
}

static int hf_conv(gvc_hifigan* c, const HfConv& w, const float* src, float* dst, int T, int B, float a_slope,
                   const float* resid, const float* resid2, float out_scale, hipStream_t s) {
    if (c->small_conv && w.Ci == w.Co && (w.Ci == 32 || w.Ci == 64 || w.Ci == 128) && T % 32 == 0 && a_slope != 0.f) {
        int tc = w.k;                                    // as many taps of W at a time as fit beside the input rows
        while (tc > 1 && conv_small_lds(w, tc) > 150 * 1024) --tc;
        if (conv_small_lds(w, tc) <= 150 * 1024) {
            ConvSmallArgs A;
            A.x = src; A.y = dst; A.w = w.w; A.b = w.b; A.resid = resid; A.resid2 = resid2; A.T = T; A.k = w.k; A.dil = w.dil;
            A.tap_chunk = tc; A.slope = a_slope; A.out_scale = out_scale;
            const size_t lds = conv_small_lds(w, tc);
            if (w.Ci == 32) hipLaunchKernelGGL((k_conv_small<32, 4>), dim3(T / 32, 1, B), dim3(256), lds, s, A);
            else if (w.Ci == 64) hipLaunchKernelGGL((k_conv_small<64, 8>), dim3(T / 32, 2, B), dim3(512), lds, s, A);
            else hipLaunchKernelGGL((k_conv_small<128, 8>), dim3(T / 32, 4, B), dim3(512), lds, s, A);
            GVC_LAUNCH_CHECK();
            return GVC_OK;
        }
    }
    GemmArgs G;
    memset(&G, 0, sizeof(G));
    const int pad = w.dil * (w.k - 1) / 2;
    const long long bs_in = (long long)(T + 2 * kHfPad) * w.Ci, bs_out = (long long)(T + 2 * kHfPad) * w.Co;
    G.A = src + (size_t)(kHfPad - pad) * w.Ci; G.lda = w.Ci; G.a_batch_stride = bs_in;
    G.conv_cin = w.Ci; G.conv_tap_stride = w.dil * w.Ci;
    G.a_act = a_slope != 0.f ? AACT_LRELU : AACT_NONE; G.a_slope = a_slope;
    G.Wt = w.w; G.ldw = w.k * w.Ci;
    G.C = dst + (size_t)kHfPad * w.Co; G.ldc = w.Co; G.c_batch_stride = bs_out;
    G.M = T; G.N = w.Co; G.K = w.k * w.Ci; G.work = c->work;
    G.e.bias = w.b;
    if (resid) { G.e.resid = resid + (size_t)kHfPad * w.Co; G.e.ldr = w.Co; G.e.resid_batch_stride = bs_out; }
    if (resid2) G.e.resid2 = resid2 + (size_t)kHfPad * w.Co;
    G.e.out_scale = out_scale;
    return launch_gemm_cap(G, B, c->work_cap, s);
}

// conv_pre .. last ResBlock sum; returns the final activation buffer and its length
static int hf_body(gvc_hifigan* c, int B, int T0, hipStream_t s, const float** x_out, int* T_out) {
    int rc;
    const gvc_hifigan_dims& D = c->dm;
    if ((rc = hf_conv(c, c->pre, c->x0, c->x1, T0, B, 0.f, nullptr, nullptr, 0.f, s))) return rc;
    const float* x = c->x1;
    int T = T0;
    for (int i = 0; i < D.n_ups; ++i) {
        const HfUp& u = c->ups[i];
        // upsample: M = T input frames, N = s*Co, window of ntap input rows starting at q + dmin
        GemmArgs G;
        memset(&G, 0, sizeof(G));
        G.A = x + (size_t)(kHfPad + u.dmin) * u.Ci; G.lda = u.Ci; G.a_batch_stride = (long long)(T + 2 * kHfPad) * u.Ci;
        G.a_act = AACT_LRELU; G.a_slope = 0.1f;
        G.Wt = u.w; G.ldw = u.ntap * u.Ci;
        const int To = T * u.s;
        G.C = c->U[i] + (size_t)kHfPad * u.Co; G.ldc = u.s * u.Co; G.c_batch_stride = (long long)(To + 2 * kHfPad) * u.Co;
        G.M = T; G.N = u.s * u.Co; G.K = u.ntap * u.Ci; G.work = c->work; G.e.bias = u.b;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
        T = To;
        float* acc[2] = {c->S0[i], c->S1[i]};
        const float* prev = nullptr;
        for (int j = 0; j < D.n_kernels; ++j) {
            const HfConv& ca = c->res[(i * D.n_kernels + j) * 2], &cb = c->res[(i * D.n_kernels + j) * 2 + 1];
            // ResBlock2: r = U + conv_a(lrelu(U)); out_j = r + conv_b(lrelu(r)); running sum over j, /n_kernels at the end
            if ((rc = hf_conv(c, ca, c->U[i], c->R[i], T, B, 0.1f, c->U[i], nullptr, 0.f, s))) return rc;
            float* dst = acc[j & 1];
            const bool last = j == D.n_kernels - 1;
            if ((rc = hf_conv(c, cb, c->R[i], dst, T, B, 0.1f, c->R[i], prev, last ? 1.0f / (float)D.n_kernels : 0.f, s)))
                return rc;
            prev = dst;
        }
        x = prev;
    }
    *x_out = x;
    *T_out = T;
    return GVC_OK;
}

static int hf_run(gvc_hifigan* c, int B, int T0, float* wav, hipStream_t s) {
    int rc;
    const float* x = nullptr;
    int T = 0;
    if (!c->use_graph) {
        if ((rc = hf_body(c, B, T0, s, &x, &T))) return rc;
    } else {
        const long long key = ((long long)B << 32) | (unsigned)T0;
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            GVC_CHECK_HIP(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
            rc = hf_body(c, B, T0, c->cap_stream, &x, &T);
            hipGraph_t graph = nullptr;
            hipError_t e = hipStreamEndCapture(c->cap_stream, &graph);
            if (rc) { if (graph) hipGraphDestroy(graph); return rc; }
            GVC_CHECK_HIP(e);
            hipGraphExec_t ge;
            e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
            hipGraphDestroy(graph);
            GVC_CHECK_HIP(e);
            it = c->graphs.emplace(key, ge).first;
        }
        GVC_CHECK_HIP(hipGraphLaunch(it->second, s));
        // geometry of the final activation (what hf_body would have returned)
        T = T0;
        for (const HfUp& u : c->ups) T *= u.s;
        const int nk = c->dm.n_kernels, last = (int)c->ups.size() - 1;
        x = ((nk - 1) & 1) ? c->S1[last] : c->S0[last];
    }
    const HfConv& p = c->post;
    hipLaunchKernelGGL(k_conv_post_tanh, dim3(cdiv(T, 256), B), dim3(256), p.k * p.Ci * sizeof(float), s, x, p.w, p.b, wav, T,
                       p.Ci, p.k, 0.01f);     // F.leaky_relu default slope before conv_post (hifigan.py:230)
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

static int hf_prepare(gvc_hifigan* c, int B, int T0, hipStream_t s) {
    GVC_REQUIRE(gvc_hifigan_missing_weights(c) == 0, GVC_ERR_STATE, "%d HiFi-GAN weight tensors are not bound",
                gvc_hifigan_missing_weights(c));
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_batch && T0 >= 1 && T0 <= c->dm.max_frames, GVC_ERR_ARG,
                "hifigan: B=%d frames=%d outside capacity (%d, %d)", B, T0, c->dm.max_batch, c->dm.max_frames);
    if (T0 != c->cur_T0 || B != c->cur_B) {
        // the zero rows around the live region depend on the geometry: re-zero everything when it changes
        size_t T = T0, ch = c->dm.up_init_ch;
        GVC_CHECK_HIP(hipMemsetAsync(c->x0, 0, (size_t)B * (T + 2 * kHfPad) * c->dm.in_dim * sizeof(float), s));
        GVC_CHECK_HIP(hipMemsetAsync(c->x1, 0, (size_t)B * (T + 2 * kHfPad) * ch * sizeof(float), s));
        for (size_t i = 0; i < c->ups.size(); ++i) {
            T *= c->ups[i].s; ch = c->ups[i].Co;
            const size_t bytes = (size_t)B * (T + 2 * kHfPad) * ch * sizeof(float);
            for (float* p : {c->U[i], c->R[i], c->S0[i], c->S1[i]}) GVC_CHECK_HIP(hipMemsetAsync(p, 0, bytes, s));
        }
        c->cur_T0 = T0; c->cur_B = B;
    }
    return GVC_OK;
}

extern "C" int gvc_hifigan_forward_latents(gvc_hifigan* c, const float* latents, int32_t B, int32_t n, int32_t scale,
                                           float* wav, gvc_stream sv) {
    GVC_REQUIRE(c && latents && wav && n >= 1 && scale >= 1, GVC_ERR_ARG, "gvc_hifigan_forward_latents: bad argument");
    hipStream_t s = (hipStream_t)sv;
    const int T0 = n * scale;
    int rc = hf_prepare(c, B, T0, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_interp_linear, dim3(T0, B), dim3(256), 0, s, latents, c->x0, n, c->dm.in_dim, scale, T0);
    GVC_LAUNCH_CHECK();
    return hf_run(c, B, T0, wav, s);
}

extern "C" int gvc_hifigan_forward(gvc_hifigan* c, const float* x, int32_t B, int32_t T, float* wav, gvc_stream sv) {
    GVC_REQUIRE(c && x && wav, GVC_ERR_ARG, "gvc_hifigan_forward: bad argument");
    hipStream_t s = (hipStream_t)sv;
    int rc = hf_prepare(c, B, T, s);
    if (rc) return rc;
    hipLaunchKernelGGL(k_cf_to_time_major, dim3(cdiv(T, 32), cdiv(c->dm.in_dim, 32), B), dim3(32, 8), 0, s, x, c->x0,
                       c->dm.in_dim, T);
    GVC_LAUNCH_CHECK();
    return hf_run(c, B, T, wav, s);
}
