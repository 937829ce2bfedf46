// HiFi-GAN generator: reference layers/hifigan.py:160-243 (HiFiGAN.forward :218-233,
// ResBlock2 :119-157) with the config of configs/vocoder_configs.py:7-20, fed by the x4 linear interpolation
// of the GPT latents (inference/inference_utils.py:81-85, 196-202).
//
// Every layer is a matrix product over time-major activations [T + 2*PAD][C] with zero rows either side:
//   * Conv1d(k, dilation)      A row t = taps at rows t - pad + j*dil (implicit im2col, no copy), W -> [Cout][k*Cin];
//   * ConvTranspose1d(k, s)    ONE product with N = s*Cout: row q of the output holds the s frames s*q .. s*q+s-1, which
//                              IS the time-major layout of the upsampled signal; W -> [s*Cout][ntap*Cin] (polyphase);
//   * leaky_relu on the conv inputs is applied while the input tile is staged; bias and residual in the epilogue.
// The streaming call (8 tokens -> 32 frames -> 8192 samples: 1.6 GFLOP, 14 MB of weights) is pure latency, so the default path
// is 12 launches of k_conv_lds (below), each ONE memory round trip deep, captured with the input staging and conv_post in one
// graph: conv_pre (K-split over 64-channel slices, combined by the last workgroup to arrive), per stage the upsampling layer,
// the first convs of the three ResBlocks in one launch and the second convs in another; the consumer of a stage adds the three
// ResBlock planes and the 1/3 while it stages its input.  184 -> 85 us per call (profiles/r03_microbench_notes.md section 9).
// Shapes k_conv_lds does not take (and GVC_VOCODER_SMALL_CONV=0) run on the tiled fp32 MFMA GEMM of gemm.h, one launch per conv.
// conv_post (Cout = 1) + tanh is a small dedicated kernel.  Weight-norm (weight_g, weight_v) is folded by the loader.
#include <stdlib.h>

#include <algorithm>
#include <map>
#include <string>
#include <vector>

#include "conv_lds.h"

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

// wav[b][t] = tanh(bias + sum_{j<k, ci} lrelu(xin[t - pad + j][ci], slope) * w[j*C + ci]),  xin = x_scale * ((p0 + p1) + p2)
// (conv_post, Cout = 1, fed by the three ResBlock planes of the last stage).  A workgroup owns 64 samples: their 64 + k - 1
// input rows go to LDS once; four lanes share a sample (a quarter of the channels each) and combine with two shuffles.
template <int NSUM>
__global__ __launch_bounds__(256) void k_conv_post_tanh(const float* x, long long x_ss, float x_scale, const float* w, const float* bias,
                                                        float* wav, int T, int C, int k, float slope) {
    extern __shared__ float ws[];                  // [k*C] weights, then [64 + k - 1][C + 1] input rows
    float* Xs = ws + k * C;
    const int tid = threadIdx.x, b = blockIdx.y, t0 = blockIdx.x * 64;
    const int pad = (k - 1) / 2, R = 64 + k - 1, C4 = C / 4, XS = C + 1;
    const int last_row = T + 2 * kHfPad - 1;
    const float* xb = x + (size_t)b * (T + 2 * kHfPad) * C;
    constexpr int UL = 3;
    for (int i0 = tid; i0 < R * C4; i0 += 256 * UL) {
        float4 xv[UL][NSUM];
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int i = min(i0 + u * 256, R * C4 - 1);
            const float* src = xb + (size_t)min(kHfPad + t0 - pad + i / C4, last_row) * C + (i % C4) * 4;
#pragma unroll
            for (int p = 0; p < NSUM; ++p) xv[u][p] = *reinterpret_cast<const float4*>(src + (size_t)p * x_ss);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UL; ++u) {
            const int i = i0 + u * 256;
            if (i < R * C4) {
                float4 v = xv[u][0];
                if constexpr (NSUM > 1) {
#pragma unroll
                    for (int p = 1; p < NSUM; ++p) { v.x += xv[u][p].x; v.y += xv[u][p].y; v.z += xv[u][p].z; v.w += xv[u][p].w; }
                    v.x *= x_scale; v.y *= x_scale; v.z *= x_scale; v.w *= x_scale;
                }
                float* d = &Xs[(i / C4) * XS + (i % C4) * 4];
                d[0] = v.x > 0.f ? v.x : v.x * slope; d[1] = v.y > 0.f ? v.y : v.y * slope;
                d[2] = v.z > 0.f ? v.z : v.z * slope; d[3] = v.w > 0.f ? v.w : v.w * slope;
            }
        }
    }
    for (int i = tid; i < k * C; i += 256) ws[i] = w[i];
    __syncthreads();
    const int t = tid >> 2, q = tid & 3, cq = C / 4;
    float acc = 0.f;
    for (int j = 0; j < k; ++j) {
        const float* xr = &Xs[(t + j) * XS + q * cq];
        const float* wr = &ws[j * C + q * cq];
        for (int c = 0; c < cq; ++c) acc = fmaf(xr[c], wr[c], acc);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    if (q == 0 && t0 + t < T) wav[(size_t)b * T + t0 + t] = tanhf(acc + bias[0]);
}

// p0 = scale * ((p0 + p1) + p2): the stage output for a consumer that cannot add the ResBlock planes itself (tiled GEMM)
__global__ void k_sum_planes(float* p, long long ps, float scale, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 a = reinterpret_cast<float4*>(p)[i];
        const float4 b = reinterpret_cast<const float4*>(p + ps)[i], c = reinterpret_cast<const float4*>(p + 2 * ps)[i];
        a.x = ((a.x + b.x) + c.x) * scale; a.y = ((a.y + b.y) + c.y) * scale;
        a.z = ((a.z + b.z) + c.z) * scale; a.w = ((a.w + b.w) + c.w) * scale;
        reinterpret_cast<float4*>(p)[i] = a;
    }
}

// [N][K] row-major -> FM16 (gemm.h)
__global__ void k_hf_to_fm16(const float* src, float* dst, int N, int K) {
    const size_t n4 = (size_t)N * K / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / (K / 4)), k = (int)(i % (K / 4)) * 4;
        *reinterpret_cast<float4*>(dst + fm16_index(n, k, K)) = *reinterpret_cast<const float4*>(src + (size_t)n * K + k);
    }
}

}  // namespace gvc

using namespace gvc;

struct HfConv { float *w = nullptr, *b = nullptr, *wp = nullptr; int Co = 0, Ci = 0, k = 0, dil = 1; };     // wp: FM16 copy for k_conv_lds, null: shape not eligible
struct HfUp { float *w = nullptr, *b = nullptr, *braw = nullptr, *wp = nullptr; int Ci = 0, Co = 0, k = 0, s = 0, pad = 0, dmin = 0, ntap = 0; };
// a stage input: one buffer, or the nsum ResBlock planes (ss floats apart) whose scaled sum it is
struct HfIn { const float* x = nullptr; int nsum = 1; long long ss = 0; float scale = 1.f; };
constexpr int kHfRing = 4;
struct HfPlan {
    hipGraph_t graph = nullptr;
    // A ring of executable graphs of the same captured graph.  The caller's two pointers are patched into an executable graph, and a
    // patch (or a destroy) must not touch one whose previous launch may still be in flight on the stream (streamed chunks are
    // enqueued back to back without a host sync): each slot remembers the pointers it holds and an event recorded behind its last
    // launch; a slot is only patched after that event has completed (normally long ago: the slot is the least recently launched).
    hipGraphExec_t ge[kHfRing] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev[kHfRing] = {nullptr, nullptr, nullptr, nullptr};
    const float* in[kHfRing] = {nullptr, nullptr, nullptr, nullptr};
    float* wav[kHfRing] = {nullptr, nullptr, nullptr, nullptr};
    unsigned long long used[kHfRing] = {0, 0, 0, 0};
    bool flying[kHfRing] = {false, false, false, false};
    HfIn out; int T = 0;                       // what conv_post reads
    // whole-call graphs: the nodes that carry the caller's pointers
    hipGraphNode_t head = nullptr, post = nullptr;
    dim3 head_grid, head_block, post_grid, post_block; unsigned post_lds = 0;
};

// executable graphs may have launches in flight: wait for them, then destroy
static void hf_destroy_plan(HfPlan& pl) {
    for (int i = 0; i < kHfRing; ++i) {
        if (pl.ev[i]) { if (pl.flying[i]) (void)hipEventSynchronize(pl.ev[i]); (void)hipEventDestroy(pl.ev[i]); }
        if (pl.ge[i]) (void)hipGraphExecDestroy(pl.ge[i]);
    }
    if (pl.graph) (void)hipGraphDestroy(pl.graph);
    (void)hipGetLastError();
}

struct gvc_hifigan {
    gvc_hifigan_dims dm;
    HfConv pre, post;
    std::vector<HfUp> ups;
    std::vector<HfConv> res;                 // [stage][kernel][2]
    std::map<std::string, int> bound;
    int n_expected = 0;
    float* x0 = nullptr;                     // interpolated input
    float* x1 = nullptr;                     // conv_pre output
    // per stage: U = upsampled signal; R, S = n_kernels planes each (outputs of the first ResBlock convs / of the ResBlocks; the
    // launch-per-conv path keeps its running sum in planes 0 and 1 of S)
    std::vector<float*> U, R, S;
    std::vector<long long> plane;            // floats per plane
    float* work = nullptr;
    long long work_cap = 0;
    float* pre_wp = nullptr;                 // conv_pre weights as 64-channel slices in FM16 (k_conv_lds<.., SPLIT>), null: shape not eligible
    int* cnt = nullptr;                      // arrival counters of the split conv_pre, zero between calls
    std::vector<void*> allocs;
    int cur_T0 = -1, cur_B = -1;
    // the chain between the input staging and conv_post only touches context buffers: it is captured once per
    // (B, frames) and replayed (a dozen launches of a few microseconds each are host-bound when launched eagerly)
    std::map<long long, HfPlan> graphs;
    unsigned long long graph_tick = 0;       // launch counter (least-recently-launched choice in a plan's ring)
    hipStream_t cap_stream = nullptr;
    int use_graph = 1;
    // GVC_VOCODER_SMALL_CONV=0: every conv through the tiled GEMM, one launch per conv; 2: only the ResBlocks on k_conv_lds (conv_pre and
    // the upsampling layers through the tiled GEMM, which reads the sum of the ResBlock planes from k_sum_planes)
    int small_conv = 1;
};

static int halloc(gvc_hifigan* c, float** p, size_t n) {
    GVC_CHECK_HIP(hipMalloc((void**)p, n * sizeof(float)));
    GVC_CHECK_HIP(hipMemset(*p, 0, n * sizeof(float)));
    c->allocs.push_back(*p);
    return GVC_OK;
}

constexpr int kPreSlice = 64, kPreCounters = 4096;

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
        if (!r && Co == Ci && conv_lds_ci_ok(Ci) && conv_lds_bytes(Ci, k, dil) <= kConvLdsMax) r = halloc(c, &w.wp, (size_t)Co * Ci * k);
        return r ? r : halloc(c, &w.b, Co);
    };
    rc = mkconv(c->pre, D.up_init_ch, D.in_dim, 7, 1);
    if (!rc && D.in_dim % kPreSlice == 0 && D.up_init_ch % 16 == 0) {
        rc = halloc(c, &c->pre_wp, (size_t)D.up_init_ch * D.in_dim * 7);
        if (!rc) rc = halloc(c, reinterpret_cast<float**>(&c->cnt), kPreCounters);
    }
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
        const size_t nw = (size_t)u.s * u.Co * u.ntap * u.Ci;
        if ((rc = halloc(c, &u.w, nw)) || (rc = halloc(c, &u.b, (size_t)u.s * u.Co)) || (rc = halloc(c, &u.braw, u.Co))) break;
        if (conv_lds_ci_ok(u.Ci) && (u.s * u.Co) % 16 == 0 && conv_lds_bytes(u.Ci, u.ntap, 1) <= kConvLdsMax && (rc = halloc(c, &u.wp, nw))) break;
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
        float *u_, *r_, *s_;
        const size_t n = B * (T + 2 * kHfPad) * ch, np = n * (D.n_kernels > 2 ? D.n_kernels : 2);
        if (!rc && !(rc = halloc(c, &u_, n)) && !(rc = halloc(c, &r_, np)) && !(rc = halloc(c, &s_, np))) {
            c->U.push_back(u_); c->R.push_back(r_); c->S.push_back(s_); c->plane.push_back((long long)n);
        }
    }
    if (!rc) rc = mkconv(c->post, 1, ch, 7, 1);
    c->work_cap = 4ll << 20;
    if (!rc) rc = halloc(c, &c->work, (size_t)c->work_cap);
    c->n_expected = 2 * (2 + D.n_ups + 2 * D.n_ups * D.n_kernels);
    if (rc) { gvc_hifigan_destroy(c); return rc; }
    if (getenv("GVC_VOCODER_GRAPH")) c->use_graph = atoi(getenv("GVC_VOCODER_GRAPH"));
    if (getenv("GVC_VOCODER_SMALL_CONV")) c->small_conv = atoi(getenv("GVC_VOCODER_SMALL_CONV"));
    conv_lds_init_attributes();
    GVC_CHECK_HIP(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    *out = c;
    return GVC_OK;
}

extern "C" int gvc_hifigan_destroy(gvc_hifigan* c) {
    if (!c) return GVC_OK;
    for (auto& kv : c->graphs) hf_destroy_plan(kv.second);
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
    if (w.wp) {
        hipLaunchKernelGGL(k_hf_to_fm16, dim3(256), dim3(256), 0, s, w.w, w.wp, w.Co, w.k * w.Ci);
        GVC_LAUNCH_CHECK();
    }
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
    if (n.rfind("conv_pre.", 0) == 0) {
        rc = hf_bind_conv(c->pre, is_bias, src, numel, name, s);
        if (!rc && is_w && c->pre_wp) {
            hipLaunchKernelGGL(k_conv_pack_slices, dim3(512), dim3(256), 0, s, src, c->pre_wp, c->pre.Co, c->pre.Ci, c->pre.k, kPreSlice);
            GVC_LAUNCH_CHECK();
        }
    }
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
            if (u.wp) {
                hipLaunchKernelGGL(k_hf_to_fm16, dim3(256), dim3(256), 0, s, u.w, u.wp, u.s * u.Co, u.ntap * u.Ci);
                GVC_LAUNCH_CHECK();
            }
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

// tiled-GEMM conv: out rows [PAD, PAD+T) = epilogue(conv(lrelu?(src)))
static int hf_conv_gemm(gvc_hifigan* c, const HfConv& w, const float* src, float* dst, int T, int B, float a_slope,
                        const float* resid, const float* resid2, float out_scale, hipStream_t s) {
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

// conv_pre .. last ResBlock; returns what conv_post reads (a buffer, or the ResBlock planes of the last stage) and its length
static int hf_body(gvc_hifigan* c, int B, int T0, hipStream_t s, HfIn* out, int* T_out) {
    int rc;
    const gvc_hifigan_dims& D = c->dm;
    const int nk = D.n_kernels;
    const HfConv& pc = c->pre;
    const int nsplit = pc.Ci / kPreSlice, pre_tiles = pc.Co / 16;
    if (c->small_conv == 1 && c->pre_wp && (long long)nsplit * B * T0 * pc.Co <= c->work_cap && B * cdiv(T0, 32) * pre_tiles <= kPreCounters) {
        // conv_pre: 16 output columns x a 64-channel slice of the 7 taps per workgroup, slices combined by the last one to finish
        ConvLdsArgs A;
        memset(&A, 0, sizeof(A));
        A.x = c->x0; A.x_bs = (long long)(T0 + 2 * kHfPad) * pc.Ci; A.x_ps = kPreSlice; A.ldx = pc.Ci; A.x_scale = 1.f; A.slope = 1.f;
        A.x_row0 = kHfPad; A.x_rows = T0 + 2 * kHfPad; A.stride = 1;
        A.y = c->work; A.y_bs = (long long)T0 * pc.Co; A.y_ps = (long long)B * T0 * pc.Co; A.ldy = pc.Co;
        A.yf = c->x1; A.yf_bs = (long long)(T0 + 2 * kHfPad) * pc.Co; A.yf_off = (long long)kHfPad * pc.Co;
        A.T = T0; A.ntiles = pre_tiles; A.split = nsplit; A.wp_js = (long long)pre_tiles * pc.k * (kPreSlice / 16) * 64; A.cnt = c->cnt;
        A.job[0].wp = reinterpret_cast<const float4*>(c->pre_wp); A.job[0].b = pc.b; A.job[0].k = pc.k; A.job[0].dil = 1;
        A.job[0].row_off = -(pc.k - 1) / 2;
        if ((rc = launch_conv_lds_split(kPreSlice, A, B, conv_lds_bytes(kPreSlice, pc.k, 1, true), s))) return rc;
    } else if ((rc = hf_conv_gemm(c, pc, c->x0, c->x1, T0, B, 0.f, nullptr, nullptr, 0.f, s))) return rc;
    HfIn in;
    in.x = c->x1;
    int T = T0;
    for (int i = 0; i < D.n_ups; ++i) {
        const HfUp& u = c->ups[i];
        const int To = T * u.s, N = u.s * u.Co;
        // upsample: T input frames x N = s*Co columns (row q of the output = frames s*q .. s*q + s - 1), ntap input rows from q + dmin
        if (c->small_conv == 1 && u.wp && (in.nsum == 1 || in.nsum == 3)) {
            ConvLdsArgs A;
            memset(&A, 0, sizeof(A));
            A.x = in.x; A.x_bs = (long long)(T + 2 * kHfPad) * u.Ci; A.ldx = u.Ci; A.x_ss = in.ss; A.x_scale = in.scale; A.slope = 0.1f;
            A.x_row0 = kHfPad; A.x_rows = T + 2 * kHfPad; A.stride = 1;
            A.y = c->U[i]; A.y_bs = (long long)(To + 2 * kHfPad) * u.Co; A.y_off = (long long)kHfPad * u.Co; A.ldy = N;
            A.T = T; A.ntiles = N / 16;
            A.job[0].wp = reinterpret_cast<const float4*>(u.wp); A.job[0].b = u.b; A.job[0].k = u.ntap; A.job[0].dil = 1; A.job[0].row_off = u.dmin;
            if ((rc = launch_conv_lds(u.Ci, in.nsum, A, 1, B, conv_lds_bytes(u.Ci, u.ntap, 1), s))) return rc;
        } else {
            if (in.nsum > 1) {         // the tiled GEMM reads one buffer: add the planes in place first
                GVC_REQUIRE(in.nsum == 3, GVC_ERR_STATE, "hifigan: %d ResBlock planes", in.nsum);
                hipLaunchKernelGGL(k_sum_planes, dim3(256), dim3(256), 0, s, const_cast<float*>(in.x), in.ss, in.scale,
                                   (size_t)B * (T + 2 * kHfPad) * u.Ci / 4);
                GVC_LAUNCH_CHECK();
            }
            GemmArgs G;
            memset(&G, 0, sizeof(G));
            G.A = in.x + (size_t)(kHfPad + u.dmin) * u.Ci; G.lda = u.Ci; G.a_batch_stride = (long long)(T + 2 * kHfPad) * u.Ci;
            G.a_act = AACT_LRELU; G.a_slope = 0.1f;
            G.Wt = u.w; G.ldw = u.ntap * u.Ci;
            G.C = c->U[i] + (size_t)kHfPad * u.Co; G.ldc = N; G.c_batch_stride = (long long)(To + 2 * kHfPad) * u.Co;
            G.M = T; G.N = N; G.K = u.ntap * u.Ci; G.work = c->work; G.e.bias = u.b;
            if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
        }
        T = To;
        const int ch = u.Co;
        const HfConv* cv = &c->res[(size_t)i * nk * 2];
        bool planes = c->small_conv && nk == 3;
        size_t lds_a = 0, lds_b = 0;
        for (int j = 0; j < nk && planes; ++j) {
            planes = cv[2 * j].wp && cv[2 * j + 1].wp;
            lds_a = std::max(lds_a, conv_lds_bytes(ch, cv[2 * j].k, cv[2 * j].dil));
            lds_b = std::max(lds_b, conv_lds_bytes(ch, cv[2 * j + 1].k, cv[2 * j + 1].dil));
        }
        if (planes) {
            // ResBlock2 j: r_j = U + conv_a(lrelu(U)); out_j = r_j + conv_b(lrelu(r_j)); the consumer adds out_0..2 and divides
            ConvLdsArgs A;
            memset(&A, 0, sizeof(A));
            A.x = c->U[i]; A.x_bs = (long long)(T + 2 * kHfPad) * ch; A.ldx = ch; A.slope = 0.1f; A.x_scale = 1.f;
            A.x_row0 = kHfPad; A.x_rows = T + 2 * kHfPad; A.stride = 1;
            A.y = c->R[i]; A.y_bs = A.x_bs; A.y_ps = c->plane[i]; A.y_off = (long long)kHfPad * ch; A.ldy = ch;
            A.resid = c->U[i]; A.T = T; A.ntiles = ch / 16;
            for (int j = 0; j < nk; ++j) {
                const HfConv& w = cv[2 * j];
                A.job[j].wp = reinterpret_cast<const float4*>(w.wp); A.job[j].b = w.b; A.job[j].k = w.k; A.job[j].dil = w.dil;
                A.job[j].row_off = -(w.dil * (w.k - 1) / 2);
            }
            if ((rc = launch_conv_lds(ch, 1, A, nk, B, lds_a, s))) return rc;
            A.x = c->R[i]; A.x_ps = c->plane[i];
            A.y = c->S[i];
            A.resid = c->R[i]; A.r_ps = c->plane[i];
            for (int j = 0; j < nk; ++j) {
                const HfConv& w = cv[2 * j + 1];
                A.job[j].wp = reinterpret_cast<const float4*>(w.wp); A.job[j].b = w.b; A.job[j].k = w.k; A.job[j].dil = w.dil;
                A.job[j].row_off = -(w.dil * (w.k - 1) / 2);
            }
            if ((rc = launch_conv_lds(ch, 1, A, nk, B, lds_b, s))) return rc;
            in.x = c->S[i]; in.nsum = nk; in.ss = c->plane[i]; in.scale = 1.0f / (float)nk;
        } else {
            // one launch per conv, running sum over j in planes 0 / 1 of S, /n_kernels folded into the last epilogue
            float* acc[2] = {c->S[i], c->S[i] + c->plane[i]};
            const float* prev = nullptr;
            for (int j = 0; j < nk; ++j) {
                if ((rc = hf_conv_gemm(c, cv[2 * j], c->U[i], c->R[i], T, B, 0.1f, c->U[i], nullptr, 0.f, s))) return rc;
                float* dst = acc[j & 1];
                if ((rc = hf_conv_gemm(c, cv[2 * j + 1], c->R[i], dst, T, B, 0.1f, c->R[i], prev, j == nk - 1 ? 1.0f / (float)nk : 0.f, s)))
                    return rc;
                prev = dst;
            }
            in = HfIn();
            in.x = prev;
        }
    }
    *out = in;
    *T_out = T;
    return GVC_OK;
}

enum HfHead { kHeadLatents = 0, kHeadChannelsFirst = 1 };

// input staging (x4 interpolation of latents [B][n][d], or a channels-first [B][d][T0] input) -> x0
static void hf_head(gvc_hifigan* c, int head, const float* in, int B, int T0, int n, int scale, hipStream_t s) {
    if (head == kHeadLatents) hipLaunchKernelGGL(k_interp_linear, dim3(T0, B), dim3(256), 0, s, in, c->x0, n, c->dm.in_dim, scale, T0);
    else hipLaunchKernelGGL(k_cf_to_time_major, dim3(cdiv(T0, 32), cdiv(c->dm.in_dim, 32), B), dim3(32, 8), 0, s, in, c->x0, c->dm.in_dim, T0);
}

static int hf_post(gvc_hifigan* c, const HfIn& x, int B, int T, float* wav, hipStream_t s) {
    const HfConv& p = c->post;
    GVC_REQUIRE(x.nsum == 1 || x.nsum == 3, GVC_ERR_STATE, "hifigan: %d ResBlock planes", x.nsum);
    const size_t lds = ((size_t)p.k * p.Ci + (size_t)(64 + p.k - 1) * (p.Ci + 1)) * sizeof(float);
    // F.leaky_relu default slope before conv_post (hifigan.py:230)
    if (x.nsum == 1)
        hipLaunchKernelGGL((k_conv_post_tanh<1>), dim3(cdiv(T, 64), B), dim3(256), lds, s, x.x, x.ss, x.scale, p.w, p.b, wav, T, p.Ci, p.k, 0.01f);
    else
        hipLaunchKernelGGL((k_conv_post_tanh<3>), dim3(cdiv(T, 64), B), dim3(256), lds, s, x.x, x.ss, x.scale, p.w, p.b, wav, T, p.Ci, p.k, 0.01f);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

static int hf_run(gvc_hifigan* c, int B, int T0, int head, const float* in, int n, int scale, float* wav, hipStream_t s);

// a node of an executable graph refused its new parameters: from now on only the conv chain is replayed from a graph
static int hf_patch_failed(gvc_hifigan* c, int B, int T0, int head, const float* in, int n, int scale, float* wav, hipStream_t s) {
    (void)hipGetLastError();
    for (auto& kv : c->graphs) hf_destroy_plan(kv.second);
    c->graphs.clear();
    c->use_graph = 2;
    hf_head(c, head, in, B, T0, n, scale, s);
    GVC_LAUNCH_CHECK();
    return hf_run(c, B, T0, head, in, n, scale, wav, s);
}

// The whole call is ONE graph per (entry point, B, frames, scale): input staging, the conv chain, conv_post.  The caller's two
// pointers (input, waveform) are parameters of the first and the last kernel node and are patched in the executable graph when
// they differ from the previous call's (a stream launch after a graph launch costs an 8 us bubble; a patch is host work).
// GVC_VOCODER_GRAPH=2: only the chain between them is captured (the round-2 scheme).
static int hf_run(gvc_hifigan* c, int B, int T0, int head, const float* in, int n, int scale, float* wav, hipStream_t s) {
    int rc;
    if (!c->use_graph) {
        HfIn x;
        int T = 0;
        hf_head(c, head, in, B, T0, n, scale, s);
        GVC_LAUNCH_CHECK();
        if ((rc = hf_body(c, B, T0, s, &x, &T))) return rc;
        return hf_post(c, x, B, T, wav, s);
    }
    const bool whole = c->use_graph == 1;
    const long long key = ((long long)(whole ? head + 1 : 0) << 60) | ((long long)B << 44) | ((long long)(whole ? scale : 0) << 32) | (unsigned)T0;
    auto it = c->graphs.find(key);
    if (it == c->graphs.end()) {
        if (c->graphs.size() >= 64) {          // bounded cache: the frame count of a non-streaming call varies freely
            for (auto& kv : c->graphs) hf_destroy_plan(kv.second);
            c->graphs.clear();
        }
        HfPlan pl;
        GVC_CHECK_HIP(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
        if (whole) hf_head(c, head, in, B, T0, n, scale, c->cap_stream);
        rc = hf_body(c, B, T0, c->cap_stream, &pl.out, &pl.T);
        if (!rc && whole) rc = hf_post(c, pl.out, B, pl.T, wav, c->cap_stream);
        hipError_t e = hipStreamEndCapture(c->cap_stream, &pl.graph);
        if (rc) { if (pl.graph) hipGraphDestroy(pl.graph); return rc; }
        GVC_CHECK_HIP(e);
        e = hipGraphInstantiate(&pl.ge[0], pl.graph, nullptr, nullptr, 0);
        if (e != hipSuccess) hipGraphDestroy(pl.graph);
        GVC_CHECK_HIP(e);
        if (whole) {
            // the two nodes that carry the caller's pointers
            size_t nn = 0;
            GVC_CHECK_HIP(hipGraphGetNodes(pl.graph, nullptr, &nn));
            std::vector<hipGraphNode_t> nodes(nn);
            GVC_CHECK_HIP(hipGraphGetNodes(pl.graph, nodes.data(), &nn));
            const void* head_fn = head == kHeadLatents ? reinterpret_cast<const void*>(&k_interp_linear) : reinterpret_cast<const void*>(&k_cf_to_time_major);
            const void* post_fn = pl.out.nsum == 1 ? reinterpret_cast<const void*>(&k_conv_post_tanh<1>) : reinterpret_cast<const void*>(&k_conv_post_tanh<3>);
            for (hipGraphNode_t nd : nodes) {
                hipGraphNodeType ty;
                hipKernelNodeParams kp;
                if (hipGraphNodeGetType(nd, &ty) != hipSuccess || ty != hipGraphNodeTypeKernel) continue;
                if (hipGraphKernelNodeGetParams(nd, &kp) != hipSuccess) continue;
                if (kp.func == head_fn) { pl.head = nd; pl.head_grid = kp.gridDim; pl.head_block = kp.blockDim; }
                if (kp.func == post_fn) { pl.post = nd; pl.post_grid = kp.gridDim; pl.post_block = kp.blockDim; pl.post_lds = kp.sharedMemBytes; }
            }
            GVC_REQUIRE(pl.head && pl.post, GVC_ERR_STATE, "hifigan: the captured graph has no input / output kernel node");
            pl.in[0] = in; pl.wav[0] = wav;
        }
        it = c->graphs.emplace(key, pl).first;
    }
    HfPlan& pl = it->second;
    // which executable graph: one that already holds the caller's pointers, else the least recently launched (instantiated on first
    // use), patched once its previous launch has finished
    int slot = -1;
    if (whole) {
        for (int i = 0; i < kHfRing && slot < 0; ++i)
            if (pl.ge[i] && pl.in[i] == in && pl.wav[i] == wav) slot = i;
        if (slot < 0) {
            slot = 0;
            for (int i = 1; i < kHfRing; ++i)
                if (pl.used[i] < pl.used[slot]) slot = i;
            if (!pl.ge[slot]) {
                GVC_CHECK_HIP(hipGraphInstantiate(&pl.ge[slot], pl.graph, nullptr, nullptr, 0));
                pl.in[slot] = nullptr;                                  // (a fresh instance holds the captured pointers: patch both)
                pl.wav[slot] = nullptr;
            }
            if (pl.flying[slot]) { GVC_CHECK_HIP(hipEventSynchronize(pl.ev[slot])); pl.flying[slot] = false; }
        }
    } else {
        slot = 0;
    }
    hipGraphExec_t ge = pl.ge[slot];
    if (whole && pl.in[slot] != in) {
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.gridDim = pl.head_grid; kp.blockDim = pl.head_block;
        float* x0 = c->x0;
        int d = c->dm.in_dim, t0 = T0, nn = n, sc = scale;
        void* a_lat[] = {&in, &x0, &nn, &d, &sc, &t0};
        void* a_cf[] = {&in, &x0, &d, &t0};
        kp.func = head == kHeadLatents ? reinterpret_cast<void*>(&k_interp_linear) : reinterpret_cast<void*>(&k_cf_to_time_major);
        kp.kernelParams = head == kHeadLatents ? a_lat : a_cf;
        if (hipGraphExecKernelNodeSetParams(ge, pl.head, &kp) != hipSuccess) return hf_patch_failed(c, B, T0, head, in, n, scale, wav, s);
        pl.in[slot] = in;
    }
    if (whole && pl.wav[slot] != wav) {
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        kp.gridDim = pl.post_grid; kp.blockDim = pl.post_block; kp.sharedMemBytes = pl.post_lds;
        const HfConv& p = c->post;
        HfIn x = pl.out;
        int T = pl.T, C = p.Ci, k = p.k;
        float slope = 0.01f;
        const float *pw = p.w, *pb = p.b;
        void* a[] = {&x.x, &x.ss, &x.scale, &pw, &pb, &wav, &T, &C, &k, &slope};
        kp.func = x.nsum == 1 ? reinterpret_cast<void*>(&k_conv_post_tanh<1>) : reinterpret_cast<void*>(&k_conv_post_tanh<3>);
        kp.kernelParams = a;
        if (hipGraphExecKernelNodeSetParams(ge, pl.post, &kp) != hipSuccess) return hf_patch_failed(c, B, T0, head, in, n, scale, wav, s);
        pl.wav[slot] = wav;
    }
    GVC_CHECK_HIP(hipGraphLaunch(ge, s));
    if (whole) {
        if (!pl.ev[slot]) GVC_CHECK_HIP(hipEventCreateWithFlags(&pl.ev[slot], hipEventDisableTiming));
        GVC_CHECK_HIP(hipEventRecord(pl.ev[slot], s));
        pl.flying[slot] = true;
        pl.used[slot] = ++c->graph_tick;
        return GVC_OK;
    }
    // round-2 scheme: staging before, conv_post after the graph (the staging kernel was enqueued by the caller of this branch)
    return hf_post(c, pl.out, B, pl.T, wav, s);
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
        const int np = c->dm.n_kernels > 2 ? c->dm.n_kernels : 2;
        for (size_t i = 0; i < c->ups.size(); ++i) {
            T *= c->ups[i].s; ch = c->ups[i].Co;
            const size_t bytes = (size_t)B * (T + 2 * kHfPad) * ch * sizeof(float);
            GVC_CHECK_HIP(hipMemsetAsync(c->U[i], 0, bytes, s));
            for (int p = 0; p < np; ++p) {
                GVC_CHECK_HIP(hipMemsetAsync(c->R[i] + (size_t)p * c->plane[i], 0, bytes, s));
                GVC_CHECK_HIP(hipMemsetAsync(c->S[i] + (size_t)p * c->plane[i], 0, bytes, s));
            }
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
    if (c->use_graph == 2) {
        hf_head(c, kHeadLatents, latents, B, T0, n, scale, s);
        GVC_LAUNCH_CHECK();
    }
    return hf_run(c, B, T0, kHeadLatents, latents, n, scale, wav, s);
}

extern "C" int gvc_hifigan_forward(gvc_hifigan* c, const float* x, int32_t B, int32_t T, float* wav, gvc_stream sv) {
    GVC_REQUIRE(c && x && wav, GVC_ERR_ARG, "gvc_hifigan_forward: bad argument");
    hipStream_t s = (hipStream_t)sv;
    int rc = hf_prepare(c, B, T, s);
    if (rc) return rc;
    if (c->use_graph == 2) {
        hf_head(c, kHeadChannelsFirst, x, B, T, T, 1, s);
        GVC_LAUNCH_CHECK();
    }
    return hf_run(c, B, T, kHeadChannelsFirst, x, T, 1, wav, s);
}
