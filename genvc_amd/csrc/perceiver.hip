// Perceiver resampler (reference layers/perceiver_encoder.py:225-319 as driven by
// GPT.get_style_emb, layers/gpt.py:351-373, mask=None).
//
// HBM layout: one buffer C[B][num_latents + F][dim] holds the latents in its first rows and the
// projected context frames behind them, so "keys = cat(latents, context)" (cross_attn_include_queries,
// perceiver_encoder.py:310-311) is simply the whole buffer and to_kv is one GEMM over it.  All matrix
// products run on the fp32 MFMA GEMM (gemm.h); attention is the shared online-softmax kernel with
// head_dim 64; GEGLU uses the exact erf GELU (perceiver_encoder.py:205-208).
#include <map>
#include <string>
#include <vector>

#include "gemm.h"
#include "gpt_kernels.h"

namespace gvc {

__global__ void k_fill_latents(float* C, const float* latents, int B, int rows_per_batch, int n_lat, int d) {
    const int row = blockIdx.x;            // b * n_lat + i
    const int b = row / n_lat, i = row - b * n_lat;
    float* dst = C + ((size_t)b * rows_per_batch + i) * d;
    const float* src = latents + (size_t)i * d;
    for (int k = threadIdx.x * 4; k < d; k += blockDim.x * 4)
        *reinterpret_cast<float4*>(dst + k) = *reinterpret_cast<const float4*>(src + k);
}

// g[m][j] = gelu_erf(h[m][ffi + j]) * h[m][j], zero in the padding columns [ffi, ffi_p)
__global__ void k_geglu(const float* h, float* g, int rows, int ffi, int ffi_p) {
    const int row = blockIdx.x;
    const float* hr = h + (size_t)row * 2 * ffi;
    float* gr = g + (size_t)row * ffi_p;
    for (int j = threadIdx.x; j < ffi_p; j += blockDim.x)
        gr[j] = j < ffi ? gelu_erf(hr[ffi + j]) * hr[j] : 0.f;
}

// out = x / max(|x|_2, 1e-12) * sqrt(d) * gamma   (RMSNorm, perceiver_encoder.py:177-179); wave per row
__global__ void k_rmsnorm_rows(const float* C, float* out, int B, int rows_per_batch, int n_lat, int d,
                               const float* gamma) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= B * n_lat) return;
    const int b = row / n_lat, i = row - b * n_lat;
    const float* x = C + ((size_t)b * rows_per_batch + i) * d;
    float* y = out + (size_t)row * d;
    float q = 0.f;
    for (int k = lane * 4; k < d; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + k);
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    const float sc = sqrtf((float)d);
    for (int k = lane * 4; k < d; k += 256) {
        float4 v = *reinterpret_cast<const float4*>(x + k);
        const float4 g = *reinterpret_cast<const float4*>(gamma + k);
        v.x = v.x / nrm * sc * g.x; v.y = v.y / nrm * sc * g.y;
        v.z = v.z / nrm * sc * g.z; v.w = v.w / nrm * sc * g.w;
        *reinterpret_cast<float4*>(y + k) = v;
    }
}

// copy a [rows][cols] matrix into a [rows][ld] one (ld >= cols), zero padding
__global__ void k_pad_cols(const float* src, float* dst, int rows, int cols, int ld) {
    const int row = blockIdx.x;
    for (int j = threadIdx.x; j < ld; j += blockDim.x)
        dst[(size_t)row * ld + j] = j < cols ? src[(size_t)row * cols + j] : 0.f;
}

}  // namespace gvc

using namespace gvc;

struct PercLayer { float *to_q, *to_kv, *to_out, *ff1_w, *ff1_b, *ff2_w, *ff2_b; };

struct gvc_perceiver {
    gvc_perceiver_dims dm;
    int inner = 0, ffi = 0, ffi_p = 0, ctx_p = 0;
    bool has_proj = false;
    float* wbase = nullptr;
    float *latents, *gamma, *proj_w, *proj_b;
    std::vector<PercLayer> layers;
    std::map<std::string, int> bound;
    int n_expected = 0;
    float *C = nullptr, *q = nullptr, *kv = nullptr, *o = nullptr, *h = nullptr, *g = nullptr, *xp = nullptr, *work = nullptr;
    long long work_cap = 0;
};

extern "C" int gvc_perceiver_create(const gvc_perceiver_dims* dims, gvc_perceiver** out) {
    GVC_REQUIRE(dims && out, GVC_ERR_ARG, "gvc_perceiver_create: null argument");
    const gvc_perceiver_dims& D = *dims;
    GVC_REQUIRE(D.dim % 256 == 0 && D.dim_head == 64 && D.heads >= 1 && D.depth >= 1 && D.num_latents >= 1,
                GVC_ERR_UNSUPPORTED, "perceiver: need dim %% 256 == 0 and dim_head == 64");
    auto* c = new gvc_perceiver();
    c->dm = D;
    c->inner = D.dim_head * D.heads;
    c->ffi = (int)((long long)D.dim * D.ff_mult * 2 / 3);
    c->ffi_p = (c->ffi + 3) & ~3;
    c->ctx_p = (D.dim_context + 3) & ~3;
    c->has_proj = D.dim_context != D.dim;
    const size_t d = D.dim, in = c->inner;
    const size_t per_layer = in * d + 2 * in * d + d * in + 2 * (size_t)c->ffi * d + 2 * c->ffi + d * c->ffi_p + d + 64;
    const size_t total = (size_t)D.num_latents * d + d + d * c->ctx_p + d + D.depth * per_layer + 64;
    GVC_CHECK_HIP(hipMalloc((void**)&c->wbase, total * sizeof(float)));
    float* p = c->wbase;
    auto take = [&](size_t n) { float* r = p; p += (n + 3) & ~(size_t)3; return r; };
    c->latents = take((size_t)D.num_latents * d); c->gamma = take(d);
    c->proj_w = take(d * c->ctx_p); c->proj_b = take(d);
    c->layers.resize(D.depth);
    for (auto& ly : c->layers) {
        ly.to_q = take(in * d); ly.to_kv = take(2 * in * d); ly.to_out = take(d * in);
        ly.ff1_w = take(2 * (size_t)c->ffi * d); ly.ff1_b = take(2 * c->ffi);
        ly.ff2_w = take(d * c->ffi_p); ly.ff2_b = take(d);
    }
    c->n_expected = 2 + (c->has_proj ? 2 : 0) + 7 * D.depth;
    const size_t B = D.max_batch, R = (size_t)D.num_latents + D.max_frames, NL = D.num_latents;
    c->work_cap = 4ll << 20;
    GVC_CHECK_HIP(hipMalloc((void**)&c->C, B * R * d * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->kv, B * R * 2 * in * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->q, B * NL * in * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->o, B * NL * in * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->h, B * NL * 2 * c->ffi * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->g, B * NL * c->ffi_p * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->xp, B * (size_t)D.max_frames * c->ctx_p * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->work, (size_t)c->work_cap * sizeof(float)));
    *out = c;
    return GVC_OK;
}

extern "C" int gvc_perceiver_destroy(gvc_perceiver* c) {
    if (!c) return GVC_OK;
    for (void* p : {(void*)c->wbase, (void*)c->C, (void*)c->q, (void*)c->kv, (void*)c->o, (void*)c->h, (void*)c->g,
                    (void*)c->xp, (void*)c->work})
        if (p) hipFree(p);
    delete c;
    return GVC_OK;
}

static int pcopy(float* dst, const float* src, int64_t numel, int64_t expect, const char* name, hipStream_t s) {
    GVC_REQUIRE(numel == expect, GVC_ERR_ARG, "%s: expected %lld elements, got %lld", name, (long long)expect,
                (long long)numel);
    GVC_CHECK_HIP(hipMemcpyAsync(dst, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
    return GVC_OK;
}

static int ppad(float* dst, const float* src, int64_t numel, int rows, int cols, int ld, const char* name, hipStream_t s) {
    GVC_REQUIRE(numel == (int64_t)rows * cols, GVC_ERR_ARG, "%s: expected %lld elements, got %lld", name,
                (long long)rows * cols, (long long)numel);
    hipLaunchKernelGGL(k_pad_cols, dim3(rows), dim3(256), 0, s, src, dst, rows, cols, ld);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

extern "C" int gvc_perceiver_bind_weight(gvc_perceiver* c, const char* name, const float* src, int64_t numel,
                                         gvc_stream sv) {
    GVC_REQUIRE(c && name && src, GVC_ERR_ARG, "gvc_perceiver_bind_weight: null argument");
    hipStream_t s = (hipStream_t)sv;
    const int64_t d = c->dm.dim, in = c->inner;
    std::string n(name);
    int rc = GVC_OK;
    bool known = true;
    if (n == "latents") rc = pcopy(c->latents, src, numel, c->dm.num_latents * d, name, s);
    else if (n == "norm.gamma") rc = pcopy(c->gamma, src, numel, d, name, s);
    else if (n == "proj_context.weight" && c->has_proj) rc = ppad(c->proj_w, src, numel, d, c->dm.dim_context, c->ctx_p, name, s);
    else if (n == "proj_context.bias" && c->has_proj) rc = pcopy(c->proj_b, src, numel, d, name, s);
    else if (n.rfind("layers.", 0) == 0) {
        const size_t dot = n.find('.', 7);
        GVC_REQUIRE(dot != std::string::npos, GVC_ERR_ARG, "malformed weight name %s", name);
        const int li = atoi(n.substr(7, dot - 7).c_str());
        GVC_REQUIRE(li >= 0 && li < c->dm.depth, GVC_ERR_ARG, "%s: layer out of range", name);
        const std::string rest = n.substr(dot + 1);
        PercLayer& ly = c->layers[li];
        if (rest == "0.to_q.weight") rc = pcopy(ly.to_q, src, numel, in * d, name, s);
        else if (rest == "0.to_kv.weight") rc = pcopy(ly.to_kv, src, numel, 2 * in * d, name, s);
        else if (rest == "0.to_out.weight") rc = pcopy(ly.to_out, src, numel, d * in, name, s);
        else if (rest == "1.0.weight") rc = pcopy(ly.ff1_w, src, numel, 2 * (int64_t)c->ffi * d, name, s);
        else if (rest == "1.0.bias") rc = pcopy(ly.ff1_b, src, numel, 2 * c->ffi, name, s);
        else if (rest == "1.2.weight") rc = ppad(ly.ff2_w, src, numel, d, c->ffi, c->ffi_p, name, s);
        else if (rest == "1.2.bias") rc = pcopy(ly.ff2_b, src, numel, d, name, s);
        else known = false;
    } else {
        known = false;
    }
    if (rc == GVC_OK && known) c->bound[n] = 1;
    return rc;
}

extern "C" int gvc_perceiver_missing_weights(gvc_perceiver* c) {
    return c ? c->n_expected - (int)c->bound.size() : -1;
}

extern "C" int gvc_perceiver_forward(gvc_perceiver* c, const float* x, int32_t B, int32_t F, float* out, gvc_stream sv) {
    GVC_REQUIRE(c && x && out, GVC_ERR_ARG, "gvc_perceiver_forward: null argument");
    GVC_REQUIRE(gvc_perceiver_missing_weights(c) == 0, GVC_ERR_STATE, "%d Perceiver weight tensors are not bound",
                gvc_perceiver_missing_weights(c));
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_batch && F >= 1 && F <= c->dm.max_frames, GVC_ERR_ARG,
                "perceiver: B=%d F=%d outside capacity (%d, %d)", B, F, c->dm.max_batch, c->dm.max_frames);
    hipStream_t s = (hipStream_t)sv;
    const int d = c->dm.dim, in = c->inner, NL = c->dm.num_latents, R = NL + F, dc = c->dm.dim_context;
    int rc;
    GemmArgs G;
    // context frames -> rows NL.. of every batch element
    if (c->has_proj) {
        const float* xa = x;
        int lda = dc;
        if (c->ctx_p != dc) {
            hipLaunchKernelGGL(k_pad_cols, dim3(B * F), dim3(128), 0, s, x, c->xp, B * F, dc, c->ctx_p);
            GVC_LAUNCH_CHECK();
            xa = c->xp;
            lda = c->ctx_p;
        }
        memset(&G, 0, sizeof(G));
        G.A = xa; G.lda = lda; G.a_batch_stride = (long long)F * lda; G.Wt = c->proj_w; G.ldw = c->ctx_p;
        G.C = c->C + (size_t)NL * d; G.ldc = d; G.c_batch_stride = (long long)R * d;
        G.M = F; G.N = d; G.K = c->ctx_p; G.work = c->work; G.e.bias = c->proj_b;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
    } else {
        for (int b = 0; b < B; ++b)
            GVC_CHECK_HIP(hipMemcpyAsync(c->C + ((size_t)b * R + NL) * d, x + (size_t)b * F * d,
                                         (size_t)F * d * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    hipLaunchKernelGGL(k_fill_latents, dim3(B * NL), dim3(256), 0, s, c->C, c->latents, B, R, NL, d);
    GVC_LAUNCH_CHECK();

    for (int l = 0; l < c->dm.depth; ++l) {
        const PercLayer& ly = c->layers[l];
        // q = latents @ to_q^T ; kv = [latents; ctx] @ to_kv^T
        memset(&G, 0, sizeof(G));
        G.A = c->C; G.lda = d; G.a_batch_stride = (long long)R * d; G.Wt = ly.to_q; G.ldw = d;
        G.C = c->q; G.ldc = in; G.c_batch_stride = (long long)NL * in; G.M = NL; G.N = in; G.K = d; G.work = c->work;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
        memset(&G, 0, sizeof(G));
        G.A = c->C; G.lda = d; G.a_batch_stride = (long long)R * d; G.Wt = ly.to_kv; G.ldw = d;
        G.C = c->kv; G.ldc = 2 * in; G.c_batch_stride = (long long)R * 2 * in; G.M = R; G.N = 2 * in; G.K = d; G.work = c->work;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;

        AttnArgs T;
        memset(&T, 0, sizeof(T));
        T.q = c->q; T.q_stride = in; T.kbase = c->kv; T.vbase = c->kv + in;
        T.k_batch_stride = (long long)R * 2 * in; T.k_head_stride = c->dm.dim_head; T.k_row_stride = 2 * in;
        T.T = NL; T.causal = 0; T.n_keys = R; T.scale = 1.0f / sqrtf((float)c->dm.dim_head);
        T.out = c->o; T.out_stride = in;
        if ((rc = launch_attention_hd(c->dm.dim_head, c->dm.heads, T, 1, B * NL, true, s))) return rc;

        // latents += o @ to_out^T
        memset(&G, 0, sizeof(G));
        G.A = c->o; G.lda = in; G.a_batch_stride = (long long)NL * in; G.Wt = ly.to_out; G.ldw = in;
        G.C = c->C; G.ldc = d; G.c_batch_stride = (long long)R * d; G.M = NL; G.N = d; G.K = in; G.work = c->work;
        G.e.resid = c->C; G.e.ldr = d; G.e.resid_batch_stride = (long long)R * d;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;

        // feed-forward: Linear -> GEGLU -> Linear, residual
        memset(&G, 0, sizeof(G));
        G.A = c->C; G.lda = d; G.a_batch_stride = (long long)R * d; G.Wt = ly.ff1_w; G.ldw = d;
        G.C = c->h; G.ldc = 2 * c->ffi; G.c_batch_stride = (long long)NL * 2 * c->ffi; G.M = NL; G.N = 2 * c->ffi; G.K = d;
        G.work = c->work; G.e.bias = ly.ff1_b;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
        hipLaunchKernelGGL(k_geglu, dim3(B * NL), dim3(256), 0, s, c->h, c->g, B * NL, c->ffi, c->ffi_p);
        GVC_LAUNCH_CHECK();
        memset(&G, 0, sizeof(G));
        G.A = c->g; G.lda = c->ffi_p; G.a_batch_stride = (long long)NL * c->ffi_p; G.Wt = ly.ff2_w; G.ldw = c->ffi_p;
        G.C = c->C; G.ldc = d; G.c_batch_stride = (long long)R * d; G.M = NL; G.N = d; G.K = c->ffi_p; G.work = c->work;
        G.e.bias = ly.ff2_b; G.e.resid = c->C; G.e.ldr = d; G.e.resid_batch_stride = (long long)R * d;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
    }
    hipLaunchKernelGGL(k_rmsnorm_rows, dim3(cdiv(B * NL, 4)), dim3(256), 0, s, c->C, out, B, R, NL, d, c->gamma);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}
