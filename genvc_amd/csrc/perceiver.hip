// Perceiver resampler (reference layers/perceiver_encoder.py:225-319 as driven by
// GPT.get_style_emb, layers/gpt.py:351-373, mask=None).
//
// Round 5: the forward is ~23 launches instead of ~40, replayed from one hipGraph per (B, F):
//   * the context frames never change across the layers, so their keys / values for ALL layers are one MFMA GEMM
//     ([F] x [depth x 2 inner] columns, a two-level batch over (batch element, layer)) in front of the layer loop;
//   * the 32 latent rows -- everything the layer loop touches -- live fragment-major (FM16, gemm.h) and go through the skinny GEMM of
//     the GPT's streaming prefill (one workgroup per 16 weight rows, weights streamed once, both operands FM16): q | k | v of the
//     latents in ONE GEMM (to_q and to_kv stacked), to_out with the residual in its epilogue, FF1 with GEGLU in its epilogue (x_j and
//     gate_j interleaved at bind time so a pair sits in neighbouring lanes), FF2 with the residual in its epilogue;
//   * cross-attention on v_mfma_f32_16x16x4_f32 (attn64.h): "keys = cat(latents, context)" (cross_attn_include_queries,
//     perceiver_encoder.py:310-311) is one buffer KV[B][32 + F][depth][q | k | v] whose first 32 rows the latents' GEMM fills.
// GEGLU uses the exact erf GELU (perceiver_encoder.py:205-208); the final RMSNorm reads the fragment-major latents.
#include <map>
#include <string>
#include <vector>

#include "gemm.h"
#include "attn64.h"

namespace gvc {

// copy a [rows][cols] matrix into a [rows][ld] one (ld >= cols), zero padding
__global__ void k_pad_cols(const float* src, float* dst, int rows, int cols, int ld) {
    const int row = blockIdx.x;
    for (int j = threadIdx.x; j < ld; j += blockDim.x)
        dst[(size_t)row * ld + j] = j < cols ? src[(size_t)row * cols + j] : 0.f;
}

// FF1 of the GEGLU feed-forward: W [2 ffi][d] (rows [0, ffi): x, rows [ffi, 2 ffi): gate) -> [2 ffi_p][d] with row 2 j = x_j, row
// 2 j + 1 = gate_j, zero rows past ffi; the bias likewise
__global__ void k_interleave_ff1(const float* w, const float* b, float* wo, float* bo, int ffi, int ffi_p, int d) {
    const int row = blockIdx.x;                         // output row
    const int j = row >> 1, src = (row & 1) * ffi + j;
    for (int k = threadIdx.x; k < d; k += blockDim.x) wo[(size_t)row * d + k] = j < ffi ? w[(size_t)src * d + k] : 0.f;
    if (threadIdx.x == 0 && b) bo[row] = j < ffi ? b[src] : 0.f;
}

// the latents parameter (FM16, 32 x d) replicated for every batch element: X[b] = latents
__global__ void k_rep_latents(const float* lat_fm, float* X, int n_per) {
    const float4* src = reinterpret_cast<const float4*>(lat_fm);
    float4* dst = reinterpret_cast<float4*>(X + (size_t)blockIdx.y * n_per);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_per / 4; i += gridDim.x * blockDim.x) dst[i] = src[i];
}

// out = x / max(|x|_2, 1e-12) * sqrt(d) * gamma   (RMSNorm, perceiver_encoder.py:177-179) over FM16 rows; wave per row
__global__ void k_rmsnorm_rows_fm16(const float* X, float* out, int rows, int d, const float* gamma) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    float q = 0.f;
    for (int k = lane * 4; k < d; k += 256) {
        const float4 v = *reinterpret_cast<const float4*>(X + fm16_index(row, k, d));
        q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    const float sc = sqrtf((float)d);
    for (int k = lane * 4; k < d; k += 256) {
        float4 v = *reinterpret_cast<const float4*>(X + fm16_index(row, k, d));
        const float4 g = *reinterpret_cast<const float4*>(gamma + k);
        v.x = v.x / nrm * sc * g.x; v.y = v.y / nrm * sc * g.y;
        v.z = v.z / nrm * sc * g.z; v.w = v.w / nrm * sc * g.w;
        *reinterpret_cast<float4*>(out + (size_t)row * d + k) = v;
    }
}

}  // namespace gvc

using namespace gvc;

// per layer: row-major staging copies as bound (w*) and the FM16 operands the forward streams (f*)
struct PercLayer {
    float *to_q, *to_kv, *to_out, *ff1_w, *ff1_b, *ff2_w, *ff2_b;      // as bound (row-major; ff2 K-padded)
    float *wqkv;                                                       // [3 inner][d] row-major: to_q over to_kv (the context GEMM reads its to_kv rows)
    float *f_qkv, *f_out, *f_ff1, *f_ff2, *b_ff1;                      // FM16 [3 inner][d], [d][inner], [2 ffi_p][d], [d][ffi_p]; interleaved bias
};

struct gvc_perceiver {
    gvc_perceiver_dims dm;
    int inner = 0, ffi = 0, ffi_p = 0, ctx_p = 0;
    bool has_proj = false;
    float* wbase = nullptr;
    float *latents, *gamma, *proj_w, *proj_b, *lat_fm;
    std::vector<PercLayer> layers;
    std::map<std::string, int> bound;
    int n_expected = 0;
    bool fm_ready = false;           // the FM16 / interleaved copies match the bound weights
    float *C = nullptr, *kv = nullptr, *X = nullptr, *o = nullptr, *g = nullptr, *xp = nullptr, *work = nullptr, *tmp = nullptr;
    long long work_cap = 0;
    std::map<long long, hipGraphExec_t> graphs;    // (B, F) -> captured body of the forward (context-owned buffers only)
    std::map<long long, unsigned long long> graph_used;   // (B, F) -> tick of its last replay (eviction: least recently used)
    unsigned long long tick = 0;
    hipStream_t cap_stream = nullptr;
    int use_graph = 1;               // GVC_PERCEIVER_GRAPH=0: eager launches
};

extern "C" int gvc_perceiver_create(const gvc_perceiver_dims* dims, gvc_perceiver** out) {
    GVC_REQUIRE(dims && out, GVC_ERR_ARG, "gvc_perceiver_create: null argument");
    const gvc_perceiver_dims& D = *dims;
    // (the latent rows live fragment-major in 16-row tiles and share one skinny GEMM of <= 128 rows per group of batch elements: any
    //  multiple of 16 up to 64 latents; GenVC's config fixes 32, gpt.py:163-170)
    GVC_REQUIRE(D.dim % 256 == 0 && D.dim_head == 64 && D.heads >= 1 && D.depth >= 1 && D.num_latents >= 16 && D.num_latents <= 64 &&
                    D.num_latents % 16 == 0,
                GVC_ERR_UNSUPPORTED, "perceiver: need dim %% 256 == 0, dim_head == 64 and 16 / 32 / 48 / 64 latents");
    GVC_REQUIRE((D.dim_head * D.heads) % 128 == 0, GVC_ERR_UNSUPPORTED, "perceiver: heads x dim_head must be a multiple of 128");
    auto* c = new gvc_perceiver();
    c->dm = D;
    c->inner = D.dim_head * D.heads;
    c->ffi = (int)((long long)D.dim * D.ff_mult * 2 / 3);
    c->ffi_p = (c->ffi + 127) & ~127;                 // K of FF2 on the skinny GEMM: eight waves x whole 16-wide k steps
    c->ctx_p = (D.dim_context + 3) & ~3;
    c->has_proj = D.dim_context != D.dim;
    const size_t d = D.dim, in = c->inner, fp = c->ffi_p;
    const size_t per_layer = in * d + 2 * in * d + d * in + 2 * (size_t)c->ffi * d + 2 * c->ffi + d * fp + d +       // as bound
                             3 * in * d + 3 * in * d + d * in + 2 * fp * d + d * fp + 2 * fp + 64;                    // wqkv + FM16 copies + bias
    const size_t total = 2 * (size_t)D.num_latents * d + d + d * c->ctx_p + d + D.depth * per_layer + 64;
    GVC_CHECK_HIP(hipMalloc((void**)&c->wbase, total * sizeof(float)));
    GVC_CHECK_HIP(hipMemset(c->wbase, 0, total * sizeof(float)));
    float* p = c->wbase;
    auto take = [&](size_t n) { float* r = p; p += (n + 3) & ~(size_t)3; return r; };
    c->latents = take((size_t)D.num_latents * d); c->lat_fm = take((size_t)D.num_latents * d); c->gamma = take(d);
    c->proj_w = take(d * c->ctx_p); c->proj_b = take(d);
    c->layers.resize(D.depth);
    // the stacked q | k | v weights of all layers are contiguous: the context GEMM walks them with one batch stride
    float* wqkv_all = take((size_t)D.depth * 3 * in * d);
    for (int l = 0; l < D.depth; ++l) {
        PercLayer& ly = c->layers[l];
        ly.wqkv = wqkv_all + (size_t)l * 3 * in * d;
        ly.to_q = ly.wqkv; ly.to_kv = ly.wqkv + in * d;             // bound straight into the stacked matrix
        ly.to_out = take(d * in);
        ly.ff1_w = take(2 * (size_t)c->ffi * d); ly.ff1_b = take(2 * c->ffi);
        ly.ff2_w = take(d * fp); ly.ff2_b = take(d);
        ly.f_qkv = take(3 * in * d); ly.f_out = take(d * in); ly.f_ff1 = take(2 * fp * d); ly.f_ff2 = take(d * fp); ly.b_ff1 = take(2 * fp);
    }
    c->n_expected = 2 + (c->has_proj ? 2 : 0) + 7 * D.depth;
    const size_t B = D.max_batch, R = (size_t)D.num_latents + D.max_frames, NL = D.num_latents;
    const size_t Mp = ((B * NL + 15) & ~(size_t)15);
    c->work_cap = 4ll << 20;
    GVC_CHECK_HIP(hipMalloc((void**)&c->C, B * R * d * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->kv, B * R * D.depth * 3 * in * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->X, Mp * d * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->o, Mp * in * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->g, Mp * fp * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->xp, B * (size_t)D.max_frames * c->ctx_p * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->work, (size_t)c->work_cap * sizeof(float)));
    GVC_CHECK_HIP(hipMalloc((void**)&c->tmp, 2 * fp * d * sizeof(float)));
    GVC_CHECK_HIP(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    if (getenv("GVC_GRAPHS")) c->use_graph = atoi(getenv("GVC_GRAPHS"));        // 0: eager launches (include/genvc_hip.h, environment switches)
    gemm_init_attributes();
    *out = c;
    return GVC_OK;
}

extern "C" int gvc_perceiver_destroy(gvc_perceiver* c) {
    if (!c) return GVC_OK;
    for (auto& kvp : c->graphs) (void)hipGraphExecDestroy(kvp.second);
    if (c->cap_stream) (void)hipStreamDestroy(c->cap_stream);
    for (void* p : {(void*)c->wbase, (void*)c->C, (void*)c->X, (void*)c->kv, (void*)c->o, (void*)c->g, (void*)c->xp, (void*)c->work,
                    (void*)c->tmp})
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
    if (rc == GVC_OK && known) { c->bound[n] = 1; c->fm_ready = false; }
    return rc;
}

extern "C" int gvc_perceiver_missing_weights(gvc_perceiver* c) {
    return c ? c->n_expected - (int)c->bound.size() : -1;
}

// fragment-major / interleaved copies of the bound weights (once after a bind, on the caller's stream, outside capture)
static int perc_prepare(gvc_perceiver* c, hipStream_t s) {
    if (c->fm_ready) return GVC_OK;
    const int d = c->dm.dim, in = c->inner, fp = c->ffi_p;
    auto fm = [&](const float* src, float* dst, int N, int K) {
        hipLaunchKernelGGL(k_to_fm16, dim3(1024), dim3(256), 0, s, src, dst, N, K);
    };
    fm(c->latents, c->lat_fm, c->dm.num_latents, d);
    for (auto& ly : c->layers) {
        fm(ly.wqkv, ly.f_qkv, 3 * in, d);
        fm(ly.to_out, ly.f_out, d, in);
        hipLaunchKernelGGL(k_interleave_ff1, dim3(2 * fp), dim3(256), 0, s, ly.ff1_w, ly.ff1_b, c->tmp, ly.b_ff1, c->ffi, fp, d);
        fm(c->tmp, ly.f_ff1, 2 * fp, d);
        fm(ly.ff2_w, ly.f_ff2, d, fp);
    }
    GVC_LAUNCH_CHECK();
    if (!c->graphs.empty()) {
        // a re-bind: the graphs were captured with the same buffers, but stay on the safe side -- and a replay may still be in flight on
        // another stream (the conditioning side stream of model_init._CondFuture): wait for the device before destroying them
        GVC_CHECK_HIP(hipDeviceSynchronize());
        for (auto& kvp : c->graphs) (void)hipGraphExecDestroy(kvp.second);
        c->graphs.clear();
        c->graph_used.clear();
    }
    c->fm_ready = true;
    return GVC_OK;
}

// the caller's frames -> the context's staging buffer (the only launch that sees the input pointer: everything behind it works on
// context-owned buffers and replays from a graph).  With a context projection: xp [B * F][ctx_p]; without: rows NL.. of C
static int perc_stage_in(gvc_perceiver* c, const float* x, int B, int F, hipStream_t s) {
    const int d = c->dm.dim, NL = c->dm.num_latents, R = NL + F, dc = c->dm.dim_context;
    if (c->has_proj) {
        hipLaunchKernelGGL(k_pad_cols, dim3(B * F), dim3(128), 0, s, x, c->xp, B * F, dc, c->ctx_p);
        GVC_LAUNCH_CHECK();
        return GVC_OK;
    }
    for (int b = 0; b < B; ++b)
        GVC_CHECK_HIP(hipMemcpyAsync(c->C + ((size_t)b * R + NL) * d, x + (size_t)b * F * d, (size_t)F * d * sizeof(float),
                                     hipMemcpyDeviceToDevice, s));
    return GVC_OK;
}

// `side`: a second stream of the same capture (null: everything in order on s)
static int perc_launch(gvc_perceiver* c, int B, int F, hipStream_t s) {
    const int d = c->dm.dim, in = c->inner, NL = c->dm.num_latents, R = NL + F, depth = c->dm.depth, fp = c->ffi_p;
    const int ldkv = depth * 3 * in;                  // floats per row of KV: [depth][q | k | v]
    int rc;
    GemmArgs G;
    // context frames -> rows NL.. of every batch element of C
    if (c->has_proj) {
        memset(&G, 0, sizeof(G));
        G.A = c->xp; G.lda = c->ctx_p; G.a_batch_stride = (long long)F * c->ctx_p; G.Wt = c->proj_w; G.ldw = c->ctx_p;
        G.C = c->C + (size_t)NL * d; G.ldc = d; G.c_batch_stride = (long long)R * d;
        G.M = F; G.N = d; G.K = c->ctx_p; G.work = c->work; G.e.bias = c->proj_b;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
    }
    // keys / values of the context rows: they do not change across the layers, so they are computed up front by the tiled MFMA GEMM,
    // batch = (batch element, layer); layer l's to_kv rows sit behind its to_q rows in the stacked matrix, its k | v columns behind its
    // q columns in a KV row.  One GEMM over (batch element, layer).  GVC_PERCEIVER_FORK=1 (measured, not the default: 275 vs 266 us per
    // forward -- the branch's GEMM and the latent chain slow each other down): layer 0's first on the main chain, the other layers' as
    // one GEMM on a parallel branch of the graph beside layer 0's latent path, joined in front of layer 1's attention
    auto ctx_kv = [&](int l0, int nl, hipStream_t st) {
        memset(&G, 0, sizeof(G));
        G.batch_inner = nl;
        G.A = c->C + (size_t)NL * d; G.lda = d; G.a_batch_stride = 0; G.a_batch_stride2 = (long long)R * d;
        G.Wt = c->layers[l0].wqkv + (size_t)in * d; G.ldw = d; G.w_batch_stride = (long long)3 * in * d;
        G.C = c->kv + (size_t)NL * ldkv + (size_t)l0 * 3 * in + in; G.ldc = ldkv; G.c_batch_stride = 3 * in; G.c_batch_stride2 = (long long)R * ldkv;
        G.M = F; G.N = 2 * in; G.K = d; G.work = nullptr;             // (no split-K: the branch must not share the work buffer)
        return launch_gemm_cap(G, B * nl, 0, st);
    };
    // (the later layers' context keys / values on a parallel graph branch: measured slower, removed -- profiles/r06_removed_experiments.patch)
    if ((rc = ctx_kv(0, depth, s))) return rc;
    // X = latents, fragment-major, one copy per batch element.  One batch element (the usual call: one reference chunk): no copy --
    // layer 0 reads the parameter itself (A operand of its q | k | v GEMM, residual of its to_out) and writes X
    if (B > 1) {
        hipLaunchKernelGGL(k_rep_latents, dim3(8, B), dim3(256), 0, s, c->lat_fm, c->X, NL * d);
        GVC_LAUNCH_CHECK();
    }

    auto skinny = [&](const float* A_fm, int M, const float* W_fm, int N, int K, float* Cp, int ldc) {
        memset(&G, 0, sizeof(G));
        G.A = A_fm; G.lda = K; G.Wt = W_fm; G.ldw = K; G.C = Cp; G.ldc = ldc; G.M = M; G.N = N; G.K = K; G.work = c->work;
    };
    for (int l = 0; l < depth; ++l) {
        const PercLayer& ly = c->layers[l];
        // q | k | v of the latent rows: one GEMM per batch element (its 32 rows are rows 0..31 of that element's KV block)
        for (int b = 0; b < B; ++b) {
            skinny(B == 1 && l == 0 ? c->lat_fm : c->X + (size_t)b * NL * d, NL, ly.f_qkv, 3 * in, d, c->kv + (size_t)b * R * ldkv + (size_t)l * 3 * in, ldkv);
            if ((rc = launch_gemm_skinny(G, 1, c->work_cap, s))) return rc;
        }
        // cross-attention of the 32 latent queries over latents + context, output fragment-major [B * 32][inner]
        {
            const float* qb = c->kv + (size_t)l * 3 * in;
            hipLaunchKernelGGL((k_attn64_mfma<false, 16>), dim3(NL / 16, c->dm.heads, B), dim3(1024), 0, s, qb, qb + in, qb + 2 * in, (long long)ldkv,
                               (long long)R * ldkv, NL, R, c->o, NL, in, 1.0f / sqrtf((float)c->dm.dim_head), 1, (const int32_t*)nullptr);
            GVC_LAUNCH_CHECK();
        }
        // the latent path below takes the batch in groups of at most 128 rows (the skinny GEMM's eight M tiles)
        const int gb = 128 / NL;                            // batch elements per group: 8 / 4 / 2 / 2 at 16 / 32 / 48 / 64 latents
        for (int b0 = 0; b0 < B; b0 += gb) {
            const int M = (B - b0 < gb ? B - b0 : gb) * NL;
            float* Xg = c->X + (size_t)b0 * NL * d;
            // latents += o @ to_out^T
            skinny(c->o + (size_t)b0 * NL * in, M, ly.f_out, d, in, Xg, d);
            G.e.c_fm16 = 1; G.e.resid = B == 1 && l == 0 ? c->lat_fm : Xg; G.e.resid_fm16 = 1; G.e.ldr = d;
            if ((rc = launch_gemm_skinny(G, 1, c->work_cap, s))) return rc;
            // feed-forward: Linear -> GEGLU (in the epilogue) -> Linear, residual
            skinny(Xg, M, ly.f_ff1, 2 * fp, d, c->g + (size_t)b0 * NL * fp, fp);
            G.e.bias = ly.b_ff1; G.e.geglu = 1;
            if ((rc = launch_gemm_skinny(G, 1, c->work_cap, s))) return rc;
            skinny(c->g + (size_t)b0 * NL * fp, M, ly.f_ff2, d, fp, Xg, d);
            G.e.bias = ly.ff2_b; G.e.c_fm16 = 1; G.e.resid = Xg; G.e.resid_fm16 = 1; G.e.ldr = d;
            if ((rc = launch_gemm_skinny(G, 1, c->work_cap, s))) return rc;
        }
    }
    return GVC_OK;
}

extern "C" int gvc_perceiver_forward(gvc_perceiver* c, const float* x, int32_t B, int32_t F, float* out, gvc_stream sv) {
    GVC_REQUIRE(c && x && out, GVC_ERR_ARG, "gvc_perceiver_forward: null argument");
    GVC_REQUIRE(gvc_perceiver_missing_weights(c) == 0, GVC_ERR_STATE, "%d Perceiver weight tensors are not bound",
                gvc_perceiver_missing_weights(c));
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_batch && F >= 1 && F <= c->dm.max_frames, GVC_ERR_ARG,
                "perceiver: B=%d F=%d outside capacity (%d, %d)", B, F, c->dm.max_batch, c->dm.max_frames);
    hipStream_t s = (hipStream_t)sv;
    int rc;
    if ((rc = perc_prepare(c, s))) return rc;
    if ((rc = perc_stage_in(c, x, B, F, s))) return rc;
    if (!c->use_graph) {
        if ((rc = perc_launch(c, B, F, s))) return rc;
    } else {
        // the body works on context-owned buffers only: one graph per (B, F), whatever the caller's pointers
        const long long key = (long long)B * 100000 + F;
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            if (c->graphs.size() >= 32) {
                // a long-running caller with ever new reference lengths: keep the cache bounded by evicting the LEAST RECENTLY USED shape (a
                // caller that alternates between a few lengths keeps its graphs); the evicted graph may still be replaying on another stream
                long long victim = c->graphs.begin()->first;
                for (auto& kvp : c->graph_used)
                    if (kvp.second < c->graph_used[victim]) victim = kvp.first;
                GVC_CHECK_HIP(hipDeviceSynchronize());
                (void)hipGraphExecDestroy(c->graphs[victim]);
                c->graphs.erase(victim);
                c->graph_used.erase(victim);
            }
            hipGraph_t graph = nullptr;
            GVC_CHECK_HIP(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
            rc = perc_launch(c, B, F, c->cap_stream);
            hipError_t e = hipStreamEndCapture(c->cap_stream, &graph);
            if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
            GVC_CHECK_HIP(e);
            hipGraphExec_t ge = nullptr;
            e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
            (void)hipGraphDestroy(graph);
            GVC_CHECK_HIP(e);
            it = c->graphs.emplace(key, ge).first;
        }
        c->graph_used[key] = ++c->tick;
        GVC_CHECK_HIP(hipGraphLaunch(it->second, s));
    }
    hipLaunchKernelGGL(k_rmsnorm_rows_fm16, dim3(cdiv(B * c->dm.num_latents, 4)), dim3(256), 0, s, c->X, out, B * c->dm.num_latents, c->dm.dim,
                       c->gamma);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}
