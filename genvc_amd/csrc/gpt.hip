// GPT context of libgenvc_hip: weight binding/repack, KV cache, prefill, decode step, latent re-pass,
// graph-replayed generation loop.  Reference seams: layers/gpt.py, layers/gpt_inference.py (see
// include/genvc_hip.h for the line-level mapping).
#include <stdlib.h>

#include <map>
#include <string>
#include <vector>

#include "gemm.h"
#include <algorithm>
#include "gpt_kernels.h"
#include "persist_kernel.h"
#include "persist_rows.h"
#include "persist_rows_b16.h"
#include "sampler.h"

namespace gvc {

// ---------------------------------------------------------------------------------------------
// row kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_ln_rows(const float* src, float* dst, int rows, int d, const float* w1, const float* b1,
                          const float* w2, const float* b2, int dst_fm16) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* x = src + (size_t)row * d;
    float* y = dst + (size_t)row * d;
    const float inv_d = 1.0f / (float)d;
    for (int pass = 0; pass < (w2 ? 2 : 1); ++pass) {
        const float* in = pass == 0 ? x : y;
        const float* gw = pass == 0 ? w1 : w2;
        const float* gb = pass == 0 ? b1 : b2;
        float s = 0.f;
        for (int k = lane * 4; k < d; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(in + k);
            s += (v.x + v.y) + (v.z + v.w);
        }
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
        for (int k = lane * 4; k < d; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(in + k);
            const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
        for (int k = lane * 4; k < d; k += 256) {
            float4 v = *reinterpret_cast<const float4*>(in + k);
            const float4 g = *reinterpret_cast<const float4*>(gw + k);
            const float4 c = *reinterpret_cast<const float4*>(gb + k);
            v.x = (v.x - mean) * rstd * g.x + c.x; v.y = (v.y - mean) * rstd * g.y + c.y;
            v.z = (v.z - mean) * rstd * g.z + c.z; v.w = (v.w - mean) * rstd * g.w + c.w;
            if (dst_fm16 && !w2) *reinterpret_cast<float4*>(dst + fm16_index(row, k, d)) = v;
            else *reinterpret_cast<float4*>(y + k) = v;
        }
    }
}

__global__ void k_embed_rows(float* x, const float* prefix_emb, int B, int T, int P, int d, const float* mel_emb,
                             const float* mel_pos, const int32_t* codes, int n, int start_tok, int stop_tok, int t_off) {
    // rows [t_off, T) of every stream (t_off > 0: the leading rows are already in the KV cache)
    const int row = blockIdx.x;
    const int Tn = T - t_off;
    const int b = row / Tn, t = row - b * Tn + t_off;
    float* dst = x + (size_t)row * d;
    if (t < P) {
        const float* src = prefix_emb + ((size_t)b * P + t) * d;
        for (int k = threadIdx.x * 4; k < d; k += blockDim.x * 4)
            *reinterpret_cast<float4*>(dst + k) = *reinterpret_cast<const float4*>(src + k);
    } else {
        const int i = t - P;
        const int tok = i == 0 ? start_tok : (i <= n ? codes[(size_t)b * n + i - 1] : stop_tok);
        const float* e = mel_emb + (size_t)tok * d;
        const float* p = mel_pos + (size_t)i * d;
        for (int k = threadIdx.x * 4; k < d; k += blockDim.x * 4) {
            const float4 a = *reinterpret_cast<const float4*>(e + k);
            const float4 c = *reinterpret_cast<const float4*>(p + k);
            *reinterpret_cast<float4*>(dst + k) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
        }
    }
}

__global__ void k_prefix_rows(float* out, const float* cond, int n_cond, const int32_t* codes, int B, int Tc,
                              int d, const float* text_emb, const float* text_pos, int start_text,
                              int stop_text) {
    const int P = n_cond + Tc + 2;
    const int row = blockIdx.x;
    const int b = row / P, t = row - b * P;
    float* dst = out + (size_t)row * d;
    if (t < n_cond) {
        const float* src = cond + ((size_t)b * n_cond + t) * d;
        for (int k = threadIdx.x * 4; k < d; k += blockDim.x * 4)
            *reinterpret_cast<float4*>(dst + k) = *reinterpret_cast<const float4*>(src + k);
    } else {
        const int i = t - n_cond;
        const int id = i == 0 ? start_text : (i <= Tc ? codes[(size_t)b * Tc + i - 1] : stop_text);
        const float* e = text_emb + (size_t)id * d;
        const float* p = text_pos + (size_t)i * d;
        for (int k = threadIdx.x * 4; k < d; k += blockDim.x * 4) {
            const float4 a = *reinterpret_cast<const float4*>(e + k);
            const float4 c = *reinterpret_cast<const float4*>(p + k);
            *reinterpret_cast<float4*>(dst + k) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
        }
    }
}

// one workgroup per stream: x[b] = mel_embedding[tok[b]] + mel_pos_embedding[mel_pos[slot[b]]]  (gpt_inference.py:92-96)
__global__ void k_embed_decode_rows(float* x, const int32_t* tok, const int32_t* slots, GptState st, const float* mel_emb,
                                    const float* mel_pos, int d, int vocab) {
    const int b = blockIdx.x;
    // (clamped: after a step that produced garbage -- a timed-out hand-off -- the sampler may hand over any id)
    const float* e = mel_emb + (size_t)min(max(tok[b], 0), vocab - 1) * d;
    const float* p = mel_pos + (size_t)st.mel_pos[slots[b]] * d;
    float* dst = x + (size_t)b * d;
    for (int k = threadIdx.x * 4; k < d; k += blockDim.x * 4) {
        const float4 a = *reinterpret_cast<const float4*>(e + k);
        const float4 c = *reinterpret_cast<const float4*>(p + k);
        *reinterpret_cast<float4*>(dst + k) = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
    }
}

// The generation loop's staging buffers (next-step logits and latent) are indexed by the position of a stream in the CALL;
// between calls they are parked per SLOT, so that consecutive calls may batch different sets of streams (streaming.py)
__global__ void k_stage_rows(float* stage, float* store, const int32_t* slots, int n, int to_store) {
    const int b = blockIdx.x;
    float* a = stage + (size_t)b * n;
    float* s = store + (size_t)slots[b] * n;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        if (to_store) s[i] = a[i];
        else a[i] = s[i];
    }
}

__global__ void k_set_state(GptState st, const int32_t* slots, int B, int seq_len, int mel_pos) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) {
        st.seq_len[slots[b]] = seq_len;
        st.mel_pos[slots[b]] = mel_pos;
    }
}

__global__ void k_gather_rows(const float* src, float* dst, int B, int T, int off, int n, int d) {
    const int row = blockIdx.x;
    const int b = row / n, i = row - b * n;
    const float* s = src + ((size_t)b * T + off + i) * d;
    float* o = dst + (size_t)row * d;
    for (int k = threadIdx.x * 4; k < d; k += blockDim.x * 4)
        *reinterpret_cast<float4*>(o + k) = *reinterpret_cast<const float4*>(s + k);
}

__global__ void k_round_bf16(float* w, unsigned short* out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned u = __float_as_uint(w[i]);
        const unsigned r = u + 0x7fffu + ((u >> 16) & 1u);            // round to nearest even (finite inputs)
        out[i] = (unsigned short)(r >> 16);
        w[i] = __uint_as_float(r & 0xffff0000u);
    }
}

__global__ void k_transpose(const float* src, float* dst, int K, int N) {
    __shared__ float tile[32][33];
    const int k0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int k = k0 + r, n = n0 + threadIdx.x;
        if (k < K && n < N) tile[r][threadIdx.x] = src[(size_t)k * N + n];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int n = n0 + r, k = k0 + threadIdx.x;
        if (k < K && n < N) dst[(size_t)n * K + k] = tile[threadIdx.x][r];
    }
}

// parameters of one gvc_gpt_generate call, resident on the device so the captured step graph
// is independent of them
struct GenCall {
    SampleCall sc;
    int32_t slots[64];
};

// Start / end of a gvc_gpt_generate call in ONE launch each (they used to be a memset, k_set_gen_call and two k_stage_rows before the
// step graphs and two k_stage_rows after them: six 4-5 us operations around every group of eight streaming steps).
// begin: workgroup b < B un-parks the next-step logits and latent of stream b; workgroup B resets the step counter and stores the
// call parameters.  end: workgroup b parks them again.
__global__ void k_gen_begin(GenCall* dst, SampleCall sc, const int32_t* slots, int B, int32_t* step_ctr, float* logits, float* slot_logits,
                            int vocab, float* latent, float* slot_latent, int d) {
    const int b = blockIdx.x;
    if (b == B) {
        if (threadIdx.x == 0) { dst->sc = sc; *step_ctr = 0; }
        if (threadIdx.x < B) dst->slots[threadIdx.x] = slots[threadIdx.x];
        return;
    }
    const size_t sl = (size_t)slots[b];
    for (int i = threadIdx.x; i < vocab; i += blockDim.x) logits[(size_t)b * vocab + i] = slot_logits[sl * vocab + i];
    for (int i = threadIdx.x; i < d; i += blockDim.x) latent[(size_t)b * d + i] = slot_latent[sl * d + i];
}

__global__ void k_gen_end(const int32_t* slots, const float* logits, float* slot_logits, int vocab, const float* latent, float* slot_latent,
                          int d) {
    const int b = blockIdx.x;
    const size_t sl = (size_t)slots[b];
    for (int i = threadIdx.x; i < vocab; i += blockDim.x) slot_logits[sl * vocab + i] = logits[(size_t)b * vocab + i];
    for (int i = threadIdx.x; i < d; i += blockDim.x) slot_latent[sl * d + i] = latent[(size_t)b * d + i];
}

}  // namespace gvc

using namespace gvc;

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct GptLayer {
    float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc_w, *fc_b, *p2_w, *p2_b;
    float *qkv_f = nullptr, *proj_f = nullptr, *fc_f = nullptr, *p2_f = nullptr;   // FM16 copies for the skinny prefill GEMM
    unsigned short *qkv_h = nullptr, *proj_h = nullptr, *fc_h = nullptr, *p2_h = nullptr;   // bf16 copies (bf16-weights contexts)
};

struct gvc_gpt {
    gvc_gpt_dims dm;
    int hd = 0, n_cu = 256;
    float* wbase = nullptr;           // one allocation for all weights
    float* wfm = nullptr;             // FM16 copies of the four per-layer matrices (prefill path)
    unsigned short* wh = nullptr;     // bf16 copies of the streamed matrices (weight_dtype = 1)
    unsigned short* head_h = nullptr;
    int bf16 = 0;
    float *mel_emb, *mel_pos, *text_emb, *text_pos, *lnf_w, *lnf_b, *fn_w, *fn_b, *head_w, *head_b;
    std::vector<GptLayer> layers;
    std::map<std::string, int> bound;  // name -> 1 once bound
    int n_expected = 0;
    float* kv = nullptr;              // [L][2][slots][H][max_seq][hd], fp32 or (kv_bf16) bf16 elements
    int kv_bf16 = 0;
    int act_bf16 = 0;                 // weight_dtype 3: the one-launch rows step rounds the activations that cross its hand-offs (persist_rows_b16.h)
    size_t kv_layer_stride = 0;       // floats per (layer, k|v)
    float *x = nullptr, *a = nullptr, *q = nullptr, *h = nullptr, *part = nullptr, *work = nullptr;
    long long work_cap = 0;
    float *x2 = nullptr, *part2 = nullptr;        // fused attention path: second residual buffer, per-head partials
    int fuse_decode = 1;                          // the fused attention + c_proj launch (k_attn_proj) serves short one-stream contexts
    int* seam_err_host = nullptr;                 // pinned, device-visible: a timed-out hand-off of the one-launch step / a full KV cache
    int* seam_err_dev = nullptr;
    int skinny_prefill = 1;                       // fragment-major weight copies + skinny / strip GEMMs (0 only when their allocation failed)
    int strip_prefill = 1;                        // more than 128 rows: strip GEMM
    int fuse_ln_rows = 8;                         // the LayerNorm-prologue GEMM serves up to this many rows (9+: separate launches win)
    int fuse_ln = 1;
    float* xalt = nullptr;                        // second residual buffer of that path [16][d]
    int rows_decode_min = 5;                      // batches of at least this many streams decode on the MFMA rows path (0: never);
                                                  // measured crossover: B=4 925 (GEMV) vs 975 us (rows), B=5 1242 vs 996 us
    float *logits = nullptr, *latent = nullptr;             // staging of the generation loop, indexed by position in the call
    float *slot_logits = nullptr, *slot_latent = nullptr;   // ... parked per slot between calls   // generate(): [slots][V], [slots][d]
    int32_t* state = nullptr;         // seq_len[slots], mel_pos[slots], tok[slots], step
    GptState st;
    int32_t *tok_buf = nullptr, *step_ctr = nullptr;
    GenCall* gen_call = nullptr;
    hipStream_t cap_stream = nullptr;
    std::map<int, hipGraphExec_t> graphs;   // 2*B + fused -> step graph
    int prof_only = -1;               // gvc_gpt_time_kernel(): launch only this kernel class
    int prof_skip_one = -1;           // ... or every class except this one
    unsigned long long* dbg = nullptr; // GVC_DEBUG_STAMPS: [launch][8] in-kernel timestamps of the eager decode step
    int dbg_n = 0;
    // one-launch decode step (persist_kernel.h), one stream
    int persist = 1;                  // GVC_PERSIST=0: launch-per-phase decode
    PersistLayer* p_layers = nullptr; // device table of per-layer pointers
    pu64* p_gran = nullptr;           // granule buffers of the in-kernel hand-offs
    unsigned* p_epoch = nullptr;      // [0] step epoch, [1] arrival counter
    unsigned long long* p_dbg = nullptr;   // GVC_PERSIST_STAMPS: wall-clock stamps of workgroup 0
    int p_ring_slots = 0, p_ascr = 0, p_hvec = 0;
    size_t p_lds = 0;
    int last_variant = 0;             // decode variant of the last gvc_gpt_generate call (gvc_gpt_decode_variant)
    int fallbacks = 0;                // hand-off timeouts that switched the one-launch steps off (gvc_gpt_health)
    int persist_cfg = 1;              // what GVC_PERSIST allowed at create (gvc_gpt_rearm restores it)
    // one-launch block stack for 2..16 rows (persist_rows.h): batched decode steps, cached chunk prefills
    int persist_rows = 1;             // GVC_PERSIST_ROWS=0: those calls keep the launch-per-phase rows path
    int persist_rows_min = 2;         // smallest row count the one-launch rows step serves
    long long r_launches = 0;         // one-launch rows steps issued (gvc_gpt_rows_step_launches; graph replays count once per capture)
    int r_ready = 0;                  // 0 not prepared yet, 1 ready, -1 unavailable on this device / for these dims
    RowsLayer* r_layers = nullptr;
    float* r_wpack = nullptr;         // packed weights [layer][256][192 KiB]
    float* r_bufs = nullptr;          // hand-off buffers, two parities
    float* r_lnfold = nullptr;        // per layer: S[3d] | C[3d] of LN1 -> c_attn and S[4d] | C[4d] of LN2 -> c_fc (k_rows_ln_fold)
    unsigned long long* r_dbg = nullptr;   // GVC_PERSIST_STAMPS
    std::vector<char> r_dirty;         // per layer: a matrix was re-bound after the pack was built (gvc_gpt_bind_weight) -> repack before use
    size_t r_lds = 0;
    int rows_keys_hint = 0;           // cached positions the longest stream of the running call reaches (set by the entry points)
    int p_xl = 1;                     // the one-stream step keeps the MLP's hidden units inside their XCD (GVC_PERSIST_XCD=0: never); cleared by
                                      // persist_prepare when the device does not deal a 256-workgroup grid as 8 XCDs x 32
    int arch_xcc = 0;                 // gcnArchName is gfx94x / gfx950: HW_REG_XCC_ID exists and means what the XL layout assumes
    long long lazy_inits = 0;         // allocations / device-wide syncs / graph captures done INSIDE a data-path call (gvc_gpt_lazy_inits)
    int in_warmup = 0;                // ... gvc_gpt_warmup's own do not count
};

static int gemv_init();

static int alloc_f(float** p, size_t n) {
    GVC_CHECK_HIP(hipMalloc((void**)p, n * sizeof(float)));
    return GVC_OK;
}

extern "C" int gvc_gpt_create(const gvc_gpt_dims* dims, gvc_gpt** out) {
    GVC_REQUIRE(dims && out, GVC_ERR_ARG, "gvc_gpt_create: null argument");
    const gvc_gpt_dims& D = *dims;
    GVC_REQUIRE(D.n_layer > 0 && D.n_head > 0 && D.d_model % D.n_head == 0, GVC_ERR_ARG, "bad GPT dims");
    const int hd = D.d_model / D.n_head;
    // the decode GEMVs keep a whole residual row in one wave (d / 256 float4 per lane): d = 256, 512, 768 or 1024
    // widths above 1024 (the reference takes every dimension from the checkpoint's config: inference/model_init.py:11-12,
    // configs/genVC_configs.py:127-139) run on the GEMM paths only -- prefill and decode alike as rows through the skinny / strip / tiled
    // GEMMs with separate LayerNorm launches, the head as a LayerNorm launch + a GEMM; the GEMV decode kernels and the one-launch steps
    // keep a residual row in one wave and stay at <= 1024
    GVC_REQUIRE(D.d_model % 256 == 0 && D.d_model >= 256 && D.d_model <= 2048, GVC_ERR_UNSUPPORTED,
                "d_model %d unsupported (a multiple of 256 up to 2048)", D.d_model);
    GVC_REQUIRE(hd == 64 || hd == 128 || hd == 256, GVC_ERR_UNSUPPORTED, "head_dim %d unsupported (64, 128 or 256)", hd);
    GVC_REQUIRE(D.max_slots >= 1 && D.max_slots <= 64, GVC_ERR_ARG, "max_slots must be in [1,64]");
    GVC_REQUIRE(D.max_rows >= D.max_slots, GVC_ERR_ARG, "max_rows must be >= max_slots");
    auto* c = new gvc_gpt();
    c->dm = D;
    c->hd = hd;
    int dev = 0;
    hipDeviceProp_t prop;
    GVC_CHECK_HIP(hipGetDevice(&dev));
    GVC_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
    c->n_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    c->arch_xcc = strncmp(prop.gcnArchName, "gfx950", 6) == 0 || strncmp(prop.gcnArchName, "gfx94", 5) == 0;
    c->p_xl = getenv("GVC_PERSIST_XCD") ? atoi(getenv("GVC_PERSIST_XCD")) : 1;

    const size_t d = D.d_model, L = D.n_layer, V = D.vocab;
    const size_t per_layer = 2 * d + 3 * d * d + 3 * d + d * d + d + 2 * d + 4 * d * d + 4 * d + 4 * d * d + d;
    const size_t total = V * d + (size_t)D.max_mel_pos * d + (size_t)D.n_text * d + (size_t)D.max_text_pos * d +
                         4 * d + V * d + V + 64 + L * per_layer;
    int rc = alloc_f(&c->wbase, total + 64);
    if (rc) { delete c; return rc; }
    float* p = c->wbase;
    auto take = [&](size_t n) { float* r = p; p += (n + 3) & ~(size_t)3; return r; };
    c->mel_emb = take(V * d); c->mel_pos = take((size_t)D.max_mel_pos * d);
    c->text_emb = take((size_t)D.n_text * d); c->text_pos = take((size_t)D.max_text_pos * d);
    c->lnf_w = take(d); c->lnf_b = take(d); c->fn_w = take(d); c->fn_b = take(d);
    c->head_w = take(V * d); c->head_b = take(V);
    c->layers.resize(L);
    for (auto& ly : c->layers) {
        ly.ln1_w = take(d); ly.ln1_b = take(d); ly.qkv_w = take(3 * d * d); ly.qkv_b = take(3 * d);
        ly.proj_w = take(d * d); ly.proj_b = take(d); ly.ln2_w = take(d); ly.ln2_b = take(d);
        ly.fc_w = take(4 * d * d); ly.fc_b = take(4 * d); ly.p2_w = take(4 * d * d); ly.p2_b = take(d);
    }
    c->n_expected = 10 + 12 * (int)L;
    gemm_init_attributes();
    GVC_CHECK_HIP(hipMalloc((void**)&c->xalt, (size_t)16 * d * sizeof(float)));
    if (getenv("GVC_ROWS_DECODE_MIN")) c->rows_decode_min = atoi(getenv("GVC_ROWS_DECODE_MIN"));
    c->bf16 = D.weight_dtype >= 1;
    c->kv_bf16 = D.weight_dtype >= 2;
    c->act_bf16 = D.weight_dtype == 3;
    GVC_REQUIRE(D.weight_dtype >= 0 && D.weight_dtype <= 3, GVC_ERR_ARG,
                "weight_dtype must be 0 (fp32), 1 (bf16 weights), 2 (bf16 weights + KV cache) or 3 (2 + bf16 activations on the one-launch rows step)");
    if (c->bf16) {
        GVC_CHECK_HIP(hipMalloc((void**)&c->wh, (L * 12 * d * d + V * d + 64) * sizeof(unsigned short)));
        unsigned short* hq = c->wh;
        for (auto& ly : c->layers) {
            ly.qkv_h = hq; hq += 3 * d * d; ly.proj_h = hq; hq += d * d; ly.fc_h = hq; hq += 4 * d * d; ly.p2_h = hq; hq += 4 * d * d;
        }
        c->head_h = hq;
    }
    if (c->skinny_prefill) {
        // bf16-weights contexts keep the FM16 copies in bf16 too (half the bytes per pass; widened in registers)
        const size_t es = c->bf16 ? 2 : 4;
        if ((rc = alloc_f(&c->wfm, (L * 12 * d * d * es + 3) / 4))) { gvc_gpt_destroy(c); return rc; }
        char* f = reinterpret_cast<char*>(c->wfm);
        for (auto& ly : c->layers) {
            ly.qkv_f = reinterpret_cast<float*>(f); f += 3 * d * d * es; ly.proj_f = reinterpret_cast<float*>(f); f += d * d * es;
            ly.fc_f = reinterpret_cast<float*>(f); f += 4 * d * d * es; ly.p2_f = reinterpret_cast<float*>(f); f += 4 * d * d * es;
        }
    }

    c->kv_layer_stride = (size_t)D.max_slots * D.n_head * D.max_seq * hd;
    const size_t rows = ((size_t)D.max_rows + 15) & ~(size_t)15;       // fragment-major activations come in 16-row tiles
    c->work_cap = 8ll << 20;
    if ((rc = alloc_f(&c->kv, (c->kv_bf16 ? 1 : 2) * L * c->kv_layer_stride)) || (rc = alloc_f(&c->x, rows * d)) ||
        (rc = alloc_f(&c->a, rows * d)) || (rc = alloc_f(&c->q, rows * d)) || (rc = alloc_f(&c->h, rows * 4 * d)) ||
        (rc = alloc_f(&c->part, (size_t)D.max_slots * D.n_head * kAttnChunks * (hd + 4))) ||
        (rc = alloc_f(&c->work, (size_t)c->work_cap)) || (rc = alloc_f(&c->logits, (size_t)D.max_slots * V)) ||
        (rc = alloc_f(&c->latent, (size_t)D.max_slots * d)) || (rc = alloc_f(&c->x2, (size_t)D.max_slots * d)) ||
        (rc = alloc_f(&c->slot_logits, (size_t)D.max_slots * V)) || (rc = alloc_f(&c->slot_latent, (size_t)D.max_slots * d)) ||
        (rc = alloc_f(&c->part2, (size_t)D.max_slots * D.n_head * d))) {
        gvc_gpt_destroy(c);
        return rc;
    }
    const size_t nstate = 3 * (size_t)D.max_slots + 4;
    GVC_CHECK_HIP(hipMalloc((void**)&c->state, nstate * sizeof(int32_t)));
    GVC_CHECK_HIP(hipMemset(c->state, 0, nstate * sizeof(int32_t)));
    c->st.seq_len = c->state;
    c->st.mel_pos = c->state + D.max_slots;
    c->tok_buf = c->state + 2 * D.max_slots;
    c->step_ctr = c->state + 3 * D.max_slots;
    GVC_CHECK_HIP(hipMalloc((void**)&c->gen_call, sizeof(GenCall)));
    GVC_CHECK_HIP(hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking));
    if ((rc = gemv_init())) { gvc_gpt_destroy(c); return rc; }
    GVC_CHECK_HIP(hipHostMalloc((void**)&c->seam_err_host, sizeof(int), hipHostMallocMapped));
    *c->seam_err_host = 0;
    GVC_CHECK_HIP(hipHostGetDevicePointer((void**)&c->seam_err_dev, c->seam_err_host, 0));
    if (getenv("GVC_PERSIST")) c->persist = atoi(getenv("GVC_PERSIST"));
    c->persist_cfg = c->persist;
    if (getenv("GVC_PERSIST_ROWS")) c->persist_rows = atoi(getenv("GVC_PERSIST_ROWS"));
    if (getenv("GVC_DEBUG_STAMPS")) {
        GVC_CHECK_HIP(hipMalloc((void**)&c->dbg, 4096 * 8 * sizeof(unsigned long long)));
        GVC_CHECK_HIP(hipMemset(c->dbg, 0, 4096 * 8 * sizeof(unsigned long long)));
    }
    *out = c;
    return GVC_OK;
}

extern "C" int gvc_gpt_destroy(gvc_gpt* c) {
    if (!c) return GVC_OK;
    for (auto& kvp : c->graphs) hipGraphExecDestroy(kvp.second);
    if (c->cap_stream) hipStreamDestroy(c->cap_stream);
    if (c->xalt) hipFree(c->xalt);
    if (c->seam_err_host) hipHostFree(c->seam_err_host);
    for (void* p : {(void*)c->p_layers, (void*)c->p_gran, (void*)c->p_epoch, (void*)c->p_dbg, (void*)c->r_layers, (void*)c->r_wpack,
                    (void*)c->r_bufs, (void*)c->r_dbg, (void*)c->r_lnfold})
        if (p) hipFree(p);
    for (void* p : {(void*)c->wbase, (void*)c->wfm, (void*)c->wh, (void*)c->kv, (void*)c->x, (void*)c->a, (void*)c->q, (void*)c->h,
                    (void*)c->part, (void*)c->work, (void*)c->logits, (void*)c->latent, (void*)c->slot_logits, (void*)c->slot_latent, (void*)c->state, (void*)c->x2, (void*)c->part2,
                    (void*)c->gen_call})
        if (p) hipFree(p);
    delete c;
    return GVC_OK;
}

static int copy_w(float* dst, const float* src, int64_t numel, int64_t expect, const char* name, hipStream_t s) {
    GVC_REQUIRE(numel == expect, GVC_ERR_ARG, "%s: expected %lld elements, got %lld", name, (long long)expect,
                (long long)numel);
    GVC_CHECK_HIP(hipMemcpyAsync(dst, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
    return GVC_OK;
}

// HF Conv1D weight [K][N] -> row-per-output [N][K]
static int transpose_w(float* dst, const float* src, int64_t numel, int K, int N, const char* name, hipStream_t s,
                       float* dst_fm16 = nullptr, unsigned short* dst_bf16 = nullptr) {
    GVC_REQUIRE(numel == (int64_t)K * N, GVC_ERR_ARG, "%s: expected %lld elements, got %lld", name,
                (long long)K * N, (long long)numel);
    hipLaunchKernelGGL(k_transpose, dim3(cdiv(N, 32), cdiv(K, 32)), dim3(32, 8), 0, s, src, dst, K, N);
    GVC_LAUNCH_CHECK();
    if (dst_bf16) {      // bf16-weights context: every path uses the SAME rounded values (fp32 copy rounded in place)
        hipLaunchKernelGGL(k_round_bf16, dim3(1024), dim3(256), 0, s, dst, dst_bf16, (size_t)K * N);
        GVC_LAUNCH_CHECK();
    }
    if (dst_fm16) {      // second copy in MFMA fragment order for the skinny prefill GEMM (bf16 elements in a bf16-weights context)
        if (dst_bf16) hipLaunchKernelGGL(k_to_fm16_bf16, dim3(1024), dim3(256), 0, s, dst, reinterpret_cast<unsigned short*>(dst_fm16), N, K);
        else hipLaunchKernelGGL(k_to_fm16, dim3(1024), dim3(256), 0, s, dst, dst_fm16, N, K);
        GVC_LAUNCH_CHECK();
    }
    return GVC_OK;
}

static void mark_rows_pack_dirty(gvc_gpt* c, int li);
extern "C" int gvc_gpt_bind_weight(gvc_gpt* c, const char* name, const float* src, int64_t numel, gvc_stream sv) {
    GVC_REQUIRE(c && name && src, GVC_ERR_ARG, "gvc_gpt_bind_weight: null argument");
    hipStream_t s = (hipStream_t)sv;
    const int64_t d = c->dm.d_model, V = c->dm.vocab;
    std::string n(name);
    int rc = GVC_OK;
    bool known = true;
    if (n == "mel_embedding.weight") rc = copy_w(c->mel_emb, src, numel, V * d, name, s);
    else if (n == "mel_pos_embedding.emb.weight") rc = copy_w(c->mel_pos, src, numel, c->dm.max_mel_pos * d, name, s);
    else if (n == "text_embedding.weight") rc = copy_w(c->text_emb, src, numel, c->dm.n_text * d, name, s);
    else if (n == "text_pos_embedding.emb.weight") rc = copy_w(c->text_pos, src, numel, c->dm.max_text_pos * d, name, s);
    else if (n == "gpt.ln_f.weight") rc = copy_w(c->lnf_w, src, numel, d, name, s);
    else if (n == "gpt.ln_f.bias") rc = copy_w(c->lnf_b, src, numel, d, name, s);
    else if (n == "final_norm.weight") rc = copy_w(c->fn_w, src, numel, d, name, s);
    else if (n == "final_norm.bias") rc = copy_w(c->fn_b, src, numel, d, name, s);
    else if (n == "mel_head.weight") {
        rc = copy_w(c->head_w, src, numel, V * d, name, s);
        if (rc == GVC_OK && c->bf16) {
            hipLaunchKernelGGL(k_round_bf16, dim3(1024), dim3(256), 0, s, c->head_w, c->head_h, (size_t)(V * d));
            GVC_LAUNCH_CHECK();
        }
    }
    else if (n == "mel_head.bias") rc = copy_w(c->head_b, src, numel, V, name, s);
    else if (n.rfind("gpt.h.", 0) == 0) {
        const size_t dot = n.find('.', 6);
        GVC_REQUIRE(dot != std::string::npos, GVC_ERR_ARG, "malformed weight name %s", name);
        const int li = atoi(n.substr(6, dot - 6).c_str());
        const std::string rest = n.substr(dot + 1);
        GVC_REQUIRE(li >= 0 && li < c->dm.n_layer, GVC_ERR_ARG, "%s: layer out of range", name);
        GptLayer& ly = c->layers[li];
        if (rest == "ln_1.weight") { rc = copy_w(ly.ln1_w, src, numel, d, name, s); mark_rows_pack_dirty(c, li); }   // (the rows step folds the
        else if (rest == "ln_1.bias") { rc = copy_w(ly.ln1_b, src, numel, d, name, s); mark_rows_pack_dirty(c, li); }    //  LayerNorm into per-row constants)
        else if (rest == "attn.c_attn.weight") { rc = transpose_w(ly.qkv_w, src, numel, d, 3 * d, name, s, ly.qkv_f, ly.qkv_h); mark_rows_pack_dirty(c, li); }
        else if (rest == "attn.c_attn.bias") { rc = copy_w(ly.qkv_b, src, numel, 3 * d, name, s); mark_rows_pack_dirty(c, li); }
        else if (rest == "attn.c_proj.weight") { rc = transpose_w(ly.proj_w, src, numel, d, d, name, s, ly.proj_f, ly.proj_h); mark_rows_pack_dirty(c, li); }
        else if (rest == "attn.c_proj.bias") rc = copy_w(ly.proj_b, src, numel, d, name, s);
        else if (rest == "ln_2.weight") { rc = copy_w(ly.ln2_w, src, numel, d, name, s); mark_rows_pack_dirty(c, li); }
        else if (rest == "ln_2.bias") { rc = copy_w(ly.ln2_b, src, numel, d, name, s); mark_rows_pack_dirty(c, li); }
        else if (rest == "mlp.c_fc.weight") { rc = transpose_w(ly.fc_w, src, numel, d, 4 * d, name, s, ly.fc_f, ly.fc_h); mark_rows_pack_dirty(c, li); }
        else if (rest == "mlp.c_fc.bias") { rc = copy_w(ly.fc_b, src, numel, 4 * d, name, s); mark_rows_pack_dirty(c, li); }
        else if (rest == "mlp.c_proj.weight") { rc = transpose_w(ly.p2_w, src, numel, 4 * d, d, name, s, ly.p2_f, ly.p2_h); mark_rows_pack_dirty(c, li); }
        else if (rest == "mlp.c_proj.bias") rc = copy_w(ly.p2_b, src, numel, d, name, s);
        else known = false;   // attn.bias / attn.masked_bias buffers of 4.33-era checkpoints
    } else {
        known = false;        // text_head.*, conditioning_perceiver.*, gpt.wte.* ... (strict=False)
    }
    if (rc == GVC_OK && known) c->bound[n] = 1;
    return rc;
}

// a block matrix of layer li was (re)bound: the one-launch rows step's packed copy of that layer must be rebuilt before its next use
static void mark_rows_pack_dirty(gvc_gpt* c, int li) {
    if (c->r_ready == 1 && li < (int)c->r_dirty.size()) c->r_dirty[li] = 1;
}

extern "C" int gvc_gpt_missing_weights(gvc_gpt* c) {
    if (!c) return -1;
    return c->n_expected - (int)c->bound.size();
}

// ---------------------------------------------------------------------------------------------
// launches
// ---------------------------------------------------------------------------------------------
struct GemvGeom { int NI, ksplit, wpb, grid; };

// launch geometry of a decode GEMV over Wt[N][K]: one wave per (row, 1024-input segment); about one
// workgroup per CU, two when that would need more than 12 compute waves
static int gemv_geom(const gvc_gpt* c, int N, int K, GemvGeom* g) {
    // a wave owns a segment of NI * 256 inputs: the whole row when K <= 1024, else the largest NI <= 4 that divides K / 256
    const int k256 = K / 256;
    g->NI = k256 <= 4 ? k256 : (k256 % 4 == 0 ? 4 : (k256 % 3 == 0 ? 3 : (k256 % 2 == 0 ? 2 : 1)));
    GVC_REQUIRE(K % 256 == 0 && g->NI >= 1, GVC_ERR_UNSUPPORTED, "gemv: K=%d unsupported", K);
    g->ksplit = K / (g->NI * 256);
    GVC_REQUIRE(g->ksplit * g->NI * 256 == K && g->ksplit <= 8, GVC_ERR_UNSUPPORTED, "gemv: K=%d unsupported", K);
    const int items = N * g->ksplit;
    constexpr int wpb_split = 12, blocks_per_cu = 1;        // (swept in round 1: profiles/r01_microbench_notes.md)
    int wpb = cdiv(items, blocks_per_cu * c->n_cu);
    if (wpb > wpb_split) wpb = cdiv(items, 2 * blocks_per_cu * c->n_cu);
    if (wpb < 4) wpb = 4;
    wpb = cdiv(wpb, g->ksplit) * g->ksplit;
    if (wpb > 15) wpb = 15 / g->ksplit * g->ksplit;
    g->wpb = wpb;
    g->grid = cdiv(items, wpb);
    return GVC_OK;
}

template <int PRO, int EPI>
static int launch_gemv(gvc_gpt* c, GemvArgs A, int B, hipStream_t s) {
    GemvGeom g;
    int rc = gemv_geom(c, A.N, A.K, &g);
    if (rc) return rc;
    if (PRO == PRO_LN_SUM && g.wpb > 8) {            // this variant is compiled for <= 512 threads
        g.wpb = 8 / g.ksplit * g.ksplit;
        g.grid = cdiv(A.N * g.ksplit, g.wpb);
    }
    const int NI = g.NI;
    A.ksplit = g.ksplit;
    A.wpb = g.wpb;
    A.B = B;
    if (c->dbg && c->dbg_n < 4096) A.dbg = c->dbg + 8 * (size_t)(c->dbg_n++);
    const int BT = B <= 1 ? 1 : (B <= 2 ? 2 : (B <= 4 ? 4 : 8));
    const int grid = g.grid;
    const int wpb = g.wpb;
    const int nthreads = wpb * 64;
    const size_t lds = ((size_t)BT * A.K + (size_t)wpb * BT) * sizeof(float);
#define GVC_GEMV_CASE(bt, ni)                                                    \
    if (BT == bt && NI == ni) {                                                  \
        if (A.Wt16) hipLaunchKernelGGL((k_gemv<bt, ni, PRO, EPI, 1>), dim3(grid), dim3(nthreads), lds, s, A); \
        else hipLaunchKernelGGL((k_gemv<bt, ni, PRO, EPI, 0>), dim3(grid), dim3(nthreads), lds, s, A); \
        GVC_LAUNCH_CHECK();                                                      \
        return GVC_OK;                                                           \
    }
    GVC_GEMV_CASE(1, 1) GVC_GEMV_CASE(1, 2) GVC_GEMV_CASE(1, 3) GVC_GEMV_CASE(1, 4)
    GVC_GEMV_CASE(2, 1) GVC_GEMV_CASE(2, 2) GVC_GEMV_CASE(2, 3) GVC_GEMV_CASE(2, 4)
    GVC_GEMV_CASE(4, 1) GVC_GEMV_CASE(4, 2) GVC_GEMV_CASE(4, 3) GVC_GEMV_CASE(4, 4)
    GVC_GEMV_CASE(8, 1) GVC_GEMV_CASE(8, 2) GVC_GEMV_CASE(8, 3) GVC_GEMV_CASE(8, 4)
#undef GVC_GEMV_CASE
    set_error("gemv: no instantiation for BT=%d NI=%d", BT, NI);
    return GVC_ERR_UNSUPPORTED;
}

// dynamic LDS above 64 KiB needs an opt-in per kernel; done once at context creation (never while a
// stream is capturing)
template <int PRO, int EPI>
static int gemv_allow_big_lds() {
#define GVC_GEMV_ATTR(bt, ni)                                                                              \
    GVC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemv<bt, ni, PRO, EPI, 0>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));            \
    GVC_CHECK_HIP(hipFuncSetAttribute((const void*)k_gemv<bt, ni, PRO, EPI, 1>,                            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 144 * 1024));
    GVC_GEMV_ATTR(1, 1) GVC_GEMV_ATTR(1, 2) GVC_GEMV_ATTR(1, 3) GVC_GEMV_ATTR(1, 4)
    GVC_GEMV_ATTR(2, 1) GVC_GEMV_ATTR(2, 2) GVC_GEMV_ATTR(2, 3) GVC_GEMV_ATTR(2, 4)
    GVC_GEMV_ATTR(4, 1) GVC_GEMV_ATTR(4, 2) GVC_GEMV_ATTR(4, 3) GVC_GEMV_ATTR(4, 4)
    GVC_GEMV_ATTR(8, 1) GVC_GEMV_ATTR(8, 2) GVC_GEMV_ATTR(8, 3) GVC_GEMV_ATTR(8, 4)
#undef GVC_GEMV_ATTR
    return GVC_OK;
}

static int gemv_init() {
    int rc;
    if ((rc = gemv_allow_big_lds<PRO_LN, EPI_QKV>())) return rc;
    if ((rc = gemv_allow_big_lds<PRO_MERGE, EPI_RESID>())) return rc;
    if ((rc = gemv_allow_big_lds<PRO_LN, EPI_GELU>())) return rc;
    if ((rc = gemv_allow_big_lds<PRO_LN_SUM, EPI_GELU>())) return rc;
    if ((rc = gemv_allow_big_lds<PRO_COPY, EPI_RESID>())) return rc;
    return gemv_allow_big_lds<PRO_LN2X, EPI_LOGITS>();
}

// K (which = 0) or V (1) cache of a layer; element size follows the context's KV dtype
static float* kv_layer(const gvc_gpt* c, int layer, int which) {
    return reinterpret_cast<float*>(reinterpret_cast<char*>(c->kv) + (size_t)(2 * layer + which) * c->kv_layer_stride * (c->kv_bf16 ? 2 : 4));
}

static GemvArgs base_args(gvc_gpt* c, const int32_t* slots, int row0) {
    GemvArgs A;
    memset(&A, 0, sizeof(A));
    A.d = c->dm.d_model;
    A.x = c->x + (size_t)row0 * A.d;
    A.x_stride = 1;
    A.x_off = 0;
    A.n_head = c->dm.n_head;
    A.head_dim = c->hd;
    A.max_seq = c->dm.max_seq;
    A.max_mel_pos = c->dm.max_mel_pos;
    A.slots = slots;
    A.st = c->st;
    A.mel_emb = c->mel_emb;
    A.mel_pos_tab = c->mel_pos;
    A.kv_bf16 = c->kv_bf16;
    A.err = c->seam_err_dev;
    return A;
}

static int launch_attention(gvc_gpt* c, AttnArgs T, int chunks, int rows, bool direct, hipStream_t s, bool wide = false) {
    // prefill-shaped calls (>= 16 rows per stream): 16-row query tiles on the matrix cores
    int rc = GVC_OK;
    if (direct && chunks == 1 && !wide && T.T >= 16 && rows % T.T == 0 &&
        launch_attention_tile(c->hd, c->dm.n_head, T, rows / T.T, T.base_len ? c->dm.max_seq : T.T, s, c->kv_bf16 != 0, &rc))
        return rc;
    return launch_attention_hd(c->hd, c->dm.n_head, T, chunks, rows, direct, s, wide && c->hd == 256, c->kv_bf16 != 0);
}

static AttnArgs gpt_attn_args(gvc_gpt* c, int layer, const int32_t* slots) {
    AttnArgs T;
    memset(&T, 0, sizeof(T));
    T.q_stride = c->dm.d_model;
    T.kbase = kv_layer(c, layer, 0);
    T.vbase = kv_layer(c, layer, 1);
    T.k_batch_stride = (long long)c->dm.n_head * c->dm.max_seq * c->hd;
    T.k_head_stride = (long long)c->dm.max_seq * c->hd;
    T.k_row_stride = c->hd;
    T.slots = slots;
    T.causal = 1;
    T.scale = 1.0f / sqrtf((float)c->hd);
    return T;
}

// one decode step for a group of <= 8 streams whose scratch rows start at row0
// gvc_gpt_time_kernel(): when prof_only >= 0 only that kernel class of the step is launched
static inline bool prof_skip(const gvc_gpt* c, int which) {
    return (c->prof_only >= 0 && c->prof_only != which) || c->prof_skip_one == which;
}

// can the fused attention + c_proj launch serve this call?  one stream, head_dim 256, and the caller
// guarantees at most 8 * kFusedMaxKeys cached positions for the whole run (gvc_gpt_generate: ids_stride)
static bool fused_ok(const gvc_gpt* c, int B, int max_keys) {
    return c->fuse_decode && B == 1 && c->hd == 256 && c->dm.d_model % 16 == 0 && c->dm.d_model <= 1024 && max_keys <= 8 * kFusedMaxKeys;
}

// ---------------------------------------------------------------------------------------------
// one-launch decode step (persist_kernel.h)
// ---------------------------------------------------------------------------------------------
// test hook (tests/test_gpu_gpt.py): GVC_PERSIST_TEST_GRID=255 launches the one-launch steps one workgroup short, which is what a
// non-resident workgroup looks like to the others -- every hand-off times out and the fallback of check_ready must take over
static int persist_test_grid() {
    const char* e = getenv("GVC_PERSIST_TEST_GRID");        // (read per launch: a test sets and clears it inside one process)
    const int g = e ? atoi(e) : kPG;
    return g > 0 && g <= kPG ? g : kPG;
}

// the instantiation that serves this context: d_model / 256, bf16 weight storage, bf16 KV cache (bf16 rows are streamed in whole
// KiB per workgroup and phase, which needs an even d_model / 256)
typedef void (*persist_fn)(const PersistArgs);
static persist_fn persist_kernel(const gvc_gpt* c) {
    const int nd = c->dm.d_model / 256;
    // p_xl = 0 (GVC_PERSIST_XCD=0, or a device that does not deal 8 XCDs x 32 workgroups: persist_prepare's probe): d_model 1024 keeps the
    // device-wide hand-off of the hidden units (the round-3 kernel)
    if (nd == 4 && c->p_xl) {
        if (c->kv_bf16) return (persist_fn)k_decode_persist<4, 1, 1, 1>;
        if (c->bf16) return (persist_fn)k_decode_persist<4, 1, 0, 1>;
        return (persist_fn)k_decode_persist<4, 0, 0, 1>;
    }
    if (c->kv_bf16) return nd == 4 ? (persist_fn)k_decode_persist<4, 1, 1> : (persist_fn)k_decode_persist<2, 1, 1>;
    if (c->bf16) return nd == 4 ? (persist_fn)k_decode_persist<4, 1, 0> : (persist_fn)k_decode_persist<2, 1, 0>;
    return nd == 4 ? (persist_fn)k_decode_persist<4> : nd == 3 ? (persist_fn)k_decode_persist<3> : nd == 2 ? (persist_fn)k_decode_persist<2>
                                                                                                              : (persist_fn)k_decode_persist<1>;
}

// one stream, fp32 weights and cache, a full MI355X (one workgroup per CU)
static bool persist_ok(const gvc_gpt* c, int B) {
    const int d = c->dm.d_model;
    return c->persist && B == 1 && (!c->bf16 || d == 512 || d == 1024) && c->n_cu >= kPG && d % 256 == 0 && d <= 1024 &&
           (c->hd == 64 || c->hd == 128 || c->hd == 256) && c->dm.n_layer < 500;
}

// Topology probe for the XCD-local hand-off (k_decode_persist<4, *, *, 1>): a grid of the one-launch step's own shape (256 workgroups, its
// threads and LDS, so one per CU), all co-resident (they wait for each other, bounded), each adding itself to its XCD's counter.
// hist[0..7]: workgroups per XCC_ID; hist[8]: arrivals; hist[9]: workgroups that gave up waiting.
__global__ __launch_bounds__(gvc::kPThreads) void k_xcd_probe(unsigned* hist) {
    if (threadIdx.x != 0) return;
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    atomicAdd(hist + (xcc & 7u), 1u);
    atomicAdd(hist + 8, 1u);
    for (unsigned i = 0; i < (1u << 16); ++i) {         // (~0.1 s at ~1.5 us per poll; a grid that is dealt one workgroup at a time still ends within half a minute)
        if (__hip_atomic_load(hist + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= gridDim.x) return;
        __builtin_amdgcn_s_sleep(8);
    }
    atomicAdd(hist + 9, 1u);
}

// does this device run a 256-workgroup grid of the one-launch step as 8 XCDs x 32 co-resident workgroups?
static bool xcd_topology_ok(gvc_gpt* c) {
    if (!c->arch_xcc) return false;
    unsigned* hist = nullptr;
    unsigned h[10] = {0};
    bool ok = hipMalloc((void**)&hist, sizeof(h)) == hipSuccess && hipMemsetAsync(hist, 0, sizeof(h), c->cap_stream) == hipSuccess &&
              hipFuncSetAttribute((const void*)k_xcd_probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->p_lds) == hipSuccess;
    if (ok) {
        // (on the context's own non-blocking stream: a launch on the legacy default stream drags every other stream's pending work in)
        hipLaunchKernelGGL(k_xcd_probe, dim3(kPG), dim3(kPThreads), c->p_lds, c->cap_stream, hist);
        ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(c->cap_stream) == hipSuccess &&
             hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (hist) (void)hipFree(hist);
    (void)hipGetLastError();
    if (!ok || h[9] != 0 || h[8] != (unsigned)kPG) return false;
    for (int x = 0; x < 8; ++x)
        if (h[x] != (unsigned)kPG / 8) return false;
    return true;
}

// a data-path call had to allocate / synchronise the device / capture a graph (first use of a path without gvc_gpt_warmup)
static inline void note_lazy(gvc_gpt* c) { if (!c->in_warmup) c->lazy_inits += 1; }

// buffers and the per-layer pointer table; called outside stream capture (synchronous copies)
static int persist_prepare(gvc_gpt* c) {
    if (c->p_layers || !c->persist) return GVC_OK;
    note_lazy(c);
    const int d = c->dm.d_model, L = c->dm.n_layer, H = c->dm.n_head;
    std::vector<PersistLayer> t(L);
    for (int l = 0; l < L; ++l) {
        const GptLayer& ly = c->layers[l];
        PersistLayer& p = t[l];
        p.ln1_w = ly.ln1_w; p.ln1_b = ly.ln1_b; p.qkv_w = ly.qkv_w; p.qkv_b = ly.qkv_b; p.proj_w = ly.proj_w; p.proj_b = ly.proj_b;
        p.ln2_w = ly.ln2_w; p.ln2_b = ly.ln2_b; p.fc_w = ly.fc_w; p.fc_b = ly.fc_b; p.p2_w = ly.p2_w; p.p2_b = ly.p2_b;
        if (c->bf16) {        // bf16-weights context: the loader streams the row-major bf16 copies (same row order, half the bytes)
            p.qkv_w = reinterpret_cast<const float*>(ly.qkv_h); p.proj_w = reinterpret_cast<const float*>(ly.proj_h);
            p.fc_w = reinterpret_cast<const float*>(ly.fc_h); p.p2_w = reinterpret_cast<const float*>(ly.p2_h);
        }
        p.kcache = kv_layer(c, l, 0); p.vcache = kv_layer(c, l, 1);
    }
    // Anything the device refuses here (memory, the LDS opt-in, residency of one workgroup per CU) switches the one-launch step off
    // for this context -- the launch-per-phase step serves it -- instead of failing the caller's decode / generate call; the
    // buffers are allocated once (a second call after a failure finds the path off and returns before this point).
    auto unavailable = [&]() {
        for (void** p : {(void**)&c->p_gran, (void**)&c->p_epoch, (void**)&c->p_dbg})
            if (*p) { (void)hipFree(*p); *p = nullptr; }
        (void)hipGetLastError();
        c->persist = 0;
        return GVC_OK;
    };
    const size_t ngran = persist_granules(d, H);
    if (hipMalloc((void**)&c->p_gran, ngran * sizeof(pu64)) != hipSuccess || hipMemset(c->p_gran, 0, ngran * sizeof(pu64)) != hipSuccess ||
        hipMalloc((void**)&c->p_epoch, 16 * sizeof(unsigned)) != hipSuccess || hipMemset(c->p_epoch, 0, 16 * sizeof(unsigned)) != hipSuccess)
        return unavailable();
    if (getenv("GVC_PERSIST_STAMPS")) {
        const size_t nb = (size_t)(20 * (L + 2) + 2 * 5 * kPG + 8 * kPG) * sizeof(unsigned long long);
        if (hipMalloc((void**)&c->p_dbg, nb) != hipSuccess || hipMemset(c->p_dbg, 0, nb) != hipSuccess) return unavailable();
    }
    // the memsets above went to the legacy default stream; the caller's stream may be a non-blocking one (no implicit ordering with it): they
    // must have landed before the first launch is enqueued.  (Device-wide sync HERE, before the context's own stream has run anything: under
    // `rocprofv3 --pmc` a device-wide sync behind the probe launch on that stream -- or the probe on the default stream -- never returned.)
    if (hipDeviceSynchronize() != hipSuccess) return unavailable();
    const int ng_hd = kPCW * 256;                       // lane-group states of the attention phase: [kPCW * 64 / (hd / 4)][hd]
    c->p_hvec = 4 * d > d + ng_hd ? 4 * d : d + ng_hd;
    c->p_ascr = 3 * c->hd + 64;
    const size_t other = ((size_t)c->p_hvec + d + c->p_ascr) * sizeof(float) + kCtlWords * sizeof(unsigned);
    // the ring takes what the device's opt-in LDS limit leaves (160 KiB on gfx950: 8 slots)
    int dev_id = 0, lds_optin = 0;
    if (hipGetDevice(&dev_id) != hipSuccess ||
        hipDeviceGetAttribute(&lds_optin, hipDeviceAttributeSharedMemPerBlockOptin, dev_id) != hipSuccess || lds_optin <= 0)
        lds_optin = 64 * 1024;
    c->p_ring_slots = 8;
    while (c->p_ring_slots > 1 && (size_t)c->p_ring_slots * kPSlot + other > (size_t)lds_optin) c->p_ring_slots >>= 1;
    c->p_lds = (size_t)c->p_ring_slots * kPSlot + other;
    if (c->p_lds > (size_t)lds_optin || c->p_ring_slots < 4) return unavailable();
    // the XCD-local layout assumes 8 XCDs x 32 resident workgroups and HW_REG_XCC_ID: probed once per context, on a grid of the
    // step's own shape; anything else (CU masking, another partition mode, another part) keeps the device-wide hand-off
    if (d == 1024 && c->p_xl && !xcd_topology_ok(c)) c->p_xl = 0;
    if (hipFuncSetAttribute((const void*)persist_kernel(c), hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->p_lds) != hipSuccess)
        return unavailable();
    // one workgroup per CU must fit (registers, LDS): otherwise the one-launch step is switched off for this context
    int per_cu = 0;
    hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, persist_kernel(c), kPThreads, c->p_lds);
    if (oe != hipSuccess || per_cu < 1) return unavailable();
    PersistLayer* dev = nullptr;
    if (hipMalloc((void**)&dev, L * sizeof(PersistLayer)) != hipSuccess ||
        hipMemcpy(dev, t.data(), L * sizeof(PersistLayer), hipMemcpyHostToDevice) != hipSuccess) {
        if (dev) (void)hipFree(dev);
        return unavailable();
    }
    c->p_layers = dev;
    return GVC_OK;
}

static int launch_persist(gvc_gpt* c, const int32_t* slots, const int32_t* tok_in, float* logits_out, float* latent_out,
                          int32_t* step_ctr, hipStream_t s) {
    GVC_REQUIRE(c->p_layers, GVC_ERR_STATE, "persistent decode step: not prepared");
    PersistArgs A;
    memset(&A, 0, sizeof(A));
    A.layers = c->p_layers; A.n_layer = c->dm.n_layer; A.d = c->dm.d_model; A.n_head = c->dm.n_head; A.head_dim = c->hd;
    A.vocab = c->dm.vocab; A.max_seq = c->dm.max_seq; A.max_mel_pos = c->dm.max_mel_pos;
    A.mel_emb = c->mel_emb; A.mel_pos = c->mel_pos; A.lnf_w = c->lnf_w; A.lnf_b = c->lnf_b; A.fn_w = c->fn_w; A.fn_b = c->fn_b;
    A.head_w = c->head_w; A.head_b = c->head_b; A.slots = slots; A.tok_in = tok_in; A.st = c->st;
    A.logits_out = logits_out; A.latent_out = latent_out; A.step_ctr = step_ctr; A.advance = 1;
    A.gran = c->p_gran; A.epoch = c->p_epoch; A.err = c->seam_err_dev; A.ring_slots = c->p_ring_slots; A.ascr_floats = c->p_ascr; A.hvec_floats = c->p_hvec;
    A.dbg = c->p_dbg;
    if (c->bf16) A.head_w = reinterpret_cast<const float*>(c->head_h);
    void* kargs[] = {&A};
    GVC_CHECK_HIP(hipLaunchKernel((const void*)persist_kernel(c), dim3(persist_test_grid()), dim3(kPThreads), kargs, c->p_lds, s));
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}


// ---------------------------------------------------------------------------------------------
// one-launch block stack for 2..16 rows (persist_rows.h)
// ---------------------------------------------------------------------------------------------
static bool rows_persist_ok(const gvc_gpt* c, int rows, const int32_t* base_len) {
    return c->persist && c->persist_rows && c->r_ready >= 0 && base_len && rows >= c->persist_rows_min && rows <= kRMaxRows &&
           c->dm.d_model == kRD && (c->hd == 64 || c->hd == 128 || c->hd == 256) && c->dm.n_head * c->hd == kRD && c->dm.n_layer % 2 == 0 &&
           c->n_cu >= kPG;
}

// key chunks per (row, head) for a call whose longest context reaches `keys` cached positions
static int rows_persist_chunks(const gvc_gpt* c, int rows, int keys) {
    (void)keys;                              // (the kernel picks the split from the contexts it finds; this is the bound)
    int nch = kRMaxChunks;
    const int R = rows <= 8 ? 8 : 16;
    while (nch > 1 && R * (kRD / kRHD) * nch > kPG) nch >>= 1;       // (phase B's workgroups: one per row, 256-dim super-head and key chunk)
    return nch;
}

// the instantiation for this context: padded rows (8 / 16), weight / cache storage, real head_dim
typedef void (*rows_fn)(const RowsArgs);
template <int HDR>
static const void* rows_kernel_hd(const gvc_gpt* c, int R) {
    rows_fn f;
    if (c->act_bf16) f = R == 8 ? (rows_fn)k_rows_persist_b16<8, HDR> : (rows_fn)k_rows_persist_b16<16, HDR>;
    else if (c->kv_bf16) f = R == 8 ? (rows_fn)k_rows_persist<8, 1, 1, HDR> : (rows_fn)k_rows_persist<16, 1, 1, HDR>;
    else if (c->bf16) f = R == 8 ? (rows_fn)k_rows_persist<8, 1, 0, HDR> : (rows_fn)k_rows_persist<16, 1, 0, HDR>;
    else f = R == 8 ? (rows_fn)k_rows_persist<8, 0, 0, HDR> : (rows_fn)k_rows_persist<16, 0, 0, HDR>;
    return (const void*)f;
}
static const void* rows_kernel(const gvc_gpt* c, int R) {
    return c->hd == 64 ? rows_kernel_hd<64>(c, R) : c->hd == 128 ? rows_kernel_hd<128>(c, R) : rows_kernel_hd<256>(c, R);
}

static void rows_persist_release(gvc_gpt* c) {
    for (void** p : {(void**)&c->r_layers, (void**)&c->r_wpack, (void**)&c->r_bufs, (void**)&c->r_dbg, (void**)&c->r_lnfold})
        if (*p) { (void)hipFree(*p); *p = nullptr; }
    (void)hipGetLastError();
    c->r_ready = -1;
}

// packed weight copy, per-layer table, hand-off buffers; called outside stream capture (synchronous).  Anything the device
// refuses (memory, LDS opt-in, residency) switches the path off for this context instead of failing the call.
static void rows_pack_layer(gvc_gpt* c, int l) {
    const GptLayer& ly = c->layers[l];
    const int wsh = c->bf16 ? 1 : 0;
    float* dst = reinterpret_cast<float*>(reinterpret_cast<char*>(c->r_wpack) + (size_t)l * kPG * (kRWgLayerBytes >> wsh));
    if (c->act_bf16) {
        // bf16-activation mode: octet-major bf16 groups with the LayerNorm gains folded into c_attn / c_fc, and the constants that go with them
        const int d = kRD;
        float* f = c->r_lnfold + (size_t)l * 14 * d;
        hipLaunchKernelGGL(k_pack_rows_weights_b16, dim3(2048), dim3(256), 0, 0, reinterpret_cast<pu32x4*>(dst), (const float*)ly.qkv_w,
                           (const float*)ly.proj_w, (const float*)ly.fc_w, (const float*)ly.p2_w, (const float*)ly.ln1_w, (const float*)ly.ln2_w);
        hipLaunchKernelGGL(k_rows_ln_fold_b16, dim3(3 * d / 4), dim3(256), 0, 0, f, f + 3 * d, (const float*)ly.qkv_w, (const float*)ly.ln1_w,
                           (const float*)ly.ln1_b, (const float*)ly.qkv_b, 3 * d, d);
        hipLaunchKernelGGL(k_rows_ln_fold_b16, dim3(4 * d / 4), dim3(256), 0, 0, f + 6 * d, f + 10 * d, (const float*)ly.fc_w, (const float*)ly.ln2_w,
                           (const float*)ly.ln2_b, (const float*)ly.fc_b, 4 * d, d);
        return;
    }
    if (wsh) hipLaunchKernelGGL(k_pack_rows_weights<1>, dim3(2048), dim3(256), 0, 0, dst, (const float*)ly.qkv_w, (const float*)ly.proj_w,
                                (const float*)ly.fc_w, (const float*)ly.p2_w);
    else hipLaunchKernelGGL(k_pack_rows_weights<0>, dim3(2048), dim3(256), 0, 0, dst, (const float*)ly.qkv_w, (const float*)ly.proj_w,
                            (const float*)ly.fc_w, (const float*)ly.p2_w);
    // LayerNorm folded into per-output-row constants (persist_rows.h, phases A / D): S_r = sum_k W_rk g_k, C_r = sum_k W_rk b_k + bias_r
    const int d = kRD;
    float* f = c->r_lnfold + (size_t)l * 14 * d;
    hipLaunchKernelGGL(k_rows_ln_fold, dim3(3 * d / 4), dim3(256), 0, 0, f, f + 3 * d, (const float*)ly.qkv_w, (const float*)ly.ln1_w,
                       (const float*)ly.ln1_b, (const float*)ly.qkv_b, 3 * d, d);
    hipLaunchKernelGGL(k_rows_ln_fold, dim3(4 * d / 4), dim3(256), 0, 0, f + 6 * d, f + 10 * d, (const float*)ly.fc_w, (const float*)ly.ln2_w,
                       (const float*)ly.ln2_b, (const float*)ly.fc_b, 4 * d, d);
}

static int rows_persist_prepare(gvc_gpt* c) {
    if (c->r_ready == 1 && !c->r_dirty.empty()) {
        // weights re-bound since the pack was built (a second GptEngine.bind on the same context): the packed copy of those layers is
        // stale.  Rare and outside capture: wait for the bind stream's transposes, repack, wait again.
        bool any = false;
        for (char f : c->r_dirty) any = any || f;
        if (any) {
            note_lazy(c);
            GVC_CHECK_HIP(hipDeviceSynchronize());
            for (int l = 0; l < c->dm.n_layer; ++l)
                if (c->r_dirty[l]) rows_pack_layer(c, l);
            GVC_CHECK_HIP(hipGetLastError());
            GVC_CHECK_HIP(hipDeviceSynchronize());
            std::fill(c->r_dirty.begin(), c->r_dirty.end(), 0);
        }
    }
    if (c->r_ready != 0) return GVC_OK;
    note_lazy(c);
    const int L = c->dm.n_layer;
    c->r_lds = (size_t)8 * kPSlot + ((size_t)kPCW * 4 * kRMaxRows * 4 + 2 * kPCW * 16 + 3 * kRMaxRows * 4 + kPCW * 64 * 4 + 256 + 64 + kPCW * 256) * sizeof(float) +
               kCtlWords * sizeof(unsigned);
    int per_cu = 0;
    const void* kern[2] = {rows_kernel(c, 8), rows_kernel(c, 16)};      // the two instantiations (8 / 16 padded rows) of this context's storage types and head_dim
    const int wsh = c->bf16 ? 1 : 0;
    if (hipFuncSetAttribute(kern[0], hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->r_lds) != hipSuccess ||
        hipFuncSetAttribute(kern[1], hipFuncAttributeMaxDynamicSharedMemorySize, (int)c->r_lds) != hipSuccess ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern[1], kPThreads, c->r_lds) != hipSuccess || per_cu < 1 ||
        hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern[0], kPThreads, c->r_lds) != hipSuccess || per_cu < 1 ||
        hipMalloc((void**)&c->r_wpack, (size_t)L * kPG * (kRWgLayerBytes >> wsh)) != hipSuccess ||
        hipMalloc((void**)&c->r_bufs, c->act_bf16 ? rows_b16_buf_bytes() : rows_buf_bytes()) != hipSuccess ||
        hipMalloc((void**)&c->r_lnfold, (size_t)L * 14 * kRD * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&c->r_layers, L * sizeof(RowsLayer)) != hipSuccess) {
        rows_persist_release(c);
        return GVC_OK;
    }
    std::vector<RowsLayer> t(L);
    for (int l = 0; l < L; ++l) {
        const GptLayer& ly = c->layers[l];
        RowsLayer& p = t[l];
        p.ln1_w = ly.ln1_w; p.ln1_b = ly.ln1_b; p.qkv_b = ly.qkv_b; p.proj_b = ly.proj_b; p.ln2_w = ly.ln2_w; p.ln2_b = ly.ln2_b;
        p.fc_b = ly.fc_b; p.p2_b = ly.p2_b; p.kcache = kv_layer(c, l, 0); p.vcache = kv_layer(c, l, 1);
        p.lnS_a = c->r_lnfold + (size_t)l * 14 * kRD; p.lnC_a = p.lnS_a + 3 * kRD; p.lnS_d = p.lnS_a + 6 * kRD; p.lnC_d = p.lnS_a + 10 * kRD;
        rows_pack_layer(c, l);
    }
    if (hipGetLastError() != hipSuccess || hipMemcpy(c->r_layers, t.data(), L * sizeof(RowsLayer), hipMemcpyHostToDevice) != hipSuccess ||
        hipMemset(c->r_bufs, 0xff, c->act_bf16 ? rows_b16_buf_bytes() : rows_buf_bytes()) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
        rows_persist_release(c);
        return GVC_OK;
    }
    if (getenv("GVC_PERSIST_STAMPS")) {        // diagnostics only: a failed allocation just leaves the stamps off
        const size_t nb = (size_t)(40 * (L + 2) + 8 * 5 * kPG) * sizeof(unsigned long long);
        if (hipMalloc((void**)&c->r_dbg, nb) != hipSuccess || hipMemset(c->r_dbg, 0, nb) != hipSuccess) {
            if (c->r_dbg) (void)hipFree(c->r_dbg);
            c->r_dbg = nullptr;
            (void)hipGetLastError();
        }
    }
    c->r_dirty.assign(L, 0);
    c->r_ready = 1;
    return GVC_OK;
}

static int launch_rows_persist(gvc_gpt* c, const int32_t* slots, int rows, int T, const int32_t* base_len, int nch, hipStream_t s,
                               const int32_t* tok_in = nullptr) {
    RowsArgs A;
    memset(&A, 0, sizeof(A));
    A.layers = c->r_layers; A.wpack = reinterpret_cast<const char*>(c->r_wpack); A.n_layer = c->dm.n_layer; A.n_head = c->dm.n_head;
    A.max_seq = c->dm.max_seq; A.rows = rows; A.T = T; A.slots = slots; A.base_len = base_len; A.x = c->x; A.bufs = c->r_bufs;
    A.err = c->seam_err_dev; A.ring_slots = 8; A.nchunks = nch; A.dbg = c->r_dbg;
    A.tok_in = tok_in; A.mel_emb = c->mel_emb; A.mel_pos = c->mel_pos; A.mel_pos_idx = c->st.mel_pos; A.vocab = c->dm.vocab;
    // keys of a (row, head) over 2 / 4 workgroups: 8 rows from 80 / 160 cached positions (one 80-key pass per workgroup; 744 vs 766 us
    // per step at 48-112 keys, 815 vs 827 at 110-250), 16 rows from 128 / 288 (their chunk merge gathers 64 KB per chunk: 1068 vs 1104 us)
    A.split1 = rows <= 8 ? 80 : 128;
    A.split2 = rows <= 8 ? 160 : 288;
    void* kargs[] = {&A};
    GVC_CHECK_HIP(hipLaunchKernel(rows_kernel(c, rows <= 8 ? 8 : 16), dim3(persist_test_grid()), dim3(kPThreads), kargs, c->r_lds, s));
    GVC_LAUNCH_CHECK();
    c->r_launches += 1;
    return GVC_OK;
}

static int decode_group(gvc_gpt* c, const int32_t* slots, int B, int row0, const int32_t* tok_in, float* logits_out,
                        float* latent_out, int32_t* step_ctr, hipStream_t s, bool fused = false) {
    const int d = c->dm.d_model;
    int rc;
    float* qb = c->q + (size_t)row0 * d;
    float* hb = c->h + (size_t)row0 * 4 * d;
    float* pb = c->part + (size_t)row0 * c->dm.n_head * kAttnChunks * (c->hd + 4);
    for (int l = 0; l < c->dm.n_layer; ++l) {
        const GptLayer& ly = c->layers[l];
        GemvArgs A = base_args(c, slots, row0);
        A.Wt = ly.qkv_w; A.Wt16 = ly.qkv_h; A.bias = ly.qkv_b; A.N = 3 * d; A.K = d;
        A.ln_w = ly.ln1_w; A.ln_b = ly.ln1_b; A.embed = l == 0; A.tok_in = tok_in;
        A.out = qb;
        A.kcache = kv_layer(c, l, 0);
        A.vcache = kv_layer(c, l, 1);
        if (!prof_skip(c, 0) && (rc = launch_gemv<PRO_LN, EPI_QKV>(c, A, B, s))) return rc;

        if (fused) {
            // attention + head-split c_proj in one launch; c_fc's prologue sums the per-head partials
            AttnProjArgs F;
            memset(&F, 0, sizeof(F));
            F.q = qb; F.kcache = A.kcache; F.vcache = A.vcache; F.slots = slots; F.seq_len = c->st.seq_len;
            F.max_seq = c->dm.max_seq; F.n_head = c->dm.n_head; F.d = d; F.scale = 1.0f / sqrtf((float)c->hd);
            F.Wp = ly.proj_w; F.part2 = c->part2 + (size_t)row0 * c->dm.n_head * d;
            if (!prof_skip(c, 1)) {
                if (c->kv_bf16) hipLaunchKernelGGL((k_attn_proj<256, 1>), dim3(d / 16, c->dm.n_head), dim3(512), 0, s, F);
                else hipLaunchKernelGGL((k_attn_proj<256, 0>), dim3(d / 16, c->dm.n_head), dim3(512), 0, s, F);
                GVC_LAUNCH_CHECK();
            }
            A = base_args(c, slots, row0);
            A.Wt = ly.fc_w; A.Wt16 = ly.fc_h; A.bias = ly.fc_b; A.N = 4 * d; A.K = d; A.ln_w = ly.ln2_w; A.ln_b = ly.ln2_b; A.out = hb;
            A.part2 = F.part2; A.pbias = ly.proj_b; A.x2 = c->x2 + (size_t)row0 * d;
            if (!prof_skip(c, 3) && (rc = launch_gemv<PRO_LN_SUM, EPI_GELU>(c, A, B, s))) return rc;
            A = base_args(c, slots, row0);
            A.Wt = ly.p2_w; A.Wt16 = ly.p2_h; A.bias = ly.p2_b; A.N = d; A.K = 4 * d; A.in = hb; A.xres = c->x2 + (size_t)row0 * d;
            if (!prof_skip(c, 4) && (rc = launch_gemv<PRO_COPY, EPI_RESID>(c, A, B, s))) return rc;
            continue;
        }
        AttnArgs T = gpt_attn_args(c, l, slots);
        T.q = qb; T.T = 1; T.base_len = c->st.seq_len; T.out = pb;
        if (!prof_skip(c, 1) && (rc = launch_attention(c, T, kAttnChunks, B, false, s))) return rc;

        A = base_args(c, slots, row0);
        A.Wt = ly.proj_w; A.Wt16 = ly.proj_h; A.bias = ly.proj_b; A.N = d; A.K = d; A.in = pb;
        if (!prof_skip(c, 2) && (rc = launch_gemv<PRO_MERGE, EPI_RESID>(c, A, B, s))) return rc;

        A = base_args(c, slots, row0);
        A.Wt = ly.fc_w; A.Wt16 = ly.fc_h; A.bias = ly.fc_b; A.N = 4 * d; A.K = d; A.ln_w = ly.ln2_w; A.ln_b = ly.ln2_b; A.out = hb;
        if (!prof_skip(c, 3) && (rc = launch_gemv<PRO_LN, EPI_GELU>(c, A, B, s))) return rc;

        A = base_args(c, slots, row0);
        A.Wt = ly.p2_w; A.Wt16 = ly.p2_h; A.bias = ly.p2_b; A.N = d; A.K = 4 * d; A.in = hb;
        if (!prof_skip(c, 4) && (rc = launch_gemv<PRO_COPY, EPI_RESID>(c, A, B, s))) return rc;
    }
    GemvArgs A = base_args(c, slots, row0);
    A.Wt = c->head_w; A.Wt16 = c->head_h; A.bias = c->head_b; A.N = c->dm.vocab; A.K = d;
    A.ln_w = c->lnf_w; A.ln_b = c->lnf_b; A.ln2_w = c->fn_w; A.ln2_b = c->fn_b;
    A.out = logits_out; A.latent_out = latent_out; A.advance = 1; A.step_ctr = step_ctr;
    if (prof_skip(c, 5)) return GVC_OK;
    return launch_gemv<PRO_LN2X, EPI_LOGITS>(c, A, B, s);
}

static int run_rows(gvc_gpt* c, const int32_t* slots, int B, int T, hipStream_t s, const int32_t* base_len = nullptr, int key_chunks = 1);

// double LayerNorm (ln_f, final_norm) of row (b * stride + off) of x -> latent[b]  (the head of a model wider than the GEMV kernels take)
__global__ void k_head_ln(const float* x, int stride, int off, float* latent, int B, int d, const float* w1, const float* b1, const float* w2,
                          const float* b2) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (b >= B) return;
    const float* in = x + ((size_t)b * stride + off) * d;
    float* y = latent + (size_t)b * d;
    const float inv_d = 1.0f / (float)d;
    for (int pass = 0; pass < 2; ++pass) {
        const float* gw = pass == 0 ? w1 : w2;
        const float* gb = pass == 0 ? b1 : b2;
        float s = 0.f;
        for (int k = lane * 4; k < d; k += 256) { const float4 v = *reinterpret_cast<const float4*>(in + k); s += (v.x + v.y) + (v.z + v.w); }
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
        for (int k = lane * 4; k < d; k += 256) {
            const float4 v = *reinterpret_cast<const float4*>(in + k);
            const float a0 = v.x - mean, a1 = v.y - mean, a2 = v.z - mean, a3 = v.w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
        for (int k = lane * 4; k < d; k += 256) {
            float4 v = *reinterpret_cast<const float4*>(in + k);
            const float4 g = *reinterpret_cast<const float4*>(gw + k), c = *reinterpret_cast<const float4*>(gb + k);
            v.x = (v.x - mean) * rstd * g.x + c.x; v.y = (v.y - mean) * rstd * g.y + c.y;
            v.z = (v.z - mean) * rstd * g.z + c.z; v.w = (v.w - mean) * rstd * g.w + c.w;
            *reinterpret_cast<float4*>(y + k) = v;
        }
        in = y;
    }
}

// what EPI_LOGITS does behind the logits: the slots' caches grew by one position
__global__ void k_head_advance(GptState st, const int32_t* slots, int B, int max_seq, int max_mel_pos, int32_t* step_ctr, int* err) {
    const int t = threadIdx.x;
    if (t < B) {
        const int slot = slots[t];
        if (st.seq_len[slot] < max_seq - 1) st.seq_len[slot] += 1;
        else if (err) *err = 950;
        if (st.mel_pos[slot] < max_mel_pos - 1) st.mel_pos[slot] += 1;
        else if (err) *err = 951;
    }
    if (step_ctr && t == 0) *step_ctr += 1;
}

// gpt_inference.py:18,111-112 for the rows (b * x_stride + x_off) of x: latent = final_norm(ln_f(h)), logits = mel_head(latent).  Up to
// d_model 1024 ONE GEMV launch with the two LayerNorms in its prologue (8 streams per launch); wider models: a LayerNorm launch + a GEMM
static int launch_head(gvc_gpt* c, const int32_t* slots, int B, int row0, const float* x, int x_stride, int x_off, float* logits_out,
                       float* latent_out, int advance, int32_t* step_ctr, hipStream_t s) {
    const int d = c->dm.d_model;
    int rc;
    if (d <= 1024) {
        for (int g = 0; g < B; g += 8) {
            const int Bg = B - g < 8 ? B - g : 8;
            GemvArgs A = base_args(c, slots + g, row0 + g);
            A.x = const_cast<float*>(x) + (size_t)g * x_stride * d; A.x_stride = x_stride; A.x_off = x_off;
            A.Wt = c->head_w; A.Wt16 = c->head_h; A.bias = c->head_b; A.N = c->dm.vocab; A.K = d;
            A.ln_w = c->lnf_w; A.ln_b = c->lnf_b; A.ln2_w = c->fn_w; A.ln2_b = c->fn_b;
            A.out = logits_out + (size_t)g * c->dm.vocab; A.latent_out = latent_out + (size_t)g * d; A.advance = advance;
            A.step_ctr = g + 8 >= B ? step_ctr : nullptr;
            if ((rc = launch_gemv<PRO_LN2X, EPI_LOGITS>(c, A, Bg, s))) return rc;
        }
        return GVC_OK;
    }
    hipLaunchKernelGGL(k_head_ln, dim3(cdiv(B, 4)), dim3(256), 0, s, x, x_stride, x_off, latent_out, B, d, c->lnf_w, c->lnf_b, c->fn_w, c->fn_b);
    GVC_LAUNCH_CHECK();
    GemmArgs G;
    memset(&G, 0, sizeof(G));
    G.A = latent_out; G.lda = d; G.Wt = c->head_w; G.ldw = d; G.C = logits_out; G.ldc = c->dm.vocab; G.M = B; G.N = c->dm.vocab; G.K = d;
    G.work = c->work; G.e.bias = c->head_b;
    if ((rc = launch_gemm_cap(G, 1, c->work_cap, s))) return rc;
    if (advance || step_ctr) {
        hipLaunchKernelGGL(k_head_advance, dim3(1), dim3(128), 0, s, c->st, slots, advance ? B : 0, c->dm.max_seq, c->dm.max_mel_pos, step_ctr,
                           c->seam_err_dev);
        GVC_LAUNCH_CHECK();
    }
    return GVC_OK;
}

// Batched decode step on the MFMA path: the B new rows go through the skinny fragment-major GEMMs (weights streamed
// once for up to 128 streams) instead of the 8-stream GEMV groups.  Same arithmetic as a prefill of one row per stream
// appended at each slot's cached length.
static bool rows_decode_ok(const gvc_gpt* c, int B) {
    if (c->r_ready >= 0 && rows_persist_ok(c, B, c->st.seq_len)) return true;       // served by the one-launch rows step (from 2 streams)
    const bool wide = c->dm.d_model > 1024;       // no GEMV path above 1024: every batch decodes as rows
    return (wide || (c->rows_decode_min > 0 && B >= c->rows_decode_min)) && B <= 128 && c->skinny_prefill && c->wfm &&
           c->dm.d_model % 256 == 0 && (long long)4 * B * c->dm.d_model <= c->work_cap / 2 && B <= c->dm.max_rows;
}

static int decode_rows(gvc_gpt* c, const int32_t* slots, int B, const int32_t* tok_in, float* logits_out, float* latent_out,
                       int32_t* step_ctr, hipStream_t s, int key_chunks = 1) {
    const int d = c->dm.d_model;
    int rc;
    if (c->r_ready == 1 && rows_persist_ok(c, B, c->st.seq_len)) {        // the one-launch rows step builds its input rows itself
        if ((rc = launch_rows_persist(c, slots, B, 1, c->st.seq_len, rows_persist_chunks(c, B, c->rows_keys_hint), s, tok_in))) return rc;
    } else {
        hipLaunchKernelGGL(k_embed_decode_rows, dim3(B), dim3(256), 0, s, c->x, tok_in, slots, c->st, c->mel_emb, c->mel_pos, d, c->dm.vocab);
        GVC_LAUNCH_CHECK();
        if ((rc = run_rows(c, slots, B, 1, s, c->st.seq_len, key_chunks))) return rc;
    }
    return launch_head(c, slots, B, 0, c->x, 1, 0, logits_out, latent_out, 1, step_ctr, s);
}

static int check_ready(gvc_gpt* c) {
    GVC_REQUIRE(c, GVC_ERR_ARG, "null context");
    const int dev_err = c->seam_err_host ? *(volatile int*)c->seam_err_host : 0;
    GVC_REQUIRE(dev_err != 950 && dev_err != 951, GVC_ERR_STATE,
                "a decode step ran with a full %s (max_seq %d, max_mel_pos %d): its position was not advanced; reset the slot",
                dev_err == 950 ? "KV cache" : "mel position table", c->dm.max_seq, c->dm.max_mel_pos);
    if (dev_err != 0) {
        // A hand-off of a one-launch step timed out: not all 256 workgroups were resident (another process or stream held CUs).
        // The call that timed out has produced garbage and this call reports it ONCE; the context itself stays usable -- both
        // one-launch steps are switched off, the captured step graphs (which contain them) are dropped, the hand-off buffers are
        // re-initialised -- and every later call runs on the launch-per-phase paths.
        (void)hipDeviceSynchronize();
        c->persist = 0;
        for (auto& kvp : c->graphs) (void)hipGraphExecDestroy(kvp.second);
        c->graphs.clear();
        if (c->r_bufs) (void)hipMemset(c->r_bufs, 0xff, c->act_bf16 ? rows_b16_buf_bytes() : rows_buf_bytes());
        // (the step epoch p_epoch[0] stays MONOTONIC: granule tags are (epoch + 1, layer, phase) and nothing zeroes the granules, so an epoch
        //  that started over could accept a stale granule of the timed-out call; only the arrival counter and the per-XCD ranks are cleared)
        if (c->p_epoch) (void)hipMemset(c->p_epoch + 1, 0, 15 * sizeof(unsigned));
        *c->seam_err_host = 0;
        c->fallbacks += 1;
        set_error("an in-kernel hand-off of a one-launch decode step timed out (code %d: were all 256 workgroups resident?); the outputs "
                  "of the previous decode / generate / cached-prefill call are invalid.  The context has switched to the launch-per-phase "
                  "paths and stays usable: reset the affected slots and repeat the call", dev_err);
        return GVC_ERR_TIMEOUT;
    }
    GVC_REQUIRE(gvc_gpt_missing_weights(c) == 0, GVC_ERR_STATE, "%d GPT weight tensors are not bound",
                gvc_gpt_missing_weights(c));
    return GVC_OK;
}

extern "C" int gvc_gpt_decode_step(gvc_gpt* c, const int32_t* slots, int32_t B, const int32_t* tok_in,
                                   float* logits_out, float* latent_out, gvc_stream sv) {
    int rc = check_ready(c);
    if (rc) return rc;
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_slots, GVC_ERR_ARG, "decode_step: B=%d outside [1,%d]", B, c->dm.max_slots);
    hipStream_t s = (hipStream_t)sv;
    if (persist_ok(c, B) && (rc = persist_prepare(c))) return rc;
    if (persist_ok(c, B)) return launch_persist(c, slots, tok_in, logits_out, latent_out, nullptr, s);
    if (rows_persist_ok(c, B, c->st.seq_len) && (rc = rows_persist_prepare(c))) return rc;
    c->rows_keys_hint = c->dm.max_seq;            // (no bound from the caller: the key split for the longest possible context)
    if (rows_decode_ok(c, B)) return decode_rows(c, slots, B, tok_in, logits_out, latent_out, nullptr, s);
    for (int g = 0; g < B; g += 8) {
        const int Bg = B - g < 8 ? B - g : 8;
        if ((rc = decode_group(c, slots + g, Bg, g, tok_in + g, logits_out + (size_t)g * c->dm.vocab,
                               latent_out + (size_t)g * c->dm.d_model, nullptr, s)))
            return rc;
    }
    return GVC_OK;
}

extern "C" int gvc_gpt_reset_slots(gvc_gpt* c, const int32_t* slots, int32_t B, gvc_stream sv) {
    GVC_REQUIRE(c && slots && B >= 1, GVC_ERR_ARG, "reset_slots: bad argument");
    if (c->seam_err_host && (*c->seam_err_host == 950 || *c->seam_err_host == 951)) *c->seam_err_host = 0;     // overflow acknowledged
    hipLaunchKernelGGL(k_set_state, dim3(cdiv(B, 64)), dim3(64), 0, (hipStream_t)sv, c->st, slots, B, 0, 0);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

extern "C" int gvc_gpt_prefix_embeddings(gvc_gpt* c, const float* cond, int32_t n_cond, const int32_t* codes,
                                         int32_t B, int32_t Tc, int32_t start_text, int32_t stop_text, float* out,
                                         gvc_stream sv) {
    int rc = check_ready(c);
    if (rc) return rc;
    GVC_REQUIRE(Tc + 2 <= c->dm.max_text_pos, GVC_ERR_ARG, "prefix: %d content codes exceed text positions", Tc);
    const int P = n_cond + Tc + 2;
    hipLaunchKernelGGL(k_prefix_rows, dim3(B * P), dim3(256), 0, (hipStream_t)sv, out, cond, n_cond, codes, B, Tc,
                       c->dm.d_model, c->text_emb, c->text_pos, start_text, stop_text);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

// block stack over B*T rows already in c->x; K/V of every row go to the slots' cache (positions 0..T-1, or
// base_len[slot] + 0..T-1 when the rows continue cached sequences: the batched decode step is T = 1).
// rows <= 128 (a streaming prefill): skinny MFMA GEMMs; the N = d projections are K-split over 4 workgroup
// rows and their raw partial sums are folded into the NEXT LayerNorm launch (k_ln_sum_rows), so a layer is
// 7 launches.  Larger row counts (batched offline prefill, latent re-pass) use the tiled GEMM.
static int run_rows(gvc_gpt* c, const int32_t* slots, int B, int T, hipStream_t s, const int32_t* base_len, int key_chunks) {
    const int d = c->dm.d_model, rows = B * T;
    int rc;
    // 2..16 rows that continue cached sequences (a batched decode step, the uncached rows of a streaming chunk): ONE launch
    if (c->r_ready == 1 && rows_persist_ok(c, rows, base_len))
        return launch_rows_persist(c, slots, rows, T, base_len, rows_persist_chunks(c, rows, c->rows_keys_hint), s);
    const bool skinny = c->skinny_prefill && c->wfm && rows <= 128 && d % 256 == 0 && (long long)4 * rows * d <= c->work_cap / 2;
    const int SKP = 4;
    float* part_proj = c->work;                                    // [SKP][rows][d]
    float* part_p2 = c->work + c->work_cap / 2;                    // [SKP][rows][d]
    int ln_rc = GVC_OK;          // (a lambda cannot return through GVC_REQUIRE: the first failure is kept and returned below)
    // batched decode over long contexts (key_chunks > 1, from the caller's key bound): the keys of a (stream, head) are split over
    // 2 or 4 workgroups of k_attention and the chunk partials are merged while the attn c_proj GEMM loads its A fragments
    const int att_nc = (skinny && base_len && T == 1 && rows <= 32 && c->hd == 256 && (key_chunks == 2 || key_chunks == 4) &&
                        rows <= c->dm.max_slots) ? key_chunks : 1;
    // <= fuse_ln_rows rows (8: a batched decode step of <= 8 streams; beyond, the prologue's L2 traffic -- rows x 6 planes in every
    // workgroup -- costs more than the two launches it saves): the row completion + LayerNorm runs
    // in the prologue of the QKV / c_fc GEMMs (5 launches per layer instead of 7); the residual stream ping-pongs between
    // c->x and c->xalt because only workgroup 0 of a launch writes the completed rows while the others still read them
    if (skinny && rows <= c->fuse_ln_rows && c->fuse_ln && d <= 1024) {
        float* X[2] = {c->x, c->xalt};
        int cur = 0;
        for (int l = 0; l < c->dm.n_layer; ++l) {
            const GptLayer& ly = c->layers[l];
            GemmArgs G;
            LnFuse P;
            memset(&G, 0, sizeof(G));
            memset(&P, 0, sizeof(P));
            P.x_in = X[cur]; P.ln_w = ly.ln1_w; P.ln_b = ly.ln1_b; P.rows = rows;
            if (l > 0) { P.part = part_p2; P.SK = SKP; P.pbias = c->layers[l - 1].p2_b; P.x_out = X[cur ^ 1]; }
            G.Wt = ly.qkv_f; G.w_bf16 = c->bf16; G.ldw = d; G.C = c->q; G.ldc = d; G.M = rows; G.N = 3 * d; G.K = d;
            G.e.bias = ly.qkv_b; G.e.qkv = 1; G.e.d = d; G.e.n_head = c->dm.n_head;
            G.e.head_dim = c->hd; G.e.max_seq = c->dm.max_seq; G.e.T = T; G.e.slots = slots; G.e.base_len = base_len;
            G.e.kcache = kv_layer(c, l, 0);
            G.e.vcache = kv_layer(c, l, 1);
            G.e.kv_bf16 = c->kv_bf16;
            if ((rc = launch_gemm_skinny_ln(G, P, s))) return rc;
            if (l > 0) cur ^= 1;

            AttnArgs At = gpt_attn_args(c, l, slots);
            At.q = c->q; At.T = T; At.base_len = base_len;
            At.out = c->a; At.out_stride = d; At.out_fm16 = 1;
            if (att_nc > 1) {
                At.out = c->part;
                if ((rc = launch_attention(c, At, att_nc, rows, false, s, true))) return rc;
            } else if ((rc = launch_attention(c, At, 1, rows, true, s, base_len && T == 1))) return rc;

            memset(&G, 0, sizeof(G));
            G.A = c->a; G.lda = d; G.Wt = ly.proj_f; G.w_bf16 = c->bf16; G.ldw = d; G.C = c->x; G.ldc = d; G.M = rows; G.N = d; G.K = d; G.work = part_proj;
            if (att_nc > 1) { G.att_part = c->part; G.att_nc = att_nc; G.att_heads = c->dm.n_head; G.att_hd = c->hd; }
            if ((rc = launch_gemm_skinny(G, SKP, c->work_cap / 2, s))) return rc;

            memset(&G, 0, sizeof(G));
            memset(&P, 0, sizeof(P));
            P.x_in = X[cur]; P.x_out = X[cur ^ 1]; P.part = part_proj; P.SK = SKP; P.pbias = ly.proj_b;
            P.ln_w = ly.ln2_w; P.ln_b = ly.ln2_b; P.rows = rows;
            G.Wt = ly.fc_f; G.w_bf16 = c->bf16; G.ldw = d; G.C = c->h; G.ldc = 4 * d; G.M = rows; G.N = 4 * d; G.K = d;
            G.e.bias = ly.fc_b; G.e.act = ACT_GELU_NEW; G.e.c_fm16 = 1;
            if ((rc = launch_gemm_skinny_ln(G, P, s))) return rc;
            cur ^= 1;

            memset(&G, 0, sizeof(G));
            G.A = c->h; G.lda = 4 * d; G.Wt = ly.p2_f; G.w_bf16 = c->bf16; G.ldw = 4 * d; G.C = c->x; G.ldc = d; G.M = rows; G.N = d; G.K = 4 * d; G.work = part_p2;
            if ((rc = launch_gemm_skinny(G, SKP, c->work_cap / 2, s))) return rc;
        }
        // fold the last layer's mlp partials into the residual stream, landing in c->x (what the head reads)
        return launch_ln_sum_rows(X[cur], c->x, c->a, part_p2, SKP, c->layers[c->dm.n_layer - 1].p2_b, rows, d, nullptr, nullptr, 1, s);
    }
    // more rows than the skinny kernels take: the strip GEMM (gemm.hip) on the same fragment-major operands; the N = d projections
    // leave raw K-split partials for the next LayerNorm launch exactly as the skinny path does (the split is the launcher's choice)
    const bool strip = !skinny && c->strip_prefill && c->skinny_prefill && c->wfm && d % 256 == 0 && d <= 1024 &&
                       (long long)rows * d <= c->work_cap / 2;
    const bool fm = skinny || strip;
    auto gemm_fm = [&](GemmArgs& G, int sk_max, int raw, int* sk_used) -> int {
        if (strip) return launch_gemm_strip(G, sk_max, sk_max > 1 ? c->work_cap / 2 : c->work_cap, raw, sk_used, s);
        if (sk_used) *sk_used = sk_max;
        return launch_gemm_skinny(G, sk_max, sk_max > 1 ? c->work_cap / 2 : c->work_cap, s);
    };
    auto ln_fm = [&](const float* w, const float* b) {
        hipLaunchKernelGGL(k_ln_rows, dim3(cdiv(rows, 4)), dim3(256), 0, s, c->x, c->a, rows, d, w, b, (const float*)nullptr,
                           (const float*)nullptr, fm ? 1 : 0);
    };
    auto ln_sum_sk = [&](const float* part, int SK, const float* bias, const float* w, const float* b) {
        const int r = launch_ln_sum_rows(c->x, c->x, c->a, part, SK, bias, rows, d, w, b, 1, s);
        if (ln_rc == GVC_OK) ln_rc = r;
    };
    int sk_p2 = SKP, sk_proj = SKP;
    for (int l = 0; l < c->dm.n_layer; ++l) {
        const GptLayer& ly = c->layers[l];
        if (fm && l > 0) ln_sum_sk(part_p2, sk_p2, c->layers[l - 1].p2_b, ly.ln1_w, ly.ln1_b);
        else ln_fm(ly.ln1_w, ly.ln1_b);
        GVC_LAUNCH_CHECK();
        GemmArgs G;
        memset(&G, 0, sizeof(G));
        G.A = c->a; G.lda = d; G.Wt = fm ? ly.qkv_f : ly.qkv_w; G.w_bf16 = fm && c->bf16; G.ldw = d; G.C = c->q; G.ldc = d; G.M = rows; G.N = 3 * d; G.K = d;
        G.work = c->work; G.e.bias = ly.qkv_b; G.e.qkv = 1; G.e.d = d; G.e.n_head = c->dm.n_head;
        G.e.head_dim = c->hd; G.e.max_seq = c->dm.max_seq; G.e.T = T; G.e.slots = slots; G.e.base_len = base_len;
        G.e.kcache = kv_layer(c, l, 0);
        G.e.vcache = kv_layer(c, l, 1);
        G.e.kv_bf16 = c->kv_bf16;
        if ((rc = fm ? gemm_fm(G, 1, 0, nullptr) : launch_gemm_cap(G, 1, c->work_cap, s))) return rc;

        AttnArgs At = gpt_attn_args(c, l, slots);
        At.q = c->q; At.T = T; At.base_len = base_len;
        At.out = c->a; At.out_stride = d; At.out_fm16 = fm ? 1 : 0;
        // one new row per cached stream (batched decode): 16 waves share the keys of a (row, head)
        if (att_nc > 1) {
            At.out = c->part;
            if ((rc = launch_attention(c, At, att_nc, rows, false, s, true))) return rc;
        } else if ((rc = launch_attention(c, At, 1, rows, true, s, base_len && T == 1))) return rc;

        memset(&G, 0, sizeof(G));
        G.A = c->a; G.lda = d; G.Wt = fm ? ly.proj_f : ly.proj_w; G.w_bf16 = fm && c->bf16; G.ldw = d; G.C = c->x; G.ldc = d; G.M = rows; G.N = d; G.K = d;
        if (fm) {
            G.work = part_proj;
            if (att_nc > 1) { G.att_part = c->part; G.att_nc = att_nc; G.att_heads = c->dm.n_head; G.att_hd = c->hd; }
            if ((rc = gemm_fm(G, strip ? 8 : SKP, 1, &sk_proj))) return rc;
            ln_sum_sk(part_proj, sk_proj, ly.proj_b, ly.ln2_w, ly.ln2_b);
        } else {
            G.work = c->work; G.e.bias = ly.proj_b; G.e.resid = c->x; G.e.ldr = d;
            if ((rc = launch_gemm_cap(G, 1, c->work_cap, s))) return rc;
            ln_fm(ly.ln2_w, ly.ln2_b);
        }
        GVC_LAUNCH_CHECK();
        memset(&G, 0, sizeof(G));
        G.A = c->a; G.lda = d; G.Wt = fm ? ly.fc_f : ly.fc_w; G.w_bf16 = fm && c->bf16; G.ldw = d; G.C = c->h; G.ldc = 4 * d; G.M = rows; G.N = 4 * d; G.K = d;
        G.work = c->work; G.e.bias = ly.fc_b; G.e.act = ACT_GELU_NEW; G.e.c_fm16 = fm ? 1 : 0;
        if ((rc = fm ? gemm_fm(G, 1, 0, nullptr) : launch_gemm_cap(G, 1, c->work_cap, s))) return rc;

        memset(&G, 0, sizeof(G));
        G.A = c->h; G.lda = 4 * d; G.Wt = fm ? ly.p2_f : ly.p2_w; G.w_bf16 = fm && c->bf16; G.ldw = 4 * d; G.C = c->x; G.ldc = d; G.M = rows; G.N = d; G.K = 4 * d;
        if (fm) {
            G.work = part_p2;
            if ((rc = gemm_fm(G, strip ? 8 : SKP, 1, &sk_p2))) return rc;
        } else {
            G.work = c->work; G.e.bias = ly.p2_b; G.e.resid = c->x; G.e.ldr = d;
            if ((rc = launch_gemm_cap(G, 1, c->work_cap, s))) return rc;
        }
    }
    if (fm) {       // fold the last layer's mlp partials into the residual stream
        ln_sum_sk(part_p2, sk_p2, c->layers[c->dm.n_layer - 1].p2_b, nullptr, nullptr);
        GVC_LAUNCH_CHECK();
    }
    return ln_rc;
}

extern "C" int gvc_gpt_prefill(gvc_gpt* c, const int32_t* slots, int32_t B, const float* prefix_emb, int32_t P,
                               int32_t start_tok, float* logits_out, float* latent_out, gvc_stream sv) {
    return gvc_gpt_prefill_cached(c, slots, B, prefix_emb, P, 0, start_tok, logits_out, latent_out, sv);
}

extern "C" int gvc_gpt_prefill_cached(gvc_gpt* c, const int32_t* slots, int32_t B, const float* prefix_emb, int32_t P,
                                      int32_t n_cached, int32_t start_tok, float* logits_out, float* latent_out, gvc_stream sv) {
    int rc = check_ready(c);
    if (rc) return rc;
    const int T = P + 1, d = c->dm.d_model;
    GVC_REQUIRE(n_cached >= 0 && n_cached <= P, GVC_ERR_ARG, "prefill: n_cached=%d outside [0,%d]", n_cached, P);
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_slots, GVC_ERR_ARG, "prefill: B=%d outside [1,%d]", B, c->dm.max_slots);
    GVC_REQUIRE(P >= 0 && B * T <= c->dm.max_rows, GVC_ERR_ARG, "prefill: %d rows exceed max_rows %d", B * T, c->dm.max_rows);
    GVC_REQUIRE(T < c->dm.max_seq, GVC_ERR_ARG, "prefill: %d rows exceed max_seq %d", T, c->dm.max_seq);
    hipStream_t s = (hipStream_t)sv;
    if (!logits_out) logits_out = c->logits;      // staging of the generation loop
    if (!latent_out) latent_out = c->latent;
    hipLaunchKernelGGL(k_embed_rows, dim3(B * (T - n_cached)), dim3(256), 0, s, c->x, prefix_emb, B, T, P, d, c->mel_emb, c->mel_pos,
                       (const int32_t*)nullptr, 0, start_tok, start_tok, n_cached);
    GVC_LAUNCH_CHECK();
    const int Tn = T - n_cached;
    if (n_cached > 0 && rows_persist_ok(c, B * Tn, c->st.seq_len) && (rc = rows_persist_prepare(c))) return rc;
    c->rows_keys_hint = T;
    if (n_cached > 0) {
        // the rows continue the cached prefix: K/V go to positions n_cached.., attention sees [0, n_cached + t]
        hipLaunchKernelGGL(k_set_state, dim3(cdiv(B, 64)), dim3(64), 0, s, c->st, slots, B, n_cached, 0);
        GVC_LAUNCH_CHECK();
        if ((rc = run_rows(c, slots, B, Tn, s, c->st.seq_len))) return rc;
    } else if ((rc = run_rows(c, slots, B, T, s))) return rc;
    if ((rc = launch_head(c, slots, B, 0, c->x, Tn, Tn - 1, logits_out, latent_out, 0, nullptr, s))) return rc;
    hipLaunchKernelGGL(k_set_state, dim3(cdiv(B, 64)), dim3(64), 0, s, c->st, slots, B, T, 1);
    GVC_LAUNCH_CHECK();
    // park the next-step logits / latent per slot: gvc_gpt_generate continues every slot from here, whoever received them
    hipLaunchKernelGGL(k_stage_rows, dim3(B), dim3(256), 0, s, logits_out, c->slot_logits, slots, c->dm.vocab, 1);
    hipLaunchKernelGGL(k_stage_rows, dim3(B), dim3(256), 0, s, latent_out, c->slot_latent, slots, d, 1);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

// The conditioning rows alone (no text rows, no start token, no head): K/V of rows 0..n_cond-1 into the slots' caches, length n_cond.
// For callers that know the target speaker before the first source segment arrives (a streaming session: the reference recomputes these
// rows inside every segment's prefill, inference_utils.py:43-66): every segment, the first one too, then takes gvc_gpt_prefill_cached with
// n_cached = n_cond.  Same kernels and per-row arithmetic as the rows of a full prefill (each row's projections depend on that row only,
// its attention on the rows in front of it).
extern "C" int gvc_gpt_prefill_cond(gvc_gpt* c, const int32_t* slots, int32_t B, const float* cond_latents, int32_t n_cond, gvc_stream sv) {
    int rc = check_ready(c);
    if (rc) return rc;
    const int d = c->dm.d_model;
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_slots && n_cond >= 1 && B * n_cond <= c->dm.max_rows && n_cond < c->dm.max_seq, GVC_ERR_ARG,
                "prefill_cond: bad B=%d n_cond=%d", B, n_cond);
    hipStream_t s = (hipStream_t)sv;
    hipLaunchKernelGGL(k_embed_rows, dim3(B * n_cond), dim3(256), 0, s, c->x, cond_latents, B, n_cond, n_cond, d, c->mel_emb, c->mel_pos,
                       (const int32_t*)nullptr, 0, 0, 0, 0);
    GVC_LAUNCH_CHECK();
    c->rows_keys_hint = n_cond;
    if ((rc = run_rows(c, slots, B, n_cond, s))) return rc;
    hipLaunchKernelGGL(k_set_state, dim3(cdiv(B, 64)), dim3(64), 0, s, c->st, slots, B, n_cond, 0);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

extern "C" int gvc_gpt_latents(gvc_gpt* c, const int32_t* slots, int32_t B, const float* prefix_emb, int32_t P,
                               const int32_t* gen_codes, int32_t n, int32_t start_tok, int32_t stop_tok, float* out,
                               gvc_stream sv) {
    int rc = check_ready(c);
    if (rc) return rc;
    const int T = P + n + 5, d = c->dm.d_model;
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_slots && n >= 1, GVC_ERR_ARG, "latents: bad B=%d n=%d", B, n);
    GVC_REQUIRE(B * T <= c->dm.max_rows, GVC_ERR_ARG, "latents: %d rows exceed max_rows %d", B * T, c->dm.max_rows);
    GVC_REQUIRE(T < c->dm.max_seq && n + 5 <= c->dm.max_mel_pos, GVC_ERR_ARG, "latents: sequence too long");
    hipStream_t s = (hipStream_t)sv;
    hipLaunchKernelGGL(k_embed_rows, dim3(B * T), dim3(256), 0, s, c->x, prefix_emb, B, T, P, d, c->mel_emb, c->mel_pos,
                       gen_codes, n, start_tok, stop_tok, 0);
    GVC_LAUNCH_CHECK();
    if ((rc = run_rows(c, slots, B, T, s))) return rc;
    hipLaunchKernelGGL(k_ln_rows, dim3(cdiv(B * T, 4)), dim3(256), 0, s, c->x, c->a, B * T, d, c->lnf_w, c->lnf_b,
                       c->fn_w, c->fn_b, 0);
    GVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gather_rows, dim3(B * n), dim3(256), 0, s, c->a, out, B, T, P, n, d);
    GVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_set_state, dim3(cdiv(B, 64)), dim3(64), 0, s, c->st, slots, B, 0, 0);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

// ---------------------------------------------------------------------------------------------
// generation loop: one captured graph = [sample -> decode step] for a fixed B, replayed n_steps times
// ---------------------------------------------------------------------------------------------
static int build_step_graph(gvc_gpt* c, int B, bool fused, int key_chunks, int n_unroll, bool greedy, hipGraphExec_t* out) {
    hipStream_t cs = c->cap_stream;
    int rc = GVC_OK;
    GVC_CHECK_HIP(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    // n_unroll consecutive steps in ONE graph: the boundary between two graph launches costs several times the boundary between
    // two kernels of a graph, and the host looks at the finished flags once per group of steps anyway
    for (int u = 0; u < n_unroll && rc == GVC_OK; ++u) {
    rc = launch_sample_indirect(&c->gen_call->sc, B, greedy, cs);
    if (rc == GVC_OK && persist_ok(c, B))
        rc = launch_persist(c, c->gen_call->slots, c->tok_buf, c->logits, c->latent, c->step_ctr, cs);
    else if (rc == GVC_OK && rows_decode_ok(c, B))
        rc = decode_rows(c, c->gen_call->slots, B, c->tok_buf, c->logits, c->latent, c->step_ctr, cs, key_chunks);
    else
    for (int g = 0; g < B && rc == GVC_OK; g += 8) {
        const int Bg = B - g < 8 ? B - g : 8;
        const bool last = g + 8 >= B;
        rc = decode_group(c, c->gen_call->slots + g, Bg, g, c->tok_buf + g, c->logits + (size_t)g * c->dm.vocab,
                          c->latent + (size_t)g * c->dm.d_model, last ? c->step_ctr : nullptr, cs, fused);
    }
    }
    hipGraph_t graph = nullptr;
    hipError_t e = hipStreamEndCapture(cs, &graph);
    if (rc != GVC_OK) {
        if (graph) hipGraphDestroy(graph);
        return rc;
    }
    GVC_CHECK_HIP(e);
    e = hipGraphInstantiate(out, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    GVC_CHECK_HIP(e);
    return GVC_OK;
}

// What a gvc_gpt_generate call over B streams that reaches `key_bound` cached positions replays: the step variant, its key split and the
// key of its captured graph.  Runs the one-time preparation of the one-launch steps first (they may switch a path off for this context --
// r_ready / persist -- so nothing is derived from rows_decode_ok / persist_ok before).
struct GenPlan {
    bool fused, greedy;
    int key_chunks, key, variant;
};

static int step_unroll() {
    // the steps of a call run as graphs of kStepUnroll consecutive steps, the remainder one by one
    return 8;   // (<= 32: bits 24..29 of the graph key)
}

static int plan_generate(gvc_gpt* c, int B, int key_bound, int top_k, GenPlan* pl) {
    int rc;
    pl->fused = fused_ok(c, B, key_bound);
    // rows mode splits the keys of long contexts over 2 / 4 attention workgroups per (stream, head): two from GVC_ROWS_KEY_SPLIT
    // cached positions on (default 144; 0: never), four beyond 320
    constexpr int key_split = 144;
    if (persist_ok(c, B) && (rc = persist_prepare(c))) return rc;
    if (!persist_ok(c, B) && rows_persist_ok(c, B, c->st.seq_len) && (rc = rows_persist_prepare(c))) return rc;
    pl->key_chunks = (key_split > 0 && rows_decode_ok(c, B) && !persist_ok(c, B) && B <= 32)
                         ? (key_bound > 320 ? 4 : (key_bound > key_split ? 2 : 1)) : 1;
    const bool rows1 = !persist_ok(c, B) && c->r_ready == 1 && rows_persist_ok(c, B, c->st.seq_len);      // one-launch rows step
    c->rows_keys_hint = key_bound;
    pl->key = B * 2 + (pl->fused ? 1 : 0) + 4096 * (rows1 ? 8 + rows_persist_chunks(c, B, key_bound) : pl->key_chunks);   // (the one-launch steps are pure functions of B [and the key split])
    pl->variant = persist_ok(c, B) ? 3 : (rows1 ? 5 : (rows_decode_ok(c, B) ? 4 : (pl->fused ? 2 : 1)));
    // (the sampler kernel is chosen at capture time: a graph serves top_k = 1 or everything else)
    pl->greedy = sample_greedy_ok(top_k, c->dm.d_model);
    return GVC_OK;
}

static int step_graph(gvc_gpt* c, int B, const GenPlan& pl, int unroll, hipGraphExec_t* ge) {
    const int k = pl.key + (unroll > 1 ? (1 << 24) * unroll : 0) + (pl.greedy ? (1 << 30) : 0);
    auto it = c->graphs.find(k);
    if (it == c->graphs.end()) {
        hipGraphExec_t g1;
        note_lazy(c);
        int r = build_step_graph(c, B, pl.fused, pl.key_chunks, unroll, pl.greedy, &g1);
        if (r) return r;
        it = c->graphs.emplace(k, g1).first;
    }
    *ge = it->second;
    return GVC_OK;
}

// Everything a later gvc_gpt_generate / gvc_gpt_decode_step / gvc_gpt_prefill_cached call of this shape would otherwise do on first use:
// the one-launch steps' buffers, weight pack, topology probe and LDS opt-in, and the captured step graphs (eight steps and one step) for
// B streams reaching `max_keys` cached positions with this top_k.  Synchronous; call it after the weights are bound, once per shape a
// deployment uses.  After it, data-path calls of that shape neither allocate nor synchronise the device (gvc_gpt_lazy_inits stays put).
extern "C" int gvc_gpt_warmup(gvc_gpt* c, int32_t B, int32_t max_keys, int32_t top_k) {
    int rc = check_ready(c);
    if (rc) return rc;
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_slots && max_keys >= 0 && max_keys < c->dm.max_seq, GVC_ERR_ARG, "warmup: bad argument");
    c->in_warmup = 1;
    struct Leave { gvc_gpt* c; ~Leave() { c->in_warmup = 0; } } leave{c};
    // the rows step also serves the <= 16 uncached rows of a cached chunk prefill, whatever B
    if (rows_persist_ok(c, 16, c->st.seq_len) && (rc = rows_persist_prepare(c))) return rc;
    GenPlan pl;
    if ((rc = plan_generate(c, B, max_keys > 0 ? max_keys : c->dm.max_seq - 1, top_k, &pl))) return rc;
    hipGraphExec_t ge;
    if (step_unroll() > 1 && (rc = step_graph(c, B, pl, step_unroll(), &ge))) return rc;
    if ((rc = step_graph(c, B, pl, 1, &ge))) return rc;
    GVC_CHECK_HIP(hipDeviceSynchronize());
    return GVC_OK;
}

extern "C" int gvc_gpt_warmup_range(gvc_gpt* c, int32_t B, int32_t min_keys, int32_t max_keys, int32_t top_k) {
    GVC_REQUIRE(c, GVC_ERR_ARG, "null context");
    GVC_REQUIRE(min_keys >= 1 && min_keys <= max_keys && max_keys < c->dm.max_seq, GVC_ERR_ARG, "warmup_range: bad key range [%d, %d]", min_keys, max_keys);
    int rc = gvc_gpt_warmup(c, B, min_keys, top_k);          // (also prepares the one-launch steps: the plans below are then pure look-ups)
    if (rc) return rc;
    c->in_warmup = 1;
    struct Leave { gvc_gpt* c; ~Leave() { c->in_warmup = 0; } } leave{c};
    int last_key = -1;
    for (int mk = min_keys; mk <= max_keys; ++mk) {
        GenPlan pl;
        if ((rc = plan_generate(c, B, mk, top_k, &pl))) return rc;
        if (pl.key == last_key) continue;
        last_key = pl.key;
        hipGraphExec_t ge;
        if (step_unroll() > 1 && (rc = step_graph(c, B, pl, step_unroll(), &ge))) return rc;
        if ((rc = step_graph(c, B, pl, 1, &ge))) return rc;
    }
    GVC_CHECK_HIP(hipDeviceSynchronize());
    return GVC_OK;
}

extern "C" long long gvc_gpt_lazy_inits(gvc_gpt* c) { return c ? c->lazy_inits : 0; }

// After a hand-off time-out the context runs on the launch-per-phase paths.  That is the right answer while another context holds CUs, and
// the wrong one for the rest of a server's life: when the caller knows the GPU is its own again (a quiet moment, the other process gone) it
// re-arms the one-launch steps here.  Synchronises the device, drops the captured step graphs (they hold the fallback launches) and resets
// the hand-off state; a later time-out simply falls back again.  Pending errors are reported first, exactly as by gvc_gpt_health.
extern "C" int gvc_gpt_rearm(gvc_gpt* c) {
    int rc = check_ready(c);
    if (rc) return rc;
    if (c->persist == c->persist_cfg) return GVC_OK;
    GVC_CHECK_HIP(hipDeviceSynchronize());
    for (auto& kvp : c->graphs) (void)hipGraphExecDestroy(kvp.second);
    c->graphs.clear();
    if (c->r_bufs) GVC_CHECK_HIP(hipMemset(c->r_bufs, 0xff, c->act_bf16 ? rows_b16_buf_bytes() : rows_buf_bytes()));
    if (c->p_epoch) GVC_CHECK_HIP(hipMemset(c->p_epoch + 1, 0, 15 * sizeof(unsigned)));      // (epoch [0] stays monotonic, see check_ready)
    GVC_CHECK_HIP(hipDeviceSynchronize());
    c->persist = c->persist_cfg;
    return GVC_OK;
}

extern "C" int gvc_gpt_generate(gvc_gpt* c, const int32_t* slots, int32_t B, int32_t* ids, int32_t ids_stride,
                                int32_t* ids_len, int32_t* finished, const gvc_sample_params* p, int32_t i0,
                                int32_t n_steps, int32_t max_keys, int32_t* tokens_out, int32_t tok_stride, float* latents_out,
                                int32_t lat_stride, gvc_stream sv) {
    int rc = check_ready(c);
    if (rc) return rc;
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_slots && p && n_steps >= 0 && max_keys >= 0, GVC_ERR_ARG, "generate: bad argument");
    // cached positions of the longest stream once this call has run: the caller's bound, else the whole ids row
    const int key_bound = max_keys > 0 ? max_keys : ids_stride;
    GVC_REQUIRE(max_keys == 0 || max_keys < c->dm.max_seq, GVC_ERR_STATE,
                "generate: %d cached positions would overflow the KV cache (max_seq %d)", max_keys, c->dm.max_seq);
    GVC_REQUIRE(p->vocab == c->dm.vocab, GVC_ERR_ARG, "generate: vocab mismatch");
    hipStream_t s = (hipStream_t)sv;
    SampleCall sc;
    memset(&sc, 0, sizeof(sc));
    sc.logits = c->logits; sc.B = B; sc.ids = ids; sc.ids_stride = ids_stride; sc.ids_len = ids_len;
    sc.finished = finished; sc.p = *p; sc.step = 0; sc.step_ptr = c->step_ctr; sc.tok_out = c->tok_buf;
    sc.tokens_out = tokens_out; sc.tok_stride = tok_stride; sc.i0 = i0; sc.latent_src = c->latent;
    sc.latents_out = latents_out; sc.lat_stride = lat_stride; sc.d = c->dm.d_model;
    hipLaunchKernelGGL(k_gen_begin, dim3(B + 1), dim3(256), 0, s, c->gen_call, sc, slots, B, c->step_ctr, c->logits, c->slot_logits,
                       c->dm.vocab, c->latent, c->slot_latent, c->dm.d_model);
    GVC_LAUNCH_CHECK();
    GenPlan pl;
    if ((rc = plan_generate(c, B, key_bound, p->top_k, &pl))) return rc;
    c->last_variant = pl.variant;
    const int kStepUnroll = step_unroll();
    int left = n_steps;
    if (kStepUnroll > 1 && left >= kStepUnroll) {
        hipGraphExec_t ge;
        if ((rc = step_graph(c, B, pl, kStepUnroll, &ge))) return rc;
        for (; left >= kStepUnroll; left -= kStepUnroll) GVC_CHECK_HIP(hipGraphLaunch(ge, s));
    }
    if (left > 0) {
        hipGraphExec_t ge;
        if ((rc = step_graph(c, B, pl, 1, &ge))) return rc;
        for (; left > 0; --left) GVC_CHECK_HIP(hipGraphLaunch(ge, s));
    }
    hipLaunchKernelGGL(k_gen_end, dim3(B), dim3(256), 0, s, slots, c->logits, c->slot_logits, c->dm.vocab, c->latent, c->slot_latent,
                       c->dm.d_model);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}


// ---------------------------------------------------------------------------------------------
// measurement hook (bench.py roofline): launch ONLY one kernel class of the decode step (0 qkv, 1 attention,
// 2 attn c_proj, 3 mlp c_fc, 4 mlp c_proj, 5 head), layer after layer as the step does (so every launch
// streams different weights, 1.5 GB apart in total: no cache reuse), n_steps times back to back on the
// caller's stream between two hipEvents; returns the mean microseconds per launch.  HIP events resolve
// single ~5 us kernels poorly (an empty event pair already measures ~4.7 us), so the mean over a long
// back-to-back run is used; it includes the same-stream launch boundary, as rocprofv3's per-kernel
// durations summed over a decode step do.  Synchronises the stream.
// ---------------------------------------------------------------------------------------------
extern "C" int gvc_gpt_time_kernel(gvc_gpt* c, int32_t which, const int32_t* slots, int32_t B, const int32_t* tok_in,
                                   int32_t n_steps, float* avg_us, int32_t* n_launches, gvc_stream sv) {
    int rc = check_ready(c);
    if (rc) return rc;
    // which 0..5: that class alone; 6: the whole step; 16 + X: the whole step WITHOUT class X (in-situ cost of X =
    // (whole - without) / launches: the class then runs behind its real predecessor, whose output it has to fetch from
    // the other XCDs, instead of re-reading its own stale inputs from L2)
    const bool whole = which == 6 || which == 7 || (which >= 16 && which <= 21);
    GVC_REQUIRE(((which >= 0 && which <= 5) || whole) && B >= 1 && B <= 8 && n_steps >= 1 && avg_us, GVC_ERR_ARG,
                "time_kernel: bad argument");
    const bool one_launch = which == 7;
    GVC_REQUIRE(!one_launch || persist_ok(c, B), GVC_ERR_UNSUPPORTED, "time_kernel: the one-launch decode step does not serve this context");
    if (one_launch && (rc = persist_prepare(c))) return rc;
    hipStream_t s = (hipStream_t)sv;
    hipEvent_t e0, e1;
    GVC_CHECK_HIP(hipEventCreate(&e0));
    GVC_CHECK_HIP(hipEventCreate(&e1));
    // one pass over the layers with only this kernel class is captured into a graph (eager launches of ~5 us
    // kernels are host-bound) and replayed n_steps times between two events on the caller's stream
    const bool fused = fused_ok(c, B, 0);       // the short-context variant bench.py's workload runs
    c->prof_only = whole ? -1 : which;
    c->prof_skip_one = which >= 16 ? which - 16 : -1;
    hipGraph_t graph = nullptr;
    hipGraphExec_t ge = nullptr;
    hipError_t e = hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal);
    if (e == hipSuccess) {
        rc = one_launch ? launch_persist(c, slots, tok_in, c->logits, c->latent, nullptr, c->cap_stream)
                        : decode_group(c, slots, B, 0, tok_in, c->logits, c->latent, nullptr, c->cap_stream, fused);
        e = hipStreamEndCapture(c->cap_stream, &graph);
    }
    c->prof_only = -1;
    c->prof_skip_one = -1;
    if (e == hipSuccess && rc == GVC_OK) e = hipGraphInstantiate(&ge, graph, nullptr, nullptr, 0);
    if (graph) (void)hipGraphDestroy(graph);
    float ms = 0.f;
    if (e == hipSuccess && rc == GVC_OK) {
        (void)hipGraphLaunch(ge, s);                                   // warm-up pass
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < n_steps; ++i) (void)hipGraphLaunch(ge, s);
        (void)hipEventRecord(e1, s);
        e = hipStreamSynchronize(s);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    }
    if (ge) (void)hipGraphExecDestroy(ge);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    GVC_CHECK_HIP(e);
    const int n = (whole ? 1 : (which == 5 ? 1 : (fused && which == 2 ? 0 : c->dm.n_layer))) * n_steps;
    *avg_us = n ? ms * 1000.0f / (float)n : 0.f;
    if (n_launches) *n_launches = n;
    return GVC_OK;
}

extern "C" int gvc_gpt_decode_variant(gvc_gpt* c) { return c ? c->last_variant : 0; }
extern "C" long long gvc_gpt_rows_step_launches(gvc_gpt* c) { return c ? c->r_launches : 0; }

// To be called after the caller has synchronised the stream of a decode / generate / cached-prefill call: 0 when that work ran
// cleanly; otherwise the state error of check_ready (a timed-out hand-off has switched the context to the launch-per-phase paths,
// a full KV cache / position table).  Without it such an error surfaces on the next library call.
extern "C" int gvc_gpt_health(gvc_gpt* c) { return check_ready(c); }

// debug: copy the in-kernel timestamps of the GEMV launches since the last call (GVC_DEBUG_STAMPS=1)
extern "C" int gvc_gpt_debug_stamps(gvc_gpt* c, unsigned long long* host_out, int32_t max_launches) {
    if (c && c->r_dbg && max_launches == -2) {    // stamps of the last one-launch rows step: workgroup 0, [(layer * 5 + phase) * 4 + k]
        (void)hipDeviceSynchronize();
        const int n = 40 * (c->dm.n_layer + 2) + 8 * 5 * kPG;
        (void)hipMemcpy(host_out, c->r_dbg, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        return n;
    }
    if (c && c->p_dbg && max_launches < 0) {     // stamps of the last one-launch decode step (layout: persist_kernel.h)
        (void)hipDeviceSynchronize();
        const int n = 20 * (c->dm.n_layer + 2) + 2 * 5 * kPG + 8 * kPG;
        (void)hipMemcpy(host_out, c->p_dbg, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        return n;
    }
    if (!c || !c->dbg) return 0;
    (void)hipDeviceSynchronize();
    const int n = c->dbg_n < max_launches ? c->dbg_n : max_launches;
    (void)hipMemcpy(host_out, c->dbg, (size_t)n * 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    c->dbg_n = 0;
    return n;
}
