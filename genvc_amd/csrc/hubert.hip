// ContentVec / HuBERT-base forward (reference layers/content_processor.py:17-31 -> fairseq
// HubertModel.extract_features(output_layer=12) + final_proj; restated in oracle/genvc_oracle.py:hubert_extract_features).
//
// Everything is time-major [B][T][C], so
//   * conv layers 1..6 (no padding, stride s, k taps) are ONE fp32 MFMA GEMM each whose A rows are overlapping
//     windows of the previous activation (row t = &x[s*t][0], K = k*C contiguous, lda = s*C), GELU in the epilogue;
//   * layer 0 (C_in = 1, k = 10) is a direct kernel that also emits per-chunk GroupNorm statistics;
//   * the grouped positional conv (k = 128, 16 groups) is a batched GEMM (batch = group) over a zero-padded copy of
//     x with implicit im2col (conv_cin = E/groups, tap stride = E), bias + GELU + residual in the epilogue;
//   * transformer layers: fused-QKV GEMM, flash attention on v_mfma_f32_16x16x4_f32 (S^T = K Q^T so the probabilities
//     come out of the MFMA already laid out as the B operand of O^T = V^T P^T: no LDS transpose), out_proj/fc2 GEMMs
//     with the residual in the epilogue, row LayerNorm kernels (post-LN).
// The middle of the forward (GroupNorm apply .. last transformer layer) is captured into a hipGraph per (B, T).
#include <map>
#include <string>
#include <vector>

#include "conv_lds.h"
#include "attn64.h"
#include <algorithm>

namespace gvc {

constexpr int kC0Chunk = 32;          // output frames per conv0 workgroup (= GroupNorm partial-statistics chunk)

// Conv1d weight [Co][Ci][k] -> [Co][k*Ci] with column j*Ci + ci
static __global__ void k_hb_repack_conv(const float* w, float* out, int Co, int Ci, int k) {
    const size_t n = (size_t)Co * Ci * k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % k);
        const int ci = (int)((i / k) % Ci);
        const int co = (int)(i / ((size_t)k * Ci));
        out[((size_t)co * k + j) * Ci + ci] = w[i];
    }
}

// layer 0: wav [B][T] -> y [B][T0][C] (raw conv), part [B][nchunk][C][2] = (sum, sum of squares) over the chunk
static __global__ __launch_bounds__(256) void k_hb_conv0(const float* wav, const float* w, float* y, float* part, int T,
                                                        int T0, int C, int k, int stride) {
    extern __shared__ float xs[];
    const int b = blockIdx.z, chunk = blockIdx.x, tid = threadIdx.x;
    const int t0 = chunk * kC0Chunk;
    const int nt = min(kC0Chunk, T0 - t0);
    const int nx = (nt - 1) * stride + k;
    const float* src = wav + (size_t)b * T + (size_t)t0 * stride;
    for (int i = tid; i < nx; i += 256) xs[i] = src[i];
    __syncthreads();
    // grid (chunks, channel blocks of 256, batch): ~200 workgroups for a 1 s chunk
    for (int c = blockIdx.y * 256 + tid; c < min(C, (int)(blockIdx.y + 1) * 256); c += 256) {
        float wr[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) wr[j] = j < k ? w[(size_t)c * k + j] : 0.f;
        float s = 0.f, ss = 0.f;
        float* yo = y + ((size_t)b * T0 + t0) * C + c;
        for (int t = 0; t < nt; ++t) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (j < k) acc = fmaf(wr[j], xs[t * stride + j], acc);
            yo[(size_t)t * C] = acc;
            s += acc; ss = fmaf(acc, acc, ss);
        }
        float* po = part + (((size_t)b * gridDim.x + chunk) * C + c) * 2;
        po[0] = s; po[1] = ss;
    }
}

// GroupNorm(C groups over C channels) statistics: per (b, channel) over time; chunks combined in double.
// A workgroup owns 8 channels; 32 lanes per channel take every 32nd chunk (all their requests in flight at once) and combine with a
// fixed-order shuffle tree (two workgroups walking the chunks serially took 9 us for a 1 s chunk: 7 dependent round trips).
static __global__ __launch_bounds__(256) void k_hb_gn_stats(const float* part, float* stats, int nchunk, int T0, int C) {
    const int b = blockIdx.y, c = blockIdx.x * 8 + (threadIdx.x >> 5), l = threadIdx.x & 31;
    if (c >= C) return;                           // (whole 32-lane groups leave together: C % 8 == 0 is not required)
    double s = 0.0, ss = 0.0;
    for (int i0 = l; i0 < nchunk; i0 += 32 * 8) {
        float2 pv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            pv[u] = *reinterpret_cast<const float2*>(part + (((size_t)b * nchunk + min(i0 + 32 * u, nchunk - 1)) * C + c) * 2);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 32 * u < nchunk) { s += (double)pv[u].x; ss += (double)pv[u].y; }
    }
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) { s += __shfl_xor(s, o, 32); ss += __shfl_xor(ss, o, 32); }
    if (l) return;
    const double mean = s / T0;
    double var = ss / T0 - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[((size_t)b * C + c) * 2] = (float)mean;
    stats[((size_t)b * C + c) * 2 + 1] = (float)(1.0 / sqrt(var + 1e-5));
}

// y = gelu((y - mean) * rstd * gamma + beta), in place
static __global__ void k_hb_gn_gelu(float* y, const float* stats, const float* gamma, const float* beta, int T0, int C) {
    const int b = blockIdx.y;
    const size_t n4 = (size_t)T0 * C / 4;
    float* yb = y + (size_t)b * T0 * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)((i * 4) % C);
        float4 v = *reinterpret_cast<float4*>(yb + i * 4);
        const float* st = stats + ((size_t)b * C + c) * 2;
        v.x = gelu_erf((v.x - st[0]) * st[1] * gamma[c] + beta[c]);
        v.y = gelu_erf((v.y - st[2]) * st[3] * gamma[c + 1] + beta[c + 1]);
        v.z = gelu_erf((v.z - st[4]) * st[5] * gamma[c + 2] + beta[c + 2]);
        v.w = gelu_erf((v.w - st[6]) * st[7] * gamma[c + 3] + beta[c + 3]);
        *reinterpret_cast<float4*>(yb + i * 4) = v;
    }
}

// dst[row] = LayerNorm(src[row]) * w + b; rows of d floats (d % 4 == 0, d <= 4096); row r of batch element
// r / rows_per_batch lives at base + (r / rpb) * batch_stride + (r % rpb) * d
// part != null: the row is first completed as src + bias + sum_s part[s][row] (raw K-split partials of the skinny GEMM);
// dst_fm16 != null: the normalised row is also written in the FM16 fragment-major layout (A operand of the next GEMM)
static __global__ __launch_bounds__(256) void k_hb_ln_rows(const float* src, long long src_bs, float* dst, long long dst_bs,
                                                          int rpb, int d, const float* w, const float* b,
                                                          const float* part = nullptr, int SK = 0, const float* pbias = nullptr,
                                                          int rows = 0, float* dst_fm16 = nullptr) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int bi = row / rpb, ri = row - bi * rpb;
    const float* xr = src + bi * src_bs + (size_t)ri * d;
    float* orow = dst + bi * dst_bs + (size_t)ri * d;
    float4 v[4];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (i * 256 + tid) * 4;
        v[i] = k < d ? *reinterpret_cast<const float4*>(xr + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (part && k < d) {
            const float4 b4 = *reinterpret_cast<const float4*>(pbias + k);
            // the partial planes are requested together (a loop with a run-time trip count made each one its own round trip);
            // added in plane order, as before
            float4 p4[8];
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx)
                p4[sidx] = *reinterpret_cast<const float4*>(part + ((size_t)min(sidx, SK - 1) * rows + row) * d + k);       // (clamped, not predicated: a branch per plane would serialise the requests again)
            v[i].x += b4.x; v[i].y += b4.y; v[i].z += b4.z; v[i].w += b4.w;
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx)
                if (sidx < SK) { v[i].x += p4[sidx].x; v[i].y += p4[sidx].y; v[i].z += p4[sidx].z; v[i].w += p4[sidx].w; }
        }
        s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float inv_d = 1.0f / (float)d;
    const float mean = block4_sum(s, red) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (i * 256 + tid) * 4;
        if (k < d) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = 1.0f / sqrtf(block4_sum(q, red) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int k = (i * 256 + tid) * 4;
        if (k < d) {
            const float4 gw = *reinterpret_cast<const float4*>(w + k);
            const float4 gb = *reinterpret_cast<const float4*>(b + k);
            float4 o;
            o.x = (v[i].x - mean) * rstd * gw.x + gb.x; o.y = (v[i].y - mean) * rstd * gw.y + gb.y;
            o.z = (v[i].z - mean) * rstd * gw.z + gb.z; o.w = (v[i].w - mean) * rstd * gw.w + gb.w;
            *reinterpret_cast<float4*>(orow + k) = o;
            if (dst_fm16) *reinterpret_cast<float4*>(dst_fm16 + fm16_index(row, k, d)) = o;
        }
    }
}

// zero `front` rows before and `back` rows after the T live rows of every batch element of buf [B][front+T+back][C], and
// the live rows whose frame is padding (fairseq TransformerEncoder.extract_features: x[padding_mask] = 0 ahead of pos_conv)
static __global__ void k_hb_zero_pad(float* buf, int C, int T, int front, int back, const int32_t* fmask) {
    const int b = blockIdx.y;
    float* base = buf + (size_t)b * (front + T + back) * C;
    const size_t nf = (size_t)front * C, nb = (size_t)back * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nf + nb; i += (size_t)gridDim.x * blockDim.x) {
        if (i < nf) base[i] = 0.f;
        else base[(size_t)(front + T) * C + (i - nf)] = 0.f;
    }
    for (int t = blockIdx.x; t < T; t += gridDim.x) {
        if (!fmask[b * T + t]) continue;                       // uniform per workgroup
        for (int i = threadIdx.x; i < C; i += blockDim.x) base[(size_t)(front + t) * C + i] = 0.f;
    }
}

// The reference's padding mask (layers/content_processor.py:24: padding_mask = (wav == 0)) reduced to frames the way
// fairseq's HubertModel.forward_padding_mask does it: the trailing T % F samples are dropped, the rest is viewed as
// [F][T / F] and a frame is padding when ALL samples of its chunk are exactly zero.  One wave per frame.
static __global__ __launch_bounds__(64) void k_hb_frame_mask(const float* wav, int32_t* fmask, int T, int F) {
    const int f = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int chunk = T / F;
    const float* p = wav + (size_t)b * T + (size_t)f * chunk;
    int nz = 0;
    for (int i = lane; i < chunk; i += 64) nz |= (p[i] != 0.f);
    nz = __any(nz);
    if (lane == 0) fmask[b * F + f] = nz ? 0 : 1;
}

}  // namespace gvc

using namespace gvc;

struct HbLin { float* w = nullptr; float* b = nullptr; int N = 0, K = 0; float* wf = nullptr; /* FM16 copy (skinny path) */ };
struct HbLn { float* w = nullptr; float* b = nullptr; };
struct HbLayer { HbLin qkv, out, fc1, fc2; HbLn ln1, ln2; };

struct gvc_hubert {
    gvc_hubert_dims dm;
    int cg = 0, pad_front = 0, pad_back = 0;
    float* conv0_w = nullptr;                     // [C0][k0]
    HbLn gn;
    std::vector<float*> conv_w;                   // layers 1.. : [Co][k*Ci]
    std::vector<float*> conv_wp;                  // ... and their FM16 copies for k_conv_lds (null: shape not eligible)
    int conv_lds = 1;                             // feature-extractor convs 1.. on k_conv_lds (few output frames) or the tiled GEMM
    int conv_lds_rows = 256;                     // layers with more output frames than this stay on the tiled GEMM
    HbLn feat_ln, enc_ln;
    HbLin proj, pos, fin;                         // pos.w: [E][kp*cg]
    std::vector<HbLayer> layers;
    std::map<std::string, int> bound;
    int n_expected = 0;
    float *act[2] = {nullptr, nullptr};           // conv activations (ping-pong)
    float *part = nullptr, *stats = nullptr;
    int32_t* fmask = nullptr;                     // [B][frames]: 1 = padding frame (all samples of its chunk are zero)
    float *xp = nullptr, *x = nullptr, *qkv = nullptr, *att = nullptr, *hbuf = nullptr, *tmp = nullptr, *work = nullptr;
    long long work_cap = 0;
    int max_frames = 0, max_t0 = 0;
    std::map<long long, hipGraphExec_t> graphs;
    hipStream_t cap_stream = nullptr;
    int use_graph = 1;
    int skinny = 1;                               // fragment-major transformer path (0 when the widths are not multiples of 128)
    int strip = 1;                                // more than 128 rows: strip GEMM
    bool fm_ready = false;                        // FM16 copies match the bound weights
    float *xf = nullptr, *af = nullptr, *hf = nullptr;   // FM16 activations of the fragment-major path (all rows, 16-row tiles)
    std::vector<void*> allocs;
};

static int hb_alloc(gvc_hubert* c, float** p, size_t n) {
    GVC_CHECK_HIP(hipMalloc((void**)p, n * sizeof(float)));
    c->allocs.push_back(*p);
    return GVC_OK;
}
static int hb_alloc_lin(gvc_hubert* c, HbLin& L, int N, int K) {
    L.N = N; L.K = K;
    int rc = hb_alloc(c, &L.w, (size_t)N * K);
    return rc ? rc : hb_alloc(c, &L.b, N);
}
static int hb_alloc_ln(gvc_hubert* c, HbLn& L, int d) {
    int rc = hb_alloc(c, &L.w, d);
    return rc ? rc : hb_alloc(c, &L.b, d);
}

static int hb_frames(const gvc_hubert_dims& D, int n, int upto) {
    for (int i = 0; i < upto; ++i) {
        if (n < D.conv_kernel[i]) return 0;
        n = (n - D.conv_kernel[i]) / D.conv_stride[i] + 1;
    }
    return n;
}

extern "C" int gvc_hubert_frames(gvc_hubert* c, int32_t n_samples) {
    return c ? hb_frames(c->dm, n_samples, c->dm.n_conv) : -1;
}

extern "C" int gvc_hubert_create(const gvc_hubert_dims* dims, gvc_hubert** out) {
    GVC_REQUIRE(dims && out, GVC_ERR_ARG, "gvc_hubert_create: null argument");
    gemm_init_attributes();            // the strip GEMM's dynamic LDS (up to 72 KiB) needs the raised limit in a process without a GPT context too
    const gvc_hubert_dims& D = *dims;
    GVC_REQUIRE(D.n_conv >= 2 && D.n_conv <= 8 && D.conv_kernel[0] <= 16, GVC_ERR_UNSUPPORTED, "hubert: need 2..8 conv layers, k0 <= 16");
    for (int i = 0; i < D.n_conv; ++i)
        GVC_REQUIRE(D.conv_dim[i] % 4 == 0 && D.conv_dim[i] <= 4096 && D.conv_kernel[i] >= 1 && D.conv_stride[i] >= 1,
                    GVC_ERR_UNSUPPORTED, "hubert: conv layer %d unsupported", i);
    GVC_REQUIRE(D.n_heads >= 1 && D.embed_dim == D.n_heads * 64, GVC_ERR_UNSUPPORTED, "hubert: head_dim must be 64");
    GVC_REQUIRE(D.embed_dim <= 4096 && D.ffn_dim % 4 == 0 && D.final_dim % 4 == 0 && D.pos_conv_groups >= 1 &&
                    D.embed_dim % D.pos_conv_groups == 0 && (D.embed_dim / D.pos_conv_groups) % 4 == 0 && D.pos_conv_kernel >= 1,
                GVC_ERR_UNSUPPORTED, "hubert: unsupported encoder dims");
    GVC_REQUIRE(D.max_batch >= 1 && hb_frames(D, D.max_samples, D.n_conv) >= 1, GVC_ERR_ARG, "hubert: bad capacity");
    auto* c = new gvc_hubert();
    c->dm = D;
    const int E = D.embed_dim, kp = D.pos_conv_kernel;
    c->cg = E / D.pos_conv_groups;
    c->pad_front = kp / 2;
    c->pad_back = kp - 1 - kp / 2;                // SamePad drops the extra output of an even kernel
    c->max_t0 = hb_frames(D, D.max_samples, 1);
    c->max_frames = hb_frames(D, D.max_samples, D.n_conv);
    int rc = hb_alloc(c, &c->conv0_w, (size_t)D.conv_dim[0] * D.conv_kernel[0]);
    if (!rc) rc = hb_alloc_ln(c, c->gn, D.conv_dim[0]);
    size_t maxact = 0;
    for (int i = 0, n = D.max_samples; i < D.n_conv; ++i) {
        n = (n - D.conv_kernel[i]) / D.conv_stride[i] + 1;
        maxact = std::max(maxact, (size_t)n * D.conv_dim[i]);
        if (i >= 1 && !rc) {
            float* w = nullptr;
            rc = hb_alloc(c, &w, (size_t)D.conv_dim[i] * D.conv_kernel[i] * D.conv_dim[i - 1]);
            c->conv_w.push_back(w);
            float* wp = nullptr;
            if (!rc && conv_lds_ci_ok(D.conv_dim[i - 1]) && D.conv_dim[i] % 16 == 0 &&
                conv_lds_bytes(D.conv_dim[i - 1], D.conv_kernel[i], 1, false, D.conv_stride[i]) <= kConvLdsMax)
                rc = hb_alloc(c, &wp, (size_t)D.conv_dim[i] * D.conv_kernel[i] * D.conv_dim[i - 1]);
            c->conv_wp.push_back(wp);
        }
    }
    const int Cl = D.conv_dim[D.n_conv - 1];
    if (!rc) rc = hb_alloc_ln(c, c->feat_ln, Cl);
    if (!rc) rc = hb_alloc_lin(c, c->proj, E, Cl);
    if (!rc) rc = hb_alloc_lin(c, c->pos, E, kp * c->cg);
    if (!rc) rc = hb_alloc_ln(c, c->enc_ln, E);
    c->layers.resize(D.n_layers);
    for (int l = 0; l < D.n_layers && !rc; ++l) {
        HbLayer& L = c->layers[l];
        if ((rc = hb_alloc_lin(c, L.qkv, 3 * E, E))) break;
        if ((rc = hb_alloc_lin(c, L.out, E, E))) break;
        if ((rc = hb_alloc_lin(c, L.fc1, D.ffn_dim, E))) break;
        if ((rc = hb_alloc_lin(c, L.fc2, E, D.ffn_dim))) break;
        if ((rc = hb_alloc_ln(c, L.ln1, E))) break;
        if ((rc = hb_alloc_ln(c, L.ln2, E))) break;
        for (HbLin* q : {&L.qkv, &L.out, &L.fc1, &L.fc2})
            if ((rc = hb_alloc(c, &q->wf, (size_t)q->N * q->K))) break;
    }
    if (!rc) rc = hb_alloc_lin(c, c->fin, D.final_dim, E);
    // conv (1 + 2 for GroupNorm) + n_conv-1 + feature LN 2 + proj 2 + pos_conv 2 + encoder LN 2 + 16 per layer + final 2
    c->n_expected = 3 + (D.n_conv - 1) + 2 + 2 + 2 + 2 + 16 * D.n_layers + 2;
    const size_t B = D.max_batch, F = c->max_frames;
    for (int i = 0; i < 2 && !rc; ++i) rc = hb_alloc(c, &c->act[i], B * maxact);
    if (!rc) rc = hb_alloc(c, &c->part, B * (size_t)cdiv(c->max_t0, kC0Chunk) * D.conv_dim[0] * 2);
    if (!rc) rc = hb_alloc(c, &c->stats, B * D.conv_dim[0] * 2);
    if (!rc) rc = hb_alloc(c, reinterpret_cast<float**>(&c->fmask), B * F);
    if (!rc) rc = hb_alloc(c, &c->xp, B * (F + kp) * E);
    if (!rc) rc = hb_alloc(c, &c->x, B * F * E);
    if (!rc) rc = hb_alloc(c, &c->tmp, B * F * E);
    if (!rc) rc = hb_alloc(c, &c->att, B * F * E);
    if (!rc) rc = hb_alloc(c, &c->qkv, B * F * 3 * E);
    if (!rc) rc = hb_alloc(c, &c->hbuf, B * F * D.ffn_dim);
    const size_t fm_rows = std::max<size_t>(128, (B * F + 15) & ~(size_t)15);      // fragment-major activations come in 16-row tiles
    if (!rc) rc = hb_alloc(c, &c->xf, fm_rows * E);
    if (!rc) rc = hb_alloc(c, &c->af, fm_rows * E);
    if (!rc) rc = hb_alloc(c, &c->hf, fm_rows * D.ffn_dim);
    c->work_cap = 16ll << 20;
    if (!rc) rc = hb_alloc(c, &c->work, (size_t)c->work_cap);
    if (!rc && hipStreamCreateWithFlags(&c->cap_stream, hipStreamNonBlocking) != hipSuccess) rc = GVC_ERR_HIP;
    if (getenv("GVC_GRAPHS")) c->use_graph = atoi(getenv("GVC_GRAPHS"));        // 0: eager launches (include/genvc_hip.h, environment switches)
    conv_lds_init_attributes();
    if (E % 128 != 0 || D.ffn_dim % 128 != 0) c->skinny = 0;
    if (rc) { gvc_hubert_destroy(c); return rc; }
    *out = c;
    return GVC_OK;
}

extern "C" int gvc_hubert_destroy(gvc_hubert* c) {
    if (!c) return GVC_OK;
    for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
    if (c->cap_stream) hipStreamDestroy(c->cap_stream);
    for (void* p : c->allocs) hipFree(p);
    delete c;
    return GVC_OK;
}

static int hb_copy(float* dst, const float* src, int64_t numel, int64_t expect, const char* name, hipStream_t s) {
    GVC_REQUIRE(numel == expect, GVC_ERR_ARG, "%s: expected %lld elements, got %lld", name, (long long)expect, (long long)numel);
    GVC_CHECK_HIP(hipMemcpyAsync(dst, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
    return GVC_OK;
}

static bool ends_with(const std::string& n, const char* suf) {
    const size_t k = strlen(suf);
    return n.size() >= k && n.compare(n.size() - k, k, suf) == 0;
}

extern "C" int gvc_hubert_bind_weight(gvc_hubert* c, const char* name, const float* src, int64_t numel, gvc_stream sv) {
    GVC_REQUIRE(c && name && src, GVC_ERR_ARG, "gvc_hubert_bind_weight: null argument");
    hipStream_t s = (hipStream_t)sv;
    const gvc_hubert_dims& D = c->dm;
    const int E = D.embed_dim;
    std::string n(name);
    const bool is_b = ends_with(n, ".bias");
    if (!is_b && !ends_with(n, ".weight")) return GVC_OK;
    int rc = GVC_OK;
    bool known = true;
    auto lin = [&](HbLin& L) { return hb_copy(is_b ? L.b : L.w, src, numel, is_b ? L.N : (int64_t)L.N * L.K, name, s); };
    auto ln = [&](HbLn& L, int d) { return hb_copy(is_b ? L.b : L.w, src, numel, d, name, s); };
    if (n.rfind("feature_extractor.conv_layers.", 0) == 0) {
        const int i = atoi(n.c_str() + 30);
        GVC_REQUIRE(i >= 0 && i < D.n_conv, GVC_ERR_ARG, "%s: layer out of range", name);
        if (ends_with(n, ".0.weight")) {
            if (i == 0) rc = hb_copy(c->conv0_w, src, numel, (int64_t)D.conv_dim[0] * D.conv_kernel[0], name, s);
            else {
                const int Co = D.conv_dim[i], Ci = D.conv_dim[i - 1], k = D.conv_kernel[i];
                GVC_REQUIRE(numel == (int64_t)Co * Ci * k, GVC_ERR_ARG, "%s: wrong size", name);
                hipLaunchKernelGGL(k_hb_repack_conv, dim3(1024), dim3(256), 0, s, src, c->conv_w[i - 1], Co, Ci, k);
                if (c->conv_wp[i - 1]) hipLaunchKernelGGL(k_to_fm16, dim3(1024), dim3(256), 0, s, c->conv_w[i - 1], c->conv_wp[i - 1], Co, k * Ci);
                GVC_LAUNCH_CHECK();
            }
        } else if (i == 0 && (ends_with(n, ".2.weight") || ends_with(n, ".2.bias"))) rc = ln(c->gn, D.conv_dim[0]);
        else known = false;
    } else if (n.rfind("layer_norm.", 0) == 0) rc = ln(c->feat_ln, D.conv_dim[D.n_conv - 1]);
    else if (n.rfind("post_extract_proj.", 0) == 0) rc = lin(c->proj);
    else if (n.rfind("final_proj.", 0) == 0) rc = lin(c->fin);
    else if (n == "encoder.pos_conv.0.bias") rc = hb_copy(c->pos.b, src, numel, E, name, s);
    else if (n == "encoder.pos_conv.0.weight") {
        GVC_REQUIRE(numel == (int64_t)E * c->cg * D.pos_conv_kernel, GVC_ERR_ARG, "%s: wrong size", name);
        hipLaunchKernelGGL(k_hb_repack_conv, dim3(1024), dim3(256), 0, s, src, c->pos.w, E, c->cg, D.pos_conv_kernel);
        GVC_LAUNCH_CHECK();
    } else if (n.rfind("encoder.layer_norm.", 0) == 0) rc = ln(c->enc_ln, E);
    else if (n.rfind("encoder.layers.", 0) == 0) {
        const int l = atoi(n.c_str() + 15);
        GVC_REQUIRE(l >= 0 && l < D.n_layers, GVC_ERR_ARG, "%s: layer out of range", name);
        HbLayer& L = c->layers[l];
        const size_t dot = n.find('.', 15);
        const std::string sub = n.substr(dot + 1);
        int which = -1;
        if (sub.rfind("self_attn.q_proj.", 0) == 0) which = 0;
        else if (sub.rfind("self_attn.k_proj.", 0) == 0) which = 1;
        else if (sub.rfind("self_attn.v_proj.", 0) == 0) which = 2;
        if (which >= 0) {
            if (is_b) rc = hb_copy(L.qkv.b + (size_t)which * E, src, numel, E, name, s);
            else rc = hb_copy(L.qkv.w + (size_t)which * E * E, src, numel, (int64_t)E * E, name, s);
        } else if (sub.rfind("self_attn.out_proj.", 0) == 0) rc = lin(L.out);
        else if (sub.rfind("self_attn_layer_norm.", 0) == 0) rc = ln(L.ln1, E);
        else if (sub.rfind("fc1.", 0) == 0) rc = lin(L.fc1);
        else if (sub.rfind("fc2.", 0) == 0) rc = lin(L.fc2);
        else if (sub.rfind("final_layer_norm.", 0) == 0) rc = ln(L.ln2, E);
        else known = false;
    } else known = false;
    if (rc == GVC_OK && known) { c->bound[n] = 1; c->fm_ready = false; }
    return rc;
}

extern "C" int gvc_hubert_missing_weights(gvc_hubert* c) { return c ? c->n_expected - (int)c->bound.size() : -1; }

static int hb_linear(gvc_hubert* c, const HbLin& L, const float* A, float* C, int M, int act, const float* resid, hipStream_t s) {
    GemmArgs G;
    memset(&G, 0, sizeof(G));
    G.A = A; G.lda = L.K; G.Wt = L.w; G.ldw = L.K; G.C = C; G.ldc = L.N;
    G.M = M; G.N = L.N; G.K = L.K; G.work = c->work;
    G.e.bias = L.b; G.e.act = act;
    if (resid) { G.e.resid = resid; G.e.ldr = L.N; }
    return launch_gemm_cap(G, 1, c->work_cap, s);
}

// GroupNorm apply .. last transformer layer (everything between the two kernels that touch caller pointers)
static int hb_body(gvc_hubert* c, int B, int T, hipStream_t s) {
    const gvc_hubert_dims& D = c->dm;
    const int E = D.embed_dim;
    int rc;
    int Tin = hb_frames(D, T, 1);
    const int C0 = D.conv_dim[0];
    const int nchunk = cdiv(Tin, kC0Chunk);
    hipLaunchKernelGGL(k_hb_gn_stats, dim3(cdiv(C0, 8), B), dim3(256), 0, s, c->part, c->stats, nchunk, Tin, C0);
    GVC_LAUNCH_CHECK();
    {
        const size_t n4 = (size_t)Tin * C0 / 4;
        const int gx = (int)std::min<size_t>((n4 + 255) / 256, 4096);
        hipLaunchKernelGGL(k_hb_gn_gelu, dim3(gx, B), dim3(256), 0, s, c->act[0], c->stats, c->gn.w, c->gn.b, Tin, C0);
        GVC_LAUNCH_CHECK();
    }
    int cur = 0;
    for (int i = 1; i < D.n_conv; ++i) {
        const int Ci = D.conv_dim[i - 1], Co = D.conv_dim[i], k = D.conv_kernel[i], st = D.conv_stride[i];
        const int Tout = (Tin - k) / st + 1;
        if (c->conv_lds && c->conv_wp[i - 1] && Tout <= c->conv_lds_rows) {
            // one memory round trip per workgroup (conv_lds.h): the layer is too small for the tiled GEMM's k-loop to pay (1 s chunk: the
            // last three layers, 13.4 / 9.8 / 9.6 us against 16.4 / 12.5 / 13.5; with more frames the 133 KB input tile per 32 x 16
            // outputs loses to the GEMM: 68 against 49 us at 1599 frames)
            ConvLdsArgs A;
            memset(&A, 0, sizeof(A));
            A.x = c->act[cur]; A.x_bs = (long long)Tin * Ci; A.ldx = Ci; A.x_scale = 1.f; A.slope = 1.f;
            A.x_rows = Tin; A.stride = st;
            A.y = c->act[cur ^ 1]; A.y_bs = (long long)Tout * Co; A.ldy = Co;
            A.T = Tout; A.ntiles = Co / 16; A.act = ACT_GELU_ERF;
            A.job[0].wp = reinterpret_cast<const float4*>(c->conv_wp[i - 1]); A.job[0].k = k; A.job[0].dil = 1;
            if ((rc = launch_conv_lds(Ci, 1, A, 1, B, conv_lds_bytes(Ci, k, 1, false, st), s))) return rc;
            cur ^= 1;
            Tin = Tout;
            continue;
        }
        GemmArgs G;
        memset(&G, 0, sizeof(G));
        G.A = c->act[cur]; G.lda = st * Ci; G.a_batch_stride = (long long)Tin * Ci;
        G.Wt = c->conv_w[i - 1]; G.ldw = k * Ci;
        G.C = c->act[cur ^ 1]; G.ldc = Co; G.c_batch_stride = (long long)Tout * Co;
        G.M = Tout; G.N = Co; G.K = k * Ci; G.work = c->work;
        G.e.act = ACT_GELU_ERF;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
        cur ^= 1;
        Tin = Tout;
    }
    const int F = Tin, Cl = D.conv_dim[D.n_conv - 1], rows = B * F;
    const int kp = D.pos_conv_kernel;
    const long long xp_bs = (long long)(F + kp - 1) * E;
    // LayerNorm(features) -> post_extract_proj -> rows [pad_front, pad_front + F) of the zero-padded xp
    hipLaunchKernelGGL(k_hb_ln_rows, dim3(rows), dim3(256), 0, s, c->act[cur], (long long)F * Cl, c->act[cur ^ 1],
                       (long long)F * Cl, F, Cl, c->feat_ln.w, c->feat_ln.b);
    GVC_LAUNCH_CHECK();
    {
        GemmArgs G;
        memset(&G, 0, sizeof(G));
        G.A = c->act[cur ^ 1]; G.lda = Cl; G.a_batch_stride = (long long)F * Cl;
        G.Wt = c->proj.w; G.ldw = Cl;
        G.C = c->xp + (size_t)c->pad_front * E; G.ldc = E; G.c_batch_stride = xp_bs;
        G.M = F; G.N = E; G.K = Cl; G.work = c->work;
        G.e.bias = c->proj.b;
        if ((rc = launch_gemm_cap(G, B, c->work_cap, s))) return rc;
    }
    // conv padding rows, and the rows of padding frames, are zero from here on
    hipLaunchKernelGGL(k_hb_zero_pad, dim3(64, B), dim3(256), 0, s, c->xp, E, F, c->pad_front, c->pad_back, c->fmask);
    GVC_LAUNCH_CHECK();
    // tmp = x + gelu(pos_conv(x) + bias): ONE batched GEMM, batch = (utterance, group)
    {
        GemmArgs G;
        memset(&G, 0, sizeof(G));
        const int cg = c->cg;
        G.batch_inner = D.pos_conv_groups;
        G.A = c->xp; G.lda = E; G.a_batch_stride = cg; G.a_batch_stride2 = xp_bs;
        G.conv_cin = cg; G.conv_tap_stride = E;
        G.Wt = c->pos.w; G.ldw = kp * cg; G.w_batch_stride = (long long)cg * kp * cg;
        G.C = c->tmp; G.ldc = E; G.c_batch_stride = cg; G.c_batch_stride2 = (long long)F * E;
        G.M = F; G.N = cg; G.K = kp * cg; G.work = c->work;
        G.e.bias = c->pos.b; G.e.bias_batch_stride = cg; G.e.act = ACT_GELU_ERF;
        G.e.resid = c->xp + (size_t)c->pad_front * E; G.e.ldr = E; G.e.resid_batch_stride = cg; G.e.resid_batch_stride2 = xp_bs;
        if ((rc = launch_gemm_cap(G, B * D.pos_conv_groups, c->work_cap, s))) return rc;
    }
    const bool skinny = c->skinny && (rows <= 128 || (c->strip && E % 64 == 0 && D.ffn_dim % 64 == 0));
    const bool strip = rows > 128;
    if (skinny) {
        // fragment-major GEMMs, K-split partials folded into the post-LN kernels: 7 launches per layer.  rows <= 128 (the 1 s
        // streaming chunk: 49 frames): the skinny kernels, weights streamed once; more rows (an utterance, a batch): the strip
        // kernel, which picks its own K split
        hipLaunchKernelGGL(k_hb_ln_rows, dim3(rows), dim3(256), 0, s, c->tmp, (long long)F * E, c->x, (long long)F * E, F, E,
                           c->enc_ln.w, c->enc_ln.b, (const float*)nullptr, 0, (const float*)nullptr, rows, c->xf);
        GVC_LAUNCH_CHECK();
        int sk_used = 1;
        auto sk_gemm = [&](const HbLin& L, const float* A_fm, float* C, int act, int c_fm16, int SK, const float* resid = nullptr) {
            GemmArgs G;
            memset(&G, 0, sizeof(G));
            G.A = A_fm; G.lda = L.K; G.Wt = L.wf; G.ldw = L.K; G.C = C; G.ldc = L.N; G.M = rows; G.N = L.N; G.K = L.K;
            G.work = c->work;
            if (strip && resid) {          // a projection back to E columns: raw partial planes, completed by the post-LN kernel
                if ((long long)rows * L.N > c->work_cap) { set_error("hubert: %d rows exceed the split-K work buffer", rows); return (int)GVC_ERR_ARG; }
                return launch_gemm_strip(G, 8, c->work_cap, 1, &sk_used, s);
            }
            sk_used = SK;
            if (SK == 1) { G.e.bias = L.b; G.e.act = act; G.e.c_fm16 = c_fm16; G.e.resid = resid; G.e.ldr = L.N; }
            if (strip) return launch_gemm_strip(G, 1, c->work_cap, 0, nullptr, s);
            return launch_gemm_skinny(G, SK, c->work_cap, s);
        };
        auto post_ln = [&](const HbLin& L, const HbLn& N, int) {       // x = LN(x + linear) (+ FM16 copy)
            const int SK = sk_used;
            if (SK > 1 || strip)
                hipLaunchKernelGGL(k_hb_ln_rows, dim3(rows), dim3(256), 0, s, c->x, (long long)F * E, c->x, (long long)F * E, F, E,
                                   N.w, N.b, c->work, SK, L.b, rows, c->xf);
            else
                hipLaunchKernelGGL(k_hb_ln_rows, dim3(rows), dim3(256), 0, s, c->tmp, (long long)F * E, c->x, (long long)F * E, F, E,
                                   N.w, N.b, (const float*)nullptr, 0, (const float*)nullptr, rows, c->xf);
        };
        const int sk_out = (E / 2) % 128 == 0 ? 2 : 1, sk_fc2 = (D.ffn_dim / 4) % 128 == 0 ? 4 : ((D.ffn_dim / 2) % 128 == 0 ? 2 : 1);
        for (int l = 0; l < D.n_layers; ++l) {
            const HbLayer& L = c->layers[l];
            if ((rc = sk_gemm(L.qkv, c->xf, c->qkv, ACT_NONE, 0, 1))) return rc;
            hipLaunchKernelGGL(k_attn64_mfma<true>, dim3(cdiv(F, 16), D.n_heads, B), dim3(256), 0, s, c->qkv, c->qkv + E, c->qkv + 2 * E, (long long)3 * E, (long long)F * 3 * E, F, F, c->af, F, E, 0.125f, 1, c->fmask);
            GVC_LAUNCH_CHECK();
            if ((rc = sk_gemm(L.out, c->af, c->tmp, ACT_NONE, 0, sk_out, c->x))) return rc;
            post_ln(L.out, L.ln1, sk_out);
            GVC_LAUNCH_CHECK();
            if ((rc = sk_gemm(L.fc1, c->xf, c->hf, ACT_GELU_ERF, 1, 1))) return rc;
            if ((rc = sk_gemm(L.fc2, c->hf, c->tmp, ACT_NONE, 0, sk_fc2, c->x))) return rc;
            post_ln(L.fc2, L.ln2, sk_fc2);
            GVC_LAUNCH_CHECK();
        }
        return GVC_OK;
    }
    hipLaunchKernelGGL(k_hb_ln_rows, dim3(rows), dim3(256), 0, s, c->tmp, (long long)F * E, c->x, (long long)F * E, F, E,
                       c->enc_ln.w, c->enc_ln.b);
    GVC_LAUNCH_CHECK();
    for (int l = 0; l < D.n_layers; ++l) {
        const HbLayer& L = c->layers[l];
        if ((rc = hb_linear(c, L.qkv, c->x, c->qkv, rows, ACT_NONE, nullptr, s))) return rc;
        hipLaunchKernelGGL(k_attn64_mfma<true>, dim3(cdiv(F, 16), D.n_heads, B), dim3(256), 0, s, c->qkv, c->qkv + E, c->qkv + 2 * E, (long long)3 * E, (long long)F * 3 * E, F, F, c->att, F, E, 0.125f, 0, c->fmask);
        GVC_LAUNCH_CHECK();
        if ((rc = hb_linear(c, L.out, c->att, c->tmp, rows, ACT_NONE, c->x, s))) return rc;
        hipLaunchKernelGGL(k_hb_ln_rows, dim3(rows), dim3(256), 0, s, c->tmp, (long long)F * E, c->x, (long long)F * E, F, E,
                           L.ln1.w, L.ln1.b);
        GVC_LAUNCH_CHECK();
        if ((rc = hb_linear(c, L.fc1, c->x, c->hbuf, rows, ACT_GELU_ERF, nullptr, s))) return rc;
        if ((rc = hb_linear(c, L.fc2, c->hbuf, c->tmp, rows, ACT_NONE, c->x, s))) return rc;
        hipLaunchKernelGGL(k_hb_ln_rows, dim3(rows), dim3(256), 0, s, c->tmp, (long long)F * E, c->x, (long long)F * E, F, E,
                           L.ln2.w, L.ln2.b);
        GVC_LAUNCH_CHECK();
    }
    return GVC_OK;
}

extern "C" int gvc_hubert_forward(gvc_hubert* c, const float* wav, int32_t B, int32_t T, float* out, gvc_stream sv) {
    GVC_REQUIRE(c && wav && out, GVC_ERR_ARG, "gvc_hubert_forward: null argument");
    GVC_REQUIRE(gvc_hubert_missing_weights(c) == 0, GVC_ERR_STATE, "%d HuBERT weight tensors are not bound",
                gvc_hubert_missing_weights(c));
    const gvc_hubert_dims& D = c->dm;
    const int F = hb_frames(D, T, D.n_conv);
    GVC_REQUIRE(B >= 1 && B <= D.max_batch && T <= D.max_samples && F >= 1, GVC_ERR_ARG,
                "hubert: B=%d samples=%d outside capacity (%d, %d) or too short", B, T, D.max_batch, D.max_samples);
    hipStream_t s = (hipStream_t)sv;
    if (c->skinny && !c->fm_ready) {
        for (HbLayer& L : c->layers)
            for (HbLin* q : {&L.qkv, &L.out, &L.fc1, &L.fc2}) {
                hipLaunchKernelGGL(k_to_fm16, dim3(1024), dim3(256), 0, s, q->w, q->wf, q->N, q->K);
                GVC_LAUNCH_CHECK();
            }
        c->fm_ready = true;
    }
    const int T0 = hb_frames(D, T, 1), C0 = D.conv_dim[0], k0 = D.conv_kernel[0], s0 = D.conv_stride[0];
    hipLaunchKernelGGL(k_hb_conv0, dim3(cdiv(T0, kC0Chunk), cdiv(C0, 256), B), dim3(256), ((kC0Chunk - 1) * s0 + k0) * sizeof(float), s, wav,
                       c->conv0_w, c->act[0], c->part, T, T0, C0, k0, s0);
    GVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_hb_frame_mask, dim3(F, B), dim3(64), 0, s, wav, c->fmask, T, F);
    GVC_LAUNCH_CHECK();
    int rc;
    if (!c->use_graph) {
        if ((rc = hb_body(c, B, T, s))) return rc;
    } else {
        const long long key = ((long long)B << 32) | (unsigned)T;
        auto it = c->graphs.find(key);
        if (it == c->graphs.end()) {
            if (c->graphs.size() >= 64) {          // bounded cache: utterance lengths vary freely
                for (auto& kv : c->graphs) hipGraphExecDestroy(kv.second);
                c->graphs.clear();
            }
            GVC_CHECK_HIP(hipStreamBeginCapture(c->cap_stream, hipStreamCaptureModeThreadLocal));
            rc = hb_body(c, B, T, c->cap_stream);
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
    }
    {
        GemmArgs G;
        memset(&G, 0, sizeof(G));
        const HbLin& L = c->fin;
        G.A = c->x; G.lda = L.K; G.Wt = L.w; G.ldw = L.K; G.C = out; G.ldc = L.N;
        G.M = B * F; G.N = L.N; G.K = L.K; G.work = c->work;
        G.e.bias = L.b;
        if ((rc = launch_gemm_cap(G, 1, c->work_cap, s))) return rc;
    }
    return GVC_OK;
}
