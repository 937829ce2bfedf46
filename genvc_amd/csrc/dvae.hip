// Content tokenizer: DiscreteVAE.get_codebook_indices (reference layers/dvae.py:324-331) = 1-D conv
// encoder (:252-291, ResBlock :172-184) + nearest-codebook search (Quantize.forward :87-93).
//
// Streaming-sized calls (B x T <= 600 frames: 124 instead of 168 us for a 1 s chunk) run every conv on the one-round-trip kernel of
// conv_lds.h in 256-channel slices (the last workgroup to arrive adds the slices, bias, skip and ReLU); larger calls:
// Activations are kept time-major [B][T + 2*pad][C] with zero rows at both ends, so a k-tap conv with
// stride s is ONE fp32 MFMA GEMM whose A rows are overlapping windows of that buffer
// (row t = &x[s*t][0], length k*C, lda = s*C) against weights repacked to [C_out][k*C_in]:
// no im2col copy, bias/ReLU/skip fused in the GEMM epilogue.  Encoder weights are ~98 MB fp32 at the
// reference size: weight-read bound at B=1.  VQ: one workgroup per frame, thread j owns code j and
// evaluates the reference's expression (|x|^2 - 2 x.e_j) + |e_j|^2 in fp32; first index wins ties.
#include <map>
#include <string>
#include <vector>

#include "conv_lds.h"

namespace gvc {

// feat [B][C][T] (reference layout) -> xpad [B][T + 2*pad][C], rows [pad, pad+T)
__global__ void k_to_time_major(const float* feat, float* xpad, int C, int T, int pad) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, c0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const float* src = feat + (size_t)b * C * T;
    float* dst = xpad + (size_t)b * (T + 2 * pad) * C;
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int c = c0 + r, t = t0 + threadIdx.x;
        if (c < C && t < T) tile[r][threadIdx.x] = src[(size_t)c * T + t];
    }
    __syncthreads();
    for (int r = threadIdx.y; r < 32; r += blockDim.y) {
        const int t = t0 + r, c = c0 + threadIdx.x;
        if (c < C && t < T) dst[(size_t)(t + pad) * C + c] = tile[threadIdx.x][r];
    }
}

// zero the `pad` rows in front of and behind the T live rows of every batch element
__global__ void k_zero_pad_rows(float* buf, int C, int T, int pad) {
    const int b = blockIdx.y;
    float* base = buf + (size_t)b * (T + 2 * pad) * C;
    const int n = pad * C;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < 2 * n; i += gridDim.x * blockDim.x) {
        if (i < n) base[i] = 0.f;
        else base[(size_t)(T + pad) * C + (i - n)] = 0.f;
    }
}

// Conv1d weight [Co][Ci][k] -> [Co][k*Ci] with column j*Ci + ci
__global__ void k_repack_conv(const float* w, float* out, int Co, int Ci, int k) {
    const size_t n = (size_t)Co * Ci * k;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(i % k);
        const int ci = (int)((i / k) % Ci);
        const int co = (int)(i / ((size_t)k * Ci));
        out[((size_t)co * k + j) * Ci + ci] = w[i];
    }
}

// ee[j] = sum_i embed[i][j]^2   (embed.pow(2).sum(0), dvae.py:88)
__global__ void k_code_norms(const float* embed, float* ee, int dim, int n_embed) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_embed) return;
    float s = 0.f;
    for (int i = 0; i < dim; ++i) { const float v = embed[(size_t)i * n_embed + j]; s += v * v; }
    ee[j] = s;
}

// one workgroup (1024 threads) per row of x [N][dim]; embed [dim][n_embed]; idx[row] = argmax_j -((xx - 2 x.e_j) + ee_j).
// Thread (g, j) accumulates the quarter g of the dot product of code j (coalesced over j); the four quarters are added
// in LDS, then the reference's expression is evaluated per code and the first index wins ties.
constexpr int kVqThreads = 1024;
__global__ __launch_bounds__(kVqThreads) void k_vq_argmin(const float* x, const float* embed, const float* ee, int dim,
                                                          int n_embed, int32_t* idx) {
    extern __shared__ float xs[];              // [dim] + [4][n_embed] partial dots
    __shared__ float rv[kVqThreads / 64];
    __shared__ int ri[kVqThreads / 64];
    __shared__ float red[kVqThreads / 64];
    const int row = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float* pd = xs + dim;
    const float* xr = x + (size_t)row * dim;
    float q = 0.f;
    for (int i = tid; i < dim; i += kVqThreads) { const float v = xr[i]; xs[i] = v; q += v * v; }
    q = wave_sum(q);
    if (lane == 0) red[wave] = q;
    __syncthreads();
    float xx = 0.f;
#pragma unroll
    for (int w = 0; w < kVqThreads / 64; ++w) xx += red[w];
    const int g = tid >> 8, jj = tid & 255;
    const int i0 = g * (dim / 4), i1 = g == 3 ? dim : (g + 1) * (dim / 4);
    for (int j = jj; j < n_embed; j += 256) {
        // 16 codebook loads in flight per thread (the rolled loop waited for every load: 50 us for 13 frames); the sum keeps
        // the element order i0, i0 + 1, ... of the plain loop
        float dot = 0.f;
        int i = i0;
        for (; i + 16 <= i1; i += 16) {
            float e[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) e[u] = embed[(size_t)(i + u) * n_embed + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) dot = fmaf(xs[i + u], e[u], dot);
        }
        for (; i < i1; ++i) dot = fmaf(xs[i], embed[(size_t)i * n_embed + j], dot);
        pd[g * n_embed + j] = dot;
    }
    __syncthreads();
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int j = tid; j < n_embed; j += kVqThreads) {
        const float dot = (pd[j] + pd[n_embed + j]) + (pd[2 * n_embed + j] + pd[3 * n_embed + j]);
        const float dist = (xx - 2.0f * dot) + ee[j];
        const float sc = -dist;
        if (sc > best || (sc == best && j < bi)) { best = sc; bi = j; }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { rv[wave] = best; ri[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kVqThreads / 64; ++w)
            if (rv[w] > best || (rv[w] == best && ri[w] < bi)) { best = rv[w]; bi = ri[w]; }
        idx[row] = bi;
    }
}

}  // namespace gvc

using namespace gvc;

struct ConvW { float* w = nullptr; float* b = nullptr; float* wp = nullptr; int Co = 0, Ci = 0, k = 0; };     // wp: 256-channel slices in FM16 (k_conv_lds)

constexpr int kDvSlice = 256, kDvCounters = 4096;

struct gvc_dvae {
    gvc_dvae_dims dm;
    int pad = 1, inner = 0;
    std::vector<ConvW> down;            // strided conv + ReLU stages
    std::vector<ConvW> res;             // 3 convs per ResBlock
    ConvW last;                         // 1x1 to codebook_dim
    float *embed = nullptr, *ee = nullptr;
    std::map<std::string, int> bound;
    int n_expected = 0;
    float *buf[3] = {nullptr, nullptr, nullptr};   // padded time-major activations (ping-pong + residual temp)
    float *enc = nullptr, *work = nullptr;
    long long work_cap = 0;
    // one-round-trip conv path (conv_lds.h; every conv's input channels a multiple of 256, as at the reference size): every conv
    // output has its own buffer, so the zero rows either side of the live region are never written (no k_zero_pad_rows launches)
    bool lds_path = false;               // streaming-sized calls: every conv on k_conv_lds
    int lds_rows = 600;                  // calls with more than this many input frames (B x T) stay on the tiled GEMM
    std::vector<float*> lbuf;            // [num_layers] stage outputs (the last one is updated in place by the ResBlocks) + 2 ResBlock temporaries
    int* cnt = nullptr;                  // arrival counters of the K-split convs
    int cur_T = -1, cur_B = -1;
    std::vector<void*> allocs;
};

static int dalloc(gvc_dvae* c, float** p, size_t n) {
    GVC_CHECK_HIP(hipMalloc((void**)p, n * sizeof(float)));
    c->allocs.push_back(*p);
    return GVC_OK;
}

extern "C" int gvc_dvae_create(const gvc_dvae_dims* dims, gvc_dvae** out) {
    GVC_REQUIRE(dims && out, GVC_ERR_ARG, "gvc_dvae_create: null argument");
    const gvc_dvae_dims& D = *dims;
    GVC_REQUIRE(D.num_layers >= 1 && D.kernel_size % 2 == 1 && D.channels % 4 == 0 && D.hidden_dim % 4 == 0 &&
                    D.codebook_dim % 4 == 0,
                GVC_ERR_UNSUPPORTED, "dvae: need num_layers >= 1, odd kernel_size, channel counts %% 4 == 0");
    auto* c = new gvc_dvae();
    c->dm = D;
    c->pad = (D.kernel_size - 1) / 2;
    GVC_REQUIRE(c->pad >= 1 || D.num_resnet_blocks == 0, GVC_ERR_UNSUPPORTED, "dvae: kernel_size 1 with ResBlocks");
    if (c->pad < 1) c->pad = 1;          // ResBlocks use k=3, pad=1
    int rc = GVC_OK;
    int cin = D.channels;
    size_t maxc = cin;
    for (int i = 0; i < D.num_layers && !rc; ++i) {
        ConvW w; w.Co = D.hidden_dim << i; w.Ci = cin; w.k = D.kernel_size;
        if (!(rc = dalloc(c, &w.w, (size_t)w.Co * w.Ci * w.k))) rc = dalloc(c, &w.b, w.Co);
        c->down.push_back(w);
        cin = w.Co;
        if ((size_t)cin > maxc) maxc = cin;
    }
    c->inner = cin;
    for (int i = 0; i < D.num_resnet_blocks && !rc; ++i)
        for (int k : {3, 3, 1}) {
            ConvW w; w.Co = cin; w.Ci = cin; w.k = k;
            if (!(rc = dalloc(c, &w.w, (size_t)w.Co * w.Ci * w.k))) rc = dalloc(c, &w.b, w.Co);
            c->res.push_back(w);
        }
    c->last.Co = D.codebook_dim; c->last.Ci = cin; c->last.k = 1;
    if (!rc && !(rc = dalloc(c, &c->last.w, (size_t)D.codebook_dim * cin))) rc = dalloc(c, &c->last.b, D.codebook_dim);
    if (!rc && !(rc = dalloc(c, &c->embed, (size_t)D.codebook_dim * D.num_tokens))) rc = dalloc(c, &c->ee, D.num_tokens);
    c->n_expected = 2 * (D.num_layers + 3 * D.num_resnet_blocks + 1) + 1;
    const size_t rows = (size_t)D.max_batch * (D.max_frames + 2 * c->pad);
    for (int i = 0; i < 3 && !rc; ++i) rc = dalloc(c, &c->buf[i], rows * maxc);
    if (!rc) rc = dalloc(c, &c->enc, (size_t)D.max_batch * D.max_frames * D.codebook_dim);
    c->work_cap = 4ll << 20;
    if (!rc) rc = dalloc(c, &c->work, (size_t)c->work_cap);
    // the k_conv_lds path: every conv in 256-channel slices
    c->lds_path = true;
    {
        std::vector<ConvW*> all;
        for (ConvW& w : c->down) all.push_back(&w);
        for (ConvW& w : c->res) all.push_back(&w);
        all.push_back(&c->last);
        for (ConvW* w : all) c->lds_path = c->lds_path && w->Ci % kDvSlice == 0 && w->Co % 16 == 0 && conv_lds_bytes(kDvSlice, w->k, 1, true, 2) <= kConvLdsMax;
        if (c->lds_path) {
            for (ConvW* w : all)
                if (!rc) rc = dalloc(c, &w->wp, (size_t)w->Co * w->Ci * w->k);
            size_t T = D.max_frames;
            for (int i = 0; i < D.num_layers && !rc; ++i) {
                T = (T + 1) / 2 + 1;         // (an upper bound of the stage's frames)
                float* p = nullptr;
                rc = dalloc(c, &p, (size_t)D.max_batch * (T + 2 * c->pad) * c->down[i].Co);
                c->lbuf.push_back(p);
            }
            for (int i = 0; i < 2 && !rc; ++i) {
                float* p = nullptr;
                rc = dalloc(c, &p, (size_t)D.max_batch * (T + 2 * c->pad) * c->inner);
                c->lbuf.push_back(p);
            }
            if (!rc) rc = dalloc(c, reinterpret_cast<float**>(&c->cnt), kDvCounters);
            if (!rc && hipMemset(c->cnt, 0, kDvCounters * sizeof(int)) != hipSuccess) rc = GVC_ERR_HIP;
            conv_lds_init_attributes();
        }
    }
    if (rc) { gvc_dvae_destroy(c); return rc; }
    *out = c;
    return GVC_OK;
}

extern "C" int gvc_dvae_destroy(gvc_dvae* c) {
    if (!c) return GVC_OK;
    for (void* p : c->allocs) hipFree(p);
    delete c;
    return GVC_OK;
}

static int bind_conv(ConvW& w, bool is_bias, const float* src, int64_t numel, const char* name, hipStream_t s) {
    if (is_bias) {
        GVC_REQUIRE(numel == w.Co, GVC_ERR_ARG, "%s: expected %d elements, got %lld", name, w.Co, (long long)numel);
        GVC_CHECK_HIP(hipMemcpyAsync(w.b, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
        return GVC_OK;
    }
    GVC_REQUIRE(numel == (int64_t)w.Co * w.Ci * w.k, GVC_ERR_ARG, "%s: expected %lld elements, got %lld", name,
                (long long)w.Co * w.Ci * w.k, (long long)numel);
    hipLaunchKernelGGL(k_repack_conv, dim3(1024), dim3(256), 0, s, src, w.w, w.Co, w.Ci, w.k);
    GVC_LAUNCH_CHECK();
    if (w.wp) {
        hipLaunchKernelGGL(k_conv_pack_slices, dim3(512), dim3(256), 0, s, src, w.wp, w.Co, w.Ci, w.k, kDvSlice);
        GVC_LAUNCH_CHECK();
    }
    return GVC_OK;
}

extern "C" int gvc_dvae_bind_weight(gvc_dvae* c, const char* name, const float* src, int64_t numel, gvc_stream sv) {
    GVC_REQUIRE(c && name && src, GVC_ERR_ARG, "gvc_dvae_bind_weight: null argument");
    hipStream_t s = (hipStream_t)sv;
    std::string n(name);
    int rc = GVC_OK;
    bool known = true;
    if (n == "codebook.embed") {
        GVC_REQUIRE(numel == (int64_t)c->dm.codebook_dim * c->dm.num_tokens, GVC_ERR_ARG, "%s: wrong size", name);
        GVC_CHECK_HIP(hipMemcpyAsync(c->embed, src, numel * sizeof(float), hipMemcpyDeviceToDevice, s));
        hipLaunchKernelGGL(k_code_norms, dim3(cdiv(c->dm.num_tokens, 256)), dim3(256), 0, s, c->embed, c->ee,
                           c->dm.codebook_dim, c->dm.num_tokens);
        GVC_LAUNCH_CHECK();
    } else if (n.rfind("encoder.", 0) == 0) {
        const size_t dot = n.find('.', 8);
        GVC_REQUIRE(dot != std::string::npos, GVC_ERR_ARG, "malformed weight name %s", name);
        const int idx = atoi(n.substr(8, dot - 8).c_str());
        const std::string rest = n.substr(dot + 1);
        const int nd = c->dm.num_layers, nr = c->dm.num_resnet_blocks;
        const bool is_bias = rest.size() >= 4 && rest.compare(rest.size() - 4, 4, "bias") == 0;
        if (idx < nd) {
            if (rest == "0.weight" || rest == "0.bias") rc = bind_conv(c->down[idx], is_bias, src, numel, name, s);
            else known = false;
        } else if (idx < nd + nr) {
            int j = -1;
            if (rest.rfind("net.0.", 0) == 0) j = 0;
            else if (rest.rfind("net.2.", 0) == 0) j = 1;
            else if (rest.rfind("net.4.", 0) == 0) j = 2;
            if (j >= 0) rc = bind_conv(c->res[3 * (idx - nd) + j], is_bias, src, numel, name, s);
            else known = false;
        } else if (idx == nd + nr && (rest == "weight" || rest == "bias")) {
            rc = bind_conv(c->last, is_bias, src, numel, name, s);
        } else {
            known = false;
        }
    } else {
        known = false;      // decoder.*, codebook.cluster_size, ... (training state)
    }
    if (rc == GVC_OK && known) c->bound[n] = 1;
    return rc;
}

extern "C" int gvc_dvae_missing_weights(gvc_dvae* c) { return c ? c->n_expected - (int)c->bound.size() : -1; }

// out rows [pad, pad+To) of `dst` (time-major, padded) = act(conv(src)) (+ skip)
static int conv_gemm(gvc_dvae* c, const ConvW& w, const float* src, int Tin, int stride, float* dst, int To, int B,
                     int act, const float* skip, bool dst_padded, hipStream_t s) {
    const int pad = c->pad;
    const int cpad = (w.k - 1) / 2;                 // this conv's own padding (<= pad)
    GemmArgs G;
    memset(&G, 0, sizeof(G));
    G.A = src + (size_t)(pad - cpad) * w.Ci;        // window of row t starts at padded row stride*t + (pad - cpad)
    G.lda = stride * w.Ci;
    G.a_batch_stride = (long long)(Tin + 2 * pad) * w.Ci;
    G.Wt = w.w; G.ldw = w.k * w.Ci;
    G.M = To; G.N = w.Co; G.K = w.k * w.Ci;
    G.ldc = w.Co;
    if (dst_padded) { G.C = dst + (size_t)pad * w.Co; G.c_batch_stride = (long long)(To + 2 * pad) * w.Co; }
    else { G.C = dst; G.c_batch_stride = (long long)To * w.Co; }
    G.work = c->work;
    G.e.bias = w.b; G.e.act = act;
    if (skip) { G.e.resid = skip + (size_t)pad * w.Co; G.e.ldr = w.Co; G.e.resid_batch_stride = (long long)(To + 2 * pad) * w.Co; }
    return launch_gemm_cap(G, B, c->work_cap, s);
}

// the same conv on k_conv_lds: Ci / 256 channel slices x Co / 16 column tiles, slices combined by the last workgroup to arrive
// (one slice: the plain kernel).  src / dst / skip as in conv_gemm (skip: a padded buffer with the layout of dst, may be dst);
// dst_padded = false: dst is [B][To][Co] without padding rows
static int conv_lds(gvc_dvae* c, const ConvW& w, const float* src, int Tin, int stride, float* dst, int To, int B, int act,
                    const float* skip, bool dst_padded, hipStream_t s) {
    const int pad = c->pad, nsplit = w.Ci / kDvSlice, tiles = w.Co / 16;
    ConvLdsArgs A;
    memset(&A, 0, sizeof(A));
    A.x = src; A.x_bs = (long long)(Tin + 2 * pad) * w.Ci; A.ldx = w.Ci; A.x_scale = 1.f; A.slope = 1.f;
    A.x_row0 = pad; A.x_rows = Tin + 2 * pad; A.stride = stride;
    A.T = To; A.ntiles = tiles; A.ldy = w.Co; A.act = act;
    A.job[0].wp = reinterpret_cast<const float4*>(w.wp); A.job[0].b = w.b; A.job[0].k = w.k; A.job[0].dil = 1;
    A.job[0].row_off = -((w.k - 1) / 2);
    const long long o_bs = dst_padded ? (long long)(To + 2 * pad) * w.Co : (long long)To * w.Co, o_off = dst_padded ? (long long)pad * w.Co : 0;
    const size_t lds = conv_lds_bytes(kDvSlice, w.k, 1, nsplit > 1, stride);
    if (nsplit == 1) {
        A.y = dst; A.y_bs = o_bs; A.y_off = o_off; A.resid = skip;
        return launch_conv_lds(kDvSlice, 1, A, 1, B, lds, s);
    }
    GVC_REQUIRE((long long)nsplit * B * To * w.Co <= c->work_cap && B * cdiv(To, 32) * tiles <= kDvCounters, GVC_ERR_ARG,
                "dvae: B=%d frames=%d exceed the K-split work buffer", B, To);
    A.x_ps = kDvSlice; A.split = nsplit; A.wp_js = (long long)tiles * w.k * (kDvSlice / 16) * 64; A.cnt = c->cnt;
    A.y = c->work; A.y_bs = (long long)To * w.Co; A.y_ps = (long long)B * To * w.Co;
    A.yf = dst; A.yf_bs = o_bs; A.yf_off = o_off; A.resid = skip;
    return launch_conv_lds_split(kDvSlice, A, B, lds, s);
}

static int zero_pads(gvc_dvae* c, float* buf, int C, int T, int B, hipStream_t s) {
    hipLaunchKernelGGL(k_zero_pad_rows, dim3(cdiv(2 * c->pad * C, 256), B), dim3(256), 0, s, buf, C, T, c->pad);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

static int dvae_run(gvc_dvae* c, int B, int T, int32_t* codes_out, float* enc_out, hipStream_t s);

static int dvae_check(gvc_dvae* c, const float* feat, int B, int T, const int32_t* codes_out) {
    GVC_REQUIRE(c && feat && codes_out, GVC_ERR_ARG, "gvc_dvae_encode: null argument");
    GVC_REQUIRE(gvc_dvae_missing_weights(c) == 0, GVC_ERR_STATE, "%d DVAE weight tensors are not bound",
                gvc_dvae_missing_weights(c));
    GVC_REQUIRE(B >= 1 && B <= c->dm.max_batch && T >= 1 && T <= c->dm.max_frames, GVC_ERR_ARG,
                "dvae: B=%d T=%d outside capacity (%d, %d)", B, T, c->dm.max_batch, c->dm.max_frames);
    return GVC_OK;
}

extern "C" int gvc_dvae_encode(gvc_dvae* c, const float* feat, int32_t B, int32_t T, int32_t* codes_out,
                               float* enc_out, gvc_stream sv) {
    int rc = dvae_check(c, feat, B, T, codes_out);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)sv;
    const int C = c->dm.channels;
    hipLaunchKernelGGL(k_to_time_major, dim3(cdiv(T, 32), cdiv(C, 32), B), dim3(32, 8), 0, s, feat, c->buf[0], C, T, c->pad);
    GVC_LAUNCH_CHECK();
    return dvae_run(c, B, T, codes_out, enc_out, s);
}

extern "C" int gvc_dvae_encode_frames(gvc_dvae* c, const float* feat, int32_t B, int32_t T, int32_t* codes_out,
                                      float* enc_out, gvc_stream sv) {
    int rc = dvae_check(c, feat, B, T, codes_out);
    if (rc) return rc;
    hipStream_t s = (hipStream_t)sv;
    const size_t C = c->dm.channels, row = C * sizeof(float);
    // [B][T][C] -> rows [pad, pad+T) of the padded time-major buffer: one strided copy
    GVC_CHECK_HIP(hipMemcpy2DAsync(c->buf[0] + (size_t)c->pad * C, (size_t)(T + 2 * c->pad) * row, feat, (size_t)T * row,
                                   (size_t)T * row, B, hipMemcpyDeviceToDevice, s));
    return dvae_run(c, B, T, codes_out, enc_out, s);
}

// input already staged in buf[0] rows [pad, pad+T)
static int dvae_run(gvc_dvae* c, int B, int T, int32_t* codes_out, float* enc_out, hipStream_t s) {
    const int pad = c->pad;
    (void)pad;
    int rc;
    float *cur = c->buf[0], *nxt = c->buf[1], *tmp = c->buf[2];
    int C = c->dm.channels, Tc = T;
    if (c->lds_path && B * T <= c->lds_rows) {
        if (T != c->cur_T || B != c->cur_B) {
            // the zero rows around the live regions depend on the geometry: everything is cleared when it changes (the input buffer's
            // live rows were just written: only its padding rows are cleared, by the kernel of the other path)
            if ((rc = zero_pads(c, cur, C, Tc, B, s))) return rc;
            size_t Tn = T;
            for (size_t i = 0; i < c->lbuf.size(); ++i) {
                const bool stage = i < c->down.size();
                if (stage) Tn = (Tn + 2 * ((c->down[i].k - 1) / 2) - c->down[i].k) / 2 + 1;
                const size_t ch = stage ? c->down[i].Co : c->inner;
                GVC_CHECK_HIP(hipMemsetAsync(c->lbuf[i], 0, (size_t)B * (Tn + 2 * pad) * ch * sizeof(float), s));
            }
            c->cur_T = T; c->cur_B = B;
        }
        for (size_t i = 0; i < c->down.size(); ++i) {      // Conv1d(k, stride 2, pad (k-1)/2) + ReLU
            const ConvW& w = c->down[i];
            const int To = (Tc + 2 * ((w.k - 1) / 2) - w.k) / 2 + 1;
            if ((rc = conv_lds(c, w, cur, Tc, 2, c->lbuf[i], To, B, ACT_RELU, nullptr, true, s))) return rc;
            cur = c->lbuf[i]; C = w.Co; Tc = To;
        }
        float *t1 = c->lbuf[c->down.size()], *t2 = c->lbuf[c->down.size() + 1];
        for (size_t r = 0; r + 2 < c->res.size(); r += 3) {   // ResBlock: conv3-ReLU-conv3-ReLU-conv1 + skip (over the block input)
            if ((rc = conv_lds(c, c->res[r], cur, Tc, 1, t1, Tc, B, ACT_RELU, nullptr, true, s))) return rc;
            if ((rc = conv_lds(c, c->res[r + 1], t1, Tc, 1, t2, Tc, B, ACT_RELU, nullptr, true, s))) return rc;
            if ((rc = conv_lds(c, c->res[r + 2], t2, Tc, 1, cur, Tc, B, ACT_NONE, cur, true, s))) return rc;
        }
        float* enc = enc_out ? enc_out : c->enc;
        if ((rc = conv_lds(c, c->last, cur, Tc, 1, enc, Tc, B, ACT_NONE, nullptr, false, s))) return rc;
        const int dim = c->dm.codebook_dim;
        hipLaunchKernelGGL(k_vq_argmin, dim3(B * Tc), dim3(kVqThreads), (dim + 4 * c->dm.num_tokens) * sizeof(float), s, enc, c->embed, c->ee, dim,
                           c->dm.num_tokens, codes_out);
        GVC_LAUNCH_CHECK();
        return GVC_OK;
    }
    c->cur_T = -1;                        // (the input buffer now has this call's geometry: the other path clears its padding rows again)
    if ((rc = zero_pads(c, cur, C, Tc, B, s))) return rc;
    for (const ConvW& w : c->down) {                  // Conv1d(k, stride 2, pad (k-1)/2) + ReLU
        const int To = (Tc + 2 * ((w.k - 1) / 2) - w.k) / 2 + 1;
        if ((rc = conv_gemm(c, w, cur, Tc, 2, nxt, To, B, ACT_RELU, nullptr, true, s))) return rc;
        if ((rc = zero_pads(c, nxt, w.Co, To, B, s))) return rc;
        std::swap(cur, nxt);
        C = w.Co; Tc = To;
    }
    for (size_t r = 0; r + 2 < c->res.size(); r += 3) {   // ResBlock: conv3-ReLU-conv3-ReLU-conv1 + skip
        if ((rc = conv_gemm(c, c->res[r], cur, Tc, 1, nxt, Tc, B, ACT_RELU, nullptr, true, s))) return rc;
        if ((rc = zero_pads(c, nxt, C, Tc, B, s))) return rc;
        if ((rc = conv_gemm(c, c->res[r + 1], nxt, Tc, 1, tmp, Tc, B, ACT_RELU, nullptr, true, s))) return rc;
        // 1x1 conv + skip, written over the block input (each element is read and written by one lane)
        if ((rc = conv_gemm(c, c->res[r + 2], tmp, Tc, 1, cur, Tc, B, ACT_NONE, cur, true, s))) return rc;
    }
    float* enc = enc_out ? enc_out : c->enc;
    if ((rc = conv_gemm(c, c->last, cur, Tc, 1, enc, Tc, B, ACT_NONE, nullptr, false, s))) return rc;
    const int dim = c->dm.codebook_dim;
    hipLaunchKernelGGL(k_vq_argmin, dim3(B * Tc), dim3(kVqThreads), (dim + 4 * c->dm.num_tokens) * sizeof(float), s, enc, c->embed, c->ee, dim,
                       c->dm.num_tokens, codes_out);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

extern "C" int gvc_vq_argmin(const float* x, const float* embed, int32_t N, int32_t dim, int32_t n_embed, int32_t* idx,
                             float* work, gvc_stream sv) {
    GVC_REQUIRE(x && embed && idx && work && N >= 1, GVC_ERR_ARG, "gvc_vq_argmin: bad argument");
    hipStream_t s = (hipStream_t)sv;
    hipLaunchKernelGGL(k_code_norms, dim3(cdiv(n_embed, 256)), dim3(256), 0, s, embed, work, dim, n_embed);
    GVC_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_vq_argmin, dim3(N), dim3(kVqThreads), (dim + 4 * n_embed) * sizeof(float), s, x, embed, work, dim, n_embed, idx);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}
