// fp32 MFMA GEMM for the batched (prefill-shaped) parts of the path:
//   C[m][n] = epilogue( sum_k A[m][k] * Wt[n][k] )      A: [M][lda], Wt: [N][ldw], both K-contiguous
// v_mfma_f32_32x32x2_f32: exact fp32 products/accumulation (bitwise an fmaf chain), 157 TFLOP/s peak.
// Tile 64x64x32, 4 waves (2x2), one 32x32 accumulator per wave, LDS double-buffered and padded to
// 36-float rows so the ds_read_b128 fragment reads are bank-conflict-free.  A lane reads 4 consecutive
// k of its row at once and feeds 4 MFMAs with the k-permutation {t, t+4} (A and B use the same one).
// Split-K (gridDim.z = batch*SK) writes raw partials; k_splitk_epilogue finishes deterministically.
#pragma once
#include "common.h"

namespace gvc {

// "FM16" fragment-major layout of an [M][K] matrix (M, K multiples of 16): each (16 rows x 16 k) block is stored as
// the 64 float4 a wave feeds to v_mfma_f32_16x16x4_f32 -- slot (m%16) + 16*((k%16)/4) holds k%16/4*4..+3 of row m -- so one
// fragment load is 1 KiB contiguous (a row-major fragment load touches 16 cache lines per quarter-wave and is
// tag-lookup bound in the texture path: 15 us instead of 4 for the prefill GEMMs).
__host__ __device__ __forceinline__ size_t fm16_index(int m, int k, int K) {
    return ((size_t)(m >> 4) * (K >> 4) + (k >> 4)) * 256 + (size_t)((((m & 15) + 16 * ((k & 15) >> 2)) << 2) + (k & 3));
}

enum GemmAct { ACT_NONE = 0, ACT_GELU_NEW = 1, ACT_RELU = 2, ACT_GELU_ERF = 3 };
enum GemmAAct { AACT_NONE = 0, AACT_LRELU = 1 };   // activation applied to A while it is staged (HiFi-GAN)

struct GemmEpi {
    const float* bias;       // [N] or null
    long long bias_batch_stride;   // grouped convolutions: batch = group
    int act;                 // GemmAct
    const float* resid;      // [M][ldr] or null (may alias C for in-place residual)
    int ldr;
    long long resid_batch_stride;
    long long resid_batch_stride2; // two-level batches (GemmArgs.batch_inner): stride of the outer index
    int resid_fm16;          // skinny kernels: `resid` is an FM16 matrix of row length N (the Perceiver keeps its latents fragment-major)
    int geglu;               // skinny kernels: columns come in (x_j, gate_j) pairs (weights interleaved at bind time); the epilogue stores
                             // gelu_erf(gate_j) * x_j as column j of an FM16 matrix of row length N / 2 (perceiver_encoder.py:205-208)
    const float* resid2;     // second residual with the layout of `resid` (HiFi-GAN resblock sum), or null
    float out_scale;         // 0 = none; otherwise the stored value is multiplied by it
    // GPT QKV scatter (prefill): n < d -> q[m][n]; else K/V cache rows
    int c_fm16;              // 1 -> C is written in FM16 layout (row length N)
    int qkv;                 // 1 -> scatter mode, C is the q buffer [M][d]
    int d, n_head, head_dim, max_seq, T;
    float* kcache; float* vcache;
    int kv_bf16;            // the cache holds bf16 (unsigned short) elements: k/v are rounded to nearest even on the way in
    const int32_t* slots;
    const int32_t* base_len;  // nullable: per-slot cached length; row t of batch b lands at position base_len[slot] + t
};

struct GemmArgs {
    const float* A; int lda; long long a_batch_stride;
    const float* Wt; int ldw; long long w_batch_stride;
    int w_bf16;             // skinny kernels only: Wt is an FM16 copy of bf16 elements (unsigned short), widened in registers
    float* C; int ldc; long long c_batch_stride;
    // two-level batches (ContentVec's grouped positional conv over a batch of utterances): batch index = outer * batch_inner + inner;
    // A / C / resid move by (inner * stride + outer * stride2), weights and bias by the inner index only.  0: one level
    int batch_inner; long long a_batch_stride2, c_batch_stride2;
    int M, N, K;
    // implicit im2col for dilated convolutions over a time-major buffer: when conv_cin > 0, column k of A is
    // (tap = k / conv_cin, ci = k % conv_cin) and lives at A[m*lda + tap*conv_tap_stride + ci]
    int conv_cin, conv_tap_stride;
    int a_act; float a_slope;       // GemmAAct applied to A elements as they are loaded
    // skinny kernels, <= 32 rows: A = the batched decode attention's per-chunk softmax partials, merged while the fragments are
    // loaded (att_part [M][att_heads][att_nc][att_hd + 4] = (o[hd], m, l, pad) as k_attention<.., DIRECT = false> writes them;
    // K = att_heads * att_hd); null: A is the FM16 operand
    const float* att_part; int att_nc, att_heads, att_hd;
    int SK;                  // split-K factor; >1: C unused, partials to `work`
    float* work;             // [batch][SK][M][N]
    GemmEpi e;
};

__device__ __forceinline__ long long gemm_boff(const GemmArgs& G, int batch, long long s1, long long s2) {
    return G.batch_inner > 0 ? (long long)(batch % G.batch_inner) * s1 + (long long)(batch / G.batch_inner) * s2 : (long long)batch * s1;
}
__device__ __forceinline__ int gemm_binner(const GemmArgs& G, int batch) { return G.batch_inner > 0 ? batch % G.batch_inner : batch; }

// everything after the bias (callers whose lanes keep one column add the bias value they loaded once)
__device__ __forceinline__ void gemm_store_nb(const GemmArgs& G, int batch, int m, int n, float v) {
    const GemmEpi& e = G.e;
    if (e.act == ACT_GELU_NEW) v = gelu_new(v);
    else if (e.act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (e.act == ACT_GELU_ERF) v = gelu_erf(v);
    if (e.qkv) {
        const int which = n / e.d;
        const int c = n - which * e.d;
        if (which == 0) {
            G.C[(size_t)m * G.ldc + c] = v;
        } else {
            const int b = m / e.T, t = m - b * e.T;
            const int slot = e.slots[b];
            const int pos = t + (e.base_len ? e.base_len[slot] : 0);
            const int h = c / e.head_dim, j = c - h * e.head_dim;
            float* cache = which == 1 ? e.kcache : e.vcache;
            const size_t at = (((size_t)slot * e.n_head + h) * e.max_seq + pos) * e.head_dim + j;
            if (e.kv_bf16) reinterpret_cast<unsigned short*>(cache)[at] = f32_to_bf16(v);
            else cache[at] = v;
        }
        return;
    }
    if (e.resid) v += e.resid[gemm_boff(G, batch, e.resid_batch_stride, e.resid_batch_stride2) + (size_t)m * e.ldr + n];
    if (e.resid2) v += e.resid2[gemm_boff(G, batch, e.resid_batch_stride, e.resid_batch_stride2) + (size_t)m * e.ldr + n];
    if (e.out_scale != 0.f) v *= e.out_scale;
    if (e.c_fm16) { G.C[fm16_index(m, n, G.N)] = v; return; }
    G.C[gemm_boff(G, batch, G.c_batch_stride, G.c_batch_stride2) + (size_t)m * G.ldc + n] = v;
}

__device__ __forceinline__ void gemm_store(const GemmArgs& G, int batch, int m, int n, float v) {
    if (G.e.bias) v += G.e.bias[gemm_binner(G, batch) * G.e.bias_batch_stride + n];
    gemm_store_nb(G, batch, m, n, v);
}

// gemm_store for four consecutive columns n .. n+3 (n % 4 == 0) of row m: the same arithmetic per element, 16-byte accesses.
// Needs ldc, ldr, d, head_dim % 4 == 0 and 16-byte aligned bases (true of every caller's buffers).
__device__ __forceinline__ void gemm_store4(const GemmArgs& G, int m, int n, float4 v) {
    const GemmEpi& e = G.e;
    if (e.bias) {
        const float4 b = *reinterpret_cast<const float4*>(e.bias + n);
        v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (e.act == ACT_GELU_NEW) { v.x = gelu_new(v.x); v.y = gelu_new(v.y); v.z = gelu_new(v.z); v.w = gelu_new(v.w); }
    else if (e.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (e.act == ACT_GELU_ERF) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
    if (e.qkv) {
        const int which = n / e.d;
        const int c = n - which * e.d;
        if (which == 0) {
            *reinterpret_cast<float4*>(G.C + (size_t)m * G.ldc + c) = v;
        } else {
            const int b = m / e.T, t = m - b * e.T;
            const int slot = e.slots[b];
            const int pos = t + (e.base_len ? e.base_len[slot] : 0);
            const int h = c / e.head_dim, j = c - h * e.head_dim;
            float* cache = which == 1 ? e.kcache : e.vcache;
            const size_t at = (((size_t)slot * e.n_head + h) * e.max_seq + pos) * e.head_dim + j;
            if (e.kv_bf16) {
                ushort4 o;
                o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
                *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(cache) + at) = o;
            } else {
                *reinterpret_cast<float4*>(cache + at) = v;
            }
        }
        return;
    }
    if (e.resid) {
        const float4 r = *reinterpret_cast<const float4*>(e.resid + (size_t)m * e.ldr + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (e.resid2) {
        const float4 r = *reinterpret_cast<const float4*>(e.resid2 + (size_t)m * e.ldr + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (e.out_scale != 0.f) { v.x *= e.out_scale; v.y *= e.out_scale; v.z *= e.out_scale; v.w *= e.out_scale; }
    if (e.c_fm16) { *reinterpret_cast<float4*>(G.C + fm16_index(m, n, G.N)) = v; return; }
    *reinterpret_cast<float4*>(G.C + (size_t)m * G.ldc + n) = v;
}

// The loads an epilogue needs that do not depend on the result -- bias, and for the QKV scatter the row's slot and cached
// length (a chain of two) -- requested BEFORE the main loop by the thread that will store the element: left in gemm_store
// they are three dependent L2 round trips behind the last MFMA.  (m < M; batch 0.)
struct EpiPre { float4 bias; int slot, base; };
__device__ __forceinline__ EpiPre gemm_prefetch(const GemmArgs& G, int m, int n, bool four) {
    EpiPre p;
    p.bias = make_float4(0.f, 0.f, 0.f, 0.f);
    p.slot = 0; p.base = 0;
    if (G.e.bias) {
        if (four) p.bias = *reinterpret_cast<const float4*>(G.e.bias + n);
        else p.bias.x = G.e.bias[n];
    }
    if (G.e.qkv && n >= G.e.d) {
        p.slot = G.e.slots[m / G.e.T];
        p.base = G.e.base_len ? G.e.base_len[p.slot] : 0;
    }
    return p;
}
// gemm_store / gemm_store4 with the prefetched values
__device__ __forceinline__ void gemm_store_pre(const GemmArgs& G, int m, int n, float v, const EpiPre& p) {
    const GemmEpi& e = G.e;
    v += p.bias.x;
    if (e.act == ACT_GELU_NEW) v = gelu_new(v);
    else if (e.act == ACT_RELU) v = fmaxf(v, 0.f);
    else if (e.act == ACT_GELU_ERF) v = gelu_erf(v);
    if (e.qkv) {
        const int which = n / e.d;
        const int c = n - which * e.d;
        if (which == 0) {
            G.C[(size_t)m * G.ldc + c] = v;
        } else {
            const int b = m / e.T, t = m - b * e.T;
            const int h = c / e.head_dim, j = c - h * e.head_dim;
            float* cache = which == 1 ? e.kcache : e.vcache;
            const size_t at = (((size_t)p.slot * e.n_head + h) * e.max_seq + (t + p.base)) * e.head_dim + j;
            if (e.kv_bf16) reinterpret_cast<unsigned short*>(cache)[at] = f32_to_bf16(v);
            else cache[at] = v;
        }
        return;
    }
    if (e.resid) v += e.resid[e.resid_fm16 ? fm16_index(m, n, G.N) : (size_t)m * e.ldr + n];
    if (e.resid2) v += e.resid2[(size_t)m * e.ldr + n];
    if (e.out_scale != 0.f) v *= e.out_scale;
    if (e.c_fm16) { G.C[fm16_index(m, n, G.N)] = v; return; }
    G.C[(size_t)m * G.ldc + n] = v;
}
__device__ __forceinline__ void gemm_store4_pre(const GemmArgs& G, int m, int n, float4 v, const EpiPre& p) {
    const GemmEpi& e = G.e;
    v.x += p.bias.x; v.y += p.bias.y; v.z += p.bias.z; v.w += p.bias.w;
    if (e.act == ACT_GELU_NEW) { v.x = gelu_new(v.x); v.y = gelu_new(v.y); v.z = gelu_new(v.z); v.w = gelu_new(v.w); }
    else if (e.act == ACT_RELU) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
    else if (e.act == ACT_GELU_ERF) { v.x = gelu_erf(v.x); v.y = gelu_erf(v.y); v.z = gelu_erf(v.z); v.w = gelu_erf(v.w); }
    if (e.qkv) {
        const int which = n / e.d;
        const int c = n - which * e.d;
        if (which == 0) {
            *reinterpret_cast<float4*>(G.C + (size_t)m * G.ldc + c) = v;
        } else {
            const int b = m / e.T, t = m - b * e.T;
            const int h = c / e.head_dim, j = c - h * e.head_dim;
            float* cache = which == 1 ? e.kcache : e.vcache;
            const size_t at = (((size_t)p.slot * e.n_head + h) * e.max_seq + (t + p.base)) * e.head_dim + j;
            if (e.kv_bf16) {
                ushort4 o;
                o.x = f32_to_bf16(v.x); o.y = f32_to_bf16(v.y); o.z = f32_to_bf16(v.z); o.w = f32_to_bf16(v.w);
                *reinterpret_cast<ushort4*>(reinterpret_cast<unsigned short*>(cache) + at) = o;
            } else {
                *reinterpret_cast<float4*>(cache + at) = v;
            }
        }
        return;
    }
    if (e.geglu) {          // (x0, gate0, x1, gate1) -> columns n / 2, n / 2 + 1 of the FM16 output (row length N / 2)
        *reinterpret_cast<float2*>(G.C + fm16_index(m, n >> 1, G.N >> 1)) = make_float2(gelu_erf(v.y) * v.x, gelu_erf(v.w) * v.z);
        return;
    }
    if (e.resid) {
        const float4 r = *reinterpret_cast<const float4*>(e.resid + (e.resid_fm16 ? fm16_index(m, n, G.N) : (size_t)m * e.ldr + n));
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (e.resid2) {
        const float4 r = *reinterpret_cast<const float4*>(e.resid2 + (size_t)m * e.ldr + n);
        v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (e.out_scale != 0.f) { v.x *= e.out_scale; v.y *= e.out_scale; v.z *= e.out_scale; v.w *= e.out_scale; }
    if (e.c_fm16) { *reinterpret_cast<float4*>(G.C + fm16_index(m, n, G.N)) = v; return; }
    *reinterpret_cast<float4*>(G.C + (size_t)m * G.ldc + n) = v;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void k_gemm_f32(const GemmArgs G);
__global__ void k_splitk_epilogue(const GemmArgs G);

// Skinny GEMM for M <= 128 rows (streaming prefill): one workgroup per 16 weight rows, its 4 or 8 waves split K and
// combine through LDS; BOTH operands are FM16 (G.A [M][K], G.Wt [N][K]) and go straight from global/L2 into MFMA
// fragments (v_mfma_f32_16x16x4_f32), 1 KiB per load instruction.
// G.SK > 1: the K range is also split over blockIdx.y and RAW partial sums go to G.work[sk][M][N] (the consumer,
// k_ln_sum_rows, adds them together with bias and residual: no separate split-K epilogue launch).
int launch_gemm_skinny(GemmArgs G, int SK, long long work_cap, hipStream_t s);
// x_out[row] = x_in[row] + bias + sum_s part[s][row];  a[row] = LayerNorm(x_out[row]) (skipped when ln_w is null); one wave per
// row, d = 256 or 1024, SK <= 8; x_out may be x_in
int launch_ln_sum_rows(const float* x_in, float* x_out, float* a, const float* part, int SK, const float* bias, int rows, int d,
                       const float* ln_w, const float* ln_b, int a_fm16, hipStream_t s);
// the same row completion + LayerNorm folded into the prologue of the skinny GEMM (<= 16 rows, K = d): see gemm.hip
struct LnFuse {
    const float* x_in; float* x_out;       // [rows][d]; x_out null: nothing to complete (part == null), keep x_in
    const float* part; int SK; const float* pbias;
    const float* ln_w; const float* ln_b;
    int rows;
};
int launch_gemm_skinny_ln(GemmArgs G, const LnFuse& P, hipStream_t s);
// Strip GEMM for more rows than the skinny kernels take (batched prefill, latent re-pass): FM16 operands, N % 64 == 0.  One
// workgroup per (m group <= 9 tiles, 64 columns, K split): A staged once per workgroup in LDS by LDS-DMA, weights streamed per
// wave, MFMA-bound inner loop (gemm.hip).  sk_max > 1 allows a K split when the grid would not fill the GPU: raw partials
// go to G.work[sk][M][N]; raw_partials = 1 leaves them for the consumer (k_ln_sum_rows) and reports the split in *sk_used,
// raw_partials = 0 runs k_splitk_epilogue (G.e applied there).
int launch_gemm_strip(GemmArgs G, int sk_max, long long work_cap, int raw_partials, int* sk_used, hipStream_t s);
void gemm_init_attributes();        // raises dynamic-LDS limits; call once, outside stream capture
// row-major [N][K] -> FM16
__global__ void k_to_fm16(const float* src, float* dst, int N, int K);
__global__ void k_to_fm16_bf16(const float* src, unsigned short* dst, int N, int K);   // src already rounded to bf16 values

// chooses the split-K factor from the shape; `work_cap` = capacity of G.work in floats
int launch_gemm_cap(GemmArgs G, int batch, long long work_cap, hipStream_t s);

}  // namespace gvc
