// One-launch block stack for 2..16 ROWS (reference: one iteration of layers/stream_generator.py:809-881 over B streams -- each
// row is one stream's new position, gpt_inference.py:92-112 -- or the <= 16 uncached rows of a streaming chunk's prefill,
// gpt_inference.py:81-91; block arithmetic: SURVEY.md Appendix A).
//
// Same engine as the one-stream step (persist_kernel.h): 256 resident workgroups, a loader wave per workgroup that streams the
// workgroup's weights through an LDS ring with LDS-DMA, eight consumer waves that gather a phase input, compute and publish.
// What is different with R rows:
//   * arithmetic on the matrix cores: v_mfma_f32_4x4x1_16b_f32 (exact fp32) multiplies FOUR weight rows by all R activation
//     rows -- its 16 blocks are (R / 4 row groups) x (16 / (R / 4) k positions) -- so the row partition of the one-stream step
//     (12 / 4 / 16 / 4 weight rows per workgroup and phase, whole K, one output plane) carries over with no padding waste;
//   * weights come from a packed copy, contiguous per (layer, workgroup): each 4-row group "k-quad-major" ([k / 4][row][4]), so a
//     ring slot read with ds_read_b128 is the A operand of four consecutive MFMAs and needs no bank padding;
//   * activations travel in the B-operand order of the NEXT phase ("frag" layout: 1 KiB = one float4 per lane of one MFMA step),
//     so a consumer wave loads its K-slice straight into registers: no activation staging in LDS, each workgroup reads a
//     hand-off buffer exactly once;
//   * hand-offs carry NO tags or flags: every element of a buffer is written by exactly one lane per layer; the buffers exist in
//     two parities (layer & 1) and a producer poisons (0xffffffff) its own elements of the OTHER parity right after publishing --
//     they were last read a layer ago -- and drains that store before its next publish.  A consumer re-reads a 16-byte piece
//     until none of its dwords is the poison pattern (scripts/ubench/seam_rows.hip: 2.4-3.9 us per 16-128 KB all-to-all, no
//     wrong value in 4e8 checked).  16-byte write-through (sc1) stores, 16-byte sc1 loads.  Why a reader can never take a stale
//     element for a fresh one: every consumer wave gathers in every phase and the first use of a gathered value waits for ALL of
//     the wave's outstanding memory operations (vmcnt(0)), stores included -- so a wave's poison stores have landed before its
//     workgroup publishes the next phase, and a reader polls an element for layer l + 2 only after it has (transitively) seen
//     publishes of every workgroup that are younger than that workgroup's poison of layer l + 1.
// Per layer: A [LN1, c_attn -> q|k|v rows, k/v appended to the cache] -> B [attention of one (row, head, key chunk) per
// workgroup] -> C [merge of the chunk partials, attn c_proj, residual] -> D [LN2, c_fc, gelu_new] -> E [mlp c_proj, residual].
// The head (double LayerNorm + mel_head) stays with the caller's launches.  d_model 1024, an even layer count; head_dim 256, 128 or 64
// (4, 8 or 16 heads -- the reference's config default is 16, configs/genVC_configs.py:132): phase B works on "super-heads" of 256
// consecutive model dims = 256 / head_dim real heads side by side in a wave (lanes of a head reduce among themselves, the softmax state
// is per lane), so the workgroup mapping and every hand-off layout are those of the 4 x 256 case.
#pragma once
#include <type_traits>

#include "persist_kernel.h"

namespace gvc {

constexpr int kRD = 1024, kRHD = 256;           // kRHD: dims of a super-head (one workgroup of phase B)
constexpr int kRMaxHeads = 16;
constexpr unsigned kRPoison = 0xffffffffu;
constexpr int kRWgLayerBytes = 192 * 1024;        // packed weights per workgroup and layer: 12 ring fills
constexpr int kRMaxChunks = 4;                    // key chunks per (row, head)
constexpr int kRMaxRows = 16;
// hand-off buffers (floats inside one parity), sized for 16 rows
constexpr int kRoffQKV = 0;
constexpr int kRoffOP = kRoffQKV + kRMaxRows * 3 * kRD;                       // [chunk][frag of R x D]
constexpr int kRoffML = kRoffOP + kRMaxChunks * kRMaxRows * kRD;              // [chunk][row][head (16 slots)] float4 {m, l, 0, 0}
constexpr int kRoffX0 = kRoffML + kRMaxChunks * kRMaxRows * kRMaxHeads * 4;
constexpr int kRoffHH = kRoffX0 + kRMaxRows * kRD;
constexpr int kRoffX1 = kRoffHH + kRMaxRows * 4 * kRD;
constexpr int kRParFloats = kRoffX1 + 2 * kRMaxRows * kRD;        // X1: two K-half planes of the mlp c_proj (their sum is x)
__host__ __device__ static inline size_t rows_buf_bytes() { return (size_t)2 * kRParFloats * sizeof(float); }

typedef float pf32x4 __attribute__((ext_vector_type(4)));

struct RowsLayer {
    const float *ln1_w, *ln1_b, *qkv_b, *proj_b, *ln2_w, *ln2_b, *fc_b, *p2_b;
    float *kcache, *vcache;         // this layer's [slot][head][max_seq][hd]
    // LayerNorm folded into the projection that follows it (k_rows_ln_fold): LN(x) W^T + bias = rstd (sum_k W_rk g_k x_k - mean S_r) + C_r
    const float *lnS_a, *lnC_a;     // [3d]: LN1 -> c_attn
    const float *lnS_d, *lnC_d;     // [4d]: LN2 -> c_fc
};

struct RowsArgs {
    const RowsLayer* layers;
    const char* wpack;              // [layer][256 workgroups][192 KiB]
    int n_layer, n_head, max_seq;
    int rows, T;                    // active rows; rows [b T, (b + 1) T) continue stream b at base_len[slot b] + 0 .. T - 1
    const int32_t* slots;
    const int32_t* base_len;        // per slot; null: 0
    float* x;                       // [rows][1024] row-major: block-stack input (unless tok_in), overwritten with its output
    const int32_t* tok_in;          // decode steps: row n enters as mel_emb[tok_in[n]] + mel_pos[mel_pos_idx[slot n]] (gpt_inference.py:92-96); null: x
    const float *mel_emb, *mel_pos;
    const int32_t* mel_pos_idx;     // per slot
    int vocab;
    float* bufs;                    // [2][kRParFloats]
    int* err;
    int ring_slots, nchunks;        // nchunks: upper bound of the key split (the kernel picks 1 / 2 / 4 from the longest context it finds)
    int split1, split2;             // cached positions from which the keys of a (row, head) take 2 / 4 workgroups
    unsigned long long* dbg;
};

// packed weights: [layer][wg]{ A: 3 groups | C: 1 | D: 4 | E: 2 half-K groups }, group = 4 rows x K as [K / 4][4 rows][4]
template <int WB>      // WB = 1: bf16 elements (the fp32 sources of a bf16-weights context are already rounded: the upper halves are exact)
__global__ void k_pack_rows_weights(float* dst, const float* qkv, const float* proj, const float* fc, const float* p2) {
    const size_t n4 = (size_t)kPG * kRWgLayerBytes / 16;                 // float4 per layer
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int wg = (int)(i / (kRWgLayerBytes / 16));
        int r = (int)(i - (size_t)wg * (kRWgLayerBytes / 16));         // float4 inside the workgroup's block
        const float* src;
        int K, row0;
        // a 4-row group over K = 1024 is 1024 float4 (one ring fill), over K = 4096 it is 4096 float4 (four fills)
        if (r < 3 * 1024) { src = qkv; K = kRD; row0 = wg * 12 + (r >> 10) * 4; r &= 1023; }
        else if (r < 4 * 1024) { src = proj; K = kRD; row0 = wg * 4; r -= 3 * 1024; }
        else if (r < 8 * 1024) { r -= 4 * 1024; src = fc; K = kRD; row0 = wg * 16 + (r >> 10) * 4; r &= 1023; }
        else {          // mlp c_proj: workgroup (cb, kh) = (wg / 2, wg % 2) owns columns [8 cb, 8 cb + 8) over K-half kh: two 4-row groups x 2048 inputs
            r -= 8 * 1024; src = p2; K = 4 * kRD; row0 = (wg >> 1) * 8 + (r >> 11) * 4; r = (r & 2047) + (wg & 1) * 2048;
        }
        const int t = r & 3, q = r >> 2;
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)(row0 + t) * K + q * 4);
        if (WB) {
            uint2 h;
            h.x = (__float_as_uint(v.x) >> 16) | (__float_as_uint(v.y) & 0xffff0000u);
            h.y = (__float_as_uint(v.z) >> 16) | (__float_as_uint(v.w) & 0xffff0000u);
            reinterpret_cast<uint2*>(dst)[i] = h;
        } else reinterpret_cast<float4*>(dst)[i] = v;
    }
}

// S_r = sum_k W_rk g_k and C_r = sum_k W_rk b_k + bias_r of one LayerNorm -> projection pair (W row-per-output [N][K], the values the
// rows step multiplies with: bf16-rounded in a bf16-weights context); double accumulation, one wave per output row
__global__ void k_rows_ln_fold(float* S, float* Cc, const float* W, const float* g, const float* b, const float* bias, int N, int K) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    double s = 0.0, c = 0.0;
    for (int k = lane; k < K; k += 64) {
        const double w = (double)W[(size_t)row * K + k];
        s += w * (double)g[k];
        c += w * (double)b[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
    if (lane == 0) { S[row] = (float)s; Cc[row] = (float)(c + (double)bias[row]); }
}

// LayerNorm statistics of row rn over the whole model dim from the eight waves' (mean, M2) partials in LDS (Chan et al.); cnt =
// elements per partial
__device__ __forceinline__ void ln_merge(const float* stat, int rn, float cnt, float inv_d, float& mean, float& rstd) {
    float m = 0.f;
#pragma unroll
    for (int w = 0; w < kPCW; ++w) m += stat[w * 16 + rn];
    m *= 1.0f / (float)kPCW;
    float M2 = 0.f, dv = 0.f;
#pragma unroll
    for (int w = 0; w < kPCW; ++w) {
        const float dm = stat[w * 16 + rn] - m;
        M2 += stat[kPCW * 16 + w * 16 + rn];
        dv += dm * dm;
    }
    const float var = (M2 + dv * cnt) * inv_d;
    rstd = 1.0f / sqrtf(var + 1e-5f);
    mean = m;
}

// four consecutive weights of one row from the ring: 16 bytes of fp32, or 8 bytes of bf16 widened in registers
template <int WB>
__device__ __forceinline__ float4 ldw4(const char* p) {
    if (WB) {
        const uint2 h = *reinterpret_cast<const uint2*>(p);
        return make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
    }
    return *reinterpret_cast<const float4*>(p);
}


__device__ __forceinline__ bool rclean(pu32x4 v) { return v.x != kRPoison && v.y != kRPoison && v.z != kRPoison && v.w != kRPoison; }
__device__ __forceinline__ float4 as_f4(pu32x4 v) {
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
__device__ __forceinline__ pu32x4 as_u4(float4 v) {
    pu32x4 r = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    return r;
}

// 16-byte write-through store of a hand-off element and the poison of the same element in the other parity
__device__ __forceinline__ void rpublish(__amdgpu_buffer_rsrc_t rs, int off_cur, int off_other, float4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(as_u4(v), rs, off_cur, 0, 16);
    // the poison pattern is re-materialised at every use (-1 is an inline constant): as a plain constant the compiler keeps the vector live across
    // the whole layer loop, and in the 16-row instantiations (168 VGPRs) spills it -- a scratch reload and an `s_waitcnt vmcnt` behind every publish
    unsigned pv;
    asm volatile("v_mov_b32 %0, -1" : "=v"(pv));
    const pu32x4 p = {pv, pv, pv, pv};
    __builtin_amdgcn_raw_buffer_store_b128(p, rs, off_other, 0, 16);
}

// NL 16-byte pieces per lane at byte offsets off + i * stride: polls piece 0, then requests the rest and re-requests the pieces
// that still hold poison
template <int NL>
__device__ __forceinline__ void rgather(PCtx& c, __amdgpu_buffer_rsrc_t rs, int off, int stride, pu32x4 (&v)[NL], int code, bool poll_all = false) {
    if (NL > 1 && poll_all) {        // every poll pass requests everything: one round trip less once the data is there
        unsigned spins = 0;
        while (true) {
            bool ok = true;
#pragma unroll
            for (int i = 0; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, i * stride, 16);
#pragma unroll
            for (int i = 0; i < NL; ++i) ok = ok && rclean(v[i]);
            if (__all(ok)) break;
            if (spin_fail(c, spins, code, 1)) break;
        }
        return;
    }
    // (no early exits: a path that leaves v[] undefined makes every gathered array live across the whole layer loop and
    //  the kernel spills ~400 registers; a dead workgroup falls through the polls within 64 spins each, spin_fail)
    unsigned spins = 0;
    while (true) {
        v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        if (__all(rclean(v[0]))) break;
        if (spin_fail(c, spins, code, 1)) break;
    }
    if (NL == 1) return;
#pragma unroll
    for (int i = 1; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, i * stride, 16);
    while (true) {
        bool again = false;
#pragma unroll
        for (int i = 1; i < NL; ++i) {
            if (__any(!rclean(v[i]))) { v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, i * stride, 16); again = true; }
        }
        if (!again) break;
        if (spin_fail(c, spins, code, 1)) break;
    }
}

// two planes of NL pieces each (K-half partial sums, `poff` bytes apart): ONE sentinel piece is polled, then both planes are requested
// together -- a second plane costs no second round trip
template <int NL>
__device__ __forceinline__ void rgather2(PCtx& c, __amdgpu_buffer_rsrc_t rs, int off, int stride, int poff, pu32x4 (&v)[NL], pu32x4 (&w)[NL], int code) {
    unsigned spins = 0;
    while (true) {
        v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        if (__all(rclean(v[0]))) break;
        if (spin_fail(c, spins, code, 1)) break;
    }
#pragma unroll
    for (int i = 1; i < NL; ++i) v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, i * stride, 16);
#pragma unroll
    for (int i = 0; i < NL; ++i) w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + poff, i * stride, 16);
    while (true) {
        bool again = false;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            if (i > 0 && __any(!rclean(v[i]))) { v[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, i * stride, 16); again = true; }
            if (__any(!rclean(w[i]))) { w[i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + poff, i * stride, 16); again = true; }
        }
        if (!again) break;
        if (spin_fail(c, spins, code, 1)) break;
    }
}

// NC key chunks of NL pieces each plus their {m, l} pieces in ONE round trip: a sentinel piece is polled, then everything is requested
// together (what is still poison is re-requested)
// (ml2 / mloff2: the {m, l} of the head of the wave's LAST steps -- with head_dim 64 a wave's K-slice spans two heads)
template <int NC, int NL>
__device__ __forceinline__ void rgather_chunks(PCtx& c, __amdgpu_buffer_rsrc_t rs, int off, int stride, int coff, int mloff, int mloff2, int mlcoff,
                                               pu32x4 (&v)[NC][NL], pu32x4 (&ml)[NC], pu32x4 (&ml2)[NC], int code) {
    unsigned spins = 0;
    while (true) {
        v[0][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 16);
        if (__all(rclean(v[0][0]))) break;
        if (spin_fail(c, spins, code, 1)) break;
    }
#pragma unroll
    for (int cc = 0; cc < NC; ++cc) {
#pragma unroll
        for (int i = 0; i < NL; ++i)
            if (cc + i > 0) v[cc][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + cc * coff, i * stride, 16);
        ml[cc] = __builtin_amdgcn_raw_buffer_load_b128(rs, mloff + cc * mlcoff, 0, 16);
        ml2[cc] = __builtin_amdgcn_raw_buffer_load_b128(rs, mloff2 + cc * mlcoff, 0, 16);
    }
    while (true) {
        bool again = false;
#pragma unroll
        for (int cc = 0; cc < NC; ++cc) {
#pragma unroll
            for (int i = 0; i < NL; ++i)
                if (cc + i > 0 && __any(!rclean(v[cc][i]))) { v[cc][i] = __builtin_amdgcn_raw_buffer_load_b128(rs, off + cc * coff, i * stride, 16); again = true; }
            if (__any(!rclean(ml[cc]))) { ml[cc] = __builtin_amdgcn_raw_buffer_load_b128(rs, mloff + cc * mlcoff, 0, 16); again = true; }
            if (__any(!rclean(ml2[cc]))) { ml2[cc] = __builtin_amdgcn_raw_buffer_load_b128(rs, mloff2 + cc * mlcoff, 0, 16); again = true; }
        }
        if (!again) break;
        if (spin_fail(c, spins, code, 1)) break;
    }
}

// sum over the lanes that share (row group, row-in-group) and differ in the k position: lane = (g KK + kk) 4 + t
template <int R>
__device__ __forceinline__ float kk_sum(float v) {
    v += dpp_mov<0x124>(v);          // row_ror:4
    v += dpp_mov<0x128>(v);          // row_ror:8
    if (R <= 8) {                    // + the other 16-lane row of the pair: v_permlane16_swap instead of a trip through the LDS crossbar
        const unsigned u = __float_as_uint(v);
        const auto sw = __builtin_amdgcn_permlane16_swap(u, u, false, false);
        v = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
    }
    if (R <= 4) v += __shfl_xor(v, 32);
    return v;
}

// ---- loader wave -------------------------------------------------------------------------------------------------
// a fill = one 4-row group over 1024 inputs = one ring slot: 16 KiB of fp32 or (WB) 8 KiB of bf16 in the slot's first half
template <int WB>
__device__ __forceinline__ void rows_loader(const RowsArgs& A, PCtx& c, char* ring) {
    constexpr int NP = 16 >> WB;                     // 1 KiB LDS-DMA instructions per fill
    const unsigned rmask = A.ring_slots - 1;
    unsigned fseq = 0;
    const char* base = A.wpack + (size_t)c.wg * (kRWgLayerBytes >> WB) + c.lane * 16;
#pragma unroll 1
    for (int l = 0; l < A.n_layer; ++l) {
        const char* lsrc = base + (size_t)l * kPG * (kRWgLayerBytes >> WB);
#pragma unroll 1
        for (int f = 0; f < kRWgLayerBytes / kPSlot; ++f) {
            unsigned spins = 0;
            bool drained = false;
            while (!c.dead) {
                unsigned m = lds_ld(c.ctl + kCtlDone);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) m = min(m, lds_ld(c.ctl + kCtlDone + w));
                if (fseq < m + (unsigned)A.ring_slots) break;
                if (!drained) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_st(c.ctl + kCtlFilled, fseq);
                    drained = true;
                }
                if (spin_fail(c, spins, 902, 1)) break;
            }
            if (c.dead) return;
            char* dst = ring + (size_t)__builtin_amdgcn_readfirstlane(fseq & rmask) * kPSlot;
            const char* src = lsrc + (size_t)f * (kPSlot >> WB);
#pragma unroll
            for (int i = 0; i < NP; ++i)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
            // two fills in flight (this one and the one before it)
            if (WB) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
            lds_st(c.ctl + kCtlFilled, fseq);
            ++fseq;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_st(c.ctl + kCtlFilled, fseq);
}

// ---- the kernel ---------------------------------------------------------------------------------------------------
template <int R, int WB, int KVB, int HDR = 256>       // padded row count (8 or 16); bf16 weight storage; bf16 KV cache; real head_dim (256 / 128 / 64)
__global__ __launch_bounds__(kPThreads) void k_rows_persist(const RowsArgs A) {
    constexpr int D = kRD, HD = kRHD;
    constexpr int G = R / 4, KK = 16 / G;            // row groups, k positions (quads) per MFMA step
    constexpr int NSX = D / 4 / KK / kPCW;           // MFMA steps (of 4 instructions) per wave over K = D      (16 rows: 8, 8 rows: 4)
    constexpr int NSH = 4 * NSX;                     // ... over K = 4 D
    constexpr int STEPB = KK * 64;                   // bytes of a 4-row weight group per step
    constexpr int LMASK = KK * 4 - 1;                // lanes that read distinct A operands
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ring = reinterpret_cast<char*>(smem);
    float* red = reinterpret_cast<float*>(ring + (size_t)A.ring_slots * kPSlot);      // [kPCW][4 groups][R] float4
    float* stat = red + kPCW * 4 * kRMaxRows * 4;    // [2][kPCW][16]
    float* resid = stat + 2 * kPCW * 16;             // [R] float4: the residual of this workgroup's four output columns
    float* resid2 = resid + kRMaxRows * 4;           // [2][R] float4: x' of the eight columns this workgroup finishes in phase E
    float* gbs = resid2 + 2 * kRMaxRows * 4;              // [kPCW][32 gain quads | 32 bias quads] of the LayerNorm a phase applies
    float* ascr = gbs + kPCW * 64 * 4;               // attention: q[256] | m_s[8][4] | l_s[8][4] | o_s[8][256]
    unsigned* ctl = reinterpret_cast<unsigned*>(ascr + 256 + 64 + kPCW * 256);
    if (threadIdx.x < kCtlWords) ctl[threadIdx.x] = 0u;
    __syncthreads();

    PCtx c;
    c.lane = threadIdx.x & 63; c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); c.wg = blockIdx.x;
    c.ctl = ctl; c.err = A.err; c.bar_target = 0; c.filled_seen = 0; c.dead = false;
    if (c.wave == kPCW) {
        rows_loader<WB>(A, c, ring);
        return;
    }
    int& lane = c.lane;
    const int wave = c.wave, wg = c.wg;
#define GVC_PHASE_BEGIN() asm volatile("" : "+v"(c.lane), "+s"(Lp))
    const unsigned rmask = A.ring_slots - 1;
    constexpr int hd = HDR;                          // real head_dim: 256, 128 or 64
    constexpr int H = D / hd;
    constexpr int lpk = hd >> 2;                     // lanes of a wave that share a real head (64, 32 or 16)
    constexpr int SH = D / HD;                       // super-heads (256 dims each): what phase B's workgroups are numbered by
    // key chunks per (row, head): from the longest context among the rows (per-slot lengths live on the device: every workgroup
    // reads the same few words and decides alike)
    int nch;
    {
        int keys = 0;
        if (c.lane < A.rows) {
            const int b0 = c.lane / A.T;
            keys = (A.base_len ? A.base_len[A.slots[b0]] : 0) + (c.lane - b0 * A.T) + 1;
        }
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) keys = max(keys, __shfl_xor(keys, o));
        keys = __builtin_amdgcn_readfirstlane(keys);
        nch = keys > A.split2 ? 4 : (keys > A.split1 ? 2 : 1);
        if (nch > A.nchunks) nch = A.nchunks;
    }
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(A.bufs, (unsigned)rows_buf_bytes());
    const float scale = 1.0f / sqrtf((float)hd);
    // stamps (GVC_PERSIST_STAMPS; wave 0, lane 0): workgroup 0, every layer: [(l * 5 + p) * 8 + k]; every workgroup at layer 2:
    // [40 (L + 2) + (wg * 5 + p) * 8 + k].  k = 0 input gathered, 1 output published, 2 LayerNorm statistics merged (B: own gathers
    // done, before the barrier), 3 weight fills waited for, 4 MFMA loop (B: score loop) done, 5 partials written to LDS, 6 barrier
    // passed, 7 final values ready (before the publish)
    const bool stamp0 = A.dbg && wave == 0 && c.lane == 0;
    const int sbase2 = 40 * (A.n_layer + 2);
    auto stamp_at = [&](int l, int p, int k) {
        if (stamp0 && wg == 0) A.dbg[(l * 5 + p) * 8 + k] = wall_clock64();
        if (stamp0 && l == 2) A.dbg[sbase2 + (wg * 5 + p) * 8 + k] = wall_clock64();
    };
    unsigned fs = 0;
    auto phase_done = [&]() { if (lane == 0) lds_st(ctl + kCtlDone + wave, fs); };
    // per-row metadata does not change during a launch: read once, not once per layer behind two dependent loads
    // (phase A's cache append: the row of lane `lane`; phase B: the row of this workgroup)
    int a_slot = 0, a_pos = 0;
    if (c.lane < A.rows) {
        const int bs = c.lane / A.T;
        a_slot = A.slots[bs];
        a_pos = (A.base_len ? A.base_len[a_slot] : 0) + (c.lane - bs * A.T);
    }
    int b_slot = 0, b_base = 0;
    {
        const int nb = wg / (nch * SH);
        if (nb < A.rows) {
            b_slot = __builtin_amdgcn_readfirstlane(A.slots[nb / A.T]);
            b_base = __builtin_amdgcn_readfirstlane(A.base_len ? A.base_len[b_slot] : 0);
        }
    }

    for (int l = 0; l < A.n_layer; ++l) {
        const RowsLayer* Lp = A.layers + l;
        const int pc = (l & 1) * kRParFloats * 4, po = ((l & 1) ^ 1) * kRParFloats * 4;      // byte offsets of the two parities
        // lane roles in the frag layout
        // =================== A: LN1 -> c_attn rows -> q | k | v ===================
        {
            GVC_PHASE_BEGIN();
            const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
            const int s0 = wave * NSX;
            float4 xv[NSX];
            // LayerNorm folded into the projection: LN1(x) W^T + bias = rstd (sum_k W_rk g_k x_k - mean S_r) + C_r.  The MFMAs run on g x
            // while the waves' (mean, M2) partials travel to LDS beside the partial sums: ONE barrier per phase, no LayerNorm merge in
            // front of the arithmetic.  S_r, C_r of the final lanes' four columns (requested ahead of the seam):
            const float4 Spre = *reinterpret_cast<const float4*>(Lp->lnS_a + wg * 12 + (wave < 3 ? wave : 0) * 4);
            const float4 Cpre = *reinterpret_cast<const float4*>(Lp->lnC_a + wg * 12 + (wave < 3 ? wave : 0) * 4);
            // LayerNorm gains of the wave's 32 k-quads -> LDS (read back one step at a time next to the MFMAs)
            if (lane < 32) *reinterpret_cast<float4*>(gbs + (wave * 64 + lane) * 4) = *reinterpret_cast<const float4*>(Lp->ln1_w + (s0 * KK + lane) * 4);
            if (l == 0) {
                if (A.tok_in) {          // a decode step: the row is built from the embedding tables here (one launch less per step)
                    const int bs = n < A.rows ? n / A.T : 0;
                    const int tok = min(max(A.tok_in[n < A.rows ? n : 0], 0), A.vocab - 1), mp = A.mel_pos_idx[A.slots[bs]];
#pragma unroll
                    for (int i = 0; i < NSX; ++i) {
                        const int q = (s0 + i) * KK + kk;
                        const float4 e = *reinterpret_cast<const float4*>(A.mel_emb + (size_t)tok * D + q * 4);
                        const float4 p = *reinterpret_cast<const float4*>(A.mel_pos + (size_t)mp * D + q * 4);
                        xv[i] = n < A.rows ? make_float4(e.x + p.x, e.y + p.y, e.z + p.z, e.w + p.w) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < NSX; ++i) {
                        const int q = (s0 + i) * KK + kk;
                        xv[i] = n < A.rows ? *reinterpret_cast<const float4*>(A.x + (size_t)n * D + q * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                }
            } else {                     // x = the two K-half planes of the previous layer's mlp c_proj (plane 0 carries residual + bias)
                pu32x4 raw[NSX], raw1[NSX];
                rgather2<NSX>(c, brs, po + kRoffX1 * 4 + s0 * 1024 + lane * 16, 1024, R * D * 4, raw, raw1, 100 + l);
#pragma unroll
                for (int i = 0; i < NSX; ++i) {
                    const float4 a = as_f4(raw[i]), b2 = as_f4(raw1[i]);
                    xv[i] = make_float4(a.x + b2.x, a.y + b2.y, a.z + b2.z, a.w + b2.w);
                }
            }
            stamp_at(l, 0, 0);
            // residual of this workgroup's output columns [4 wg, 4 wg + 4) of phase C: k-quad wg of x
            {
                const int sq = wg / KK;
                if (sq >= s0 && sq < s0 + NSX && kk == wg % KK) {
#pragma unroll
                    for (int i = 0; i < NSX; ++i)
                        if (i == sq - s0) *reinterpret_cast<float4*>(resid + n * 4) = xv[i];
                }
            }
            wait_fill(c, fs + 2);
            stamp_at(l, 0, 3);
            const char* wbase = ring + ((s0 * STEPB + (lane & LMASK) * 16) >> WB);
            const float* gb = gbs + (wave * 64 + kk) * 4;
            float4 wc[3], wn[3], gc, gn;
#pragma unroll
            for (int rg = 0; rg < 3; ++rg) wc[rg] = ldw4<WB>(wbase + (size_t)((fs + rg) & rmask) * kPSlot);
            gc = *reinterpret_cast<const float4*>(gb);
            // LayerNorm statistics: per-wave (mean, M2) of the wave's K-slice -> LDS, merged by the final lanes (Chan et al.).  Their three
            // dependent pieces (sum + lane reduction, squared deviations, lane reduction + store) ride in the shadow of the first three
            // MFMA steps: the matrix pipe is busy with the step's 12 instructions while the vector ALU does them
            float ln_mw = 0.f, ln_m2 = 0.f;
            pf32x4 acc[3];
#pragma unroll
            for (int rg = 0; rg < 3; ++rg) acc[rg] = (pf32x4){0.f, 0.f, 0.f, 0.f};
            {
                // one step (four MFMAs per row group) at a time, the next step's operands requested a step ahead: left alone, the
                // scheduler hoists every LDS read of the unrolled loop to its top (128 registers of weights) and spills
#pragma unroll
                for (int i = 0; i < NSX; ++i) {
                    if (i + 1 < NSX) {
#pragma unroll
                        for (int rg = 0; rg < 3; ++rg) wn[rg] = ldw4<WB>(wbase + (size_t)((fs + rg) & rmask) * kPSlot + (((i + 1) * STEPB) >> WB));
                        gn = *reinterpret_cast<const float4*>(gb + (i + 1) * KK * 4);
                    }
                    const float x0 = xv[i].x * gc.x, x1 = xv[i].y * gc.y, x2 = xv[i].z * gc.z, x3 = xv[i].w * gc.w;
#pragma unroll
                    for (int rg = 0; rg < 3; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[rg].x, x0, acc[rg], 0, 0, 0);
#pragma unroll
                    for (int rg = 0; rg < 3; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[rg].y, x1, acc[rg], 0, 0, 0);
#pragma unroll
                    for (int rg = 0; rg < 3; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[rg].z, x2, acc[rg], 0, 0, 0);
#pragma unroll
                    for (int rg = 0; rg < 3; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[rg].w, x3, acc[rg], 0, 0, 0);
                    // LayerNorm statistics of the wave's K-slice, one dependent piece per step
                    if (i == 0) {
                        float sm = 0.f;
#pragma unroll
                        for (int j = 0; j < NSX; ++j) sm += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
                        ln_mw = kk_sum<R>(sm) * (1.0f / (float)(NSX * KK * 4));
                    }
                    if (i == 1) {
#pragma unroll
                        for (int j = 0; j < NSX; ++j) {
                            const float a0 = xv[j].x - ln_mw, a1 = xv[j].y - ln_mw, a2 = xv[j].z - ln_mw, a3 = xv[j].w - ln_mw;
                            ln_m2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                        }
                    }
                    if (i == 2) {
                        ln_m2 = kk_sum<R>(ln_m2);
                        if (kk == 0) { stat[wave * 16 + n] = ln_mw; stat[kPCW * 16 + wave * 16 + n] = ln_m2; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rg = 0; rg < 3; ++rg) wc[rg] = wn[rg];
                    gc = gn;
                }
            }
            stamp_at(l, 0, 4);
#pragma unroll
            for (int rg = 0; rg < 3; ++rg) {
                const float4 r = make_float4(kk_sum<R>(acc[rg][0]), kk_sum<R>(acc[rg][1]), kk_sum<R>(acc[rg][2]), kk_sum<R>(acc[rg][3]));
                if (kk == 0) *reinterpret_cast<float4*>(red + ((wave * 4 + rg) * kRMaxRows + n) * 4) = r;
            }
            fs += 3;
            phase_done();
            stamp_at(l, 0, 5);
            cbar(c);
            stamp_at(l, 0, 6);
            if (wave < 3 && lane < R) {                       // wave rg finishes row group rg for row `lane`
                const int rg = wave, rn = lane;
                float4 s = *reinterpret_cast<const float4*>(red + ((0 * 4 + rg) * kRMaxRows + rn) * 4);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(red + ((w * 4 + rg) * kRMaxRows + rn) * 4);
                    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
                }
                const int col = wg * 12 + rg * 4;
                float mean, rstd;
                ln_merge(stat, rn, (float)(NSX * KK * 4), 1.0f / (float)D, mean, rstd);
                s.x = (s.x - mean * Spre.x) * rstd + Cpre.x; s.y = (s.y - mean * Spre.y) * rstd + Cpre.y;
                s.z = (s.z - mean * Spre.z) * rstd + Cpre.z; s.w = (s.w - mean * Spre.w) * rstd + Cpre.w;
                if (KVB && col >= D) {                        // a bf16 cache: k and v are rounded where they enter it, and this step's
                    s.x = bf16_round(s.x); s.y = bf16_round(s.y); s.z = bf16_round(s.z); s.w = bf16_round(s.w);      // attention reads the same values
                }
                stamp_at(l, 0, 7);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's poison of the previous layer has landed
                const int eo = (kRoffQKV + rn * 3 * D + col) * 4;
                rpublish(brs, pc + eo, po + eo, s);
                stamp_at(l, 0, 1);
                if (col >= D && rn < A.rows) {                // append k / v of this row to its stream's cache (read by later launches)
                    const int which = col / D, ci = col - which * D, h = ci / hd, j = ci - h * hd;
                    const int slot = a_slot, pos = a_pos;
                    if (pos < A.max_seq) {
                        float* cache = which == 1 ? Lp->kcache : Lp->vcache;
                        const size_t e = (((size_t)slot * H + h) * A.max_seq + pos) * hd + j;
                        if (KVB) {
                            uint2 hv;
                            hv.x = (__float_as_uint(s.x) >> 16) | (__float_as_uint(s.y) & 0xffff0000u);
                            hv.y = (__float_as_uint(s.z) >> 16) | (__float_as_uint(s.w) & 0xffff0000u);
                            *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(cache) + e) = hv;
                        } else *reinterpret_cast<float4*>(cache + e) = s;
                    } else *A.err = 950;                      // KV cache full: GVC_ERR_STATE on the host's next call
                }
            }
        }
        // =================== B: attention of one (row, head, key chunk) per workgroup ===================
        if (wg < R * SH * nch) {
            GVC_PHASE_BEGIN();
            const int ch = wg % nch, h = (wg / nch) % SH, n = wg / (nch * SH);           // h: super-head
            const int sub = lane / lpk, hreal = h * (HD / hd) + sub, dl = (lane - sub * lpk) * 4;    // this lane's real head, its dims inside it
            const bool active = n < A.rows;
            const int bstream = active ? n / A.T : 0, t = active ? n - bstream * A.T : 0, r0 = bstream * A.T;
            const int slot = b_slot;
            const int base = b_base;
            const int k0 = active ? (int)(((long long)base * ch) / nch) : 0, k1 = active ? (int)(((long long)base * (ch + 1)) / nch) : 0;
            const bool last = active && ch == nch - 1;           // the new rows [r0, n] of this very step belong to the last chunk
            constexpr int ESZ = KVB ? 2 : 4;                     // bytes per cache element
            // the slot's rows of the layer's K / V cache as buffers ([head][max_seq][hd]); a lane addresses its own real head
            const unsigned slot_bytes = (unsigned)H * A.max_seq * hd * ESZ;
            const size_t slot_off = (size_t)slot * H * A.max_seq * hd * ESZ;
            const __amdgpu_buffer_rsrc_t krs = make_rsrc(reinterpret_cast<const char*>(Lp->kcache) + slot_off, slot_bytes);
            const __amdgpu_buffer_rsrc_t vrs = make_rsrc(reinterpret_cast<const char*>(Lp->vcache) + slot_off, slot_bytes);
            constexpr int U = 10;                                // keys per wave and pass: 80 keys of the chunk per pass
            float4 kr[U], vr[U];
            auto load_pass = [&](int kb) {                       // keys kb + wave + 8 u of the cache (written by earlier launches)
                const int voff = ((hreal * A.max_seq + kb + wave) * hd + dl) * ESZ;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (KVB) {                                   // 8 bytes = 4 bf16 per lane, widened at use
                        pu32x2 kq = {0u, 0u}, vq = {0u, 0u};
                        if (kb + wave + u * kPCW < k1) {
                            kq = __builtin_amdgcn_raw_buffer_load_b64(krs, voff, u * kPCW * hd * ESZ, 0);
                            vq = __builtin_amdgcn_raw_buffer_load_b64(vrs, voff, u * kPCW * hd * ESZ, 0);
                        }
                        kr[u] = make_float4(__uint_as_float(kq.x << 16), __uint_as_float(kq.x & 0xffff0000u), __uint_as_float(kq.y << 16), __uint_as_float(kq.y & 0xffff0000u));
                        vr[u] = make_float4(__uint_as_float(vq.x << 16), __uint_as_float(vq.x & 0xffff0000u), __uint_as_float(vq.y << 16), __uint_as_float(vq.y & 0xffff0000u));
                    } else {
                        pu32x4 kq = {0u, 0u, 0u, 0u}, vq = {0u, 0u, 0u, 0u};
                        if (kb + wave + u * kPCW < k1) {
                            kq = __builtin_amdgcn_raw_buffer_load_b128(krs, voff, u * kPCW * hd * ESZ, 0);
                            vq = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, u * kPCW * hd * ESZ, 0);
                        }
                        kr[u] = as_f4(kq); vr[u] = as_f4(vq);
                    }
                }
            };
            load_pass(k0);                                       // requested ahead of the seam
            // q of (row, super-head): 256 floats = one 16-byte piece per lane, gathered by EVERY wave for itself (1 KiB per wave from L2
            // instead of wave 0 -> LDS -> barrier: one barrier less on the critical path of the phase).
            // New rows r0 + j <= n of this very step: k and v from the hand-off buffer; row j belongs to wave (j + 1) % 8.  A decode
            // step's single new key (j = 0: the row itself) comes in the same round trip as q: q | k | v are D floats apart.
            float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 kn[2], vn[2];
            bool has_new[2] = {false, false};
            kn[0] = vn[0] = kn[1] = vn[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int j0 = (wave + kPCW - 1) & (kPCW - 1);
            const int qoff = pc + (kRoffQKV + n * 3 * D + h * HD) * 4 + lane * 16;
            if (last && j0 <= t && r0 + j0 == n) {               // (wave-uniform)
                has_new[0] = true;
                pu32x4 qkv[3];
                rgather<3>(c, brs, qoff, D * 4, qkv, 200 + l, true);
                q4 = as_f4(qkv[0]); kn[0] = as_f4(qkv[1]); vn[0] = as_f4(qkv[2]);
            } else {
                if (active) {
                    pu32x4 qv[1];
                    rgather<1>(c, brs, qoff, 0, qv, 200 + l);
                    q4 = as_f4(qv[0]);
                }
                if (last && j0 <= t) {
                    has_new[0] = true;
                    pu32x4 kv[2];
                    rgather<2>(c, brs, pc + (kRoffQKV + (r0 + j0) * 3 * D + D + h * HD) * 4 + lane * 16, D * 4, kv, 210 + l, true);
                    kn[0] = as_f4(kv[0]); vn[0] = as_f4(kv[1]);
                }
            }
            if (last && j0 + kPCW <= t) {
                has_new[1] = true;
                pu32x4 kv[2];
                rgather<2>(c, brs, pc + (kRoffQKV + (r0 + j0 + kPCW) * 3 * D + D + h * HD) * 4 + lane * 16, D * 4, kv, 210 + l, true);
                kn[1] = as_f4(kv[0]); vn[1] = as_f4(kv[1]);
            }
            stamp_at(l, 1, 0);
            float m = -INFINITY, lsum = 0.f;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            // online softmax over a batch of NB scores (-inf: no key): one rescale per batch
            auto fold = [&](auto& sc, auto& vv, auto nb) {
                constexpr int NB = decltype(nb)::value;
                float mn = m;
#pragma unroll
                for (int u = 0; u < NB; ++u) mn = fmaxf(mn, sc[u]);
                if (mn > -INFINITY) {                            // (wave-uniform)
                    const float alpha = __expf(m - mn);
                    lsum *= alpha; o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
#pragma unroll
                    for (int u = 0; u < NB; ++u) {
                        const float p = __expf(sc[u] - mn);
                        lsum += p;
                        o.x = fmaf(p, vv[u].x, o.x); o.y = fmaf(p, vv[u].y, o.y);
                        o.z = fmaf(p, vv[u].z, o.z); o.w = fmaf(p, vv[u].w, o.w);
                    }
                    m = mn;
                }
            };
            for (int kb = k0; kb < k1; kb += U * kPCW) {
                if (kb > k0) load_pass(kb);
                float sc[U];
#pragma unroll
                for (int u = 0; u < U; ++u) sc[u] = dot4(q4, kr[u]);
#pragma unroll
                for (int u = 0; u < U; ++u) sc[u] = kb + wave + u * kPCW < k1 ? group_sum(sc[u], lpk) * scale : -INFINITY;
                fold(sc, vr, std::integral_constant<int, U>());
            }
            {
                float sc[2];
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) sc[jj] = has_new[jj] ? group_sum(dot4(q4, kn[jj]), lpk) * scale : -INFINITY;
                fold(sc, vn, std::integral_constant<int, 2>());
            }
            stamp_at(l, 1, 4);
            float* m_s = ascr + 256;                             // [kPCW][4 real heads of the super-head]
            float* l_s = m_s + kPCW * 4;
            float* o_s = l_s + kPCW * 4;
            if (dl == 0) { m_s[wave * 4 + sub] = m; l_s[wave * 4 + sub] = lsum; }
            *reinterpret_cast<float4*>(o_s + wave * 256 + lane * 4) = o;
            stamp_at(l, 1, 5);
            cbar(c);
            stamp_at(l, 1, 6);
            if (wave == 0) {
                float M = -INFINITY;
#pragma unroll
                for (int i = 0; i < kPCW; ++i) M = fmaxf(M, m_s[i * 4 + sub]);
                float Lt = 0.f;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (M > -INFINITY) {                                       // (uniform: every real head sees the same keys)
#pragma unroll
                    for (int i = 0; i < kPCW; ++i) {
                        const float wgt = __expf(m_s[i * 4 + sub] - M);    // waves without a key: exp(-inf) = 0
                        const float4 oi = *reinterpret_cast<const float4*>(o_s + i * 256 + lane * 4);
                        Lt += wgt * l_s[i * 4 + sub];
                        acc.x = fmaf(wgt, oi.x, acc.x); acc.y = fmaf(wgt, oi.y, acc.y);
                        acc.z = fmaf(wgt, oi.z, acc.z); acc.w = fmaf(wgt, oi.w, acc.w);
                    }
                    const float inv = 1.0f / Lt;
                    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
                } else { M = -1e30f; Lt = 0.f; }                           // a chunk without keys (or a padding row): weight 0
                stamp_at(l, 1, 7);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                // frag position of (row n, k-quad h * 64 + lane) in C's B-operand order
                const int q = h * 64 + lane;
                const int fl = (q / KK) * 64 + (((n >> 2) * KK + (q % KK)) * 4 + (n & 3));
                const int eo = (kRoffOP + ch * R * D) * 4 + fl * 16;
                rpublish(brs, pc + eo, po + eo, acc);
                if (dl == 0) {                                             // one {m, l} piece per real head
                    const int mo = (kRoffML + ((ch * kRMaxRows + n) * kRMaxHeads + hreal) * 4) * 4;
                    rpublish(brs, pc + mo, po + mo, make_float4(M, Lt, 0.f, 0.f));
                }
            }
            stamp_at(l, 1, 1);
        }
        // =================== C: merge chunk partials -> attn c_proj -> x' = x + ... ===================
        {
            GVC_PHASE_BEGIN();
            const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
            const int s0 = wave * NSX;
            float4 ov[NSX];
            const float4 bpre = *reinterpret_cast<const float4*>(Lp->proj_b + wg * 4);
            if (nch == 1) {                                  // one chunk: its partial IS the head output (already normalised)
                pu32x4 raw[NSX];
                rgather<NSX>(c, brs, pc + kRoffOP * 4 + s0 * 1024 + lane * 16, 1024, raw, 300 + l);
#pragma unroll
                for (int i = 0; i < NSX; ++i) ov[i] = as_f4(raw[i]);
            } else {
                // several chunks: merge their (o, m, l) in chunk order.  The wave's 32 k-quads (128 dims) lie in ONE real head with
                // head_dim 256 / 128 and in TWO with head_dim 64 (the first and the second half of its steps): two {m, l} sets, `a` for
                // steps < NSX / 2 and `b` for the rest (the same head twice unless head_dim is 64).  All chunks are requested in one
                // round trip (16 rows x 4 chunks: two trips of two chunks, 64 registers each)
                const int ha = (wave * 128) / hd, hb = (wave * 128 + 64) / hd;
                const int ooff = pc + kRoffOP * 4 + s0 * 1024 + lane * 16, ocoff = R * D * 4;
                const int moff = pc + (kRoffML + (n * kRMaxHeads + ha) * 4) * 4, moff2 = pc + (kRoffML + (n * kRMaxHeads + hb) * 4) * 4;
                const int mcoff = kRMaxRows * kRMaxHeads * 4 * 4;
                float Ma = 0.f, Wa = 0.f, Mb = 0.f, Wb = 0.f;            // running max, sum of weights (at that max), per half
                auto merge = [&](const pu32x4 (&raw)[NSX], pu32x4 mla, pu32x4 mlb, bool first) {
                    if (first) {
#pragma unroll
                        for (int i = 0; i < NSX; ++i) ov[i] = as_f4(raw[i]);
                        Ma = __uint_as_float(mla.x); Wa = __uint_as_float(mla.y);
                        Mb = __uint_as_float(mlb.x); Wb = __uint_as_float(mlb.y);
                        return;
                    }
                    float fa[2], fb[2];
                    {
                        const float mc = __uint_as_float(mla.x), lc = __uint_as_float(mla.y);
                        const float Mn = fmaxf(Ma, mc);
                        const float wa = Wa * __expf(Ma - Mn), wb = lc * __expf(mc - Mn);
                        const float tot = wa + wb;
                        fa[0] = tot > 0.f ? wa / tot : 0.f; fb[0] = tot > 0.f ? wb / tot : 0.f;
                        Ma = Mn; Wa = tot;
                    }
                    {
                        const float mc = __uint_as_float(mlb.x), lc = __uint_as_float(mlb.y);
                        const float Mn = fmaxf(Mb, mc);
                        const float wa = Wb * __expf(Mb - Mn), wb = lc * __expf(mc - Mn);
                        const float tot = wa + wb;
                        fa[1] = tot > 0.f ? wa / tot : 0.f; fb[1] = tot > 0.f ? wb / tot : 0.f;
                        Mb = Mn; Wb = tot;
                    }
#pragma unroll
                    for (int i = 0; i < NSX; ++i) {
                        const float4 oc = as_f4(raw[i]);
                        const float a = fa[i < NSX / 2 ? 0 : 1], b = fb[i < NSX / 2 ? 0 : 1];
                        ov[i].x = a * ov[i].x + b * oc.x; ov[i].y = a * ov[i].y + b * oc.y;
                        ov[i].z = a * ov[i].z + b * oc.z; ov[i].w = a * ov[i].w + b * oc.w;
                    }
                };
                if (nch == 2) {
                    pu32x4 raw[2][NSX], ml[2], ml2[2];
                    rgather_chunks<2, NSX>(c, brs, ooff, 1024, ocoff, moff, moff2, mcoff, raw, ml, ml2, 320 + l);
                    merge(raw[0], ml[0], ml2[0], true); merge(raw[1], ml[1], ml2[1], false);
                } else if constexpr (NSX <= 4) {
                    pu32x4 raw[4][NSX], ml[4], ml2[4];
                    rgather_chunks<4, NSX>(c, brs, ooff, 1024, ocoff, moff, moff2, mcoff, raw, ml, ml2, 320 + l);
                    merge(raw[0], ml[0], ml2[0], true); merge(raw[1], ml[1], ml2[1], false); merge(raw[2], ml[2], ml2[2], false); merge(raw[3], ml[3], ml2[3], false);
                } else {
#pragma unroll 1
                    for (int c2 = 0; c2 < 4; c2 += 2) {
                        pu32x4 raw[2][NSX], ml[2], ml2[2];
                        rgather_chunks<2, NSX>(c, brs, ooff + c2 * ocoff, 1024, ocoff, moff + c2 * mcoff, moff2 + c2 * mcoff, mcoff, raw, ml, ml2, 320 + l);
                        merge(raw[0], ml[0], ml2[0], c2 == 0); merge(raw[1], ml[1], ml2[1], false);
                    }
                }
            }
            stamp_at(l, 2, 0);
            pf32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            wait_fill(c, fs);
            stamp_at(l, 2, 3);
            {
                const char* wbase = ring + (size_t)(fs & rmask) * kPSlot + ((s0 * STEPB + (lane & LMASK) * 16) >> WB);
                float4 wc = ldw4<WB>(wbase), wn;
#pragma unroll
                for (int i = 0; i < NSX; ++i) {
                    if (i + 1 < NSX) wn = ldw4<WB>(wbase + (((i + 1) * STEPB) >> WB));
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wc.x, ov[i].x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wc.y, ov[i].y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(wc.z, ov[i].z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(wc.w, ov[i].w, acc1, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    wc = wn;
                }
            }
            stamp_at(l, 2, 4);
            {
                const float4 r = make_float4(kk_sum<R>(acc0[0] + acc1[0]), kk_sum<R>(acc0[1] + acc1[1]), kk_sum<R>(acc0[2] + acc1[2]),
                                             kk_sum<R>(acc0[3] + acc1[3]));
                if (kk == 0) *reinterpret_cast<float4*>(red + ((wave * 4 + 0) * kRMaxRows + n) * 4) = r;
            }
            fs += 1;
            phase_done();
            stamp_at(l, 2, 5);
            cbar(c);
            stamp_at(l, 2, 6);
            if (wave == 0 && lane < R) {
                const int rn = lane;
                float4 s = *reinterpret_cast<const float4*>(red + ((0 * 4 + 0) * kRMaxRows + rn) * 4);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(red + ((w * 4 + 0) * kRMaxRows + rn) * 4);
                    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
                }
                const float4 bi = bpre;
                const float4 xr = *reinterpret_cast<const float4*>(resid + rn * 4);
                s.x = xr.x + (s.x + bi.x); s.y = xr.y + (s.y + bi.y); s.z = xr.z + (s.z + bi.z); s.w = xr.w + (s.w + bi.w);
                *reinterpret_cast<float4*>(resid + rn * 4) = s;           // x': the residual of phase E
                stamp_at(l, 2, 7);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int fl = (wg / KK) * 64 + (((rn >> 2) * KK + (wg % KK)) * 4 + (rn & 3));
                const int eo = kRoffX0 * 4 + fl * 16;
                rpublish(brs, pc + eo, po + eo, s);
            }
            stamp_at(l, 2, 1);
        }
        // =================== D: LN2 -> c_fc rows -> gelu_new ===================
        {
            GVC_PHASE_BEGIN();
            const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
            const int s0 = wave * NSX;
            float4 xv[NSX];
            // (LayerNorm folded into c_fc as in phase A)
            const float4 Spre = *reinterpret_cast<const float4*>(Lp->lnS_d + wg * 16 + (wave & 3) * 4);
            const float4 Cpre = *reinterpret_cast<const float4*>(Lp->lnC_d + wg * 16 + (wave & 3) * 4);
            if (lane < 32) *reinterpret_cast<float4*>(gbs + (wave * 64 + lane) * 4) = *reinterpret_cast<const float4*>(Lp->ln2_w + (s0 * KK + lane) * 4);
            {
                pu32x4 raw[NSX];
                rgather<NSX>(c, brs, pc + kRoffX0 * 4 + s0 * 1024 + lane * 16, 1024, raw, 400 + l);
#pragma unroll
                for (int i = 0; i < NSX; ++i) xv[i] = as_f4(raw[i]);
            }
            stamp_at(l, 3, 0);
            // residual of phase E: this workgroup finishes columns [8 (wg / 2), +8) there = k-quads 2 (wg / 2) + {0, 1} of x'
#pragma unroll
            for (int ge = 0; ge < 2; ++ge) {
                const int qe = (wg >> 1) * 2 + ge, sq = qe / KK;
                if (sq >= s0 && sq < s0 + NSX && kk == qe % KK) {
#pragma unroll
                    for (int i = 0; i < NSX; ++i)
                        if (i == sq - s0) *reinterpret_cast<float4*>(resid2 + (ge * kRMaxRows + n) * 4) = xv[i];
                }
            }
            wait_fill(c, fs + 3);
            stamp_at(l, 3, 3);
            const char* wbase = ring + ((s0 * STEPB + (lane & LMASK) * 16) >> WB);
            const float* gb = gbs + (wave * 64 + kk) * 4;
            float4 wc[4], wn[4], gc, gn;
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) wc[rg] = ldw4<WB>(wbase + (size_t)((fs + rg) & rmask) * kPSlot);
            gc = *reinterpret_cast<const float4*>(gb);
            float ln_mw = 0.f, ln_m2 = 0.f;          // (statistics in the shadow of the MFMA steps, as in phase A)
            pf32x4 acc[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[rg] = (pf32x4){0.f, 0.f, 0.f, 0.f};
            {
                // one step (four MFMAs per row group) at a time, the next step's operands requested a step ahead: left alone, the
                // scheduler hoists every LDS read of the unrolled loop to its top (128 registers of weights) and spills
#pragma unroll
                for (int i = 0; i < NSX; ++i) {
                    if (i + 1 < NSX) {
#pragma unroll
                        for (int rg = 0; rg < 4; ++rg) wn[rg] = ldw4<WB>(wbase + (size_t)((fs + rg) & rmask) * kPSlot + (((i + 1) * STEPB) >> WB));
                        gn = *reinterpret_cast<const float4*>(gb + (i + 1) * KK * 4);
                    }
                    const float x0 = xv[i].x * gc.x, x1 = xv[i].y * gc.y, x2 = xv[i].z * gc.z, x3 = xv[i].w * gc.w;
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[rg].x, x0, acc[rg], 0, 0, 0);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[rg].y, x1, acc[rg], 0, 0, 0);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[rg].z, x2, acc[rg], 0, 0, 0);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc[rg].w, x3, acc[rg], 0, 0, 0);
                    // LayerNorm statistics of the wave's K-slice, one dependent piece per step
                    if (i == 0) {
                        float sm = 0.f;
#pragma unroll
                        for (int j = 0; j < NSX; ++j) sm += (xv[j].x + xv[j].y) + (xv[j].z + xv[j].w);
                        ln_mw = kk_sum<R>(sm) * (1.0f / (float)(NSX * KK * 4));
                    }
                    if (i == 1) {
#pragma unroll
                        for (int j = 0; j < NSX; ++j) {
                            const float a0 = xv[j].x - ln_mw, a1 = xv[j].y - ln_mw, a2 = xv[j].z - ln_mw, a3 = xv[j].w - ln_mw;
                            ln_m2 += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
                        }
                    }
                    if (i == 2) {
                        ln_m2 = kk_sum<R>(ln_m2);
                        if (kk == 0) { stat[wave * 16 + n] = ln_mw; stat[kPCW * 16 + wave * 16 + n] = ln_m2; }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int rg = 0; rg < 4; ++rg) wc[rg] = wn[rg];
                    gc = gn;
                }
            }
            stamp_at(l, 3, 4);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 r = make_float4(kk_sum<R>(acc[rg][0]), kk_sum<R>(acc[rg][1]), kk_sum<R>(acc[rg][2]), kk_sum<R>(acc[rg][3]));
                if (kk == 0) *reinterpret_cast<float4*>(red + ((wave * 4 + rg) * kRMaxRows + n) * 4) = r;
            }
            fs += 4;
            phase_done();
            stamp_at(l, 3, 5);
            cbar(c);
            stamp_at(l, 3, 6);
            if (wave < 4 && lane < R) {
                const int rg = wave, rn = lane;
                float4 s = *reinterpret_cast<const float4*>(red + ((0 * 4 + rg) * kRMaxRows + rn) * 4);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(red + ((w * 4 + rg) * kRMaxRows + rn) * 4);
                    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
                }
                float mean, rstd;
                ln_merge(stat, rn, (float)(NSX * KK * 4), 1.0f / (float)D, mean, rstd);
                s.x = gelu_new((s.x - mean * Spre.x) * rstd + Cpre.x); s.y = gelu_new((s.y - mean * Spre.y) * rstd + Cpre.y);
                s.z = gelu_new((s.z - mean * Spre.z) * rstd + Cpre.z); s.w = gelu_new((s.w - mean * Spre.w) * rstd + Cpre.w);
                stamp_at(l, 3, 7);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int q = wg * 4 + rg;                   // k-quad of the hidden units [16 wg + 4 rg, +4)
                const int fl = (q / KK) * 64 + (((rn >> 2) * KK + (q % KK)) * 4 + (rn & 3));
                const int eo = kRoffHH * 4 + fl * 16;
                rpublish(brs, pc + eo, po + eo, s);
            }
            stamp_at(l, 3, 1);
        }
        // =================== E: mlp c_proj, columns [8 cb, +8) over K-half kh -> plane kh of x = x' + ... ===================
        // (workgroup (cb, kh) = (wg / 2, wg % 2): half of h to gather -- 128 KB instead of 256 KB at 16 rows, the largest hand-off of the
        //  layer -- for one more plane in the next phase-A gather; two independent accumulator groups)
        {
            GVC_PHASE_BEGIN();
            const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
            constexpr int NSE = NSH / 2;                             // MFMA steps per wave over 2048 inputs (16 rows: 16, 8 rows: 8)
            const int cb = wg >> 1, kh = wg & 1;
            const int sl = wave * NSE;                               // first step inside the K-half
            const float4 bpre = *reinterpret_cast<const float4*>(Lp->p2_b + cb * 8 + (wave & 1) * 4);
            pf32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            pu32x4 raw[NSE];
            rgather<NSE>(c, brs, pc + kRoffHH * 4 + (kh * kPCW * NSE + sl) * 1024 + lane * 16, 1024, raw, 500 + l);
            stamp_at(l, 4, 0);
            {
                // group ge lies in fills fs + 2 ge, fs + 2 ge + 1 (32 KiB of fp32); the wave's NSE steps sit inside one fill
                const unsigned boff = (unsigned)sl * STEPB;
                wait_fill(c, fs + 2 + (boff >> 14));
                stamp_at(l, 4, 3);
                const char* wb0 = ring + (((boff & 16383u) + (lane & LMASK) * 16) >> WB);
                const char* w0 = wb0 + (size_t)((fs + (boff >> 14)) & rmask) * kPSlot;
                const char* w1 = wb0 + (size_t)((fs + 2 + (boff >> 14)) & rmask) * kPSlot;
                float4 wc0 = ldw4<WB>(w0), wc1 = ldw4<WB>(w1), wn0, wn1;
#pragma unroll
                for (int i = 0; i < NSE; ++i) {
                    if (i + 1 < NSE) { wn0 = ldw4<WB>(w0 + (((i + 1) * STEPB) >> WB)); wn1 = ldw4<WB>(w1 + (((i + 1) * STEPB) >> WB)); }
                    const float4 hv = as_f4(raw[i]);
                    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc0.x, hv.x, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc1.x, hv.x, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc0.y, hv.y, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc1.y, hv.y, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc0.z, hv.z, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc1.z, hv.z, acc[1], 0, 0, 0);
                    acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc0.w, hv.w, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(wc1.w, hv.w, acc[1], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    wc0 = wn0; wc1 = wn1;
                }
            }
            stamp_at(l, 4, 4);
#pragma unroll
            for (int ge = 0; ge < 2; ++ge) {
                const float4 r = make_float4(kk_sum<R>(acc[ge][0]), kk_sum<R>(acc[ge][1]), kk_sum<R>(acc[ge][2]), kk_sum<R>(acc[ge][3]));
                if (kk == 0) *reinterpret_cast<float4*>(red + ((wave * 4 + ge) * kRMaxRows + n) * 4) = r;
            }
            fs += 4;
            phase_done();
            stamp_at(l, 4, 5);
            cbar(c);
            stamp_at(l, 4, 6);
            if (wave < 2 && lane < R) {
                const int ge = wave, rn = lane;
                float4 s = *reinterpret_cast<const float4*>(red + ((0 * 4 + ge) * kRMaxRows + rn) * 4);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(red + ((w * 4 + ge) * kRMaxRows + rn) * 4);
                    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
                }
                if (kh == 0) {                               // plane 0 carries the residual x' and the bias
                    const float4 bi = bpre;
                    const float4 xr = *reinterpret_cast<const float4*>(resid2 + (ge * kRMaxRows + rn) * 4);
                    s.x = xr.x + (s.x + bi.x); s.y = xr.y + (s.y + bi.y); s.z = xr.z + (s.z + bi.z); s.w = xr.w + (s.w + bi.w);
                }
                stamp_at(l, 4, 7);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int q = cb * 2 + ge;
                const int fl = (q / KK) * 64 + (((rn >> 2) * KK + (q % KK)) * 4 + (rn & 3));
                const int eo = (kRoffX1 + kh * R * D) * 4 + fl * 16;
                rpublish(brs, pc + eo, po + eo, s);
            }
            stamp_at(l, 4, 1);
        }
    }
    // =================== output rows: x of the last layer = its two planes, written row-major by workgroup 0 ===================
    if (wg == 0) {
        asm volatile("" : "+v"(c.lane));
        const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
        const int s0 = wave * NSX;
        const int pl = ((A.n_layer - 1) & 1) * kRParFloats * 4;
        pu32x4 raw[NSX], raw1[NSX];
        rgather2<NSX>(c, brs, pl + kRoffX1 * 4 + s0 * 1024 + lane * 16, 1024, R * D * 4, raw, raw1, 600);
        if (n < A.rows) {
#pragma unroll
            for (int i = 0; i < NSX; ++i) {
                const float4 a = as_f4(raw[i]), b2 = as_f4(raw1[i]);
                *reinterpret_cast<float4*>(A.x + (size_t)n * D + ((s0 + i) * KK + kk) * 4) = make_float4(a.x + b2.x, a.y + b2.y, a.z + b2.z, a.w + b2.w);
            }
        }
    }
#undef GVC_PHASE_BEGIN
}

}  // namespace gvc
