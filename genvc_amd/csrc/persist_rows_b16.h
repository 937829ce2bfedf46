// One-launch block stack for 2..16 rows, bf16-ACTIVATION mode (weight_dtype 3 of include/genvc_hip.h; BASELINE configs[3]: "GenVC_large
// streaming bf16, 8 concurrent streams").  Reference arithmetic: one iteration of layers/stream_generator.py:809-881 over B streams, each
// row one stream's new position (gpt_inference.py:92-112), block math SURVEY.md Appendix A.
//
// Same engine and the same five phases per layer as k_rows_persist (persist_rows.h).  Measured there (profiles/r05_microbench_notes.md
// section 1): ~11 us of a 23 us layer are hand-off "waits" that are really the TRANSFER TIME of the 192 KB every workgroup gathers per
// layer through device-scope loads, and the projections run fp32 4x4x1 MFMAs at the vector rate on operands widened from bf16.  Here:
//   * the four activations that cross a hand-off -- x into LN1, the attention output, x' into LN2, the gelu output -- are rounded to bf16
//     ONCE, by their producer, where they are published (8-byte write-through stores); a consumer lane's 16-byte piece holds EIGHT
//     consecutive inputs of one row.  The gathers shrink from 192 KB to 112 KB per workgroup and layer at 8 rows;
//   * the projections run on v_mfma_f32_4x4x4_16b_bf16 with fp32 accumulation: a 16-byte LDS read of the ring is the A operand of two MFMAs, a
//     gathered piece their B operand -- no widening, no per-element multiply: the LayerNorm gain is folded into the packed weights
//     (W' = bf16(W g)) and the LayerNorm bias into a per-output constant, the statistics come from the rounded row;
//   * the mlp c_proj is NOT split over K-halves any more (bf16 halves that gather anyway): workgroup wg owns output columns [4 wg, 4 wg + 4) in
//     phases C and E alike, so the fp32 residual stream of those columns never leaves the workgroup (LDS), x crosses as ONE bf16 plane, and the
//     last layer's finalising lanes write the output rows themselves (no closing gather);
//   * q stays fp32; k / v are rounded to bf16 where they enter the cache (as in mode 2) and this step's own attention reads those values;
//   * key chunks: with one chunk per (row, head) the normalised head output is published in bf16; with 2 / 4 chunks the partials stay fp32
//     (merge weights need them) and the merged output is rounded by the consumer.
// What a bf16-autocast run of the reference would round is LN(x) rather than x: the same 2^-9 relative step one normalisation earlier.  The
// oracle restates exactly these rounding points (oracle/genvc_oracle.py: dims["act_bf16"]); bf16 cannot claim bit-exactness (SURVEY.md section
// 7): a value within float rounding of a bf16 boundary lands one bf16 ulp apart in two summation orders -- tests assert an agreement rate and
// a tolerance.  Hand-off protocol (poison / two parities), loader, LDS map, error handling: persist_rows.h.
#pragma once
#include "persist_rows.h"

namespace gvc {

typedef short ps16x4 __attribute__((ext_vector_type(4)));

constexpr int kBWgLayerBytes = 96 * 1024;         // packed bf16 weights per workgroup and layer: 12 ring fills of 8 KiB
// hand-off buffers (BYTES inside one parity).  The four bf16 buffers every workgroup gathers exist in kBCopies COPIES, one per XCD: a
// producer stores its element into every copy, a consumer reads the copy of the XCD it runs on (XCC_ID) -- 32 readers per line instead
// of 256.  (A 16 KB buffer read by all 256 workgroups at once lives in a handful of memory-side channels: the gather time is set by
// those channels, not by the byte count -- halving the bytes alone moved the step by 5 %, profiles/r06_microbench_notes.md.)
constexpr int kBCopies = 8;
constexpr int kBoffQKV = 0;                                                       // fp32 [row][3 D]
constexpr int kBoffOP = kBoffQKV + kRMaxRows * 3 * kRD * 4;                       // fp32 [chunk][frag of R x D]: chunk partials (2 / 4 key chunks)
constexpr int kBoffML = kBoffOP + kRMaxChunks * kRMaxRows * kRD * 4;              // [chunk][row][head (16 slots)] float4 {m, l, 0, 0}
constexpr int kBoffRep = kBoffML + kRMaxChunks * kRMaxRows * kRMaxHeads * 16;     // [copy]{ OPH | X0 | X1 | HH }
constexpr int kBoffOPH = 0;                                                       // (inside a copy) bf16 frag of R x D: head outputs, one key chunk
constexpr int kBoffX0 = kBoffOPH + kRMaxRows * kRD * 2;                           // bf16 frag of R x D: x'
constexpr int kBoffX1 = kBoffX0 + kRMaxRows * kRD * 2;                            // bf16 frag of R x D: x
constexpr int kBoffHH = kBoffX1 + kRMaxRows * kRD * 2;                            // bf16 frag of R x 4 D
constexpr int kBCopyBytes = kBoffHH + kRMaxRows * 4 * kRD * 2;
constexpr int kBParBytes = kBoffRep + kBCopies * kBCopyBytes;
__host__ __device__ static inline size_t rows_b16_buf_bytes() { return (size_t)2 * kBParBytes; }

__device__ __forceinline__ ps16x4 b16lo(pu32x4 v) { const pu32x2 t = {v.x, v.y}; return __builtin_bit_cast(ps16x4, t); }
__device__ __forceinline__ ps16x4 b16hi(pu32x4 v) { const pu32x2 t = {v.z, v.w}; return __builtin_bit_cast(ps16x4, t); }
__device__ __forceinline__ unsigned pack_bf16(float a, float b) { return (unsigned)f32_to_bf16(a) | ((unsigned)f32_to_bf16(b) << 16); }
__device__ __forceinline__ pu32x2 pack_bf16x4(float4 v) { const pu32x2 r = {pack_bf16(v.x, v.y), pack_bf16(v.z, v.w)}; return r; }

// frag position (bytes) of (row rn, k-quad q) in the B-operand order of the consuming phase: a 16-byte piece per lane = one k-OCTET of one
// row, lane = ((rn / 4) KK + octet % KK) 4 + rn % 4, MFMA step = octet / KK (1 KiB per step), the quad's half inside the piece = q & 1
template <int KK>
__device__ __forceinline__ int b16_off(int rn, int q) {
    const int oct = q >> 1;
    return (oct / KK) * 1024 + ((((rn >> 2) * KK + (oct % KK)) * 4 + (rn & 3)) << 4) + (q & 1) * 8;
}
// the same (row, quad) in the fp32 frag of the chunk partials: pieces 2 step + (q & 1), one float4 per lane
template <int KK>
__device__ __forceinline__ int f32oct_off(int rn, int q) {
    const int oct = q >> 1;
    return ((oct / KK) * 2 + (q & 1)) * 1024 + ((((rn >> 2) * KK + (oct % KK)) * 4 + (rn & 3)) << 4);
}

// 8-byte write-through stores of a hand-off element (four bf16) and the poison of the same element in the other parity, the per-XCD copies dealt
// over lanes: lane group `cp0` of `ncp` groups stores copies cp0, cp0 + ncp, ...
__device__ __forceinline__ void rpublish8_dealt(__amdgpu_buffer_rsrc_t rs, int off_cur, int off_other, pu32x2 v, int ncopy, int cp0, int ncp) {
    for (int cp = cp0; cp < ncopy; cp += ncp) __builtin_amdgcn_raw_buffer_store_b64(v, rs, off_cur, cp * kBCopyBytes, 16);
    unsigned pv;
    asm volatile("v_mov_b32 %0, -1" : "=v"(pv));
    const pu32x2 p = {pv, pv};
    for (int cp = cp0; cp < ncopy; cp += ncp) __builtin_amdgcn_raw_buffer_store_b64(p, rs, off_other, cp * kBCopyBytes, 16);
}

// packed weights: [layer][wg]{ A: 3 groups of 8 KiB | C: 1 | D: 4 | E: 1 group over K = 4096 (four fills) }, a group = 4 rows x K as
// [K / 8][4 rows][8 bf16]; the c_attn / c_fc groups hold bf16(W g) (LayerNorm gain folded in; W is already bf16-rounded in this context)
__global__ void k_pack_rows_weights_b16(pu32x4* dst, const float* qkv, const float* proj, const float* fc, const float* p2, const float* g1,
                                        const float* g2) {
    constexpr int PW = kBWgLayerBytes / 16;                               // 16-byte pieces per workgroup block; 512 per fill
    const size_t n16 = (size_t)kPG * PW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const int wg = (int)(i / PW);
        int r = (int)(i - (size_t)wg * PW);
        const float *src, *g = nullptr;
        int K, row0;
        if (r < 3 * 512) { src = qkv; g = g1; K = kRD; row0 = wg * 12 + (r >> 9) * 4; r &= 511; }
        else if (r < 4 * 512) { src = proj; K = kRD; row0 = wg * 4; r -= 3 * 512; }
        else if (r < 8 * 512) { r -= 4 * 512; src = fc; g = g2; K = kRD; row0 = wg * 16 + (r >> 9) * 4; r &= 511; }
        else { r -= 8 * 512; src = p2; K = 4 * kRD; row0 = wg * 4; }
        const int t = r & 3, oct = r >> 2;
        const float* p = src + (size_t)(row0 + t) * K + oct * 8;
        float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        if (g) {
            const float4 ga = *reinterpret_cast<const float4*>(g + oct * 8), gb = *reinterpret_cast<const float4*>(g + oct * 8 + 4);
            a.x *= ga.x; a.y *= ga.y; a.z *= ga.z; a.w *= ga.w; b.x *= gb.x; b.y *= gb.y; b.z *= gb.z; b.w *= gb.w;
        }
        const pu32x4 o = {pack_bf16(a.x, a.y), pack_bf16(a.z, a.w), pack_bf16(b.x, b.y), pack_bf16(b.z, b.w)};
        dst[i] = o;
    }
}

// S_r = sum_k bf16(W_rk g_k) and C_r = sum_k W_rk b_k + bias_r (the constants of the folded LayerNorm for the bf16(W g) weights above)
__global__ void k_rows_ln_fold_b16(float* S, float* Cc, const float* W, const float* g, const float* b, const float* bias, int N, int K) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= N) return;
    double s = 0.0, c = 0.0;
    for (int k = lane; k < K; k += 64) {
        const float w = W[(size_t)row * K + k];
        s += (double)bf16_round(w * g[k]);
        c += (double)w * (double)b[k];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
    if (lane == 0) { S[row] = (float)s; Cc[row] = (float)(c + (double)bias[row]); }
}

// (mean, M2) contribution of the eight bf16 values of a piece, widened exactly
__device__ __forceinline__ float b16_sum8(pu32x4 v) {
    return ((__uint_as_float(v.x << 16) + __uint_as_float(v.x & 0xffff0000u)) + (__uint_as_float(v.y << 16) + __uint_as_float(v.y & 0xffff0000u))) +
           ((__uint_as_float(v.z << 16) + __uint_as_float(v.z & 0xffff0000u)) + (__uint_as_float(v.w << 16) + __uint_as_float(v.w & 0xffff0000u)));
}
__device__ __forceinline__ float b16_sq8(pu32x4 v, float m) {
    const float a0 = __uint_as_float(v.x << 16) - m, a1 = __uint_as_float(v.x & 0xffff0000u) - m, a2 = __uint_as_float(v.y << 16) - m,
                a3 = __uint_as_float(v.y & 0xffff0000u) - m, a4 = __uint_as_float(v.z << 16) - m, a5 = __uint_as_float(v.z & 0xffff0000u) - m,
                a6 = __uint_as_float(v.w << 16) - m, a7 = __uint_as_float(v.w & 0xffff0000u) - m;
    return ((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3)) + ((a4 * a4 + a5 * a5) + (a6 * a6 + a7 * a7));
}

template <int R, int HDR = 256>       // padded row count (8 or 16); real head_dim (256 / 128 / 64)
__global__ __launch_bounds__(kPThreads) void k_rows_persist_b16(const RowsArgs A) {
    constexpr int D = kRD, HD = kRHD;
    constexpr int G = R / 4, KK = 16 / G;            // row groups, k positions (OCTETS) per MFMA step
    constexpr int NSX = D / 8 / KK / kPCW;           // steps (two MFMAs per row group) per wave over K = D      (16 rows: 4, 8 rows: 2)
    constexpr int NSE = 4 * NSX;                     // ... over K = 4 D
    constexpr int STEPB = KK * 64;                   // bytes of a 4-row weight group per step
    constexpr int LMASK = KK * 4 - 1;                // lanes that read distinct A operands
    extern __shared__ __attribute__((aligned(16))) float smem[];
    char* ring = reinterpret_cast<char*>(smem);
    float* red = reinterpret_cast<float*>(ring + (size_t)A.ring_slots * kPSlot);      // [kPCW][4 groups][R] float4
    float* stat = red + kPCW * 4 * kRMaxRows * 4;    // [2][kPCW][16]
    float* resid = stat + 2 * kPCW * 16;             // [R] float4: the fp32 residual stream of this workgroup's four columns
    float* resid2 = resid + kRMaxRows * 4;           // (unused here: same LDS map as k_rows_persist)
    float* gbs = resid2 + 2 * kRMaxRows * 4;
    float* ascr = gbs + kPCW * 64 * 4;               // attention: q[256] | m_s[8][4] | l_s[8][4] | o_s[8][256]
    unsigned* ctl = reinterpret_cast<unsigned*>(ascr + 256 + 64 + kPCW * 256);
    if (threadIdx.x < kCtlWords) ctl[threadIdx.x] = 0u;
    __syncthreads();

    PCtx c;
    c.lane = threadIdx.x & 63; c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); c.wg = blockIdx.x;
    c.ctl = ctl; c.err = A.err; c.bar_target = 0; c.filled_seen = 0; c.dead = false;
    if (c.wave == kPCW) {
        rows_loader<1>(A, c, ring);
        return;
    }
    int& lane = c.lane;
    const int wave = c.wave, wg = c.wg;
#define GVC_PHASE_BEGIN() asm volatile("" : "+v"(c.lane), "+s"(Lp))
    const unsigned rmask = A.ring_slots - 1;
    constexpr int hd = HDR;
    constexpr int H = D / hd;
    constexpr int lpk = hd >> 2;
    constexpr int SH = D / HD;
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
    const __amdgpu_buffer_rsrc_t brs = make_rsrc(A.bufs, (unsigned)rows_b16_buf_bytes());
    // the copy of the replicated hand-off buffers this workgroup reads: the one of its XCD
    constexpr int ncopy = kBCopies;
    int rep;
    {
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        rep = kBoffRep + (ncopy > 1 ? (int)(xcc & (unsigned)(ncopy - 1)) : 0) * kBCopyBytes;
    }
    const float scale = 1.0f / sqrtf((float)hd);
    // stamps (GVC_PERSIST_STAMPS; wave 0, lane 0 of workgroup 0, every layer): [(l * 5 + p) * 8 + k], k = 0 input gathered, 1 output published,
    // 3 weight fills waited for, 4 MFMA loop done, 6 barrier passed
    const bool stamp0 = A.dbg && wave == 0 && c.lane == 0 && wg == 0;
    auto stamp_at = [&](int l, int p, int k) { if (stamp0) A.dbg[(l * 5 + p) * 8 + k] = wall_clock64(); };
    unsigned fs = 0;
    auto phase_done = [&]() { if (lane == 0) lds_st(ctl + kCtlDone + wave, fs); };
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
    const int pstride = kBParBytes;

    for (int l = 0; l < A.n_layer; ++l) {
        const RowsLayer* Lp = A.layers + l;
        const int pc = (l & 1) * pstride, po = ((l & 1) ^ 1) * pstride;
        // =================== A: LN1 (folded) -> c_attn rows -> q | k | v ===================
        {
            GVC_PHASE_BEGIN();
            const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
            const int s0 = wave * NSX;
            pu32x4 xr[NSX];
            const float4 Spre = *reinterpret_cast<const float4*>(Lp->lnS_a + wg * 12 + (wave < 3 ? wave : 0) * 4);
            const float4 Cpre = *reinterpret_cast<const float4*>(Lp->lnC_a + wg * 12 + (wave < 3 ? wave : 0) * 4);
            if (l == 0) {
                // the block-stack input in fp32 (embedding rows of a decode step, or the caller's rows): rounded here, once, exactly as a
                // producer would; this workgroup's own four columns start its fp32 residual stream
                const int bs = n < A.rows ? n / A.T : 0;
                const int tok = A.tok_in ? min(max(A.tok_in[n < A.rows ? n : 0], 0), A.vocab - 1) : 0;
                const int mp = A.tok_in ? A.mel_pos_idx[A.slots[bs]] : 0;
#pragma unroll
                for (int i = 0; i < NSX; ++i) {
                    const int o = (s0 + i) * KK + kk;
                    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
                    if (n < A.rows) {
                        if (A.tok_in) {
                            const float4 e0 = *reinterpret_cast<const float4*>(A.mel_emb + (size_t)tok * D + o * 8);
                            const float4 e1 = *reinterpret_cast<const float4*>(A.mel_emb + (size_t)tok * D + o * 8 + 4);
                            const float4 p0 = *reinterpret_cast<const float4*>(A.mel_pos + (size_t)mp * D + o * 8);
                            const float4 p1 = *reinterpret_cast<const float4*>(A.mel_pos + (size_t)mp * D + o * 8 + 4);
                            a = make_float4(e0.x + p0.x, e0.y + p0.y, e0.z + p0.z, e0.w + p0.w);
                            b = make_float4(e1.x + p1.x, e1.y + p1.y, e1.z + p1.z, e1.w + p1.w);
                        } else {
                            a = *reinterpret_cast<const float4*>(A.x + (size_t)n * D + o * 8);
                            b = *reinterpret_cast<const float4*>(A.x + (size_t)n * D + o * 8 + 4);
                        }
                    }
                    if (o == (wg >> 1)) *reinterpret_cast<float4*>(resid + n * 4) = (wg & 1) ? b : a;
                    const pu32x2 pa = pack_bf16x4(a), pb = pack_bf16x4(b);
                    xr[i] = (pu32x4){pa.x, pa.y, pb.x, pb.y};
                }
            } else {
                rgather<NSX>(c, brs, po + rep + kBoffX1 + s0 * 1024 + lane * 16, 1024, xr, 100 + l, NSX <= 2);
            }
            stamp_at(l, 0, 0);
            wait_fill(c, fs + 2);
            stamp_at(l, 0, 3);
            const char* wbase = ring + s0 * STEPB + (lane & LMASK) * 16;
            pf32x4 acc[3];
#pragma unroll
            for (int rg = 0; rg < 3; ++rg) acc[rg] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NSX; ++i) {
                pu32x4 wv[3];
#pragma unroll
                for (int rg = 0; rg < 3; ++rg) wv[rg] = *reinterpret_cast<const pu32x4*>(wbase + (size_t)((fs + rg) & rmask) * kPSlot + i * STEPB);
#pragma unroll
                for (int rg = 0; rg < 3; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(b16lo(wv[rg]), b16lo(xr[i]), acc[rg], 0, 0, 0);
#pragma unroll
                for (int rg = 0; rg < 3; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(b16hi(wv[rg]), b16hi(xr[i]), acc[rg], 0, 0, 0);
            }
            // LayerNorm statistics of the wave's K-slice of the rounded row -> LDS (merged by the final lanes, Chan et al.); issued behind the
            // MFMAs, which do not need them
            {
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < NSX; ++i) sm += b16_sum8(xr[i]);
                const float mw = kk_sum<R>(sm) * (1.0f / (float)(NSX * KK * 8));
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < NSX; ++i) m2 += b16_sq8(xr[i], mw);
                m2 = kk_sum<R>(m2);
                if (kk == 0) { stat[wave * 16 + n] = mw; stat[kPCW * 16 + wave * 16 + n] = m2; }
            }
            stamp_at(l, 0, 4);
#pragma unroll
            for (int rg = 0; rg < 3; ++rg) {
                const float4 r = make_float4(kk_sum<R>(acc[rg][0]), kk_sum<R>(acc[rg][1]), kk_sum<R>(acc[rg][2]), kk_sum<R>(acc[rg][3]));
                if (kk == 0) *reinterpret_cast<float4*>(red + ((wave * 4 + rg) * kRMaxRows + n) * 4) = r;
            }
            fs += 3;
            phase_done();
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
                ln_merge(stat, rn, (float)(NSX * KK * 8), 1.0f / (float)D, mean, rstd);
                s.x = (s.x - mean * Spre.x) * rstd + Cpre.x; s.y = (s.y - mean * Spre.y) * rstd + Cpre.y;
                s.z = (s.z - mean * Spre.z) * rstd + Cpre.z; s.w = (s.w - mean * Spre.w) * rstd + Cpre.w;
                if (col >= D) {                               // k and v are rounded where they enter the bf16 cache; this step's attention reads the same values
                    s.x = bf16_round(s.x); s.y = bf16_round(s.y); s.z = bf16_round(s.z); s.w = bf16_round(s.w);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's poison of the previous layer has landed
                const int eo = kBoffQKV + (rn * 3 * D + col) * 4;
                rpublish(brs, pc + eo, po + eo, s);
                stamp_at(l, 0, 1);
                if (col >= D && rn < A.rows) {                // append k / v of this row to its stream's cache (read by later launches)
                    const int which = col / D, ci = col - which * D, h = ci / hd, j = ci - h * hd;
                    if (a_pos < A.max_seq) {
                        float* cache = which == 1 ? Lp->kcache : Lp->vcache;
                        const size_t e = (((size_t)a_slot * H + h) * A.max_seq + a_pos) * hd + j;
                        *reinterpret_cast<pu32x2*>(reinterpret_cast<unsigned short*>(cache) + e) = pack_bf16x4(s);
                    } else *A.err = 950;                      // KV cache full: GVC_ERR_STATE on the host's next call
                }
            }
        }
        // =================== B: attention of one (row, head, key chunk) per workgroup ===================
        if (wg < R * SH * nch) {
            GVC_PHASE_BEGIN();
            const int ch = wg % nch, h = (wg / nch) % SH, n = wg / (nch * SH);           // h: super-head
            const int sub = lane / lpk, hreal = h * (HD / hd) + sub, dl = (lane - sub * lpk) * 4;
            const bool active = n < A.rows;
            const int bstream = active ? n / A.T : 0, t = active ? n - bstream * A.T : 0, r0 = bstream * A.T;
            const int slot = b_slot;
            const int base = b_base;
            const int k0 = active ? (int)(((long long)base * ch) / nch) : 0, k1 = active ? (int)(((long long)base * (ch + 1)) / nch) : 0;
            const bool last = active && ch == nch - 1;           // the new rows [r0, n] of this very step belong to the last chunk
            constexpr int ESZ = 2;
            const unsigned slot_bytes = (unsigned)H * A.max_seq * hd * ESZ;
            const size_t slot_off = (size_t)slot * H * A.max_seq * hd * ESZ;
            const __amdgpu_buffer_rsrc_t krs = make_rsrc(reinterpret_cast<const char*>(Lp->kcache) + slot_off, slot_bytes);
            const __amdgpu_buffer_rsrc_t vrs = make_rsrc(reinterpret_cast<const char*>(Lp->vcache) + slot_off, slot_bytes);
            constexpr int U = 10;                                // keys per wave and pass: 80 keys of the chunk per pass
            float4 kr[U], vr[U];
            auto load_pass = [&](int kb) {
                const int voff = ((hreal * A.max_seq + kb + wave) * hd + dl) * ESZ;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    pu32x2 kq = {0u, 0u}, vq = {0u, 0u};
                    if (kb + wave + u * kPCW < k1) {
                        kq = __builtin_amdgcn_raw_buffer_load_b64(krs, voff, u * kPCW * hd * ESZ, 0);
                        vq = __builtin_amdgcn_raw_buffer_load_b64(vrs, voff, u * kPCW * hd * ESZ, 0);
                    }
                    kr[u] = bf16x4_to_f4(kq);
                    vr[u] = bf16x4_to_f4(vq);
                }
            };
            load_pass(k0);                                       // requested ahead of the seam
            float4 q4 = make_float4(0.f, 0.f, 0.f, 0.f);
            float4 kn[2], vn[2];
            bool has_new[2] = {false, false};
            kn[0] = vn[0] = kn[1] = vn[1] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int j0 = (wave + kPCW - 1) & (kPCW - 1);
            const int qoff = pc + kBoffQKV + (n * 3 * D + h * HD) * 4 + lane * 16;
            if (last && j0 <= t && r0 + j0 == n) {               // (wave-uniform) a decode step's own key: q | k | v in one round trip
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
                    rgather<2>(c, brs, pc + kBoffQKV + ((r0 + j0) * 3 * D + D + h * HD) * 4 + lane * 16, D * 4, kv, 210 + l, true);
                    kn[0] = as_f4(kv[0]); vn[0] = as_f4(kv[1]);
                }
            }
            if (last && j0 + kPCW <= t) {
                has_new[1] = true;
                pu32x4 kv[2];
                rgather<2>(c, brs, pc + kBoffQKV + ((r0 + j0 + kPCW) * 3 * D + D + h * HD) * 4 + lane * 16, D * 4, kv, 210 + l, true);
                kn[1] = as_f4(kv[0]); vn[1] = as_f4(kv[1]);
            }
            stamp_at(l, 1, 0);
            float m = -INFINITY, lsum = 0.f;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
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
            cbar(c);
            stamp_at(l, 1, 6);
            // one key chunk: wave w merges for itself and publishes copy w of the head output (the eight copies leave in parallel);
            // several chunks: wave 0 publishes the fp32 partial
            if (wave == 0 || (nch == 1 && wave < ncopy)) {
                float M = -INFINITY;
#pragma unroll
                for (int i = 0; i < kPCW; ++i) M = fmaxf(M, m_s[i * 4 + sub]);
                float Lt = 0.f;
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                if (M > -INFINITY) {
#pragma unroll
                    for (int i = 0; i < kPCW; ++i) {
                        const float wgt = __expf(m_s[i * 4 + sub] - M);
                        const float4 oi = *reinterpret_cast<const float4*>(o_s + i * 256 + lane * 4);
                        Lt += wgt * l_s[i * 4 + sub];
                        acc.x = fmaf(wgt, oi.x, acc.x); acc.y = fmaf(wgt, oi.y, acc.y);
                        acc.z = fmaf(wgt, oi.z, acc.z); acc.w = fmaf(wgt, oi.w, acc.w);
                    }
                    const float inv = 1.0f / Lt;
                    acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
                } else { M = -1e30f; Lt = 0.f; }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int q = h * 64 + lane;                               // k-quad of C's input this lane holds for row n
                if (nch == 1) {                                            // the head output itself: rounded here, once
                    const int eo = kBoffRep + kBoffOPH + b16_off<KK>(n, q);
                    rpublish8_dealt(brs, pc + eo, po + eo, pack_bf16x4(acc), ncopy, wave, kPCW);
                } else {
                    const int eo = kBoffOP + ch * R * D * 4 + f32oct_off<KK>(n, q);
                    rpublish(brs, pc + eo, po + eo, acc);
                    if (dl == 0) {
                        const int mo = kBoffML + ((ch * kRMaxRows + n) * kRMaxHeads + hreal) * 16;
                        rpublish(brs, pc + mo, po + mo, make_float4(M, Lt, 0.f, 0.f));
                    }
                }
            }
            stamp_at(l, 1, 1);
        }
        // =================== C: (merge chunk partials ->) attn c_proj -> x' = x + ... ===================
        {
            GVC_PHASE_BEGIN();
            const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
            const int s0 = wave * NSX;
            pu32x4 ovh[NSX];
            const float4 bpre = *reinterpret_cast<const float4*>(Lp->proj_b + wg * 4);
            if (nch == 1) {
                rgather<NSX>(c, brs, pc + rep + kBoffOPH + s0 * 1024 + lane * 16, 1024, ovh, 300 + l, NSX <= 2);
            } else {
                // several chunks: fp32 partials (two pieces per step: the octet's two quads) merged in chunk order, then rounded
                constexpr int NP = 2 * NSX;
                float4 ov[NP];
                const int ha = (wave * 128) / hd, hb = (wave * 128 + 64) / hd;
                const int ooff = pc + kBoffOP + 2 * s0 * 1024 + lane * 16, ocoff = R * D * 4;
                const int moff = pc + kBoffML + (n * kRMaxHeads + ha) * 16, moff2 = pc + kBoffML + (n * kRMaxHeads + hb) * 16;
                const int mcoff = kRMaxRows * kRMaxHeads * 16;
                float Ma = 0.f, Wa = 0.f, Mb = 0.f, Wb = 0.f;
                auto merge = [&](const pu32x4 (&raw)[NP], pu32x4 mla, pu32x4 mlb, bool first) {
                    if (first) {
#pragma unroll
                        for (int i = 0; i < NP; ++i) ov[i] = as_f4(raw[i]);
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
                    for (int i = 0; i < NP; ++i) {
                        const float4 oc = as_f4(raw[i]);
                        const float a = fa[i < NP / 2 ? 0 : 1], b = fb[i < NP / 2 ? 0 : 1];
                        ov[i].x = a * ov[i].x + b * oc.x; ov[i].y = a * ov[i].y + b * oc.y;
                        ov[i].z = a * ov[i].z + b * oc.z; ov[i].w = a * ov[i].w + b * oc.w;
                    }
                };
                if (nch == 2) {
                    pu32x4 raw[2][NP], ml[2], ml2[2];
                    rgather_chunks<2, NP>(c, brs, ooff, 1024, ocoff, moff, moff2, mcoff, raw, ml, ml2, 320 + l);
                    merge(raw[0], ml[0], ml2[0], true); merge(raw[1], ml[1], ml2[1], false);
                } else {
#pragma unroll 1
                    for (int c2 = 0; c2 < 4; c2 += 2) {
                        pu32x4 raw[2][NP], ml[2], ml2[2];
                        rgather_chunks<2, NP>(c, brs, ooff + c2 * ocoff, 1024, ocoff, moff + c2 * mcoff, moff2 + c2 * mcoff, mcoff, raw, ml, ml2, 320 + l);
                        merge(raw[0], ml[0], ml2[0], c2 == 0); merge(raw[1], ml[1], ml2[1], false);
                    }
                }
#pragma unroll
                for (int i = 0; i < NSX; ++i) {
                    const pu32x2 pa = pack_bf16x4(ov[2 * i]), pb = pack_bf16x4(ov[2 * i + 1]);
                    ovh[i] = (pu32x4){pa.x, pa.y, pb.x, pb.y};
                }
            }
            stamp_at(l, 2, 0);
            pf32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            wait_fill(c, fs);
            stamp_at(l, 2, 3);
            {
                const char* wbase = ring + (size_t)(fs & rmask) * kPSlot + s0 * STEPB + (lane & LMASK) * 16;
#pragma unroll
                for (int i = 0; i < NSX; ++i) {
                    const pu32x4 wv = *reinterpret_cast<const pu32x4*>(wbase + i * STEPB);
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(b16lo(wv), b16lo(ovh[i]), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(b16hi(wv), b16hi(ovh[i]), acc1, 0, 0, 0);
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
            cbar(c);
            stamp_at(l, 2, 6);
            if (wave == 0) {                                  // lane = (copy group, row): every group finishes the row, each stores its own copies
                const int rn = lane & (R - 1), cp0 = lane / R;
                float4 s = *reinterpret_cast<const float4*>(red + ((0 * 4 + 0) * kRMaxRows + rn) * 4);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(red + ((w * 4 + 0) * kRMaxRows + rn) * 4);
                    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
                }
                const float4 xr = *reinterpret_cast<const float4*>(resid + rn * 4);
                s.x = xr.x + (s.x + bpre.x); s.y = xr.y + (s.y + bpre.y); s.z = xr.z + (s.z + bpre.z); s.w = xr.w + (s.w + bpre.w);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");         // every lane of the wave has read the residual before lane group 0 replaces it
                if (cp0 == 0) *reinterpret_cast<float4*>(resid + rn * 4) = s;           // x' in fp32: the residual of phase E, never leaves the workgroup
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int eo = kBoffRep + kBoffX0 + b16_off<KK>(rn, wg);
                rpublish8_dealt(brs, pc + eo, po + eo, pack_bf16x4(s), ncopy, cp0, 64 / R);
            }
            stamp_at(l, 2, 1);
        }
        // =================== D: LN2 (folded) -> c_fc rows -> gelu_new ===================
        {
            GVC_PHASE_BEGIN();
            const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
            const int s0 = wave * NSX;
            pu32x4 xr[NSX];
            const float4 Spre = *reinterpret_cast<const float4*>(Lp->lnS_d + wg * 16 + (wave & 3) * 4);
            const float4 Cpre = *reinterpret_cast<const float4*>(Lp->lnC_d + wg * 16 + (wave & 3) * 4);
            rgather<NSX>(c, brs, pc + rep + kBoffX0 + s0 * 1024 + lane * 16, 1024, xr, 400 + l, NSX <= 2);
            stamp_at(l, 3, 0);
            wait_fill(c, fs + 3);
            stamp_at(l, 3, 3);
            const char* wbase = ring + s0 * STEPB + (lane & LMASK) * 16;
            pf32x4 acc[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) acc[rg] = (pf32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < NSX; ++i) {
                pu32x4 wv[4];
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) wv[rg] = *reinterpret_cast<const pu32x4*>(wbase + (size_t)((fs + rg) & rmask) * kPSlot + i * STEPB);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(b16lo(wv[rg]), b16lo(xr[i]), acc[rg], 0, 0, 0);
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) acc[rg] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(b16hi(wv[rg]), b16hi(xr[i]), acc[rg], 0, 0, 0);
            }
            {
                float sm = 0.f;
#pragma unroll
                for (int i = 0; i < NSX; ++i) sm += b16_sum8(xr[i]);
                const float mw = kk_sum<R>(sm) * (1.0f / (float)(NSX * KK * 8));
                float m2 = 0.f;
#pragma unroll
                for (int i = 0; i < NSX; ++i) m2 += b16_sq8(xr[i], mw);
                m2 = kk_sum<R>(m2);
                if (kk == 0) { stat[wave * 16 + n] = mw; stat[kPCW * 16 + wave * 16 + n] = m2; }
            }
            stamp_at(l, 3, 4);
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const float4 r = make_float4(kk_sum<R>(acc[rg][0]), kk_sum<R>(acc[rg][1]), kk_sum<R>(acc[rg][2]), kk_sum<R>(acc[rg][3]));
                if (kk == 0) *reinterpret_cast<float4*>(red + ((wave * 4 + rg) * kRMaxRows + n) * 4) = r;
            }
            fs += 4;
            phase_done();
            cbar(c);
            stamp_at(l, 3, 6);
            if (wave < 4) {
                const int rg = wave, rn = lane & (R - 1), cp0 = lane / R;
                float4 s = *reinterpret_cast<const float4*>(red + ((0 * 4 + rg) * kRMaxRows + rn) * 4);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(red + ((w * 4 + rg) * kRMaxRows + rn) * 4);
                    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
                }
                float mean, rstd;
                ln_merge(stat, rn, (float)(NSX * KK * 8), 1.0f / (float)D, mean, rstd);
                s.x = gelu_new((s.x - mean * Spre.x) * rstd + Cpre.x); s.y = gelu_new((s.y - mean * Spre.y) * rstd + Cpre.y);
                s.z = gelu_new((s.z - mean * Spre.z) * rstd + Cpre.z); s.w = gelu_new((s.w - mean * Spre.w) * rstd + Cpre.w);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int eo = kBoffRep + kBoffHH + b16_off<KK>(rn, wg * 4 + rg);       // hidden units [16 wg + 4 rg, +4)
                rpublish8_dealt(brs, pc + eo, po + eo, pack_bf16x4(s), ncopy, cp0, 64 / R);
            }
            stamp_at(l, 3, 1);
        }
        // =================== E: mlp c_proj, columns [4 wg, +4) over the whole K = 4 D -> x = x' + ... ===================
        {
            GVC_PHASE_BEGIN();
            const int kk = (lane >> 2) & (KK - 1), n = (lane / (KK * 4)) * 4 + (lane & 3);
            const int sl = wave * NSE;                               // first step of the wave: 512 inputs per wave, two waves per 8 KiB fill
            const float4 bpre = *reinterpret_cast<const float4*>(Lp->p2_b + wg * 4);
            pf32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            pu32x4 hv[NSE];
            rgather<NSE>(c, brs, pc + rep + kBoffHH + sl * 1024 + lane * 16, 1024, hv, 500 + l);
            stamp_at(l, 4, 0);
            {
                wait_fill(c, fs + (wave >> 1));
                stamp_at(l, 4, 3);
                const char* wbase = ring + (size_t)((fs + (wave >> 1)) & rmask) * kPSlot + (wave & 1) * NSE * STEPB + (lane & LMASK) * 16;
#pragma unroll
                for (int i = 0; i < NSE; ++i) {
                    const pu32x4 wv = *reinterpret_cast<const pu32x4*>(wbase + i * STEPB);
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(b16lo(wv), b16lo(hv[i]), acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(b16hi(wv), b16hi(hv[i]), acc1, 0, 0, 0);
                }
            }
            stamp_at(l, 4, 4);
            {
                const float4 r = make_float4(kk_sum<R>(acc0[0] + acc1[0]), kk_sum<R>(acc0[1] + acc1[1]), kk_sum<R>(acc0[2] + acc1[2]),
                                             kk_sum<R>(acc0[3] + acc1[3]));
                if (kk == 0) *reinterpret_cast<float4*>(red + ((wave * 4 + 0) * kRMaxRows + n) * 4) = r;
            }
            fs += 4;
            phase_done();
            cbar(c);
            stamp_at(l, 4, 6);
            if (wave == 0) {
                const int rn = lane & (R - 1), cp0 = lane / R;
                float4 s = *reinterpret_cast<const float4*>(red + ((0 * 4 + 0) * kRMaxRows + rn) * 4);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) {
                    const float4 p = *reinterpret_cast<const float4*>(red + ((w * 4 + 0) * kRMaxRows + rn) * 4);
                    s.x += p.x; s.y += p.y; s.z += p.z; s.w += p.w;
                }
                const float4 xr = *reinterpret_cast<const float4*>(resid + rn * 4);
                s.x = xr.x + (s.x + bpre.x); s.y = xr.y + (s.y + bpre.y); s.z = xr.z + (s.z + bpre.z); s.w = xr.w + (s.w + bpre.w);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (cp0 == 0) {
                    *reinterpret_cast<float4*>(resid + rn * 4) = s;       // x in fp32: the residual of the next layer's phase C
                    if (l == A.n_layer - 1 && rn < A.rows) *reinterpret_cast<float4*>(A.x + (size_t)rn * D + wg * 4) = s;      // the block stack's output rows
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int eo = kBoffRep + kBoffX1 + b16_off<KK>(rn, wg);
                rpublish8_dealt(brs, pc + eo, po + eo, pack_bf16x4(s), ncopy, cp0, 64 / R);
            }
            stamp_at(l, 4, 1);
        }
    }
#undef GVC_PHASE_BEGIN
}

}  // namespace gvc
