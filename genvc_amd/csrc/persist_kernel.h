// One-launch decode step for ONE stream (reference: one iteration of layers/gpt_inference.py:92-112 over the
// HF GPT-2 block stack).
//
// Why one launch: with one stream a decode step is ~120 dependent weight-streaming phases of 4-17 MB each; as
// separate launches every phase pays a boundary (1.6 us) + the reload of its input vector (1.2 us) + the ramp and
// tail of its own HBM burst, and the step sits at ~0.3 of the HBM roofline.  Here 256 resident workgroups (one per
// CU) run the whole step:
//   * a LOADER wave per workgroup streams that workgroup's weight rows (contiguous in the row-per-output layout) into
//     an LDS ring with LDS-DMA (global_load_lds, non-temporal), at most two 16 KiB fills in flight, up to 8 fills
//     (~1.5 phases) ahead of the consumers: the HBM stream keeps running across the dependency seams;
//   * eight CONSUMER waves gather the phase input, do the LayerNorm / attention / dot products out of LDS and publish
//     the phase output.  Consumer waves never have a weight load outstanding (vmcnt retires in order per wave), so
//     they can poll.  A single wave retires an instruction every ~5-10 ns, so the work of a phase is spread over
//     all eight waves (rows, or K-halves of a row whose two partial sums are added by the NEXT phase's gather);
//   * a seam (all-to-all hand-off of a phase output) is a sweep over 8-byte {value, tag} granules: every output
//     element is written by ONE write-through (sc1) store and polled with sc1 loads (16 bytes = two granules per lane
//     and load) until its tag is the expected one.  A sweep of one or two loads per lane re-reads everything in every poll
//     pass (one round trip once the data is there); larger sweeps poll one sentinel load per lane and read the rest when it
//     has arrived.  The tag encodes (step epoch, layer, phase); nothing is zeroed between steps, the epoch lives in
//     device memory and is bumped by the last workgroup to finish the step.
// Per layer: A [LN1, c_attn] -> B [attention, split over <= 8 key chunks per head, a few workgroups] ->
//            C [merge of the chunk partials, attn c_proj, residual] -> D [LN2, c_fc, gelu_new] -> E [mlp c_proj, residual];
// then the double-LayerNorm head.  Contexts of <= 80 keys (head_dim 256, <= 4 heads) fuse B and C (persist_fused_attn).
// Every spin is bounded: on a timeout the workgroup sets *err, and the host raises
// GVC_ERR_STATE on the next call.
//
// XL (round 4, d_model 1024): the hidden units of the MLP never leave their XCD.  A hand-off between workgroups that share an L2
// costs 1.0 us idle / 2.6 us beside the weight stream when the producer uses PLAIN stores (they stay in the XCD's L2) and the
// consumers sc1 loads, against 2.4 / 3.7 us for the device-wide write-through hand-off (scripts/ubench/seam_xcd.hip).  Which XCD a
// workgroup runs on is read from the hardware (XCC_ID: the dispatcher deals workgroups round-robin but does not start every grid at
// XCD 0) and its rank j among the XCD's 32 workgroups comes from an atomic counter per XCD -- with one workgroup per CU resident
// every XCD holds exactly 32.  So phase D of workgroup (x, j) produces hidden units [512 x + 16 j, +16) and
// phase E multiplies the XCD's 512 hidden units into ITS K-slice of mlp c_proj for output rows [32 j, +32): the 4096-wide h hand-off
// (the dearest of the layer: 3.8 us) becomes a 512-wide XCD-local one, and x leaves phase E as eight partial planes (one per XCD)
// that the next phase A adds in its gather (64 KB instead of 16 KB per workgroup: +0.5 us).
#pragma once
#include "gpt_kernels.h"

namespace gvc {

constexpr int kPG = 256;            // workgroups: one per CU, all co-resident
constexpr int kPCW = 8;             // consumer waves per workgroup (wave kPCW is the loader)
constexpr int kPThreads = (kPCW + 1) * 64;
constexpr int kPSlot = 16384;       // bytes per ring slot = one LDS-DMA fill (16 x 1 KiB wave-instructions)
constexpr int kPMaxChunks = 8;      // key chunks per head
constexpr int kPU = 4;              // keys per lane group held in registers per pass
constexpr int kPUF = 10;            // ... in the fused attention + projection phase (short contexts: kPUF * kPCW keys)
constexpr unsigned kPSpinLimit = 400000;
constexpr int kPStampLayer = 2;     // GVC_PERSIST_STAMPS: the layer whose phases every workgroup stamps
// LDS control words
constexpr int kCtlFilled = 0, kCtlDone = 1, kCtlArrive = 1 + kPCW, kCtlAbort = 2 + kPCW, kCtlXcd = 3 + kPCW, kCtlWords = 16;

typedef unsigned long long pu64;
typedef unsigned int pu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int pu32x2 __attribute__((ext_vector_type(2)));

struct PersistLayer {
    const float *ln1_w, *ln1_b, *qkv_w, *qkv_b, *proj_w, *proj_b, *ln2_w, *ln2_b, *fc_w, *fc_b, *p2_w, *p2_b;
    float *kcache, *vcache;         // this layer's [slot][head][max_seq][hd]
};

struct PersistArgs {
    const PersistLayer* layers;
    int n_layer, d, n_head, head_dim, vocab, max_seq, max_mel_pos;
    const float *mel_emb, *mel_pos, *lnf_w, *lnf_b, *fn_w, *fn_b, *head_w, *head_b;
    const int32_t* slots;           // [1]
    const int32_t* tok_in;          // [1]
    GptState st;
    float* logits_out;              // [vocab]
    float* latent_out;              // [d]
    int32_t* step_ctr;              // nullable
    int advance;
    pu64* gran;                     // granule buffers: Q[3d] | P[kPMaxChunks][d + 2H] | X0[4][d] | HH[4d] | X1[8][d] (2 planes used without XL)
    unsigned* epoch;                // [0] step epoch, [1] arrival counter of the running step, [4..11] XL: workgroups seen per XCD (32 per launch)
    int* err;                       // device-visible host word: != 0 after a timeout
    int ring_slots;                 // power of two
    int hvec_floats, ascr_floats;
    unsigned long long* dbg;        // nullable: wall-clock stamps
};

// granules of the hand-off buffers (host: allocation size)
static inline size_t persist_granules(int d, int H) { return (size_t)3 * d + (size_t)kPMaxChunks * (d + 2 * H) + 4 * d + 4 * d + 8 * d; }

// Short contexts of the trained GenVC shape (head_dim 256, <= 4 heads): attention and the attn c_proj run as ONE phase --
// workgroup (i, h) recomputes head h's attention (<= kPUF * kPCW keys, K/V rows from L2) and multiplies it into its rows of the
// head's K-slice of c_proj; the per-head partial sums are planes of X0 that D's gather adds.  Four hand-offs per layer instead of five.
__host__ __device__ static inline bool persist_fused_attn(int n_head, int head_dim, int n_keys) {
    return head_dim == 256 && n_head <= 4 && (kPG % n_head) == 0 && n_keys <= kPUF * kPCW;
}

struct PCtx {
    int lane, wave, wg;
    unsigned* ctl;                  // LDS control words (kCtl*)
    int* err;
    unsigned bar_target;
    unsigned filled_seen;
    bool dead;
};

__device__ __forceinline__ unsigned lds_ld(const unsigned* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one failed poll: back off; gives up (and flags the whole workgroup) after ~0.2 s
__device__ __forceinline__ bool spin_fail(PCtx& c, unsigned& spins, int code, int sleep) {
    if (sleep == 0) __builtin_amdgcn_s_sleep(0);
    else if (sleep == 1) __builtin_amdgcn_s_sleep(1);
    else __builtin_amdgcn_s_sleep(3);
    ++spins;
    if ((spins & 63u) == 0u && lds_ld(c.ctl + kCtlAbort)) { c.dead = true; return true; }
    if (spins > kPSpinLimit) {
        c.dead = true;
        lds_st(c.ctl + kCtlAbort, 1u);
        if (c.lane == 0) __hip_atomic_store(c.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
    return false;
}

// barrier of the consumer waves through an LDS arrival counter (s_barrier would stop the loader wave too)
__device__ __forceinline__ void cbar(PCtx& c) {
    c.bar_target += kPCW;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c.lane == 0) __hip_atomic_fetch_add(c.ctl + kCtlArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    // tight poll first (an LDS round trip per iteration, no sleep, no bookkeeping): the waves of a workgroup arrive within a
    // fraction of a microsecond of each other; the bounded back-off loop only takes over after ~64 fast polls
    bool met = false;
    for (int i = 0; i < 64 && !c.dead; ++i)
        if (lds_ld(c.ctl + kCtlArrive) >= c.bar_target) { met = true; break; }
    unsigned spins = 0;
    while (!met && !c.dead && lds_ld(c.ctl + kCtlArrive) < c.bar_target)
        if (spin_fail(c, spins, 900, 0)) break;
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void wait_fill(PCtx& c, unsigned seq) {
    if (c.dead || c.filled_seen > seq) return;
    unsigned spins = 0;
    while (true) {
        const unsigned f = lds_ld(c.ctl + kCtlFilled);
        if (f > seq) { c.filled_seen = f; break; }
        if (spin_fail(c, spins, 901, 1)) break;
    }
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// one {value, tag} granule = ONE 8-byte write-through store (index = granule number inside the allocation)
__device__ __forceinline__ void publish(__amdgpu_buffer_rsrc_t rs, int index, unsigned tag, float v) {
    pu32x2 g;
    g.x = __float_as_uint(v); g.y = tag;
    __builtin_amdgcn_raw_buffer_store_b64(g, rs, index * 8, 0, 16);
}

// the same granule for consumers on the producer's own XCD: a plain store (it stays in the shared L2; readers use sc1 loads)
__device__ __forceinline__ void publish_local(__amdgpu_buffer_rsrc_t rs, int index, unsigned tag, float v) {
    pu32x2 g;
    g.x = __float_as_uint(v); g.y = tag;
    __builtin_amdgcn_raw_buffer_store_b64(g, rs, index * 8, 0, 0);
}

// Sweep of a phase input by the consumer waves.  The input has NP planes of n granules (a K-split producer publishes
// one partial sum per plane; the value is their sum, plane 0 first); item t of plane p sits at granule base + p * n + t.
// A lane reads TWO adjacent granules per load (16 bytes): lane (wave, lane) owns items 2 (wave * 64 + lane) + {0, 1} +
// j * 128 kPCW, j < NJ.  Sweeps of more than two loads per lane poll a sentinel first (load j = 0 of the last plane) and issue
// the other loads when its tags have arrived (everything is re-issued if a tag is still old); one- and two-load sweeps re-read
// everything in every poll pass, which saves a round trip (measured: 612 vs 630 us per step at 110-250 keys).  n is a multiple of 128.
// gd (diagnostics, GVC_PERSIST_STAMPS): [0] entry, [1] sentinel seen, [2] done (wall clock), [3] sentinel polls, [4] sweep passes
template <int NJ, int NP>
__device__ __forceinline__ void gather(PCtx& c, __amdgpu_buffer_rsrc_t rs, int base, int n, unsigned tag, float* dst, int code, int sleep = 3,
                                       unsigned long long* gd = nullptr) {
    if (c.dead || c.wave * 128 >= n) return;         // a wave is all in or all out
    const int t0 = 2 * (c.wave * 64 + c.lane);
    const int voff = (base + t0) * 8;
    constexpr int JT = kPCW * 128;                   // items between loads j and j + 1 of a lane
    unsigned spins = 0;
    pu32x4 v[NJ][NP];
    constexpr bool kSentinel = NJ > 1 || NP > 2;      // one load per lane on <= 2 planes: every poll pass reads everything (one round trip less)
    unsigned npoll = 0, npass = 0;
    if (gd) gd[0] = wall_clock64();
    while (kSentinel) {
        v[0][NP - 1] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (NP - 1) * n * 8, 16);
        ++npoll;
        if (__all(v[0][NP - 1].y == tag && v[0][NP - 1].w == tag)) break;
        if (spin_fail(c, spins, code, 3)) return;
    }
    if (gd) { gd[1] = wall_clock64(); gd[3] = npoll; }
    while (true) {
        unsigned bad = 0;
        ++npass;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j == 0 || c.wave * 128 + j * JT < n) {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    if (!(kSentinel && j == 0 && p == NP - 1)) v[j][p] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, (j * JT + p * n) * 8, 16);
            }
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            if (j == 0 || c.wave * 128 + j * JT < n) {
#pragma unroll
                for (int p = 0; p < NP; ++p) bad |= (v[j][p].y ^ tag) | (v[j][p].w ^ tag);
            }
        }
        if (!__any(bad != 0u)) break;
        if (spin_fail(c, spins, code, sleep)) return;
    }
    if (gd) { gd[2] = wall_clock64(); gd[4] = npass; }
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        if (j == 0 || c.wave * 128 + j * JT < n) {
            float s0 = __uint_as_float(v[j][0].x), s1 = __uint_as_float(v[j][0].z);
#pragma unroll
            for (int p = 1; p < NP; ++p) { s0 += __uint_as_float(v[j][p].x); s1 += __uint_as_float(v[j][p].z); }
            *reinterpret_cast<float2*>(dst + t0 + j * JT) = make_float2(s0, s1);
        }
    }
}

__device__ __forceinline__ float bf16_round(float v) { return __uint_as_float((unsigned)f32_to_bf16(v) << 16); }
__device__ __forceinline__ float4 bf16x4_to_f4(pu32x2 h) {
    return make_float4(__uint_as_float(h.x << 16), __uint_as_float(h.x & 0xffff0000u), __uint_as_float(h.y << 16), __uint_as_float(h.y & 0xffff0000u));
}

template <int VN>
__device__ __forceinline__ void vec_from_lds(const float* src, int lane, float4 (&v)[VN]) {
#pragma unroll
    for (int i = 0; i < VN; ++i) v[i] = *reinterpret_cast<const float4*>(src + i * 256 + lane * 4);
}

// per-lane partial dot product of VN KiB of one weight row (in the ring from byte `off` of the segment whose first fill is
// s0; the caller has waited for the fills) with vec
// (offsets are written in fp32 bytes throughout; WB = 1: the ring holds bf16 rows at half of every offset, widened here)
template <int VN, int WB = 0>
__device__ __forceinline__ float row_partial(const char* ring, unsigned rmask, unsigned s0, unsigned off, int lane,
                                             const float4 (&vec)[VN]) {
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
#pragma unroll
    for (int j = 0; j < VN; ++j) {
        const unsigned o = (off + j * 1024u) >> WB;
        const unsigned slot = (s0 + (o >> 14)) & rmask;
        float4 w;
        if (WB) w = bf16x4_to_f4(*reinterpret_cast<const pu32x2*>(ring + slot * kPSlot + (o & 16383u) + lane * 8));
        else w = *reinterpret_cast<const float4*>(ring + slot * kPSlot + (o & 16383u) + lane * 16);
        sx = fmaf(w.x, vec[j].x, sx); sy = fmaf(w.y, vec[j].y, sy);
        sz = fmaf(w.z, vec[j].z, sz); sw = fmaf(w.w, vec[j].w, sw);
    }
    return (sx + sy) + (sz + sw);
}

template <int ND>
__device__ __forceinline__ void layer_norm_regs(float4 (&v)[ND], const float4 (&g)[ND], const float4 (&b)[ND]) {
    const float inv_d = 1.0f / (float)(256 * ND);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    // (two passes, the reference's form; the one-pass E[x^2] - mean^2 variant was measured and dropped: profiles/r06_removed_experiments.patch)
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        v[i].x -= mean; v[i].y -= mean; v[i].z -= mean; v[i].w -= mean;
        q += (v[i].x * v[i].x + v[i].y * v[i].y) + (v[i].z * v[i].z + v[i].w * v[i].w);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        v[i].x = v[i].x * rstd * g[i].x + b[i].x; v[i].y = v[i].y * rstd * g[i].y + b[i].y;
        v[i].z = v[i].z * rstd * g[i].z + b[i].z; v[i].w = v[i].w * rstd * g[i].w + b[i].w;
    }
}

template <int ND>
__device__ __forceinline__ void load_gb(const float* gw, const float* gb, int lane, float4 (&g)[ND], float4 (&b)[ND]) {
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        g[i] = *reinterpret_cast<const float4*>(gw + i * 256 + lane * 4);
        b[i] = *reinterpret_cast<const float4*>(gb + i * 256 + lane * 4);
    }
}

// sum over the LPK lanes that share a key (LPK = 16, 32 or 64 consecutive lanes)
__device__ __forceinline__ float group_sum(float v, int lpk) {
    if (lpk == 64) return wave_sum(v);
    v = row16_sum(v);
    if (lpk == 32) v += __shfl_xor(v, 16);
    return v;
}

// Phase C input: the <= kPMaxChunks key-chunk partials (o, m, l) of the head that owns flat output dims [e, e + 4): reads every
// chunk after polling the {m, l} granules of the LAST chunk (sentinel) and merges the partial softmax states.  NCT >= nchunks
// is a compile-time bound (registers are statically indexed); slots past nchunks re-read the last chunk with weight 0.
template <int NCT>
__device__ __forceinline__ float4 merge_chunks(PCtx& c, __amdgpu_buffer_rsrc_t grs, int iP, int PS, int D, int nchunks, int e, int h,
                                               unsigned tg, int code) {
    pu32x4 oa[NCT], ob[NCT], ml[NCT];
    unsigned spins = 0;
    const int ml_last = (iP + (nchunks - 1) * PS + D + 2 * h) * 8;
    while (true) {
        ml[0] = __builtin_amdgcn_raw_buffer_load_b128(grs, ml_last, 0, 16);
        if (__all(ml[0].y == tg && ml[0].w == tg)) break;
        if (spin_fail(c, spins, code, 3)) return make_float4(0.f, 0.f, 0.f, 0.f);
    }
    while (true) {
        unsigned bad = 0;
#pragma unroll
        for (int ch = 0; ch < NCT; ++ch) {
            const int cc = ch < nchunks ? ch : nchunks - 1;
            const int off = (iP + cc * PS + e) * 8;
            oa[ch] = __builtin_amdgcn_raw_buffer_load_b128(grs, off, 0, 16);
            ob[ch] = __builtin_amdgcn_raw_buffer_load_b128(grs, off + 16, 0, 16);
            ml[ch] = __builtin_amdgcn_raw_buffer_load_b128(grs, (iP + cc * PS + D + 2 * h) * 8, 0, 16);
        }
#pragma unroll
        for (int ch = 0; ch < NCT; ++ch)
            bad |= (oa[ch].y ^ tg) | (oa[ch].w ^ tg) | (ob[ch].y ^ tg) | (ob[ch].w ^ tg) | (ml[ch].y ^ tg) | (ml[ch].w ^ tg);
        if (!__any(bad != 0u)) break;
        if (spin_fail(c, spins, code, 3)) return make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float M = -INFINITY;
#pragma unroll
    for (int ch = 0; ch < NCT; ++ch) M = fmaxf(M, __uint_as_float(ml[ch].x));
    float Lt = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ch = 0; ch < NCT; ++ch) {
        const float wgt = ch < nchunks ? __expf(__uint_as_float(ml[ch].x) - M) : 0.f;
        Lt += wgt * __uint_as_float(ml[ch].z);
        o.x = fmaf(wgt, __uint_as_float(oa[ch].x), o.x); o.y = fmaf(wgt, __uint_as_float(oa[ch].z), o.y);
        o.z = fmaf(wgt, __uint_as_float(ob[ch].x), o.z); o.w = fmaf(wgt, __uint_as_float(ob[ch].z), o.w);
    }
    const float inv = 1.0f / Lt;
    return make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
}

// ---- loader wave: the workgroup's weight rows, phase after phase, into the ring ---------------------------------
template <int ND, int WB, int XL>
__device__ __forceinline__ void persist_loader(const PersistArgs& A, PCtx& c, char* ring, int xx, int jj) {
    constexpr unsigned D = 256 * ND;
    const unsigned rmask = A.ring_slots - 1;
    const int rm = A.vocab / kPG, rem = A.vocab - rm * kPG;
    const bool fused = persist_fused_attn(A.n_head, A.head_dim, A.st.seq_len[A.slots[0]] + 1);
    unsigned fseq = 0;
    const int n_seg = 4 * A.n_layer + 2;
#pragma unroll 1
    for (int sgi = 0; sgi < n_seg; ++sgi) {
        // segment = this workgroup's rows of one matrix: contiguous in the row-per-output layout
        // WB = 1: the matrices are bf16 (2 bytes per element, PersistLayer pointers then address the bf16 copies)
        const char* base;
        unsigned bytes;
        unsigned kib_stride = 1024;                  // global bytes between consecutive KiB of the segment (contiguous rows)
        unsigned lane_off = c.lane * 16;             // this lane's 16 bytes inside a KiB of the segment
        bool rowpair = false;                        // a row piece is TWO consecutive KiB (kib_stride then counts rows)
        auto wptr = [&](const float* w, size_t elems) { return reinterpret_cast<const char*>(w) + ((elems * 4) >> WB); };
        if (sgi < 4 * A.n_layer) {
            const PersistLayer& Ly = A.layers[sgi >> 2];
            const int ph = sgi & 3;
            if (ph == 0) { base = wptr(Ly.qkv_w, (size_t)c.wg * (3 * ND) * D); bytes = (3 * ND * D * 4) >> WB; }
            else if (ph == 1 && fused) {             // rows [i RF, (i+1) RF) x the 256-element K-slice of head h
                const int h = c.wg % A.n_head, i = c.wg / A.n_head, RF = ND * A.n_head;
                base = wptr(Ly.proj_w, (size_t)i * RF * D + h * 256); bytes = (RF * 1024) >> WB; kib_stride = D * 4;
                // fp32: one KiB per row; bf16: a KiB of the segment is TWO rows' 512-byte slices (lanes 32.. take the second)
                if (WB) lane_off = (c.lane >> 5) * (D * 2) + (c.lane & 31) * 16;
            }
            else if (ph == 1) { base = wptr(Ly.proj_w, (size_t)c.wg * ND * D); bytes = (ND * D * 4) >> WB; }
            else if (ph == 2) {
                const int dwg = XL ? xx * 32 + jj : c.wg;          // XL: the XCD's workgroups own 512 consecutive hidden units
                base = wptr(Ly.fc_w, (size_t)dwg * (4 * ND) * D); bytes = (4 * ND * D * 4) >> WB;
            }
            else if (XL) {       // rows [32 j, +32) x the XCD's K-slice [512 x, +512) of mlp c_proj: 2 KiB (bf16: 1 KiB) per row
                base = wptr(Ly.p2_w, (size_t)(32 * jj) * (4 * D) + 512 * xx); bytes = (32 * 2048) >> WB;
                kib_stride = (4 * D * 4) >> WB; rowpair = !WB;
            }
            else { base = wptr(Ly.p2_w, (size_t)c.wg * ND * (4 * D)); bytes = (ND * 4 * D * 4) >> WB; }
        } else if (sgi == 4 * A.n_layer) {
            base = wptr(A.head_w, (size_t)c.wg * rm * D); bytes = (rm * D * 4) >> WB;
        } else {
            base = wptr(A.head_w, (size_t)(rm * kPG + c.wg) * D); bytes = c.wg < rem ? (D * 4) >> WB : 0;
        }
#pragma unroll 1
        for (unsigned off = 0; off < bytes; off += kPSlot) {
            // slot free?  (every consumer wave is past fill fseq - ring_slots)
            unsigned spins = 0;
            bool drained = false;
            while (!c.dead) {
                unsigned m = lds_ld(c.ctl + kCtlDone);
#pragma unroll
                for (int w = 1; w < kPCW; ++w) m = min(m, lds_ld(c.ctl + kCtlDone + w));
                if (fseq < m + (unsigned)A.ring_slots) break;
                if (!drained) {      // blocked: whatever was issued is landed and announced before waiting
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_st(c.ctl + kCtlFilled, fseq);
                    drained = true;
                }
                if (spin_fail(c, spins, 902, 1)) break;
            }
            if (c.dead) return;
            const unsigned n = (min((unsigned)kPSlot, bytes - off)) >> 10;
            const unsigned slot = __builtin_amdgcn_readfirstlane(fseq & rmask);
            char* dst = ring + slot * kPSlot;
            const char* src = base + (size_t)(rowpair ? off >> 11 : off >> 10) * kib_stride + lane_off;
            if (n == 16 && rowpair) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)(i >> 1) * kib_stride + (i & 1) * 1024),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
                // ONE fill in flight: enough to keep up (11 KB/us per CU) and it leaves the memory queue to the consumers' polls (two / three in
                // flight: +0.5 % / +4 %, profiles/r04_microbench_notes.md)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_st(c.ctl + kCtlFilled, fseq + 1);
            } else if (n == 16) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (size_t)i * kib_stride),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
                // ONE fill in flight: enough to keep up (11 KB/us per CU) and it leaves the memory queue to the consumers' polls (two / three in
                // flight: +0.5 % / +4 %, profiles/r04_microbench_notes.md)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_st(c.ctl + kCtlFilled, fseq + 1);
            } else {
#pragma unroll 1
                for (unsigned i = 0; i < n; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(rowpair ? src + (size_t)(i >> 1) * kib_stride + (i & 1) * 1024
                                                                                                             : src + (size_t)i * kib_stride),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_st(c.ctl + kCtlFilled, fseq + 1);
            }
            ++fseq;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_st(c.ctl + kCtlFilled, fseq);
}

// ---- decode-step kernel --------------------------------------------------------------------------------------
template <int ND, int WB = 0, int KVB = 0, int XL = 0>      // d_model = 256 * ND; bf16 weight storage; bf16 KV cache; XCD-local MLP hand-off
__global__ __launch_bounds__(kPThreads) void k_decode_persist(const PersistArgs A) {
    static_assert(!XL || ND == 4, "the XCD-local MLP hand-off is laid out for d_model 1024 (8 XCDs x 32 workgroups)");
    constexpr int D = 256 * ND;
    constexpr int NPX = XL ? 8 : 2;              // planes of X1 (x as it leaves phase E): one per XCD, or the two K-halves
    constexpr int KSC = ND % 2 == 0 ? 2 : 1;     // K-split of an attn c_proj row over waves (partial sums = planes of X0)
    constexpr int KSE = 2;                       // K-split of an mlp c_proj row (planes of X1)
    constexpr int NJX = (D + kPCW * 128 - 1) / (kPCW * 128);        // 16-byte loads per lane and plane in a sweep over a d-vector
    extern __shared__ __attribute__((aligned(16))) float smem[];      // (same declaration as k_gemv's)
    char* ring = reinterpret_cast<char*>(smem);
    float* hvec = reinterpret_cast<float*>(ring + (size_t)A.ring_slots * kPSlot);      // [4D] mlp hidden units
    float* ovec = hvec;                          // [D] merged attention output: hvec is idle between E of a layer and E of the next
    float* o_s = hvec + D;                       // attention lane-group states [NG][hd] (phase B; same idle window)
    float* xvec = hvec + A.hvec_floats;          // [D] phase input (x / x')
    float* ascr = xvec + D;                      // attention: q_h | k_h | v_h of this step, then m_s, l_s
    unsigned* ctl = reinterpret_cast<unsigned*>(ascr + A.ascr_floats);
    if (threadIdx.x < kCtlWords) ctl[threadIdx.x] = 0u;
    // XL: this workgroup's XCD and its rank among the XCD's workgroups (requested here, used after the barrier)
    unsigned xcc = 0, xrank = 0;
    if (XL && threadIdx.x == kPCW * 64) {
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 7u;
        xrank = __hip_atomic_fetch_add(A.epoch + 4 + xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // more than 32 workgroups of this grid on one XCD (CU masking, another partition mode, an uneven deal): the XCD-local layout does
        // not hold.  Reported like a hand-off time-out (the ranks that are missing elsewhere time out anyway): the host drops the call
        if (xrank >= 32u) __hip_atomic_store(A.err, 961, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();

    PCtx c;
    c.lane = threadIdx.x & 63; c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); c.wg = blockIdx.x;
    c.ctl = ctl; c.err = A.err; c.bar_target = 0; c.filled_seen = 0; c.dead = false;
    const unsigned epoch = __hip_atomic_load(A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    if (c.wave == kPCW) {
        int xx = 0, jj = 0;
        if (XL) {
            xx = __builtin_amdgcn_readfirstlane((int)xcc);
            jj = __builtin_amdgcn_readfirstlane((int)(xrank & 31u));
            if (c.lane == 0) lds_st(ctl + kCtlXcd, 0x10000u | (unsigned)(xx << 8) | (unsigned)jj);
        }
        persist_loader<ND, WB, XL>(A, c, ring, xx, jj);
    } else {
        // `lane` is re-defined through an empty asm at every phase: without it the compiler hoists the per-lane addresses of all
        // phases out of the layer loop and spills them to scratch (vector memory behind the loader's DMA queue)
        int& lane = c.lane;
        const int wave = c.wave, wg = c.wg;
#define GVC_PHASE_BEGIN() asm volatile("" : "+v"(c.lane))
        const unsigned rmask = A.ring_slots - 1;
        const int H = A.n_head, hd = A.head_dim;
        const int slot = A.slots[0];
        const int S = A.st.seq_len[slot];                // cached positions; this step's key goes to index S
        const int n_keys = S + 1;
        const int lpk = hd >> 2, kpw = 64 / lpk;         // lanes per key row, keys per wave-instruction
        const int pass_keys = kPU * kPCW * kpw;
        const int kc = max(pass_keys, (n_keys + kPMaxChunks - 1) / kPMaxChunks);
        const int nchunks = (n_keys + kc - 1) / kc;
        const bool fused = persist_fused_attn(H, hd, n_keys);
        const bool is_attn = !fused && wg < H * nchunks;
        const int ah = wg / nchunks, ac = wg - ah * nchunks;
        const float scale = 1.0f / sqrtf((float)hd);
        const int PS = D + 2 * H;                        // granules per key chunk: o[D] | {m, l}[H]
        // granule indices of the five hand-off buffers inside the allocation
        const int iQ = 0, iP = 3 * D, iX0 = iP + kPMaxChunks * PS, iH = iX0 + 4 * D, iX1 = iH + 4 * D;
        const __amdgpu_buffer_rsrc_t grs = make_rsrc(A.gran, (unsigned)(iX1 + 8 * D) * 8u);
        int xx = 0, jj = 0;              // XL: XCD and rank inside it (the loader wave publishes them through LDS)
        auto need_xcd = [&]() {
            if (!XL || xx >= 0x100) return;
            unsigned v = 0, spins = 0;
            while (!c.dead && !((v = lds_ld(ctl + kCtlXcd)) & 0x10000u))
                if (spin_fail(c, spins, 960, 1)) break;
            xx = (int)((v >> 8) & 7u) | 0x100;       // (bit 8: known)
            jj = (int)(v & 31u);
        };
        const unsigned tbase = ((epoch + 1u) & 0xfffffu) << 12;
        auto tag_of = [&](int l, int p) { return tbase | (unsigned)(l * 8 + p + 1); };
        // stamps (GVC_PERSIST_STAMPS): workgroup 0, every layer: [(l * 5 + p) * 4 + k], k = 0 input ready, 1 output published,
        // 2 / 3 extra; every workgroup, layer 2: [base2 + (wg * 5 + p) * 2 + k]
        const bool stamp0 = A.dbg && wave == 0 && lane == 0;
        const int base2 = 4 * 5 * (A.n_layer + 2);
        auto stamp_at = [&](int l, int p, int k) {
            if (stamp0 && wg == 0) A.dbg[(l * 5 + p) * 4 + k] = wall_clock64();
            if (stamp0 && l == kPStampLayer && k < 2) A.dbg[base2 + (wg * 5 + p) * 2 + k] = wall_clock64();
        };
        unsigned fs = 0;                                 // first fill of the current weight segment
        constexpr unsigned nfA = (((3 * ND * D * 4) >> WB) + kPSlot - 1) / kPSlot, nfC = (((ND * D * 4) >> WB) + kPSlot - 1) / kPSlot,
                           nfD = (((4 * ND * D * 4) >> WB) + kPSlot - 1) / kPSlot;
        constexpr int FSH = 14 + WB;                 // fill index of a byte offset written in fp32 bytes
        auto phase_done = [&]() { if (lane == 0) lds_st(ctl + kCtlDone + wave, fs); };

        // ---- x = mel_embedding[tok] + mel_pos_embedding[pos]  (gpt_inference.py:92-96), by every workgroup ----
        if (wave < ND) {
            const float* e = A.mel_emb + (size_t)min(max(A.tok_in[0], 0), A.vocab - 1) * D + wave * 256 + lane * 4;
            const float* p = A.mel_pos + (size_t)A.st.mel_pos[slot] * D + wave * 256 + lane * 4;
            const float4 ev = *reinterpret_cast<const float4*>(e), pv = *reinterpret_cast<const float4*>(p);
            *reinterpret_cast<float4*>(xvec + wave * 256 + lane * 4) = make_float4(ev.x + pv.x, ev.y + pv.y, ev.z + pv.z, ev.w + pv.w);
        }
        if (stamp0 && wg == 0) A.dbg[4 * 5 * (A.n_layer + 1)] = wall_clock64();      // kernel entry

        for (int l = 0; l < A.n_layer; ++l) {
            const PersistLayer& Ly = A.layers[l];
            // =================== A: LN1 -> c_attn rows -> q | k | v ===================
            // row (wave + 8 i) of the workgroup's 3 ND rows belongs to wave `wave`
            {
                GVC_PHASE_BEGIN();
                float4 g[ND], b[ND], xv[ND];
                load_gb<ND>(Ly.ln1_w, Ly.ln1_b, lane, g, b);
                constexpr int RA = 3 * ND, UPW = (RA + kPCW - 1) / kPCW;
                const int nmy = wave < RA ? (RA - wave + kPCW - 1) / kPCW : 0;
                const int row_g = wg * RA + wave + kPCW * lane;
                const float bias = lane < nmy ? Ly.qkv_b[row_g] : 0.f;
                if (l > 0) gather<NJX, NPX>(c, grs, iX1, D, tag_of(l - 1, 4), xvec, 100 + l, 3, nullptr);
                cbar(c);
                stamp_at(l, 0, 0);
                float val = 0.f;
                if (nmy > 0) {
                    vec_from_lds<ND>(xvec, lane, xv);
                    layer_norm_regs<ND>(xv, g, b);
                    stamp_at(l, 0, 2);
                    wait_fill(c, fs + ((unsigned)((wave + kPCW * (nmy - 1)) * D * 4 + D * 4 - 1) >> FSH));
                    float part[UPW];
#pragma unroll
                    for (int i = 0; i < UPW; ++i)
                        part[i] = i < nmy ? row_partial<ND, WB>(ring, rmask, fs, (unsigned)(wave + kPCW * i) * D * 4, lane, xv) : 0.f;
#pragma unroll
                    for (int i = 0; i < UPW; ++i) {
                        const float s = wave_sum(part[i]);
                        if (lane == i) val = s;
                    }
                }
                stamp_at(l, 0, 3);
                if (lane < nmy) {
                    val += bias;
                    if (KVB && row_g >= D) val = bf16_round(val);      // a bf16 cache: k / v rounded where they enter it, this step's attention reads the same values
                    publish(grs, iQ + row_g, tag_of(l, 0), val);
                    if (row_g >= D) {            // append k / v of this position to the cache (read by later steps)
                        const int which = row_g / D, ci = row_g - which * D;
                        const int h = ci / hd, j = ci - h * hd;
                        float* cache = which == 1 ? Ly.kcache : Ly.vcache;
                        const size_t e = (((size_t)slot * H + h) * A.max_seq + S) * hd + j;
                        if (KVB) reinterpret_cast<unsigned short*>(cache)[e] = (unsigned short)(__float_as_uint(val) >> 16);
                        else cache[e] = val;
                    }
                }
                fs += nfA;
                phase_done();
                stamp_at(l, 0, 1);
            }
            // =================== B: attention over one key chunk of one head (a few workgroups) ===================
            if (is_attn) {
                GVC_PHASE_BEGIN();
                const int kl = lane / lpk, dl = (lane - kl * lpk) * 4;
                const int k0 = ac * kc, k1 = min(S, k0 + kc);            // cached keys of this chunk
                // this head's rows of the layer's K / V cache as buffers: one per-lane offset serves every key of a pass
                constexpr int ESZ = KVB ? 2 : 4;                             // bytes per cache element
                const unsigned head_bytes = (unsigned)A.max_seq * hd * ESZ;
                const size_t head_off = ((size_t)slot * H + ah) * A.max_seq * hd * ESZ;
                const __amdgpu_buffer_rsrc_t krs = make_rsrc(reinterpret_cast<const char*>(Ly.kcache) + head_off, head_bytes);
                const __amdgpu_buffer_rsrc_t vrs = make_rsrc(reinterpret_cast<const char*>(Ly.vcache) + head_off, head_bytes);
                const int kv_step = kPCW * kpw * hd * ESZ;                   // bytes between the keys u and u + 1 of a lane
                float4 kr[kPU], vr[kPU];
                auto load_pass = [&](int kbase) {
                    const int key0 = kbase + wave * kpw + kl;
                    const int voff = (key0 * hd + dl) * ESZ;
#pragma unroll
                    for (int u = 0; u < kPU; ++u) {
                        const bool ok = key0 + u * kPCW * kpw < k1;
                        if (KVB) {
                            pu32x2 kk = {0u, 0u}, vv = {0u, 0u};
                            if (ok) {
                                kk = __builtin_amdgcn_raw_buffer_load_b64(krs, voff, u * kv_step, 0);
                                vv = __builtin_amdgcn_raw_buffer_load_b64(vrs, voff, u * kv_step, 0);
                            }
                            kr[u] = bf16x4_to_f4(kk); vr[u] = bf16x4_to_f4(vv);
                        } else {
                            pu32x4 kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
                            if (ok) {
                                kk = __builtin_amdgcn_raw_buffer_load_b128(krs, voff, u * kv_step, 0);
                                vv = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, u * kv_step, 0);
                            }
                            kr[u] = make_float4(__uint_as_float(kk.x), __uint_as_float(kk.y), __uint_as_float(kk.z), __uint_as_float(kk.w));
                            vr[u] = make_float4(__uint_as_float(vv.x), __uint_as_float(vv.y), __uint_as_float(vv.z), __uint_as_float(vv.w));
                        }
                    }
                };
                load_pass(k0);          // requested ahead of the seam: these rows were written by earlier launches
                const bool last_chunk = ac == nchunks - 1;
                const int nseg = last_chunk ? 3 : 1;     // q_h, and the k_h / v_h rows of this very step
                {   // nseg * hd granules as pairs: ONE 16-byte load per lane (a single round trip), item t = 2 (wave * 64 + lane)
                    const int t = 2 * (wave * 64 + lane);
                    if (t - 2 * lane < nseg * hd && !c.dead) {           // (wave-uniform)
                        const bool in = t < nseg * hd;
                        const int sg = in ? t / hd : 0, tt = in ? t : 0;
                        const int off = (iQ + sg * D + ah * hd + (tt - sg * hd)) * 8;
                        const unsigned tg = tag_of(l, 0);
                        unsigned spins = 0;
                        pu32x4 v;
                        while (true) {
                            v = __builtin_amdgcn_raw_buffer_load_b128(grs, off, 0, 16);
                            if (__all(!in || (v.y == tg && v.w == tg))) break;
                            if (spin_fail(c, spins, 200 + l, 3)) break;
                        }
                        if (in) *reinterpret_cast<float2*>(ascr + t) = make_float2(__uint_as_float(v.x), __uint_as_float(v.z));
                    }
                }
                cbar(c);
                stamp_at(l, 1, 0);
                const float4 q4 = *reinterpret_cast<const float4*>(ascr + dl);
                float m = -INFINITY, lsum = 0.f;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                auto fold = [&](const float (&s)[kPU]) {
                    float mn = m;
#pragma unroll
                    for (int u = 0; u < kPU; ++u) mn = fmaxf(mn, s[u]);
                    if (mn > -INFINITY) {
                        const float alpha = __expf(m - mn);
                        lsum *= alpha; o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
#pragma unroll
                        for (int u = 0; u < kPU; ++u) {
                            const float p = __expf(s[u] - mn);
                            lsum += p;
                            o.x = fmaf(p, vr[u].x, o.x); o.y = fmaf(p, vr[u].y, o.y);
                            o.z = fmaf(p, vr[u].z, o.z); o.w = fmaf(p, vr[u].w, o.w);
                        }
                        m = mn;
                    }
                };
                for (int kbase = k0; kbase < k1; kbase += pass_keys) {
                    if (kbase > k0) load_pass(kbase);
                    float s[kPU];
#pragma unroll
                    for (int u = 0; u < kPU; ++u) s[u] = dot4(q4, kr[u]);
#pragma unroll
                    for (int u = 0; u < kPU; ++u) {
                        const int key = kbase + (u * kPCW + wave) * kpw + kl;
                        const float dsum = group_sum(s[u], lpk);
                        s[u] = key < k1 ? dsum * scale : -INFINITY;
                    }
                    fold(s);
                }
                if (last_chunk && wave == kPCW - 1) {    // the key of this step: lane group 0 of the last consumer wave
                    float s[kPU];
#pragma unroll
                    for (int u = 0; u < kPU; ++u) {
                        s[u] = -INFINITY;
                        vr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    const float4 kn = *reinterpret_cast<const float4*>(ascr + hd + dl);
                    vr[0] = *reinterpret_cast<const float4*>(ascr + 2 * hd + dl);
                    const float dsum = group_sum(dot4(q4, kn), lpk);
                    if (kl == 0) s[0] = dsum * scale;
                    fold(s);
                }
                stamp_at(l, 1, 2);
                // merge the kPCW * kpw lane-group states of the workgroup
                const int NG = kPCW * kpw;
                float* m_s = ascr + 3 * hd;              // [NG]
                float* l_s = m_s + NG;                   // [NG]
                const int gidx = wave * kpw + kl;
                if (dl == 0) { m_s[gidx] = m; l_s[gidx] = lsum; }
                *reinterpret_cast<float4*>(o_s + gidx * hd + dl) = o;
                cbar(c);
                stamp_at(l, 1, 3);
                const int tid = wave * 64 + lane;
                if (tid < hd) {
                    float M = -INFINITY;
                    for (int i = 0; i < NG; ++i) M = fmaxf(M, m_s[i]);
                    float Lp = 0.f, acc = 0.f;
                    for (int i = 0; i < NG; ++i) {
                        const float wgt = __expf(m_s[i] - M);        // groups without a key: exp(-inf) = 0
                        Lp += wgt * l_s[i];
                        acc += wgt * o_s[i * hd + tid];
                    }
                    const unsigned tg = tag_of(l, 1);
                    publish(grs, iP + ac * PS + ah * hd + tid, tg, acc);
                    if (tid == 0) {
                        publish(grs, iP + ac * PS + D + 2 * ah, tg, M);
                        publish(grs, iP + ac * PS + D + 2 * ah + 1, tg, Lp);
                    }
                }
                stamp_at(l, 1, 1);
            }
            // =================== BC (short contexts): attention of head h + its K-slice of attn c_proj, every workgroup ===================
            if (fused) {
                GVC_PHASE_BEGIN();
                const int fh = wg % H, fi = wg / H, RF = ND * H;        // head, row block, rows of the block
                constexpr int ESZ = KVB ? 2 : 4;
                const unsigned head_bytes = (unsigned)A.max_seq * 256u * ESZ;
                const size_t head_off = ((size_t)slot * H + fh) * A.max_seq * 256 * ESZ;
                const __amdgpu_buffer_rsrc_t krs = make_rsrc(reinterpret_cast<const char*>(Ly.kcache) + head_off, head_bytes);
                const __amdgpu_buffer_rsrc_t vrs = make_rsrc(reinterpret_cast<const char*>(Ly.vcache) + head_off, head_bytes);
                float4 kr[kPUF], vr[kPUF];
                {       // cached rows of the whole head (written by earlier launches), requested ahead of the seam: key u * 8 + wave
                    const int voff = (wave * 256 + lane * 4) * ESZ;
#pragma unroll
                    for (int u = 0; u < kPUF; ++u) {
                        if (KVB) {
                            pu32x2 kk = {0u, 0u}, vv = {0u, 0u};
                            if (u * kPCW + wave < S) {
                                kk = __builtin_amdgcn_raw_buffer_load_b64(krs, voff, u * kPCW * 256 * ESZ, 0);
                                vv = __builtin_amdgcn_raw_buffer_load_b64(vrs, voff, u * kPCW * 256 * ESZ, 0);
                            }
                            kr[u] = bf16x4_to_f4(kk); vr[u] = bf16x4_to_f4(vv);
                        } else {
                            pu32x4 kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
                            if (u * kPCW + wave < S) {
                                kk = __builtin_amdgcn_raw_buffer_load_b128(krs, voff, u * kPCW * 256 * ESZ, 0);
                                vv = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, u * kPCW * 256 * ESZ, 0);
                            }
                            kr[u] = make_float4(__uint_as_float(kk.x), __uint_as_float(kk.y), __uint_as_float(kk.z), __uint_as_float(kk.w));
                            vr[u] = make_float4(__uint_as_float(vv.x), __uint_as_float(vv.y), __uint_as_float(vv.z), __uint_as_float(vv.w));
                        }
                    }
                }
                const int r0 = wave, r1 = wave + kPCW;                   // this wave's rows of the block (RF <= 16)
                const float bias0 = fh == 0 && r0 < RF ? Ly.proj_b[fi * RF + r0] : 0.f;
                const float bias1 = fh == 0 && r1 < RF ? Ly.proj_b[fi * RF + r1] : 0.f;
                // q_h | k_h | v_h of this step: 768 granules = 384 pairs, ONE 16-byte load per lane of waves 0..5 (a single round trip)
                if (wave < 6 && !c.dead) {
                    const int t = 2 * (wave * 64 + lane), sg = t >> 8;
                    const int off = (iQ + sg * D + fh * 256 + (t & 255)) * 8;
                    const unsigned tg = tag_of(l, 0);
                    unsigned spins = 0;
                    pu32x4 v;
                    while (true) {
                        v = __builtin_amdgcn_raw_buffer_load_b128(grs, off, 0, 16);
                        if (__all(v.y == tg && v.w == tg)) break;
                        if (spin_fail(c, spins, 200 + l, 3)) break;
                    }
                    *reinterpret_cast<float2*>(ascr + t) = make_float2(__uint_as_float(v.x), __uint_as_float(v.z));
                }
                cbar(c);
                stamp_at(l, 1, 0);
                const float4 q4 = *reinterpret_cast<const float4*>(ascr + lane * 4);
                float sc[kPUF];
#pragma unroll
                for (int u = 0; u < kPUF; ++u) {
                    if (u * kPCW + wave == S) {                          // the key of this step comes from the gathered granules
                        kr[u] = *reinterpret_cast<const float4*>(ascr + 256 + lane * 4);
                        vr[u] = *reinterpret_cast<const float4*>(ascr + 512 + lane * 4);
                    }
                    sc[u] = dot4(q4, kr[u]);
                }
                float m = -INFINITY;
#pragma unroll
                for (int u = 0; u < kPUF; ++u) {
                    if (u * kPCW + wave <= S) {                          // (wave-uniform: slots past the context cost nothing)
                        sc[u] = wave_sum(sc[u]) * scale;
                        m = fmaxf(m, sc[u]);
                    } else sc[u] = -INFINITY;
                }
                float lsum = 0.f;
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int u = 0; u < kPUF; ++u) {
                    const float p = __expf(sc[u] - m);                   // (every wave owns at least key `wave` <= S... see below)
                    lsum += p;
                    o.x = fmaf(p, vr[u].x, o.x); o.y = fmaf(p, vr[u].y, o.y);
                    o.z = fmaf(p, vr[u].z, o.z); o.w = fmaf(p, vr[u].w, o.w);
                }
                // a wave without any key (S + 1 < 8 never happens after a prefill, but stay safe): m = -inf -> p = NaN; zero it
                if (!(m > -INFINITY)) { lsum = 0.f; o = make_float4(0.f, 0.f, 0.f, 0.f); }
                float* m_s = ascr + 3 * 256;
                float* l_s = m_s + kPCW;
                if (lane == 0) { m_s[wave] = m; l_s[wave] = lsum; }
                *reinterpret_cast<float4*>(o_s + wave * 256 + lane * 4) = o;
                cbar(c);
                stamp_at(l, 1, 1);
                // every wave merges the eight states into the normalised head output it multiplies with
                float mm[kPCW], ll[kPCW];
                float4 oo[kPCW];
#pragma unroll
                for (int g = 0; g < kPCW; ++g) { mm[g] = m_s[g]; ll[g] = l_s[g]; oo[g] = *reinterpret_cast<const float4*>(o_s + g * 256 + lane * 4); }
                float M = mm[0];
#pragma unroll
                for (int g = 1; g < kPCW; ++g) M = fmaxf(M, mm[g]);
                float Lt = 0.f;
                float4 oh[1] = {make_float4(0.f, 0.f, 0.f, 0.f)};
#pragma unroll
                for (int g = 0; g < kPCW; ++g) {
                    const float wgt = __expf(mm[g] - M);
                    Lt += wgt * ll[g];
                    oh[0].x = fmaf(wgt, oo[g].x, oh[0].x); oh[0].y = fmaf(wgt, oo[g].y, oh[0].y);
                    oh[0].z = fmaf(wgt, oo[g].z, oh[0].z); oh[0].w = fmaf(wgt, oo[g].w, oh[0].w);
                }
                const float inv = 1.0f / Lt;
                oh[0].x *= inv; oh[0].y *= inv; oh[0].z *= inv; oh[0].w *= inv;
                stamp_at(l, 2, 0);
                if (r0 < RF) {
                    wait_fill(c, fs + (((unsigned)(r1 < RF ? r1 : r0) * 1024u + 1023u) >> FSH));
                    const float p0 = row_partial<1, WB>(ring, rmask, fs, (unsigned)r0 * 1024u, lane, oh);
                    const float p1 = r1 < RF ? row_partial<1, WB>(ring, rmask, fs, (unsigned)r1 * 1024u, lane, oh) : 0.f;
                    float s0 = wave_sum(p0), s1 = wave_sum(p1);
                    if (fh == 0) {                                       // plane 0 carries the residual (x of this layer is still in xvec) and the bias
                        s0 = xvec[fi * RF + r0] + (s0 + bias0);
                        if (r1 < RF) s1 = xvec[fi * RF + r1] + (s1 + bias1);
                    }
                    if (lane == 0) publish(grs, iX0 + fh * D + fi * RF + r0, tag_of(l, 2), s0);
                    if (lane == 1 && r1 < RF) publish(grs, iX0 + fh * D + fi * RF + r1, tag_of(l, 2), s1);
                }
                fs += nfC;
                phase_done();
                stamp_at(l, 2, 1);
            } else
            // =================== C: merge chunk partials -> attn c_proj (row, K-half) units -> x' = x + ... ===================
            {
                GVC_PHASE_BEGIN();
                constexpr int VN = ND / KSC;             // KiB of a row per unit
                const int crow = wave / KSC, cks = wave - crow * KSC;
                const bool unit = wave < ND * KSC;
                const float bias = unit && cks == 0 ? Ly.proj_b[wg * ND + crow] : 0.f;
                if (wave < ND && !c.dead) {
                    const int e = wave * 256 + lane * 4, h = e / hd;
                    const unsigned tg = tag_of(l, 1);
                    float4 o;
                    if (nchunks <= 2) o = merge_chunks<2>(c, grs, iP, PS, D, nchunks, e, h, tg, 300 + l);
                    else if (nchunks <= 4) o = merge_chunks<4>(c, grs, iP, PS, D, nchunks, e, h, tg, 300 + l);
                    else o = merge_chunks<kPMaxChunks>(c, grs, iP, PS, D, nchunks, e, h, tg, 300 + l);
                    *reinterpret_cast<float4*>(ovec + e) = o;
                }
                cbar(c);
                stamp_at(l, 2, 0);
                if (unit) {
                    float4 ov[VN];
                    vec_from_lds<VN>(ovec + cks * VN * 256, lane, ov);
                    const unsigned off = (unsigned)crow * D * 4 + (unsigned)cks * VN * 1024u;
                    wait_fill(c, fs + ((off + VN * 1024u - 1u) >> FSH));
                    float s = wave_sum(row_partial<VN, WB>(ring, rmask, fs, off, lane, ov));
                    if (cks == 0) s = xvec[wg * ND + crow] + (s + bias);       // residual: x of this layer (A's input) is still in xvec
                    if (lane == 0) publish(grs, iX0 + cks * D + wg * ND + crow, tag_of(l, 2), s);
                }
                fs += nfC;
                phase_done();
                stamp_at(l, 2, 1);
            }
            // =================== D: LN2 -> c_fc rows -> gelu_new ===================
            {
                GVC_PHASE_BEGIN();
                float4 g[ND], b[ND], xv[ND];
                load_gb<ND>(Ly.ln2_w, Ly.ln2_b, lane, g, b);
                constexpr int RD = 4 * ND, UPW = (RD + kPCW - 1) / kPCW;
                const int nmy = wave < RD ? (RD - wave + kPCW - 1) / kPCW : 0;
                need_xcd();
                const int dwg = XL ? (xx & 7) * 32 + jj : wg;                // XL: hidden units [512 x + 16 j, +16) of XCD x
                const int row_g = dwg * RD + wave + kPCW * lane;
                const float bias = lane < nmy ? Ly.fc_b[row_g] : 0.f;
                // (diagnostics: wave 0's gather of layer 2 in every workgroup -> [base3 + wg * 8 + ..], and the workgroup's XCD / rank)
                unsigned long long* gdD = (A.dbg && l == kPStampLayer && wave == 0) ? A.dbg + base2 + 2 * 5 * kPG + wg * 8 : nullptr;
                if (gdD && lane == 0) gdD[5] = XL ? (unsigned long long)(((xx & 7) << 8) | jj) : 0xffffull;
                if (!fused) gather<NJX, KSC>(c, grs, iX0, D, tag_of(l, 2), xvec, 400 + l);
                else if (H == 4) gather<NJX, 4>(c, grs, iX0, D, tag_of(l, 2), xvec, 400 + l, 3, gdD);
                else if (H == 2) gather<NJX, 2>(c, grs, iX0, D, tag_of(l, 2), xvec, 400 + l);
                else gather<NJX, 1>(c, grs, iX0, D, tag_of(l, 2), xvec, 400 + l);
                cbar(c);
                stamp_at(l, 3, 0);
                float val = 0.f;
                if (nmy > 0) {
                    vec_from_lds<ND>(xvec, lane, xv);
                    layer_norm_regs<ND>(xv, g, b);
                    wait_fill(c, fs + ((unsigned)((wave + kPCW * (nmy - 1)) * D * 4 + D * 4 - 1) >> FSH));
                    float part[UPW];
#pragma unroll
                    for (int i = 0; i < UPW; ++i)
                        part[i] = i < nmy ? row_partial<ND, WB>(ring, rmask, fs, (unsigned)(wave + kPCW * i) * D * 4, lane, xv) : 0.f;
#pragma unroll
                    for (int i = 0; i < UPW; ++i) {
                        const float s = wave_sum(part[i]);
                        if (lane == i) val = s;
                    }
                }
                if (lane < nmy) {
                    if (XL) publish_local(grs, iH + row_g, tag_of(l, 3), gelu_new(val + bias));       // read by this XCD's workgroups only
                    else publish(grs, iH + row_g, tag_of(l, 3), gelu_new(val + bias));
                }
                fs += nfD;
                phase_done();
                stamp_at(l, 3, 1);
            }
            // =================== E (XL): the XCD's 512 hidden units x its K-slice of mlp c_proj, 32 output rows -> plane x of X1 ===================
            if constexpr (XL != 0) {
                GVC_PHASE_BEGIN();
                const int x8 = xx & 7;
                const int orow = 32 * jj + 4 * wave + lane;                  // (lanes 0..3: the wave's four rows)
                const float bias = x8 == 0 && lane < 4 ? Ly.p2_b[orow] : 0.f;
                gather<1, 1>(c, grs, iH + 512 * x8, 512, tag_of(l, 3), hvec, 500 + l, 3);       // one 16-byte load per lane of waves 0..3
                cbar(c);
                stamp_at(l, 4, 0);
                float4 hv[2];
                vec_from_lds<2>(hvec, lane, hv);
                wait_fill(c, fs + (((unsigned)(4 * wave + 3) * 2048u + 2047u) >> FSH));
                float part[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) part[i] = row_partial<2, WB>(ring, rmask, fs, (unsigned)(4 * wave + i) * 2048u, lane, hv);
                float val = 0.f;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float sm = wave_sum(part[i]);
                    if (lane == i) val = sm;
                }
                if (lane < 4) {
                    if (x8 == 0) val = xvec[orow] + (val + bias);            // plane 0 carries the residual (x' is still in xvec) and the bias
                    publish(grs, iX1 + x8 * D + orow, tag_of(l, 4), val);
                }
                fs += nfD;
                phase_done();
                stamp_at(l, 4, 1);
            } else
            // =================== E: mlp c_proj (row, K-half) units -> x = x' + ... ===================
            {
                GVC_PHASE_BEGIN();
                constexpr int VN = 4 * ND / KSE;         // KiB of a row per unit
                const int erow = wave / KSE, eks = wave - erow * KSE;
                const bool unit = wave < ND * KSE;
                const float bias = unit && eks == 0 ? Ly.p2_b[wg * ND + erow] : 0.f;
                gather<(4 * D + kPCW * 128 - 1) / (kPCW * 128), 1>(c, grs, iH, 4 * D, tag_of(l, 3), hvec, 500 + l);
                cbar(c);
                stamp_at(l, 4, 0);
                if (unit) {
                    float4 hv[VN];
                    vec_from_lds<VN>(hvec + eks * VN * 256, lane, hv);
                    const unsigned off = (unsigned)erow * 4 * D * 4 + (unsigned)eks * VN * 1024u;
                    wait_fill(c, fs + ((off + VN * 1024u - 1u) >> FSH));
                    float s = wave_sum(row_partial<VN, WB>(ring, rmask, fs, off, lane, hv));
                    if (eks == 0) s = xvec[wg * ND + erow] + (s + bias);       // residual: x' (D's input) is still in xvec
                    if (lane == 0) publish(grs, iX1 + eks * D + wg * ND + erow, tag_of(l, 4), s);
                }
                fs += nfD;
                phase_done();
                stamp_at(l, 4, 1);
            }
        }
        // =================== head: ln_f -> final_norm -> latent -> mel_head rows ===================
        {
            GVC_PHASE_BEGIN();
            const int L = A.n_layer;
            float4 g[ND], b[ND], g2[ND], b2[ND], xv[ND];
            load_gb<ND>(A.lnf_w, A.lnf_b, lane, g, b);
            load_gb<ND>(A.fn_w, A.fn_b, lane, g2, b2);
            const int rm = A.vocab / kPG, rem = A.vocab - rm * kPG;
            const int nmy = wave < rm ? (rm - wave + kPCW - 1) / kPCW : 0;
            const int row_g = wg * rm + wave + kPCW * lane;
            const float bias = lane < nmy ? A.head_b[row_g] : 0.f;
            const bool tail = wave == kPCW - 1 && wg < rem;
            const float tbias = tail ? A.head_b[rm * kPG + wg] : 0.f;
            gather<NJX, NPX>(c, grs, iX1, D, tag_of(L - 1, 4), xvec, 600);
            cbar(c);
            stamp_at(L, 0, 0);
            vec_from_lds<ND>(xvec, lane, xv);
            layer_norm_regs<ND>(xv, g, b);
            layer_norm_regs<ND>(xv, g2, b2);
            if (wg == 0 && wave == 0) {
#pragma unroll
                for (int i = 0; i < ND; ++i) *reinterpret_cast<float4*>(A.latent_out + i * 256 + lane * 4) = xv[i];
            }
            float val = 0.f;
            for (int i = 0; i < nmy; ++i) {
                const unsigned off = (unsigned)(wave + kPCW * i) * D * 4;
                wait_fill(c, fs + ((off + D * 4 - 1) >> FSH));
                const float s = wave_sum(row_partial<ND, WB>(ring, rmask, fs, off, lane, xv));
                if (lane == i) val = s;
            }
            if (lane < nmy) A.logits_out[row_g] = val + bias;
            if (rm > 0) fs += (((rm * D * 4) >> WB) + kPSlot - 1) / kPSlot;
            if (tail) {
                wait_fill(c, fs);
                const float s = wave_sum(row_partial<ND, WB>(ring, rmask, fs, 0u, lane, xv));
                if (lane == 0) A.logits_out[rm * kPG + wg] = s + tbias;
            }
            if (wg < rem) fs += (((D * 4) >> WB) + kPSlot - 1) / kPSlot;
            phase_done();
            stamp_at(L, 0, 1);
            if (wg == 0 && wave == 0 && lane == 0) {
                if (A.advance) {
                    if (A.st.seq_len[slot] < A.max_seq - 1) A.st.seq_len[slot] += 1;
                    else *A.err = 950;           // KV cache full: GVC_ERR_STATE on the host's next call
                    if (A.st.mel_pos[slot] < A.max_mel_pos - 1) A.st.mel_pos[slot] += 1;
                    else *A.err = 951;
                }
                if (A.step_ctr) *A.step_ctr += 1;
            }
        }
#undef GVC_PHASE_BEGIN
    }
    // ---- end of step: the last workgroup to arrive opens the next epoch ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(A.epoch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(A.epoch + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the per-XCD rank counters start every launch at zero: a workgroup's rank is its arrival order inside THIS grid, whatever
            // earlier launches (other grid sizes, an uneven deal) left behind
            if (XL) {
#pragma unroll
                for (int x = 0; x < 8; ++x) __hip_atomic_store(A.epoch + 4 + x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            __hip_atomic_store(A.epoch, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace gvc
