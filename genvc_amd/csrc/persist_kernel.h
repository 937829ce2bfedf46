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
//   * four CONSUMER waves gather the phase input, do the LayerNorm / attention / dot products out of LDS and publish
//     the phase output.  Consumer waves never have a weight load outstanding (vmcnt retires in order per wave), so
//     they can poll;
//   * a seam (all-to-all hand-off of a phase output) is a sweep over 8-byte {value, tag} granules: every output
//     element is written by ONE write-through (sc1) store and polled with sc1 loads until its tag is the expected one.
//     The tag encodes (step epoch, layer, phase); nothing is zeroed between steps, the epoch lives in device memory and
//     is bumped by the last workgroup to finish the step.
// Per layer: A [LN1, c_attn] -> B [attention, split over <= 8 key chunks per head, a few workgroups] ->
//            C [merge of the chunk partials, attn c_proj, residual] -> D [LN2, c_fc, gelu_new] -> E [mlp c_proj, residual];
// then the double-LayerNorm head.  Every spin is bounded: on a timeout the workgroup sets *err, and the host raises
// GVC_ERR_STATE on the next call.
#pragma once
#include "gpt_kernels.h"

namespace gvc {

constexpr int kPG = 256;            // workgroups: one per CU, all co-resident
constexpr int kPCW = 4;             // consumer waves per workgroup (wave kPCW is the loader)
constexpr int kPThreads = (kPCW + 1) * 64;
constexpr int kPSlot = 16384;       // bytes per ring slot = one LDS-DMA fill (16 x 1 KiB wave-instructions)
constexpr int kPMaxChunks = 8;      // key chunks per head
constexpr int kPU = 8;              // keys per lane group held in registers per pass
constexpr unsigned kPSpinLimit = 400000;

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
    pu64* gran;                     // granule buffers: Q[3d] | P[kPMaxChunks][d + 2H] | X0[d] | HH[4d] | X1[d]
    unsigned* epoch;                // [0] step epoch, [1] arrival counter of the running step
    int* err;                       // device-visible host word: != 0 after a timeout
    int ring_slots;                 // power of two
    int ascr_floats;
    unsigned long long* dbg;        // nullable: wall-clock stamps of workgroup 0, [phase][2]
};

struct PCtx {
    int lane, wave, wg;
    unsigned* ctl;                  // LDS: [0] filled, [1..4] done per consumer wave, [5] arrive, [6] abort
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
    if (sleep == 1) __builtin_amdgcn_s_sleep(1);
    else __builtin_amdgcn_s_sleep(4);
    ++spins;
    if ((spins & 63u) == 0u && lds_ld(c.ctl + 6)) { c.dead = true; return true; }
    if (spins > kPSpinLimit) {
        c.dead = true;
        lds_st(c.ctl + 6, 1u);
        if (c.lane == 0) __hip_atomic_store(c.err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return true;
    }
    return false;
}

// barrier of the consumer waves through an LDS arrival counter (s_barrier would stop the loader wave too)
__device__ __forceinline__ void cbar(PCtx& c) {
    c.bar_target += kPCW;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (c.lane == 0) __hip_atomic_fetch_add(c.ctl + 5, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    unsigned spins = 0;
    while (!c.dead && lds_ld(c.ctl + 5) < c.bar_target)
        if (spin_fail(c, spins, 900, 1)) break;
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ void wait_fill(PCtx& c, unsigned seq) {
    if (c.dead || c.filled_seen > seq) return;
    unsigned spins = 0;
    while (true) {
        const unsigned f = lds_ld(c.ctl);
        if (f > seq) { c.filled_seen = f; break; }
        if (spin_fail(c, spins, 901, 1)) break;
    }
    asm volatile("" ::: "memory");
}

// one {value, tag} granule = ONE 8-byte write-through store (index = granule number inside the allocation)
__device__ __forceinline__ void publish(__amdgpu_buffer_rsrc_t rs, int index, unsigned tag, float v) {
    pu32x2 g;
    g.x = __float_as_uint(v); g.y = tag;
    __builtin_amdgcn_raw_buffer_store_b64(g, rs, index * 8, 0, 16);
}

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}

// The consumer waves sweep n consecutive granules (starting `base` bytes into the granule allocation) until every tag
// matches; values land in dst[t].  Load j of a lane covers item (j * kPCW + wave) * 64 + lane; only loads that still
// hold an old tag are re-issued.  Buffer addressing: ONE per-lane offset register serves every load of the sweep.
template <int MAXL>
__device__ __forceinline__ void gather(PCtx& c, __amdgpu_buffer_rsrc_t rs, int base, int n, unsigned tag, float* dst, int code) {
    if (c.dead) return;
    unsigned pend = 0;
#pragma unroll
    for (int j = 0; j < MAXL; ++j)
        if ((j * kPCW + c.wave) * 64 < n) pend |= 1u << j;
    const int t0 = c.wave * 64 + c.lane;
    const int voff = t0 * 8;
    pu32x2 v[MAXL];
    unsigned spins = 0;
    while (pend) {
#pragma unroll
        for (int j = 0; j < MAXL; ++j)
            if (pend & (1u << j)) v[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff, base + j * (kPCW * 64 * 8), 16);
        unsigned np = 0;
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            if (pend & (1u << j)) {
                const int t = t0 + j * (kPCW * 64);
                const bool ok = t >= n || v[j].y == tag;
                if (t < n && ok) dst[t] = __uint_as_float(v[j].x);
                if (!__all(ok)) np |= 1u << j;
            }
        }
        pend = np;
        if (pend && spin_fail(c, spins, code, 4)) break;
    }
}

// the same for up to 3 * 64 * kPCW items whose granule index comes from a map (attention: q_h | k_h | v_h slices)
template <typename Map>
__device__ __forceinline__ void gather_mapped(PCtx& c, __amdgpu_buffer_rsrc_t rs, int n, unsigned tag, float* dst, int code, Map map) {
    if (c.dead) return;
    constexpr int MAXL = 3;
    unsigned pend = 0;
#pragma unroll
    for (int j = 0; j < MAXL; ++j)
        if ((j * kPCW + c.wave) * 64 < n) pend |= 1u << j;
    pu32x2 v[MAXL];
    unsigned spins = 0;
    while (pend) {
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            if (pend & (1u << j)) {
                const int t = (j * kPCW + c.wave) * 64 + c.lane;
                v[j] = __builtin_amdgcn_raw_buffer_load_b64(rs, map(t < n ? t : 0) * 8, 0, 16);
            }
        }
        unsigned np = 0;
#pragma unroll
        for (int j = 0; j < MAXL; ++j) {
            if (pend & (1u << j)) {
                const int t = (j * kPCW + c.wave) * 64 + c.lane;
                const bool ok = t >= n || v[j].y == tag;
                if (t < n && ok) dst[t] = __uint_as_float(v[j].x);
                if (!__all(ok)) np |= 1u << j;
            }
        }
        pend = np;
        if (pend && spin_fail(c, spins, code, 4)) break;
    }
}

template <int VN>
__device__ __forceinline__ void vec_from_lds(const float* src, int lane, float4 (&v)[VN]) {
#pragma unroll
    for (int i = 0; i < VN; ++i) v[i] = *reinterpret_cast<const float4*>(src + i * 256 + lane * 4);
}

// dot product of one weight row (VN KiB in the ring, starting at byte `off` of the segment whose first fill is s0)
template <int VN>
__device__ __forceinline__ float row_dot(PCtx& c, const char* ring, unsigned rmask, unsigned s0, unsigned off,
                                         const float4 (&vec)[VN]) {
    wait_fill(c, s0 + ((off + VN * 1024u - 1u) >> 14));
    float sx = 0.f, sy = 0.f, sz = 0.f, sw = 0.f;
#pragma unroll
    for (int j = 0; j < VN; ++j) {
        const unsigned o = off + j * 1024u;
        const unsigned slot = (s0 + (o >> 14)) & rmask;
        const float4 w = *reinterpret_cast<const float4*>(ring + slot * kPSlot + (o & 16383u) + c.lane * 16);
        sx = fmaf(w.x, vec[j].x, sx); sy = fmaf(w.y, vec[j].y, sy);
        sz = fmaf(w.z, vec[j].z, sz); sw = fmaf(w.w, vec[j].w, sw);
    }
    return wave_sum((sx + sy) + (sz + sw));
}

template <int ND>
__device__ __forceinline__ void layer_norm_regs(float4 (&v)[ND], const float4 (&g)[ND], const float4 (&b)[ND]) {
    const float inv_d = 1.0f / (float)(256 * ND);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < ND; ++i) {
        v[i].x = (v[i].x - mean) * rstd * g[i].x + b[i].x; v[i].y = (v[i].y - mean) * rstd * g[i].y + b[i].y;
        v[i].z = (v[i].z - mean) * rstd * g[i].z + b[i].z; v[i].w = (v[i].w - mean) * rstd * g[i].w + b[i].w;
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

// ---- loader wave: the workgroup's weight rows, phase after phase, into the ring ---------------------------------
template <int ND>
__device__ __forceinline__ void persist_loader(const PersistArgs& A, PCtx& c, char* ring) {
    constexpr unsigned D = 256 * ND;
    const unsigned rmask = A.ring_slots - 1;
    const int rm = A.vocab / kPG, rem = A.vocab - rm * kPG;
    unsigned fseq = 0;
    const int n_seg = 4 * A.n_layer + 2;
#pragma unroll 1
    for (int sgi = 0; sgi < n_seg; ++sgi) {
        // segment = this workgroup's rows of one matrix: contiguous in the row-per-output layout
        const float* base;
        unsigned bytes;
        if (sgi < 4 * A.n_layer) {
            const PersistLayer& Ly = A.layers[sgi >> 2];
            const int ph = sgi & 3;
            if (ph == 0) { base = Ly.qkv_w + (size_t)c.wg * (3 * ND) * D; bytes = 3 * ND * D * 4; }
            else if (ph == 1) { base = Ly.proj_w + (size_t)c.wg * ND * D; bytes = ND * D * 4; }
            else if (ph == 2) { base = Ly.fc_w + (size_t)c.wg * (4 * ND) * D; bytes = 4 * ND * D * 4; }
            else { base = Ly.p2_w + (size_t)c.wg * ND * (4 * D); bytes = ND * 4 * D * 4; }
        } else if (sgi == 4 * A.n_layer) {
            base = A.head_w + (size_t)c.wg * rm * D; bytes = rm * D * 4;
        } else {
            base = A.head_w + (size_t)(rm * kPG + c.wg) * D; bytes = c.wg < rem ? D * 4 : 0;
        }
#pragma unroll 1
        for (unsigned off = 0; off < bytes; off += kPSlot) {
            // slot free?  (every consumer wave is past fill fseq - ring_slots)
            unsigned spins = 0;
            bool drained = false;
            while (!c.dead) {
                const unsigned m = min(min(lds_ld(c.ctl + 1), lds_ld(c.ctl + 2)), min(lds_ld(c.ctl + 3), lds_ld(c.ctl + 4)));
                if (fseq < m + (unsigned)A.ring_slots) break;
                if (!drained) {      // blocked: whatever was issued is landed and announced before waiting
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    lds_st(c.ctl, fseq);
                    drained = true;
                }
                if (spin_fail(c, spins, 902, 1)) break;
            }
            if (c.dead) return;
            const unsigned n = (min((unsigned)kPSlot, bytes - off)) >> 10;
            const unsigned slot = __builtin_amdgcn_readfirstlane(fseq & rmask);
            char* dst = ring + slot * kPSlot;
            const char* src = reinterpret_cast<const char*>(base) + off + c.lane * 16;
            if (n == 16) {           // thinned: at most this fill and the one before it in flight
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                lds_st(c.ctl, fseq);
            } else {
#pragma unroll 1
                for (unsigned i = 0; i < n; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                lds_st(c.ctl, fseq + 1);
            }
            ++fseq;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_st(c.ctl, fseq);
}

// ---- decode-step kernel --------------------------------------------------------------------------------------
template <int ND>      // d_model = 256 * ND
__global__ __launch_bounds__(kPThreads) void k_decode_persist(const PersistArgs A) {
    constexpr int D = 256 * ND;
    extern __shared__ __attribute__((aligned(16))) float smem[];      // (same declaration as k_gemv's)
    char* ring = reinterpret_cast<char*>(smem);
    float* hvec = reinterpret_cast<float*>(ring + (size_t)A.ring_slots * kPSlot);      // [4D] mlp hidden units
    float* xvec = hvec + 4 * D;                                                        // [D] phase input (x / x')
    float* ovec = hvec;                                                                // [D] merged attention output: aliases hvec, which
                                                                                       // is idle between E of a layer and E of the next
    float* ascr = xvec + D;                                                            // attention scratch
    unsigned* ctl = reinterpret_cast<unsigned*>(ascr + A.ascr_floats);
    if (threadIdx.x < 8) ctl[threadIdx.x] = 0u;
    __syncthreads();

    PCtx c;
    c.lane = threadIdx.x & 63; c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); c.wg = blockIdx.x;
    c.ctl = ctl; c.err = A.err; c.bar_target = 0; c.filled_seen = 0; c.dead = false;
    const unsigned epoch = __hip_atomic_load(A.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    if (c.wave == kPCW) {
        persist_loader<ND>(A, c, ring);
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
        const bool is_attn = wg < H * nchunks;
        const int ah = wg / nchunks, ac = wg - ah * nchunks;
        const float scale = 1.0f / sqrtf((float)hd);
        const int PS = D + 2 * H;                        // granules per key chunk: o[D] | {m, l}[H]
        // granule indices of the five hand-off buffers inside the allocation
        const int iQ = 0, iP = 3 * D, iX0 = iP + kPMaxChunks * PS, iH = iX0 + D, iX1 = iH + 4 * D;
        const __amdgpu_buffer_rsrc_t grs = make_rsrc(A.gran, (unsigned)(iX1 + D) * 8u);
        const unsigned tbase = ((epoch + 1u) & 0xfffffu) << 12;
        auto tag_of = [&](int l, int p) { return tbase | (unsigned)(l * 8 + p + 1); };
        const bool stamp = A.dbg && wg == 0 && wave == 0 && lane == 0;
        auto stamp_at = [&](int l, int p, int k) { if (stamp) A.dbg[(l * 5 + p) * 2 + k] = wall_clock64(); };
        unsigned fs = 0;                                 // first fill of the current weight segment
        constexpr unsigned nfA = (3 * ND * D * 4 + kPSlot - 1) / kPSlot, nfC = (ND * D * 4 + kPSlot - 1) / kPSlot,
                           nfD = (4 * ND * D * 4 + kPSlot - 1) / kPSlot;

        // ---- x = mel_embedding[tok] + mel_pos_embedding[pos]  (gpt_inference.py:92-96), by every workgroup ----
        if (wave < ND) {
            const float* e = A.mel_emb + (size_t)A.tok_in[0] * D + wave * 256 + lane * 4;
            const float* p = A.mel_pos + (size_t)A.st.mel_pos[slot] * D + wave * 256 + lane * 4;
            const float4 ev = *reinterpret_cast<const float4*>(e), pv = *reinterpret_cast<const float4*>(p);
            *reinterpret_cast<float4*>(xvec + wave * 256 + lane * 4) = make_float4(ev.x + pv.x, ev.y + pv.y, ev.z + pv.z, ev.w + pv.w);
        }
        float xres = 0.f;                                // residual element wg * ND + wave (waves < ND)
        if (stamp) A.dbg[2 * (5 * A.n_layer + 2)] = wall_clock64();

        for (int l = 0; l < A.n_layer; ++l) {
            const PersistLayer& Ly = A.layers[l];
            // =================== A: LN1 -> c_attn rows -> q | k | v ===================
            GVC_PHASE_BEGIN();
            {
                float4 g[ND], b[ND], xv[ND];
                load_gb<ND>(Ly.ln1_w, Ly.ln1_b, lane, g, b);
                constexpr int RA = 3 * ND, UPW = (RA + kPCW - 1) / kPCW;
                const int r0 = wave * UPW, nmy = max(0, min(RA, r0 + UPW) - r0);
                const int row_g = wg * RA + r0 + lane;
                const float bias = lane < nmy ? Ly.qkv_b[row_g] : 0.f;
                if (l > 0) gather<ND>(c, grs, iX1 * 8, D, tag_of(l - 1, 4), xvec, 100 + l);
                cbar(c);
                stamp_at(l, 0, 0);
                vec_from_lds<ND>(xvec, lane, xv);
                if (l == 0 && wave < ND) xres = xvec[wg * ND + wave];
                layer_norm_regs<ND>(xv, g, b);
                float val = 0.f;
#pragma unroll
                for (int i = 0; i < UPW; ++i) {
                    if (i < nmy) {
                        const float s = row_dot<ND>(c, ring, rmask, fs, (unsigned)(r0 + i) * D * 4, xv);
                        if (lane == i) val = s;
                    }
                }
                if (lane < nmy) {
                    val += bias;
                    publish(grs, iQ + row_g, tag_of(l, 0), val);
                    if (row_g >= D) {            // append k / v of this position to the cache (read by later steps)
                        const int which = row_g / D, ci = row_g - which * D;
                        const int h = ci / hd, j = ci - h * hd;
                        float* cache = which == 1 ? Ly.kcache : Ly.vcache;
                        cache[(((size_t)slot * H + h) * A.max_seq + S) * hd + j] = val;
                    }
                }
                fs += nfA;
                if (lane == 0) lds_st(ctl + 1 + wave, fs);
                stamp_at(l, 0, 1);
            }
            // =================== B: attention over one key chunk of one head (a few workgroups) ===================
            GVC_PHASE_BEGIN();
            if (is_attn) {
                const int kl = lane / lpk, dl = (lane - kl * lpk) * 4;
                const int k0 = ac * kc, k1 = min(S, k0 + kc);            // cached keys of this chunk
                // this head's rows of the layer's K / V cache as buffers: one per-lane offset serves every key of a pass
                const unsigned head_bytes = (unsigned)A.max_seq * hd * 4u;
                const __amdgpu_buffer_rsrc_t krs = make_rsrc(Ly.kcache + ((size_t)slot * H + ah) * A.max_seq * hd, head_bytes);
                const __amdgpu_buffer_rsrc_t vrs = make_rsrc(Ly.vcache + ((size_t)slot * H + ah) * A.max_seq * hd, head_bytes);
                const int kv_step = kPCW * kpw * hd * 4;                     // bytes between the keys u and u + 1 of a lane
                float4 kr[kPU], vr[kPU];
                auto load_pass = [&](int kbase) {
                    const int key0 = kbase + wave * kpw + kl;
                    const int voff = (key0 * hd + dl) * 4;
#pragma unroll
                    for (int u = 0; u < kPU; ++u) {
                        const bool ok = key0 + u * kPCW * kpw < k1;
                        pu32x4 kk = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
                        if (ok) {
                            kk = __builtin_amdgcn_raw_buffer_load_b128(krs, voff, u * kv_step, 0);
                            vv = __builtin_amdgcn_raw_buffer_load_b128(vrs, voff, u * kv_step, 0);
                        }
                        kr[u] = make_float4(__uint_as_float(kk.x), __uint_as_float(kk.y), __uint_as_float(kk.z), __uint_as_float(kk.w));
                        vr[u] = make_float4(__uint_as_float(vv.x), __uint_as_float(vv.y), __uint_as_float(vv.z), __uint_as_float(vv.w));
                    }
                };
                load_pass(k0);          // requested ahead of the seam: these rows were written by earlier launches
                const bool last_chunk = ac == nchunks - 1;
                const int nseg = last_chunk ? 3 : 1;     // q_h, and the k_h / v_h rows of this very step
                gather_mapped(c, grs, nseg * hd, tag_of(l, 0), ascr, 200 + l,
                              [&](int t) { const int sg = t / hd; return iQ + sg * D + ah * hd + (t - sg * hd); });
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
                    for (int u = 0; u < kPU; ++u) {
                        const int key = kbase + (u * kPCW + wave) * kpw + kl;
                        const float dsum = group_sum(dot4(q4, kr[u]), lpk);
                        s[u] = key < k1 ? dsum * scale : -INFINITY;
                    }
                    fold(s);
                }
                if (last_chunk && wave == 0) {           // the key of this step: lane group 0 of wave 0
                    float s[kPU];
#pragma unroll
                    for (int u = 0; u < kPU; ++u) {
                        s[u] = -INFINITY;
                        kr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                        vr[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                    }
                    kr[0] = *reinterpret_cast<const float4*>(ascr + hd + dl);
                    vr[0] = *reinterpret_cast<const float4*>(ascr + 2 * hd + dl);
                    const float dsum = group_sum(dot4(q4, kr[0]), lpk);
                    if (kl == 0) s[0] = dsum * scale;
                    fold(s);
                }
                // merge the kPCW * kpw lane-group states of the workgroup
                const int NG = kPCW * kpw;
                float* o_s = ascr + 3 * hd;              // [NG][hd]
                float* m_s = o_s + NG * hd;              // [NG]
                float* l_s = m_s + NG;                   // [NG]
                const int gidx = wave * kpw + kl;
                if (dl == 0) { m_s[gidx] = m; l_s[gidx] = lsum; }
                *reinterpret_cast<float4*>(o_s + gidx * hd + dl) = o;
                cbar(c);
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
            // =================== C: merge chunk partials -> attn c_proj rows -> x' = x + ... ===================
            GVC_PHASE_BEGIN();
            {
                const float bias = wave < ND ? Ly.proj_b[wg * ND + wave] : 0.f;
                if (wave < ND && !c.dead) {
                    const int e = wave * 256 + lane * 4, h = e / hd;
                    const unsigned tg = tag_of(l, 1);
                    pu32x4 oa[kPMaxChunks], ob[kPMaxChunks], ml[kPMaxChunks];
                    unsigned pend = (1u << nchunks) - 1u;
                    unsigned spins = 0;
                    while (pend) {
#pragma unroll
                        for (int ch = 0; ch < kPMaxChunks; ++ch) {
                            if (pend & (1u << ch)) {
                                const int off = (iP + ch * PS + e) * 8;
                                oa[ch] = __builtin_amdgcn_raw_buffer_load_b128(grs, off, 0, 16);
                                ob[ch] = __builtin_amdgcn_raw_buffer_load_b128(grs, off + 16, 0, 16);
                                ml[ch] = __builtin_amdgcn_raw_buffer_load_b128(grs, (iP + ch * PS + D + 2 * h) * 8, 0, 16);
                            }
                        }
                        unsigned np = 0;
#pragma unroll
                        for (int ch = 0; ch < kPMaxChunks; ++ch) {
                            if (pend & (1u << ch)) {
                                const bool ok = oa[ch].y == tg && oa[ch].w == tg && ob[ch].y == tg && ob[ch].w == tg &&
                                                ml[ch].y == tg && ml[ch].w == tg;
                                if (!__all(ok)) np |= 1u << ch;
                            }
                        }
                        pend = np;
                        if (pend && spin_fail(c, spins, 300 + l, 4)) break;
                    }
                    float M = -INFINITY;
#pragma unroll
                    for (int ch = 0; ch < kPMaxChunks; ++ch)
                        if (ch < nchunks) M = fmaxf(M, __uint_as_float(ml[ch].x));
                    float Lt = 0.f;
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int ch = 0; ch < kPMaxChunks; ++ch) {
                        if (ch < nchunks) {
                            const float wgt = __expf(__uint_as_float(ml[ch].x) - M);
                            Lt += wgt * __uint_as_float(ml[ch].z);
                            o.x = fmaf(wgt, __uint_as_float(oa[ch].x), o.x); o.y = fmaf(wgt, __uint_as_float(oa[ch].z), o.y);
                            o.z = fmaf(wgt, __uint_as_float(ob[ch].x), o.z); o.w = fmaf(wgt, __uint_as_float(ob[ch].z), o.w);
                        }
                    }
                    const float inv = 1.0f / Lt;
                    *reinterpret_cast<float4*>(ovec + e) = make_float4(o.x * inv, o.y * inv, o.z * inv, o.w * inv);
                }
                cbar(c);
                stamp_at(l, 2, 0);
                if (wave < ND) {
                    float4 ov[ND];
                    vec_from_lds<ND>(ovec, lane, ov);
                    const float s = row_dot<ND>(c, ring, rmask, fs, (unsigned)wave * D * 4, ov);
                    xres = xres + (s + bias);
                    if (lane == 0) publish(grs, iX0 + wg * ND + wave, tag_of(l, 2), xres);
                }
                fs += nfC;
                if (lane == 0) lds_st(ctl + 1 + wave, fs);
                stamp_at(l, 2, 1);
            }
            // =================== D: LN2 -> c_fc rows -> gelu_new ===================
            GVC_PHASE_BEGIN();
            {
                float4 g[ND], b[ND], xv[ND];
                load_gb<ND>(Ly.ln2_w, Ly.ln2_b, lane, g, b);
                constexpr int RD = 4 * ND, UPW = RD / kPCW;      // ND rows per wave
                const int r0 = wave * UPW;
                const int row_g = wg * RD + r0 + lane;
                const float bias = lane < UPW ? Ly.fc_b[row_g] : 0.f;
                gather<ND>(c, grs, iX0 * 8, D, tag_of(l, 2), xvec, 400 + l);
                cbar(c);
                stamp_at(l, 3, 0);
                vec_from_lds<ND>(xvec, lane, xv);
                layer_norm_regs<ND>(xv, g, b);
                float val = 0.f;
#pragma unroll
                for (int i = 0; i < UPW; ++i) {
                    const float s = row_dot<ND>(c, ring, rmask, fs, (unsigned)(r0 + i) * D * 4, xv);
                    if (lane == i) val = s;
                }
                if (lane < UPW) publish(grs, iH + row_g, tag_of(l, 3), gelu_new(val + bias));
                fs += nfD;
                if (lane == 0) lds_st(ctl + 1 + wave, fs);
                stamp_at(l, 3, 1);
            }
            // =================== E: mlp c_proj rows -> x = x' + ... ===================
            GVC_PHASE_BEGIN();
            {
                const float bias = wave < ND ? Ly.p2_b[wg * ND + wave] : 0.f;
                gather<4 * ND>(c, grs, iH * 8, 4 * D, tag_of(l, 3), hvec, 500 + l);
                cbar(c);
                stamp_at(l, 4, 0);
                if (wave < ND) {
                    float4 hv[4 * ND];
                    vec_from_lds<4 * ND>(hvec, lane, hv);
                    const float s = row_dot<4 * ND>(c, ring, rmask, fs, (unsigned)wave * 4 * D * 4, hv);
                    xres = xres + (s + bias);
                    if (lane == 0) publish(grs, iX1 + wg * ND + wave, tag_of(l, 4), xres);
                }
                fs += nfD;
                if (lane == 0) lds_st(ctl + 1 + wave, fs);
                stamp_at(l, 4, 1);
            }
        }
        // =================== head: ln_f -> final_norm -> latent -> mel_head rows ===================
        GVC_PHASE_BEGIN();
        {
            const int L = A.n_layer;
            float4 g[ND], b[ND], g2[ND], b2[ND], xv[ND];
            load_gb<ND>(A.lnf_w, A.lnf_b, lane, g, b);
            load_gb<ND>(A.fn_w, A.fn_b, lane, g2, b2);
            const int rm = A.vocab / kPG, rem = A.vocab - rm * kPG;
            const int upw = (rm + kPCW - 1) / kPCW;
            const int r0 = wave * upw, nmy = max(0, min(rm, r0 + upw) - r0);
            const int row_g = wg * rm + r0 + lane;
            const float bias = lane < nmy ? A.head_b[row_g] : 0.f;
            const bool tail = wave == 0 && wg < rem;
            const float tbias = tail ? A.head_b[rm * kPG + wg] : 0.f;
            gather<ND>(c, grs, iX1 * 8, D, tag_of(L - 1, 4), xvec, 600);
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
                const float s = row_dot<ND>(c, ring, rmask, fs, (unsigned)(r0 + i) * D * 4, xv);
                if (lane == i) val = s;
            }
            if (lane < nmy) A.logits_out[row_g] = val + bias;
            if (rm > 0) fs += (rm * D * 4 + kPSlot - 1) / kPSlot;
            if (tail) {
                const float s = row_dot<ND>(c, ring, rmask, fs, 0u, xv);
                if (lane == 0) A.logits_out[rm * kPG + wg] = s + tbias;
            }
            if (wg < rem) fs += (D * 4 + kPSlot - 1) / kPSlot;
            if (lane == 0) lds_st(ctl + 1 + wave, fs);
            stamp_at(L, 0, 1);
            if (wg == 0 && wave == 0 && lane == 0) {
                if (A.advance) {
                    if (A.st.seq_len[slot] < A.max_seq - 1) A.st.seq_len[slot] += 1;
                    if (A.st.mel_pos[slot] < A.max_mel_pos - 1) A.st.mel_pos[slot] += 1;
                }
                if (A.step_ctr) *A.step_ctr += 1;
            }
        }
    }
#undef GVC_PHASE_BEGIN
    // ---- end of step: the last workgroup to arrive opens the next epoch ----
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned prev = __hip_atomic_fetch_add(A.epoch + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev == gridDim.x - 1) {
            __hip_atomic_store(A.epoch + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(A.epoch, epoch + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

}  // namespace gvc
