// Device kernels of the GPT hot path (decode-step GEMVs, attention, row ops).
// Layouts (all fp32):
//   weights     row-per-output "Wt[N][K]" (HF Conv1D [in,out] is transposed once at bind time), so a
//               wave streams one output row as 1 KiB-per-instruction coalesced float4 loads;
//   KV cache    [layer][k|v][slot][head][max_seq][head_dim]: a key/value row is head_dim*4 contiguous
//               bytes (1 KiB at head_dim 256) and rows of one head are adjacent;
//   residual x  [row][d]; q [row][d]; mlp hidden [row][4d];
//   attention partials [row][head][chunk][head_dim + 4]  (o[hd], m, l, pad) for split-key decode.
#pragma once
#include "common.h"

namespace gvc {

constexpr int kAttnChunks = 8;   // key-axis split of one decode attention row (flash-decoding style)

struct GptState {                // device-resident step state, one entry per slot
    int32_t* seq_len;            // cached positions
    int32_t* mel_pos;            // index into mel_pos_embedding of the next decode input
};

// see prefetch_wave() below
struct Prefetch {
    const char* base;    // null: nothing to prefetch
    int chunk_bytes;     // bytes read by one workgroup of the next launch (contiguous)
    int n_chunks;
};

enum Prologue { PRO_LN = 0, PRO_MERGE = 1, PRO_COPY = 2, PRO_LN2X = 3, PRO_LN_SUM = 4 };
enum Epilogue { EPI_QKV = 0, EPI_RESID = 1, EPI_GELU = 2, EPI_LOGITS = 3 };

struct GemvArgs {
    // GEMV operand
    const float* Wt;     // [N][K]
    const unsigned short* Wt16;   // bf16 copy of Wt (bf16-weights contexts), else null
    const float* bias;   // [N]
    int N, K;
    int wpb;             // waves per block
    int ksplit;          // waves per output row (each owns K/ksplit consecutive inputs)
    int B;               // live streams (<= BT)
    // row addressing of the residual stream: row(b) = b * x_stride + x_off
    float* x;
    int x_stride, x_off;
    int d;
    // prologue operands
    const float* ln_w; const float* ln_b;       // PRO_LN / first LN of PRO_LN2X
    const float* ln2_w; const float* ln2_b;     // second LN of PRO_LN2X
    const float* in;                            // PRO_COPY: [B][K]; PRO_MERGE: attention partials
    int embed;                                  // PRO_LN: 1 -> x = mel_emb[tok] + mel_pos[pos] first
    const float* mel_emb; const float* mel_pos_tab; const int32_t* tok_in;
    int n_head, head_dim;
    // epilogue operands
    float* out;                                 // q [B][d] / hidden [B][N] / logits [B][N]
    float* latent_out;                          // PRO_LN2X: [B][d]
    float* kcache; float* vcache;               // this layer's [slot][head][max_seq][hd]
    int kv_bf16;                                // the cache holds bf16 elements (weight_dtype 2)
    int max_seq, max_mel_pos;
    const int32_t* slots;
    GptState st;
    int advance;                                // EPI_LOGITS: 1 -> seq_len++, mel_pos++ per slot
    int32_t* step_ctr;                          // EPI_LOGITS: nullable, ++ once per launch (generation loop)
    Prefetch pf;                                // next launch's weights (L2 warm-up by the extra wave)
    unsigned long long* dbg;                    // null, or 8 timestamps (100 MHz wall clock) of this launch
    int32_t* prog;                              // null, or the step's progress counter (++ when this launch starts)
    // fused attention path (k_attn_proj): PRO_LN_SUM builds x' = x + pbias + sum_h part2[b][h] and stores it to x2;
    // EPI_RESID then reads its residual from xres (= x2) instead of x
    const float* part2; const float* pbias; float* x2; const float* xres;
    int* err;                                   // device-visible host word: set when a slot's KV cache is full (EPI_LOGITS)
};

// ---------------------------------------------------------------------------------------------
// decode-step GEMV: out[b][n] = epilogue( sum_k Wt[n][k] * a[b][k] + bias[n] ),  a = prologue(x)
//
// One wave owns one (output row, K-segment of 256*NI inputs).  Program order per wave is
//   prologue input loads (L2-resident, tiny)  ->  all NI float4 weight loads (1 KiB per
//   wave-instruction, straight to VGPRs: a GEMV operand is streamed once and never shared, so an LDS
//   round trip would be pure overhead)  ->  prologue math (LayerNorm / attention merge) while the
//   HBM stream is in flight  ->  LDS-staged input vectors read back as conflict-free ds_read_b128.
// vmcnt retires in order, hence the small prologue loads are issued BEFORE the weight stream.
// HBM-bound: N*K*4 algorithmic bytes per launch.  BT = streams per launch (padded), B live ones.
// ---------------------------------------------------------------------------------------------
// Warm the XCD-local L2 with the weights the NEXT kernel will stream.  Executed by one extra wave per
// workgroup that returns right afterwards: its loads are fire-and-forget (s_endpgm waits for them), so
// the compute waves' in-order vmcnt never sees them and the kernel overlaps its own latency chain with
// the next operand's HBM transfer.  Chunk cb is what workgroup cb of the next launch reads; it is
// fetched by a workgroup with the same index modulo 8 (observed XCD placement; affects speed only).
__device__ __forceinline__ void prefetch_wave(const Prefetch& P, int lane, int block, int nblocks) {
    const int r = block & 7;
    const int nb_r = (nblocks - r + 7) >> 3;            // workgroups of this launch with residue r
    const int i = block >> 3;
    // hipcc does not track an asm load: the destination is a register kept live ("+v") up to our own
    // wait, so it can never be re-used for an address while a load is still in flight
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t sink = {0.f, 0.f, 0.f, 0.f};
    for (int cb = r + 8 * i; cb < P.n_chunks; cb += 8 * nb_r) {
        const char* p = P.base + (size_t)cb * P.chunk_bytes;
        for (int off = lane * 16; off < P.chunk_bytes; off += 64 * 16)
            asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(p + off) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
}

// PRO_LN_SUM keeps more loads in flight per lane: it is launched with at most 8 waves (256-VGPR budget)
// WB = 1: the matrix is streamed from its bf16 copy (Wt16, same [N][K] layout; half the HBM bytes), products and
// accumulation stay fp32.  Lane l then owns inputs [512j + 8l, +8) of its segment instead of [256i + 4l, +4).
template <int BT, int NI, int PRO, int EPI, int WB = 0>
__global__ __launch_bounds__(PRO == PRO_LN_SUM ? 512 : 1024) void k_gemv(const GemvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwave = A.wpb;                   // compute waves; wave == nwave is the prefetcher
    if (wave >= nwave) {
        if (A.pf.base) prefetch_wave(A.pf, lane, blockIdx.x, gridDim.x);
        return;                                // a terminated wave no longer counts at s_barrier
    }
    constexpr int KSEG = NI * 256;
    if (A.prog && blockIdx.x == 0 && threadIdx.x == 0)
        __hip_atomic_fetch_add(A.prog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool stamp = A.dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1);
    unsigned long long* dbg = A.dbg + (blockIdx.x == 0 ? 0 : 4);
    if (stamp) dbg[0] = wall_clock64();
    float* a_lds = smem;                       // [BT][K]
    float* red = smem + BT * A.K;              // [wpb][BT] cross-wave partial sums (ksplit > 1)

    const int item = blockIdx.x * nwave + wave;
    const int row = item / A.ksplit;
    const int seg = item - row * A.ksplit;
    const bool live = row < A.N;
    const float* wp = A.Wt + (size_t)(live ? row : 0) * A.K + seg * KSEG + lane * 4;
    float4 w[NI];
    constexpr int NB = (NI + 1) / 2;           // 16-byte bf16 loads per lane (8 inputs each)
    uint4 wq[NB];
    const unsigned short* wp16 = A.Wt16 + (size_t)(live ? row : 0) * A.K + seg * KSEG + lane * 8;

    // ---- epilogue operands are requested first: nothing is loaded after the reduction ----
    const bool fin = live && seg == 0 && lane < A.B;     // lane b finishes stream b
    float e_bias = A.bias[live ? row : 0];
    float e_res = 0.f;
    int e_pos = 0;
    if constexpr (EPI == EPI_RESID) {
        if (fin) e_res = (A.xres ? A.xres : A.x)[(size_t)(lane * A.x_stride + A.x_off) * A.d + row];
    }
    if constexpr (EPI == EPI_QKV) {
        if (fin && row >= A.d) e_pos = A.st.seq_len[A.slots[lane]];
    }

    // weights are read exactly once per step by exactly one wave: non-temporal (evict-first) loads keep
    // them from displacing the activations / KV cache in L2 and MALL
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x4_nt __attribute__((ext_vector_type(4)));
    auto load_w = [&]() {
        if constexpr (WB == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(wp + i * 256));
                w[i] = make_float4(t.x, t.y, t.z, t.w);
            }
        } else {
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                wq[j] = make_uint4(0u, 0u, 0u, 0u);
                if (j * 512 + lane * 8 < KSEG) {
                    const u32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const u32x4_nt*>(wp16 + j * 512));
                    wq[j] = make_uint4(t.x, t.y, t.z, t.w);
                }
            }
        }
    };
    // residual-stream row of stream b (layer 0 of a decode step builds it from the embeddings)
    auto load_x = [&](int b, float4 (&v)[NI]) {
        float* xr = A.x + (size_t)(b * A.x_stride + A.x_off) * A.d;
        if (PRO == PRO_LN && A.embed) {
            const int slot = A.slots[b];
            const float* e = A.mel_emb + (size_t)A.tok_in[b] * A.d;
            const float* p = A.mel_pos_tab + (size_t)A.st.mel_pos[slot] * A.d;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float4 ev = *reinterpret_cast<const float4*>(e + i * 256 + lane * 4);
                const float4 pv = *reinterpret_cast<const float4*>(p + i * 256 + lane * 4);
                v[i] = make_float4(ev.x + pv.x, ev.y + pv.y, ev.z + pv.z, ev.w + pv.w);
                if (blockIdx.x == 0) *reinterpret_cast<float4*>(xr + i * 256 + lane * 4) = v[i];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NI; ++i) v[i] = *reinterpret_cast<const float4*>(xr + i * 256 + lane * 4);
        }
    };
    // one LayerNorm of v in registers
    auto layer_norm = [&](float4 (&v)[NI], const float4 (&g)[NI], const float4 (&c)[NI]) {
        const float inv_d = 1.0f / (float)A.d;
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        const float mean = wave_sum(s) * inv_d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            v[i].x = (v[i].x - mean) * rstd * g[i].x + c[i].x;
            v[i].y = (v[i].y - mean) * rstd * g[i].y + c[i].y;
            v[i].z = (v[i].z - mean) * rstd * g[i].z + c[i].z;
            v[i].w = (v[i].w - mean) * rstd * g[i].w + c[i].w;
        }
    };
    auto load_gb = [&](const float* gw, const float* gb, float4 (&g)[NI], float4 (&c)[NI]) {
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            g[i] = *reinterpret_cast<const float4*>(gw + i * 256 + lane * 4);
            c[i] = *reinterpret_cast<const float4*>(gb + i * 256 + lane * 4);
        }
    };
    auto ln_store = [&](int b, float4 (&v)[NI], const float4 (&g)[NI], const float4 (&c)[NI]) {
        layer_norm(v, g, c);
        if constexpr (PRO == PRO_LN2X) {
            float4 g2[NI], c2[NI];
            load_gb(A.ln2_w, A.ln2_b, g2, c2);
            layer_norm(v, g2, c2);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            *reinterpret_cast<float4*>(a_lds + b * A.K + i * 256 + lane * 4) = v[i];
            if (PRO == PRO_LN2X && blockIdx.x == 0)
                *reinterpret_cast<float4*>(A.latent_out + (size_t)b * A.d + i * 256 + lane * 4) = v[i];
        }
    };

    if constexpr (PRO == PRO_LN_SUM) {
        // ONE stream (fused attention path): x' = x + c_proj bias + sum_h part2[h], then LayerNorm.  Wave g owns
        // outputs [256g, 256g+256): all its operands are requested at once (one round trip), the row statistics
        // are combined through LDS.
        const bool has = wave < NI;
        const int col = wave * 256 + lane * 4;
        float4 t = make_float4(0.f, 0.f, 0.f, 0.f), g4 = t, c4 = t;
        if (has) {
            const float* xr = A.x + (size_t)A.x_off * A.d;
            t = *reinterpret_cast<const float4*>(xr + col);
            const float4 pb = *reinterpret_cast<const float4*>(A.pbias + col);
            g4 = *reinterpret_cast<const float4*>(A.ln_w + col);
            c4 = *reinterpret_cast<const float4*>(A.ln_b + col);
            float4 ph[16];
#pragma unroll
            for (int h = 0; h < 16; ++h)
                if (h < A.n_head) ph[h] = *reinterpret_cast<const float4*>(A.part2 + (size_t)h * A.d + col);
            t.x += pb.x; t.y += pb.y; t.z += pb.z; t.w += pb.w;
#pragma unroll
            for (int h = 0; h < 16; ++h)
                if (h < A.n_head) { t.x += ph[h].x; t.y += ph[h].y; t.z += ph[h].z; t.w += ph[h].w; }
        }
        load_w();
        const float inv_d = 1.0f / (float)A.d;
        if (has) {
            if (blockIdx.x == 0) *reinterpret_cast<float4*>(A.x2 + (size_t)A.x_off * A.d + col) = t;
            const float s1 = wave_sum((t.x + t.y) + (t.z + t.w));
            if (lane == 0) red[wave] = s1;
        }
        __syncthreads();
        float mean = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) mean += red[i];
        mean *= inv_d;
        if (has) {
            const float a0 = t.x - mean, a1 = t.y - mean, a2 = t.z - mean, a3 = t.w - mean;
            const float s2 = wave_sum((a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3));
            if (lane == 0) red[NI + wave] = s2;
        }
        __syncthreads();
        float var = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) var += red[NI + i];
        const float rstd = 1.0f / sqrtf(var * inv_d + 1e-5f);
        if (has) {
            t.x = (t.x - mean) * rstd * g4.x + c4.x; t.y = (t.y - mean) * rstd * g4.y + c4.y;
            t.z = (t.z - mean) * rstd * g4.z + c4.z; t.w = (t.w - mean) * rstd * g4.w + c4.w;
            *reinterpret_cast<float4*>(a_lds + col) = t;
        }
    } else if constexpr (PRO == PRO_LN || PRO == PRO_LN2X) {
        // one wave per stream (K == d == 256*NI); row, gain and bias are requested before the weights
        float4 v[NI], g[NI], c[NI];
        const bool mine = wave < A.B;
        if (mine) { load_x(wave, v); load_gb(A.ln_w, A.ln_b, g, c); }
        load_w();
        if (mine) ln_store(wave, v, g, c);
        for (int b = wave + nwave; b < A.B; b += nwave) { load_x(b, v); ln_store(b, v, g, c); }
        for (int b = A.B + wave; b < BT; b += nwave)
#pragma unroll
            for (int i = 0; i < NI; ++i)
                *reinterpret_cast<float4*>(a_lds + b * A.K + i * 256 + lane * 4) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else if constexpr (PRO == PRO_COPY) {
        // a = in[b][0..K): every thread moves float4s; BT*K/4 of them, at most 8 per thread in flight
        const int total4 = BT * A.K / 4;
        const int nthr = nwave * 64;
        constexpr int CU = 8;
        bool first = true;
        for (int base = threadIdx.x; base < total4; base += nthr * CU) {
            float4 t[CU];
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int idx = base + u * nthr;
                const int b = (idx * 4) / A.K;
                t[u] = (idx < total4 && b < A.B) ? *reinterpret_cast<const float4*>(A.in + (size_t)idx * 4)
                                                 : make_float4(0.f, 0.f, 0.f, 0.f);
            }
            if (first) { load_w(); first = false; }
#pragma unroll
            for (int u = 0; u < CU; ++u) {
                const int idx = base + u * nthr;
                if (idx < total4) *reinterpret_cast<float4*>(a_lds + (size_t)idx * 4) = t[u];
            }
        }
        if (first) load_w();
    } else {  // PRO_MERGE: combine the kAttnChunks partial softmax states of each head
        // work item = (stream b, group of 256 outputs); items are spread over the workgroup's waves so the
        // dependent loads of one item are a single round trip per wave
        const int pstride = A.head_dim + 4;   // (o[hd], m, l, pad): keeps float4 alignment
        const int ngrp = A.K / 256;
        bool first = true;
        for (int item2 = wave; item2 < BT * ngrp; item2 += nwave) {
            const int b = item2 / ngrp;
            const int k = (item2 - b * ngrp) * 256 + lane * 4;
            float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
            float mc[kAttnChunks], lc[kAttnChunks];
            float4 oc[kAttnChunks];
            if (b < A.B) {
                const int h = k / A.head_dim;
                const int j = k - h * A.head_dim;
                const float* pp = A.in + ((size_t)(b * A.n_head + h) * kAttnChunks) * pstride;
#pragma unroll
                for (int c = 0; c < kAttnChunks; ++c) {
                    mc[c] = pp[c * pstride + A.head_dim];
                    lc[c] = pp[c * pstride + A.head_dim + 1];
                    oc[c] = *reinterpret_cast<const float4*>(pp + c * pstride + j);
                }
            }
            if (first) { load_w(); first = false; }     // partials requested, then the weight stream
            if (b < A.B) {
                float m = mc[0];
#pragma unroll
                for (int c = 1; c < kAttnChunks; ++c) m = fmaxf(m, mc[c]);
                float L = 0.f;
#pragma unroll
                for (int c = 0; c < kAttnChunks; ++c) {
                    const float wgt = __expf(mc[c] - m);
                    L += wgt * lc[c];
                    o.x += wgt * oc[c].x; o.y += wgt * oc[c].y; o.z += wgt * oc[c].z; o.w += wgt * oc[c].w;
                }
                const float inv = 1.0f / L;
                o.x *= inv; o.y *= inv; o.z *= inv; o.w *= inv;
            }
            *reinterpret_cast<float4*>(a_lds + b * A.K + k) = o;
        }
        if (first) load_w();
    }
    if (stamp) dbg[1] = wall_clock64();         // prologue of this wave done
    __syncthreads();
    if (stamp) dbg[2] = wall_clock64();         // input vectors staged by the whole workgroup

    // ---- dot products against the staged vectors ----
    float acc[BT];
    const float* ap = a_lds + seg * KSEG + lane * 4;
#pragma unroll
    for (int b = 0; b < BT; ++b) {
        float s = 0.f;
        if constexpr (WB == 0) {
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float4 av = *reinterpret_cast<const float4*>(ap + b * A.K + i * 256);
                s = fmaf(w[i].x, av.x, s); s = fmaf(w[i].y, av.y, s);
                s = fmaf(w[i].z, av.z, s); s = fmaf(w[i].w, av.w, s);
            }
        } else {
            const float* ap8 = a_lds + seg * KSEG + lane * 8 + b * A.K;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (j * 512 + lane * 8 < KSEG) {
                    const float4 a0 = *reinterpret_cast<const float4*>(ap8 + j * 512);
                    const float4 a1 = *reinterpret_cast<const float4*>(ap8 + j * 512 + 4);
                    // a bf16 is the upper half of an fp32: element 2i sits in the low 16 bits of word i
                    s = fmaf(__uint_as_float(wq[j].x << 16), a0.x, s); s = fmaf(__uint_as_float(wq[j].x & 0xffff0000u), a0.y, s);
                    s = fmaf(__uint_as_float(wq[j].y << 16), a0.z, s); s = fmaf(__uint_as_float(wq[j].y & 0xffff0000u), a0.w, s);
                    s = fmaf(__uint_as_float(wq[j].z << 16), a1.x, s); s = fmaf(__uint_as_float(wq[j].z & 0xffff0000u), a1.y, s);
                    s = fmaf(__uint_as_float(wq[j].w << 16), a1.z, s); s = fmaf(__uint_as_float(wq[j].w & 0xffff0000u), a1.w, s);
                }
            }
        }
        acc[b] = wave_sum(s);
    }
    if (A.ksplit > 1) {      // combine the K-segments of a row (adjacent waves of this block)
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < BT; ++b) red[wave * BT + b] = acc[b];
        }
        __syncthreads();
        if (seg == 0) {
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                float s = 0.f;
                for (int j = 0; j < A.ksplit; ++j) s += red[(wave + j) * BT + b];
                acc[b] = s;
            }
        }
    }

    if (stamp) dbg[3] = wall_clock64();         // weights consumed, reductions done
    // ---- epilogue: lane b finishes stream b ----
    if (fin) {
        float val = 0.f;
#pragma unroll
        for (int b = 0; b < BT; ++b)
            if (lane == b) val = acc[b];
        const int b = lane;
        val += e_bias;
        if constexpr (EPI == EPI_QKV) {
            const int which = row / A.d;
            const int c = row - which * A.d;
            if (which == 0) {
                A.out[(size_t)b * A.d + c] = val;
            } else {
                const int slot = A.slots[b];
                const int h = c / A.head_dim;
                const int j = c - h * A.head_dim;
                float* cache = which == 1 ? A.kcache : A.vcache;
                const size_t at = (((size_t)slot * A.n_head + h) * A.max_seq + e_pos) * A.head_dim + j;
                if (A.kv_bf16) reinterpret_cast<unsigned short*>(cache)[at] = f32_to_bf16(val);
                else cache[at] = val;
            }
        } else if constexpr (EPI == EPI_RESID) {
            A.x[(size_t)(b * A.x_stride + A.x_off) * A.d + row] = e_res + val;
        } else if constexpr (EPI == EPI_GELU) {
            A.out[(size_t)b * A.N + row] = gelu_new(val);
        } else {
            A.out[(size_t)b * A.N + row] = val;
        }
    }
    if constexpr (EPI == EPI_LOGITS) {
        // the slot's cache grew by one position; the last cache row is re-used once the slot is full
        if (A.advance && blockIdx.x == 0 && threadIdx.x < A.B) {
            const int slot = A.slots[threadIdx.x];
            // a full cache: the position is not advanced (no write past the slot) and the host is told (GVC_ERR_STATE on its next call)
            if (A.st.seq_len[slot] < A.max_seq - 1) A.st.seq_len[slot] += 1;
            else if (A.err) *A.err = 950;
            if (A.st.mel_pos[slot] < A.max_mel_pos - 1) A.st.mel_pos[slot] += 1;
            else if (A.err) *A.err = 951;
        }
        if (A.step_ctr && blockIdx.x == 0 && threadIdx.x == 0) *A.step_ctr += 1;
    }
}

// ---------------------------------------------------------------------------------------------
// attention over a cached key range with online softmax.
//   grid (chunks, heads, rows); 256 threads.  A key/value row is HD floats: HD/4 lanes cover it with
//   float4 loads (a wave handles 64/(HD/4) keys per instruction); K and V rows of a key are fetched
//   together so a chunk costs one memory round trip.  Each lane-group keeps a running (m, l, o) state,
//   the block merges them through LDS.  DIRECT: one chunk, normalised output row; otherwise the
//   (o, m, l) partial of the chunk is stored for the consumer GEMV's merge prologue.
// ---------------------------------------------------------------------------------------------
struct AttnArgs {
    const float* q;            // [rows][q_stride]; head h at + h*HD
    int q_stride;
    const void* kbase;         // GPT: layer K cache [slot][head][max_seq][HD]; generic: see k_* strides.  fp32, or bf16 (KVB = 1)
    const void* vbase;
    long long k_batch_stride;  // elements between batch elements (GPT: slot stride)
    long long k_head_stride;   // elements between heads
    int k_row_stride;          // elements between consecutive keys
    int T;                     // rows per batch element (row = b*T + t)
    const int32_t* slots;      // nullable: batch index -> slot
    const int32_t* base_len;   // nullable: per-slot cached length before this call
    int causal;                // 1: keys [0, base + t + 1); 0: keys [0, n_keys)
    int n_keys;
    float scale;
    float* out;                // DIRECT: [rows][out_stride] ; else partials [rows][heads][chunks][HD+4]
    int out_stride;
    Prefetch pf;               // next launch's weights, fetched by the 5th wave (blockDim 320)
    int32_t* prog;             // null, or the step's progress counter (++ when this launch starts)
    int out_fm16;              // DIRECT: write `out` in the FM16 layout of gemm.h (row length out_stride)
};

// NW = waves that share the keys of one (chunk, head, row): 4, or 16 for the batched decode rows (one new row over a
// long cached context: 16 waves x U keys are requested per round trip instead of 4 x U)
// four consecutive cache elements of a key row: requested as stored (KVB = 1: bf16, 8 bytes per lane) and widened to fp32 only
// where they are used -- a conversion next to the load would make every request wait for the one before it
template <int KVB> struct KvRaw { typedef float4 T; };
template <> struct KvRaw<1> { typedef uint2 T; };

template <int KVB>
__device__ __forceinline__ typename KvRaw<KVB>::T load_kv_raw(const char* p) {
    return *reinterpret_cast<const typename KvRaw<KVB>::T*>(p);
}
__device__ __forceinline__ float4 kv_f4(const float4& r) { return r; }
__device__ __forceinline__ float4 kv_f4(const uint2& u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <int HD, bool DIRECT, int NW = 4, int KVB = 0>
__global__ __launch_bounds__(NW == 4 ? 320 : NW * 64) void k_attention(const AttnArgs A) {
    if (NW == 4 && threadIdx.x >= 256) {      // optional prefetcher wave, see prefetch_wave()
        if (A.pf.base)
            prefetch_wave(A.pf, threadIdx.x & 63, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z),
                          gridDim.x * gridDim.y * gridDim.z);
        return;
    }
    if (A.prog && threadIdx.x == 0 && blockIdx.x + blockIdx.y + blockIdx.z == 0)
        __hip_atomic_fetch_add(A.prog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    constexpr int LPK = HD / 4;          // lanes per key
    constexpr int KPW = 64 / LPK;        // keys per wave-instruction
    constexpr int NG = NW * KPW;         // softmax states per block
    __shared__ float m_s[NG], l_s[NG];
    __shared__ __attribute__((aligned(16))) float o_s[NG][HD];

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int kl = lane / LPK;                 // key sub-slot inside the wave
    const int dl = (lane % LPK) * 4;           // first dim owned by the lane
    const int chunk = blockIdx.x, h = blockIdx.y, row = blockIdx.z;
    const int nchunk = gridDim.x, nhead = gridDim.y;
    const int b = row / A.T, t = row - b * A.T;
    const int bi = A.slots ? A.slots[b] : b;
    const int base = A.base_len ? A.base_len[bi] : 0;
    const int nk = A.causal ? base + t + 1 : A.n_keys;
    const int cs = (nk + nchunk - 1) / nchunk;
    const int k0 = chunk * cs;
    const int k1 = min(nk, k0 + cs);

    const float4 q4 = *reinterpret_cast<const float4*>(A.q + (size_t)row * A.q_stride + h * HD + dl);
    constexpr int ES = KVB ? 2 : 4;            // bytes per cache element
    const char* kp = reinterpret_cast<const char*>(A.kbase) + (bi * A.k_batch_stride + h * A.k_head_stride + dl) * ES;
    const char* vp = reinterpret_cast<const char*>(A.vbase) + (bi * A.k_batch_stride + h * A.k_head_stride + dl) * ES;

    float m = -INFINITY, l = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
    constexpr int U = NW == 4 ? 4 : 8;
    for (int kb = k0 + wave * KPW + kl; kb < k1 + (U * NW * KPW); kb += U * NW * KPW) {
        if (kb - kl - wave * KPW >= k1) break;           // uniform per block iteration
        typename KvRaw<KVB>::T kv[U], vv[U];
        float s[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int key = kb + u * NW * KPW;
            const bool ok = key < k1;
            const size_t off = (size_t)(ok ? key : k0) * A.k_row_stride * ES;
            kv[u] = load_kv_raw<KVB>(kp + off);
            vv[u] = load_kv_raw<KVB>(vp + off);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float d = dot4(q4, kv_f4(kv[u]));
            if constexpr (LPK == 64) d = wave_sum(d);
            else d = row16_sum(d);                         // LPK == 16
            s[u] = (kb + u * NW * KPW < k1) ? d * A.scale : -INFINITY;
        }
        float mn = m;
#pragma unroll
        for (int u = 0; u < U; ++u) mn = fmaxf(mn, s[u]);
        if (mn > -INFINITY) {
            const float alpha = __expf(m - mn);            // m = -inf -> 0
            l *= alpha; o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float p = __expf(s[u] - mn);         // masked keys: exp(-inf) = 0
                const float4 v4 = kv_f4(vv[u]);
                l += p;
                o.x = fmaf(p, v4.x, o.x); o.y = fmaf(p, v4.y, o.y);
                o.z = fmaf(p, v4.z, o.z); o.w = fmaf(p, v4.w, o.w);
            }
            m = mn;
        }
    }
    const int g = wave * KPW + kl;
    if ((lane % LPK) == 0) { m_s[g] = m; l_s[g] = l; }
    *reinterpret_cast<float4*>(&o_s[g][dl]) = o;
    __syncthreads();
    if (threadIdx.x < HD) {
        const int j = threadIdx.x;
        float M = -INFINITY;
#pragma unroll
        for (int i = 0; i < NG; ++i) M = fmaxf(M, m_s[i]);
        float L = 0.f, acc = 0.f;
        if (M > -INFINITY) {
#pragma unroll
            for (int i = 0; i < NG; ++i) {
                const float wgt = __expf(m_s[i] - M);
                L += wgt * l_s[i];
                acc += wgt * o_s[i][j];
            }
        }
        if constexpr (DIRECT) {
            if (A.out_fm16) {
                const int m = row, k = h * HD + j, K16 = A.out_stride >> 4;
                A.out[((size_t)(m >> 4) * K16 + (k >> 4)) * 256 + ((((m & 15) + 16 * ((k & 15) >> 2)) << 2) + (k & 3))] = acc / L;
            } else
            A.out[(size_t)row * A.out_stride + h * HD + j] = acc / L;
        } else {
            float* pp = A.out + (((size_t)row * nhead + h) * nchunk + chunk) * (HD + 4);
            pp[j] = acc;
            if (j == 0) { pp[HD] = M; pp[HD + 1] = L; }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Fused decode attention + attn c_proj for ONE stream with a short context (keys <= 8 * kFusedMaxKeys).
// grid (d/16, heads), 8 waves.  Workgroup (i, h) recomputes the attention of head h (all its K/V rows are
// requested up front: one memory round trip, the rows come from L2 for all but the first workgroup of a
// head) and multiplies o_h into its 16 rows of the c_proj slice Wp[n][h*HD .. (h+1)*HD): a K-split of the
// projection by head.  The `heads` partial vectors are summed by the next launch's prologue (PRO_LN_SUM), so
// the separate attention launch and its boundary disappear.  Redundant attention work grows with the
// context, hence the host only takes this path while 2*S*HD*4 bytes per workgroup stay small.
// ---------------------------------------------------------------------------------------------
constexpr int kFusedMaxKeys = 16;       // keys per wave

struct AttnProjArgs {
    const float* q;            // [d] of the stream
    const void* kcache; const void* vcache;     // layer base [slot][head][max_seq][HD], fp32 or bf16 (KVB = 1)
    const int32_t* slots; const int32_t* seq_len;
    int max_seq, n_head, d;
    float scale;
    const float* Wp;           // attn c_proj, row-per-output [d][d]
    float* part2;              // [heads][d] partial projections
    int32_t* prog;
};

template <int HD, int KVB = 0>
__global__ __launch_bounds__(512, 2) void k_attn_proj(const AttnProjArgs A) {
    static_assert(HD == 256, "one key row = one float4 per lane");
    __shared__ float sc[8 * kFusedMaxKeys];
    __shared__ float l_s[8];
    __shared__ __attribute__((aligned(16))) float o_s[8][HD];
    __shared__ __attribute__((aligned(16))) float o_f[HD];
    if (A.prog && threadIdx.x == 0 && blockIdx.x + blockIdx.y == 0)
        __hip_atomic_fetch_add(A.prog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int h = blockIdx.y;
    const int slot = A.slots[0];
    const int nk = A.seq_len[slot] + 1;                      // the key appended by this step included
    const float4 q4 = *reinterpret_cast<const float4*>(A.q + h * HD + lane * 4);
    constexpr int ES = KVB ? 2 : 4;
    const size_t head_off = (((size_t)slot * A.n_head + h) * A.max_seq * HD + lane * 4) * ES;
    const char* kp = reinterpret_cast<const char*>(A.kcache) + head_off;
    const char* vp = reinterpret_cast<const char*>(A.vcache) + head_off;
    typename KvRaw<KVB>::T kr[kFusedMaxKeys], vr[kFusedMaxKeys];
#pragma unroll
    for (int j = 0; j < kFusedMaxKeys; ++j) {
        const int key = wave + 8 * j;            // wave-uniform: rows past the context are not requested at all
        kr[j] = typename KvRaw<KVB>::T{};        // all-zero bits are 0.0f in both storage types
        if (key < nk) kr[j] = load_kv_raw<KVB>(kp + (size_t)key * HD * ES);
    }
#pragma unroll
    for (int j = 0; j < kFusedMaxKeys; ++j) {
        const int key = wave + 8 * j;
        vr[j] = typename KvRaw<KVB>::T{};
        if (key < nk) vr[j] = load_kv_raw<KVB>(vp + (size_t)key * HD * ES);
    }
    // this wave's two rows of the head's c_proj slice (streamed once, non-temporal)
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));
    const int row0 = blockIdx.x * 16 + wave * 2;
    f32x4_nt w0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(A.Wp + (size_t)row0 * A.d + h * HD + lane * 4));
    f32x4_nt w1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(A.Wp + (size_t)(row0 + 1) * A.d + h * HD + lane * 4));

    float s[kFusedMaxKeys];
#pragma unroll
    for (int j = 0; j < kFusedMaxKeys; ++j) {
        const int key = wave + 8 * j;
        s[j] = key < nk ? wave_sum(dot4(q4, kv_f4(kr[j]))) * A.scale : -INFINITY;
        if (lane == 0) sc[key] = s[j];
    }
    __syncthreads();
    float m = fmaxf(sc[lane], sc[lane + 64]);                // 8 * 16 = 128 scores
    m = wave_max(m);
    float l = 0.f;
    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int j = 0; j < kFusedMaxKeys; ++j) {
        const float p = __expf(s[j] - m);                    // masked keys: exp(-inf) = 0
        const float4 v4 = kv_f4(vr[j]);
        l += p;
        o.x = fmaf(p, v4.x, o.x); o.y = fmaf(p, v4.y, o.y);
        o.z = fmaf(p, v4.z, o.z); o.w = fmaf(p, v4.w, o.w);
    }
    *reinterpret_cast<float4*>(&o_s[wave][lane * 4]) = o;
    if (lane == 0) l_s[wave] = l;
    __syncthreads();
    if (threadIdx.x < HD) {
        float L = 0.f, acc = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { L += l_s[w]; acc += o_s[w][threadIdx.x]; }
        o_f[threadIdx.x] = acc / L;
    }
    __syncthreads();
    const float4 of = *reinterpret_cast<const float4*>(&o_f[lane * 4]);
    float d0 = fmaf(w0.x, of.x, fmaf(w0.y, of.y, fmaf(w0.z, of.z, w0.w * of.w)));
    float d1 = fmaf(w1.x, of.x, fmaf(w1.y, of.y, fmaf(w1.z, of.z, w1.w * of.w)));
    d0 = wave_sum(d0);
    d1 = wave_sum(d1);
    if (lane == 0) {
        A.part2[(size_t)h * A.d + row0] = d0;
        A.part2[(size_t)h * A.d + row0 + 1] = d1;
    }
}

// ---------------------------------------------------------------------------------------------
// Fused MLP block of the one-stream decode step: [x' = x + attn c_proj bias + sum_h partials; LN2; c_fc; gelu_new]
// -> in-kernel all-to-all of the 4d hidden units -> [mlp c_proj; x = x' + ...], ONE launch instead of two.
//   * grid = d/4 workgroups (one per CU at d = 1024), all co-resident; 8 compute waves + 8 auxiliary waves.
//   * the compute waves request BOTH weight slices of their workgroup at kernel start (16 hidden rows of c_fc and
//     4 output rows of c_proj, 64 KiB each at d = 1024), so the c_proj stream runs underneath the c_fc phase and the
//     exchange; nothing else is ever loaded by them (vmcnt retires in order per wave).
//   * the auxiliary waves (no weight load outstanding) do the prologue and the exchange: every hidden unit is published
//     as an 8-byte {tag, value} granule with an agent-scope store and gathered with agent-scope loads (the only
//     hand-off that crosses XCDs: L2s are not coherent with each other); tag = launch epoch, kept in device memory
//     and bumped by workgroup 0 at its end, after it has seen every workgroup's granules.
//   * spins are bounded: a timeout raises *err (checked by the host) instead of hanging the GPU.
// Workgroup barriers are bare s_barrier + lgkmcnt wait, so they do not drain the weight stream.
// ---------------------------------------------------------------------------------------------
struct MlpArgs {
    float* x;                    // residual row of the stream [d], updated in place
    const float* part2;          // [heads][d] per-head attention-projection partials (k_attn_proj)
    const float* pbias;          // attn c_proj bias [d]
    const float* ln_w; const float* ln_b;
    const float* Wfc; const float* bfc;      // [4d][d], [4d]
    const float* Wp2; const float* bp2;      // [d][4d], [d]
    int d, n_head;
    unsigned long long* gran;    // [4d] granules
    unsigned* epoch;             // tag of the last launch that used `gran`
    int* err;
    int32_t* prog;
    unsigned long long* dbg;     // null, or 8 timestamps: workgroup 0 {entry, LN2 ready, published, gathered}, last workgroup the same
};

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int NI>        // d = 128 * NI
__global__ __launch_bounds__(1024) void k_mlp_fused(const MlpArgs A) {
    constexpr int D = 128 * NI, F = 4 * D, D4 = D / 4;
    constexpr int WPR1 = D4 / 64;          // waves per c_fc row (K = d)
    constexpr int WPR2 = D / 64;           // wave-partials per c_proj row (K = 4d): 2 * NI
    __shared__ __attribute__((aligned(16))) float a_s[D];      // LN2(x')
    __shared__ __attribute__((aligned(16))) float xp_s[D];     // x'
    __shared__ __attribute__((aligned(16))) float h_s[F];      // gathered hidden units
    __shared__ float red1[16][WPR1 > 0 ? WPR1 : 1];
    __shared__ float red2[4][WPR2];
    __shared__ float stat1[8], stat2[8];
    __shared__ unsigned tag_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    typedef float f32x4_nt __attribute__((ext_vector_type(4)));
    if (wave < 8) {
        // ---------------- compute waves: weights only ----------------
        f32x4_nt w1[NI], w2[NI];
        const f32x4_nt* p1 = reinterpret_cast<const f32x4_nt*>(A.Wfc + (size_t)wg * 16 * D) + tid;
        const f32x4_nt* p2 = reinterpret_cast<const f32x4_nt*>(A.Wp2 + (size_t)wg * 4 * F) + tid;
        // epilogue operands first (vmcnt retires in order: nothing may be requested behind the weight stream)
        const float e_b1 = tid < 16 ? A.bfc[wg * 16 + tid] : 0.f;
        const float e_b2 = tid < 4 ? A.bp2[wg * 4 + tid] : 0.f;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NI; ++i) w1[i] = __builtin_nontemporal_load(p1 + i * 512);
        __builtin_amdgcn_sched_barrier(0);
        lds_barrier();               // b0
        lds_barrier();               // b0'
        lds_barrier();               // b1: a_s ready
        // the c_proj slice is requested only now: asked for at kernel start it doubles the queue in front of the prologue's
        // small loads (they then return after ~6 us instead of ~2); from here it streams underneath phase 1 and the exchange
#pragma unroll
        for (int i = 0; i < NI; ++i) w2[i] = __builtin_nontemporal_load(p2 + i * 512);
        __builtin_amdgcn_sched_barrier(0);
        // phase 1: 16 hidden rows of c_fc; float4 f = i*512 + tid sits in row f / D4 at column 4 * (f % D4)
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = i * 512 + tid;
            const float4 av = *reinterpret_cast<const float4*>(&a_s[(f % D4) * 4]);
            float s = fmaf(w1[i].x, av.x, 0.f); s = fmaf(w1[i].y, av.y, s); s = fmaf(w1[i].z, av.z, s); s = fmaf(w1[i].w, av.w, s);
            s = wave_sum(s);
            if (lane == 0) red1[(i * 512 + wave * 64) / D4][((i * 512 + wave * 64) % D4) / 64] = s;
        }
        lds_barrier();               // b2: row partials ready
        if (tid < 16) {
            float v = e_b1;
#pragma unroll
            for (int q = 0; q < WPR1; ++q) v += red1[tid][q];
            v = gelu_new(v);
            const unsigned tag = tag_s;
            __hip_atomic_store(A.gran + wg * 16 + tid, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        lds_barrier();               // b3: h_s ready
        // phase 2: 4 output rows of c_proj; float4 f sits in row f / D at column 4 * (f % D) of the 4d inputs
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int f = i * 512 + tid;
            const float4 hv = *reinterpret_cast<const float4*>(&h_s[(f % D) * 4]);
            float s = fmaf(w2[i].x, hv.x, 0.f); s = fmaf(w2[i].y, hv.y, s); s = fmaf(w2[i].z, hv.z, s); s = fmaf(w2[i].w, hv.w, s);
            s = wave_sum(s);
            if (lane == 0) red2[(i * 512 + wave * 64) / D][((i * 512 + wave * 64) % D) / 64] = s;
        }
        lds_barrier();               // b4
        if (tid < 4) {
            const int row = wg * 4 + tid;
            float v = e_b2;
#pragma unroll
            for (int q = 0; q < WPR2; ++q) v += red2[tid][q];
            A.x[row] = xp_s[row] + v;
        }
        return;
    }
    // ---------------- auxiliary waves: prologue + exchange ----------------
    const int at = tid - 512;                     // 0..511
    const int aw = wave - 8;                      // 0..7
    const bool stamp = A.dbg && at == 0 && (wg == 0 || wg == (int)gridDim.x - 1);
    unsigned long long* dbg = A.dbg + (wg == 0 ? 0 : 4);
    if (stamp) dbg[0] = wall_clock64();
    if (at == 0) {
        if (A.prog && wg == 0) __hip_atomic_fetch_add(A.prog, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tag_s = *A.epoch + 1u;
    }
    // x' = x + pbias + sum_h part2[h]: thread `at` owns floats [4*at, 4*at+4) when 4*at < D
    const bool has = at * 4 < D;
    const int col = at * 4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f), g4 = t, c4 = t;
    if (has) {
        t = *reinterpret_cast<const float4*>(A.x + col);
        const float4 pb = *reinterpret_cast<const float4*>(A.pbias + col);
        g4 = *reinterpret_cast<const float4*>(A.ln_w + col);
        c4 = *reinterpret_cast<const float4*>(A.ln_b + col);
        float4 ph[16];
#pragma unroll
        for (int h = 0; h < 16; ++h)
            if (h < A.n_head) ph[h] = *reinterpret_cast<const float4*>(A.part2 + (size_t)h * D + col);
        t.x += pb.x; t.y += pb.y; t.z += pb.z; t.w += pb.w;
#pragma unroll
        for (int h = 0; h < 16; ++h)
            if (h < A.n_head) { t.x += ph[h].x; t.y += ph[h].y; t.z += ph[h].z; t.w += ph[h].w; }
        *reinterpret_cast<float4*>(&xp_s[col]) = t;
    }
    constexpr int AWV = D / 256;                  // auxiliary waves that hold data (4 at d = 1024, 1 at d = 256)
    const float inv_d = 1.0f / (float)D;
    {
        const float s1 = wave_sum((t.x + t.y) + (t.z + t.w));
        if (lane == 0) stat1[aw] = s1;
    }
    lds_barrier();                   // b0
    float mean = 0.f;
#pragma unroll
    for (int i = 0; i < AWV; ++i) mean += stat1[i];
    mean *= inv_d;
    {
        float s2 = 0.f;
        if (has) {
            const float a0 = t.x - mean, a1 = t.y - mean, a2 = t.z - mean, a3 = t.w - mean;
            s2 = (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
        s2 = wave_sum(s2);
        if (lane == 0) stat2[aw] = s2;
    }
    lds_barrier();                   // b0'
    float var = 0.f;
#pragma unroll
    for (int i = 0; i < AWV; ++i) var += stat2[i];
    const float rstd = 1.0f / sqrtf(var * inv_d + 1e-5f);
    if (has) {
        float4 o;
        o.x = (t.x - mean) * rstd * g4.x + c4.x; o.y = (t.y - mean) * rstd * g4.y + c4.y;
        o.z = (t.z - mean) * rstd * g4.z + c4.z; o.w = (t.w - mean) * rstd * g4.w + c4.w;
        *reinterpret_cast<float4*>(&a_s[col]) = o;
    }
    lds_barrier();                   // b1
    if (stamp) dbg[1] = wall_clock64();
    lds_barrier();                   // b2 (the compute waves publish right after it)
    if (stamp) dbg[2] = wall_clock64();
    // gather the 4d hidden units of all workgroups
    {
        const unsigned tag = tag_s;
        int spins = 0;
        for (int idx = at; idx < F; idx += 512) {
            unsigned long long g;
            while (true) {
                g = __hip_atomic_load(A.gran + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(g >> 32) == tag)) break;
                if (++spins > 400000) { if (lane == 0) *A.err = 1; break; }      // ~0.1 s: not all workgroups resident?
                __builtin_amdgcn_s_sleep(1);
            }
            h_s[idx] = __uint_as_float((unsigned)g);
        }
    }
    if (stamp) dbg[3] = wall_clock64();
    lds_barrier();                   // b3
    lds_barrier();                   // b4
    // every workgroup has published (this one gathered all of them) and read the epoch long ago: bump it for the next launch
    if (wg == 0 && at == 0) *A.epoch = tag_s;
}

// ---------------------------------------------------------------------------------------------
// Step-long weight prefetcher (one launch per decode step on a side stream, concurrent with the step graph).
// One single-wave workgroup per CU polls the step's progress counter; when launch number `trigger` has
// started it pulls the operand of a LATER launch from HBM into its XCD's L2 (chunk cb goes to a workgroup
// with cb % 8 == its own index % 8, matching the consumer's observed placement), so the consumer's
// non-temporal weight loads hit L2 and HBM keeps streaming across launch boundaries.  Pure loads: results
// never depend on it; the spin is bounded.
// ---------------------------------------------------------------------------------------------
struct PrefetchEntry {
    int trigger;            // issue when *prog >= trigger
    Prefetch pf;
};

static __global__ __launch_bounds__(64) void k_step_prefetcher(const PrefetchEntry* sched, int n_entries, const int32_t* prog,
                                                               int base, int max_polls, int fetch) {
    const int lane = threadIdx.x;
    typedef float f32x4_t __attribute__((ext_vector_type(4)));
    f32x4_t sink = {0.f, 0.f, 0.f, 0.f};
    const int r = blockIdx.x & 7;
    const int nb_r = (gridDim.x - r + 7) >> 3;
    const int i = blockIdx.x >> 3;
    for (int e = 0; e < n_entries; ++e) {
        const int trig = base + sched[e].trigger;     // the counter is monotone over the steps of a context
        int polls = 0;
        while (__hip_atomic_load(prog, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - trig < 0) {
            __builtin_amdgcn_s_sleep(8);
            if (++polls > max_polls) return;        // the step is not progressing: give up quietly
        }
        const Prefetch P = sched[e].pf;
        if (!fetch) continue;
        for (int cb = r + 8 * i; cb < P.n_chunks; cb += 8 * nb_r) {
            const char* p = P.base + (size_t)cb * P.chunk_bytes;
            for (int off = lane * 16; off < P.chunk_bytes; off += 64 * 16)
                asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(sink) : "v"(p + off) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
    }
}

// host launcher shared by the GPT and Perceiver contexts
static inline int launch_attention_hd(int head_dim, int n_head, const AttnArgs& T, int chunks, int rows, bool direct,
                                      hipStream_t s, bool wide = false, bool kv_bf16 = false) {
    dim3 grid(chunks, n_head, rows);
    dim3 block(T.pf.base ? 320 : 256);
    if (kv_bf16 && head_dim == 256) {
        if (direct && wide) hipLaunchKernelGGL((k_attention<256, true, 16, 1>), grid, dim3(1024), 0, s, T);
        else if (direct) hipLaunchKernelGGL((k_attention<256, true, 4, 1>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<256, false, 4, 1>), grid, block, 0, s, T);
    } else if (kv_bf16 && head_dim == 64) {
        if (direct) hipLaunchKernelGGL((k_attention<64, true, 4, 1>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<64, false, 4, 1>), grid, block, 0, s, T);
    } else if (head_dim == 256) {
        if (direct && wide) hipLaunchKernelGGL((k_attention<256, true, 16>), grid, dim3(1024), 0, s, T);
        else if (direct) hipLaunchKernelGGL((k_attention<256, true>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<256, false>), grid, block, 0, s, T);
    } else if (head_dim == 64) {
        if (direct) hipLaunchKernelGGL((k_attention<64, true>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<64, false>), grid, block, 0, s, T);
    } else {
        set_error("attention: head_dim %d unsupported (64 or 256)", head_dim);
        return GVC_ERR_UNSUPPORTED;
    }
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

// ---------------------------------------------------------------------------------------------
// row kernels for the prefill / re-pass path (one wave per row of d floats)
// ---------------------------------------------------------------------------------------------
// LayerNorm (optionally twice: ln_f then final_norm) of rows src -> dst
__global__ void k_ln_rows(const float* src, float* dst, int rows, int d, const float* w1, const float* b1,
                          const float* w2, const float* b2, int dst_fm16);
// build GPT input rows: t < P -> prefix_emb[b][t]; else mel_embedding[tok] + mel_pos[t - P]
//   tok: t == P -> start_tok; 1 <= t-P <= n -> codes[b][t-P-1]; beyond -> stop_tok
__global__ void k_embed_rows(float* x, const float* prefix_emb, int B, int T, int P, int d, const float* mel_emb,
                             const float* mel_pos, const int32_t* codes, int n, int start_tok, int stop_tok, int t_off);
// prefix rows of GPT.compute_embeddings
__global__ void k_prefix_rows(float* out, const float* cond, int n_cond, const int32_t* codes, int B, int Tc,
                              int d, const float* text_emb, const float* text_pos, int start_text,
                              int stop_text);
__global__ void k_set_state(GptState st, const int32_t* slots, int B, int seq_len, int mel_pos);
// gather rows [b][off .. off+n) of src [B][T][d] into dst [B][n][d]
__global__ void k_gather_rows(const float* src, float* dst, int B, int T, int off, int n, int d);
// round fp32 values to bf16 (nearest even) in place and write the packed bf16 copy
__global__ void k_round_bf16(float* w, unsigned short* out, size_t n);
// transpose Conv1D weight [K][N] -> [N][K] (LDS-tiled, 32x32)
__global__ void k_transpose(const float* src, float* dst, int K, int N);

}  // namespace gvc
