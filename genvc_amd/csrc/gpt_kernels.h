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
    unsigned long long* dbg;                    // null, or 8 timestamps (100 MHz wall clock) of this launch
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
// PRO_LN_SUM keeps more loads in flight per lane: it is launched with at most 8 waves (256-VGPR budget)
// WB = 1: the matrix is streamed from its bf16 copy (Wt16, same [N][K] layout; half the HBM bytes), products and
// accumulation stay fp32.  Lane l then owns inputs [512j + 8l, +8) of its segment instead of [256i + 4l, +4).
template <int BT, int NI, int PRO, int EPI, int WB = 0>
__global__ __launch_bounds__(PRO == PRO_LN_SUM ? 512 : 1024) void k_gemv(const GemvArgs A) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int nwave = A.wpb;
    constexpr int KSEG = NI * 256;
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
__global__ __launch_bounds__(NW * 64) void k_attention(const AttnArgs A) {
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
        // (left alone, the scheduler sinks the V requests below the score phase: a second memory round trip per pass)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            float d = dot4(q4, kv_f4(kv[u]));
            if constexpr (LPK == 64) d = wave_sum(d);
            else if constexpr (LPK == 32) { d = row16_sum(d); d += __shfl_xor(d, 16); }
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
};

template <int HD, int KVB = 0>
__global__ __launch_bounds__(512, 2) void k_attn_proj(const AttnProjArgs A) {
    static_assert(HD == 256, "one key row = one float4 per lane");
    __shared__ float sc[8 * kFusedMaxKeys];
    __shared__ float l_s[8];
    __shared__ __attribute__((aligned(16))) float o_s[8][HD];
    __shared__ __attribute__((aligned(16))) float o_f[HD];
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
// Prefill attention on the matrix cores: one workgroup (8 waves) per (16 query rows, head, stream).
// k_attention gives every (row, head) its own workgroup, so the K/V rows of a head are pulled through L2 once per
// query row (240 MB per layer at 5 x 110 rows) and every key costs a wave reduction.  Here a tile of 16 query rows
//   1. S = scale * Q K^T : the key tiles are shared out over the waves; Q fragments stay in registers, K fragments
//      come straight from the cache rows (a float4 of 4 dims per lane), v_mfma_f32_16x16x4_f32; S lands in LDS;
//   2. softmax per row in LDS (two rows per wave): p = exp(s - max), masked past the row's last key, and 1 / sum;
//   3. O = P V : V is staged through LDS 64 keys at a time (requested a chunk ahead, coalesced key rows); every wave
//      owns HD/128 column tiles of the head; the normalised tile leaves through LDS as float4 (row-major or FM16).
// Same arithmetic as the reference's softmax(QK^T / sqrt(hd)) V (GPT2Attention._attn through
// /root/reference/layers/gpt_inference.py:81-91); sums in MFMA order, within the tests' 1e-4 of the oracle.
// ---------------------------------------------------------------------------------------------

typedef float at_f32x4 __attribute__((ext_vector_type(4)));

template <int HD, int KVB, int VC>          // VC: keys per V chunk
__global__ __launch_bounds__(512) void k_attention_tile(const AttnArgs A, int nkp) {
    constexpr int KB = HD / 16;                    // 16-dim blocks of a head
    constexpr int NTW = KB >= 8 ? KB / 8 : 1;      // output column tiles per wave
    constexpr int LDV = HD + 4;
    constexpr int ES = KVB ? 2 : 4;
    constexpr int VPT = VC * (HD / 4) / 512;       // float4 of a V chunk per thread
    extern __shared__ __attribute__((aligned(16))) float at_lds[];
    const int lds_s = nkp + 4;
    float* S = at_lds;                              // [16][lds_s]
    float* linv = S + 16 * lds_s;                   // [16]
    float* Vs = linv + 16;                          // [VC][LDV]; the output tile [16][LDV] afterwards
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int t0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int bi = A.slots ? A.slots[b] : b;
    const int base = A.base_len ? A.base_len[bi] : 0;
    const int nk = A.causal ? base + min(t0 + 16, A.T) : A.n_keys;       // keys any row of the tile sees
    const int nkt = (nk + 15) >> 4;
    const char* kp = reinterpret_cast<const char*>(A.kbase) + (bi * A.k_batch_stride + h * A.k_head_stride) * ES;
    const char* vp = reinterpret_cast<const char*>(A.vbase) + (bi * A.k_batch_stride + h * A.k_head_stride) * ES;
    const size_t krow = (size_t)A.k_row_stride * ES;

    // the first V chunk is requested before anything else: it arrives under phases 1 and 2
    typename KvRaw<KVB>::T vreg[VPT];
    auto v_request = [&](int c0) {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = tid + 512 * i;
            const int key = idx / (HD / 4), d4 = idx - key * (HD / 4);
            vreg[i] = load_kv_raw<KVB>(vp + (size_t)min(c0 + key, nk - 1) * krow + (size_t)d4 * 4 * ES);
        }
    };
    auto v_commit = [&]() {
#pragma unroll
        for (int i = 0; i < VPT; ++i) {
            const int idx = tid + 512 * i;
            const int key = idx / (HD / 4), d4 = idx - key * (HD / 4);
            *reinterpret_cast<float4*>(Vs + key * LDV + d4 * 4) = kv_f4(vreg[i]);
        }
    };
    v_request(0);

    // ---- 1. S = scale * Q K^T ----
    float4 qf[KB];
    {
        const int t = min(t0 + r16, A.T - 1);
        const float* qp = A.q + (size_t)(b * A.T + t) * A.q_stride + h * HD + 4 * g;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) qf[kb] = *reinterpret_cast<const float4*>(qp + kb * 16);
    }
    for (int kt = wave; kt < nkt; kt += 8) {
        const int key = min(kt * 16 + r16, nk - 1);
        const char* kr = kp + (size_t)key * krow + (size_t)(4 * g) * ES;
        typename KvRaw<KVB>::T kraw[KB];
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) kraw[kb] = load_kv_raw<KVB>(kr + (size_t)kb * 16 * ES);
        at_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // (two chains: a dependent MFMA waits ~40 cycles)
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float4 kf = kv_f4(kraw[kb]);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb].x, kf.x, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb].y, kf.y, acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb].z, kf.z, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf[kb].w, kf.w, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) S[(4 * g + q) * lds_s + kt * 16 + r16] = (acc[q] + acc1[q]) * A.scale;
    }
    __syncthreads();

    // ---- 2. softmax rows 2*wave, 2*wave + 1 ----
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        const int lim = A.causal ? min(base + t0 + r + 1, nk) : nk;
        float* sr = S + r * lds_s;
        float mx = -INFINITY;
        for (int k = lane; k < nkt * 16; k += 64) mx = fmaxf(mx, k < lim ? sr[k] : -INFINITY);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int k = lane; k < nkt * 16; k += 64) {
            const float pr = k < lim ? __expf(sr[k] - mx) : 0.f;
            sr[k] = pr;
            sum += pr;
        }
        sum = wave_sum(sum);
        if (lane == 0) linv[r] = 1.0f / sum;
    }

    // ---- 3. O = P V ----
    at_f32x4 oacc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) oacc[i] = {0.f, 0.f, 0.f, 0.f};
    const bool pv_wave = wave * NTW < KB;            // (HD = 64: four column tiles, waves 4..7 only help staging)
    for (int c0 = 0; c0 < nkt * 16; c0 += VC) {
        __syncthreads();                             // the previous chunk is consumed (first pass: S is complete)
        v_commit();
        __syncthreads();
        if (c0 + VC < nkt * 16) v_request(c0 + VC);
        if (pv_wave) {
#pragma unroll
            for (int kb = 0; kb < VC / 16; ++kb) {
                if (c0 + kb * 16 < nkt * 16) {
                    const float4 pf = *reinterpret_cast<const float4*>(S + r16 * lds_s + c0 + kb * 16 + 4 * g);
#pragma unroll
                    for (int i = 0; i < NTW; ++i) {
                        const float* vcol = Vs + (kb * 16 + 4 * g) * LDV + (wave * NTW + i) * 16 + r16;
                        oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.x, vcol[0], oacc[i], 0, 0, 0);
                        oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.y, vcol[LDV], oacc[i], 0, 0, 0);
                        oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.z, vcol[2 * LDV], oacc[i], 0, 0, 0);
                        oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.w, vcol[3 * LDV], oacc[i], 0, 0, 0);
                    }
                }
            }
        }
    }
    __syncthreads();
    // the normalised tile, row-major in LDS, then float4 stores
    float* Os = Vs;
    if (pv_wave) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) Os[(4 * g + q) * LDV + (wave * NTW + i) * 16 + r16] = oacc[i][q] * linv[4 * g + q];
    }
    __syncthreads();
    if (A.out_fm16) {
        // 16-dim block kb of the head: lane L carries slot L of the fragment-major block (row L & 15, dims 4 (L >> 4) ..)
        for (int kb = wave; kb < KB; kb += 8) {
            const int t = t0 + r16;
            if (t < A.T) {
                const float4 v = *reinterpret_cast<const float4*>(Os + r16 * LDV + kb * 16 + 4 * g);
                const int m = b * A.T + t, k = h * HD + kb * 16 + 4 * g, K16 = A.out_stride >> 4;
                *reinterpret_cast<float4*>(A.out + ((size_t)(m >> 4) * K16 + (k >> 4)) * 256 + (((m & 15) + 16 * ((k & 15) >> 2)) << 2)) = v;
            }
        }
    } else {
        for (int idx = tid; idx < 16 * (HD / 4); idx += 512) {
            const int r = idx / (HD / 4), d4 = idx - r * (HD / 4);
            if (t0 + r < A.T)
                *reinterpret_cast<float4*>(A.out + (size_t)(b * A.T + t0 + r) * A.out_stride + h * HD + d4 * 4) =
                    *reinterpret_cast<const float4*>(Os + r * LDV + d4 * 4);
        }
    }
}

// The common prefill case -- every row of the tile sees at most 128 keys (a segment's prefix without a cached context): Q, the
// whole K and the whole V of the head are requested up front with coalesced row loads (one memory round trip for the kernel),
// K and V take turns in one LDS buffer, and every fragment comes from LDS.  One key tile per wave in phase 1; the P V loop has
// no tail case (S is zero-filled up to 128 keys).
template <int HD, int KVB>
__global__ __launch_bounds__(512) void k_attention_tile_short(const AttnArgs A) {
    constexpr int KB = HD / 16, NTW = KB >= 8 ? KB / 8 : 1, LDV = HD + 4, ES = KVB ? 2 : 4;
    constexpr int NKP = 128, LDS_S = NKP + 4;
    constexpr int KPT = NKP * (HD / 4) / 512;      // float4 of K (or V) per thread
    constexpr int QPT = (16 * (HD / 4) + 511) / 512;
    extern __shared__ __attribute__((aligned(16))) float at_lds[];
    float* S = at_lds;                              // [16][LDS_S]
    float* linv = S + 16 * LDS_S;                   // [16]
    float* KV = linv + 16;                          // [128][LDV]: K, then V, then the output tile
    float* Qs = KV + NKP * LDV;                     // [16][LDV]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, g = lane >> 4;
    const int t0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const int bi = A.slots ? A.slots[b] : b;
    const int base = A.base_len ? A.base_len[bi] : 0;
    const int nk = min(NKP, A.causal ? base + min(t0 + 16, A.T) : A.n_keys);
    const int nkt = (nk + 15) >> 4;
    const char* kp = reinterpret_cast<const char*>(A.kbase) + (bi * A.k_batch_stride + h * A.k_head_stride) * ES;
    const char* vp = reinterpret_cast<const char*>(A.vbase) + (bi * A.k_batch_stride + h * A.k_head_stride) * ES;
    const size_t krow = (size_t)A.k_row_stride * ES;

    typename KvRaw<KVB>::T kreg[KPT], vreg[KPT];
    float4 qreg[QPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int idx = tid + 512 * i;
        const int key = idx / (HD / 4), d4 = idx - key * (HD / 4);
        kreg[i] = load_kv_raw<KVB>(kp + (size_t)min(key, nk - 1) * krow + (size_t)d4 * 4 * ES);
    }
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int idx = tid + 512 * i;
        const int r = min(idx / (HD / 4), 15), d4 = idx % (HD / 4);
        qreg[i] = *reinterpret_cast<const float4*>(A.q + (size_t)(b * A.T + min(t0 + r, A.T - 1)) * A.q_stride + h * HD + d4 * 4);
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int idx = tid + 512 * i;
        const int key = idx / (HD / 4), d4 = idx - key * (HD / 4);
        vreg[i] = load_kv_raw<KVB>(vp + (size_t)min(key, nk - 1) * krow + (size_t)d4 * 4 * ES);
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int idx = tid + 512 * i;
        const int key = idx / (HD / 4), d4 = idx - key * (HD / 4);
        *reinterpret_cast<float4*>(KV + key * LDV + d4 * 4) = kv_f4(kreg[i]);
    }
#pragma unroll
    for (int i = 0; i < QPT; ++i) {
        const int idx = tid + 512 * i;
        if (idx < 16 * (HD / 4)) *reinterpret_cast<float4*>(Qs + (idx / (HD / 4)) * LDV + (idx % (HD / 4)) * 4) = qreg[i];
    }
    __syncthreads();

    // ---- 1. S = scale * Q K^T: key tile `wave` ----
    if (wave < nkt) {
        at_f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};       // (two chains: a dependent MFMA waits ~40 cycles)
        const float* qrow = Qs + r16 * LDV + 4 * g;
        const float* krw = KV + (wave * 16 + r16) * LDV + 4 * g;
#pragma unroll
        for (int kb = 0; kb < KB; ++kb) {
            const float4 qf = *reinterpret_cast<const float4*>(qrow + kb * 16);
            const float4 kf = *reinterpret_cast<const float4*>(krw + kb * 16);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.x, kf.x, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.y, kf.y, acc1, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.z, kf.z, acc, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(qf.w, kf.w, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) S[(4 * g + q) * LDS_S + wave * 16 + r16] = (acc[q] + acc1[q]) * A.scale;
    }
    __syncthreads();                  // S complete; K is consumed

    // ---- 2. V takes the buffer; softmax rows 2*wave, 2*wave + 1 (zero past the row's last key, up to 128) ----
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const int idx = tid + 512 * i;
        const int key = idx / (HD / 4), d4 = idx - key * (HD / 4);
        *reinterpret_cast<float4*>(KV + key * LDV + d4 * 4) = kv_f4(vreg[i]);
    }
#pragma unroll
    for (int rr = 0; rr < 2; ++rr) {
        const int r = 2 * wave + rr;
        const int lim = A.causal ? min(base + t0 + r + 1, nk) : nk;
        float* sr = S + r * LDS_S;
        const float s0 = lane < lim ? sr[lane] : -INFINITY, s1 = lane + 64 < lim ? sr[lane + 64] : -INFINITY;
        const float mx = wave_max(fmaxf(s0, s1));
        const float p0 = lane < lim ? __expf(s0 - mx) : 0.f, p1 = lane + 64 < lim ? __expf(s1 - mx) : 0.f;
        sr[lane] = p0;
        sr[lane + 64] = p1;
        const float sum = wave_sum(p0 + p1);
        if (lane == 0) linv[r] = 1.0f / sum;
    }
    __syncthreads();

    // ---- 3. O = P V ----
    at_f32x4 oacc[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) oacc[i] = {0.f, 0.f, 0.f, 0.f};
    const bool pv_wave = wave * NTW < KB;
    if (pv_wave) {
#pragma unroll
        for (int kb = 0; kb < NKP / 16; ++kb) {
            const float4 pf = *reinterpret_cast<const float4*>(S + r16 * LDS_S + kb * 16 + 4 * g);
#pragma unroll
            for (int i = 0; i < NTW; ++i) {
                const float* vcol = KV + (kb * 16 + 4 * g) * LDV + (wave * NTW + i) * 16 + r16;
                oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.x, vcol[0], oacc[i], 0, 0, 0);
                oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.y, vcol[LDV], oacc[i], 0, 0, 0);
                oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.z, vcol[2 * LDV], oacc[i], 0, 0, 0);
                oacc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(pf.w, vcol[3 * LDV], oacc[i], 0, 0, 0);
            }
        }
    }
    // the normalised tile goes out through the Q buffer (nobody reads Q any more): row-major in LDS, then float4 stores
    float* Os = Qs;
    if (pv_wave) {
#pragma unroll
        for (int i = 0; i < NTW; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) Os[(4 * g + q) * LDV + (wave * NTW + i) * 16 + r16] = oacc[i][q] * linv[4 * g + q];
    }
    __syncthreads();
    if (A.out_fm16) {
        for (int kb = wave; kb < KB; kb += 8) {
            const int t = t0 + r16;
            if (t < A.T) {
                const float4 v = *reinterpret_cast<const float4*>(Os + r16 * LDV + kb * 16 + 4 * g);
                const int m = b * A.T + t, k = h * HD + kb * 16 + 4 * g, K16 = A.out_stride >> 4;
                *reinterpret_cast<float4*>(A.out + ((size_t)(m >> 4) * K16 + (k >> 4)) * 256 + (((m & 15) + 16 * ((k & 15) >> 2)) << 2)) = v;
            }
        }
    } else {
        for (int idx = tid; idx < 16 * (HD / 4); idx += 512) {
            const int r = idx / (HD / 4), d4 = idx - r * (HD / 4);
            if (t0 + r < A.T)
                *reinterpret_cast<float4*>(A.out + (size_t)(b * A.T + t0 + r) * A.out_stride + h * HD + d4 * 4) =
                    *reinterpret_cast<const float4*>(Os + r * LDV + d4 * 4);
        }
    }
}

// launch when the call is a prefill-shaped causal/direct one (T >= 16 rows per stream); false: the caller uses k_attention
static inline bool launch_attention_tile(int head_dim, int n_head, const AttnArgs& T, int batch, int max_keys, hipStream_t s,
                                         bool kv_bf16, int* rc) {
    // (few tiles -- one stream's segment prefix -- leave most CUs idle behind one latency chain: k_attention's row-per-workgroup grid wins)
    if (T.T < 16 || batch * ((T.T + 15) / 16) < 12 || (head_dim != 64 && head_dim != 128 && head_dim != 256)) return false;
    const int nkp = (max_keys + 15) & ~15;
    if (nkp <= 128) {
        const size_t lds_short = ((size_t)16 * 132 + 16 + (size_t)(128 + 16) * (head_dim + 4)) * sizeof(float);
        const dim3 grid_s((T.T + 15) / 16, n_head, batch);
#define GVC_ATT_SHORT(hd, kvb)                                                                                                    \
    {                                                                                                                             \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_tile_short<hd, kvb>),                           \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                   \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((k_attention_tile_short<hd, kvb>), grid_s, dim3(512), lds_short, s, T);                          \
    }
        if (head_dim == 256) { if (kv_bf16) GVC_ATT_SHORT(256, 1) else GVC_ATT_SHORT(256, 0) }
        else if (head_dim == 128) { if (kv_bf16) GVC_ATT_SHORT(128, 1) else GVC_ATT_SHORT(128, 0) }
        else { if (kv_bf16) GVC_ATT_SHORT(64, 1) else GVC_ATT_SHORT(64, 0) }
#undef GVC_ATT_SHORT
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("kernel launch failed: %s (attention tile)", hipGetErrorString(e)); *rc = GVC_ERR_HIP; return true; }
        *rc = GVC_OK;
        return true;
    }
    const int vc = 64;
    const size_t lds = ((size_t)16 * (nkp + 4) + 16 + (size_t)vc * (head_dim + 4)) * sizeof(float);
    if (lds > 160 * 1024) return false;
    const dim3 grid((T.T + 15) / 16, n_head, batch);
#define GVC_ATT_TILE(hd, kvb)                                                                                                     \
    {                                                                                                                             \
        static bool attr = false;                                                                                                 \
        if (!attr) {                                                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_attention_tile<hd, kvb, 64>),                             \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);                                   \
            attr = true;                                                                                                          \
        }                                                                                                                         \
        hipLaunchKernelGGL((k_attention_tile<hd, kvb, 64>), grid, dim3(512), lds, s, T, nkp);                                    \
    }
    if (head_dim == 256) { if (kv_bf16) GVC_ATT_TILE(256, 1) else GVC_ATT_TILE(256, 0) }
    else if (head_dim == 128) { if (kv_bf16) GVC_ATT_TILE(128, 1) else GVC_ATT_TILE(128, 0) }
    else { if (kv_bf16) GVC_ATT_TILE(64, 1) else GVC_ATT_TILE(64, 0) }
#undef GVC_ATT_TILE
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("kernel launch failed: %s (attention tile)", hipGetErrorString(e)); *rc = GVC_ERR_HIP; return true; }
    *rc = GVC_OK;
    return true;
}

// host launcher shared by the GPT and Perceiver contexts
static inline int launch_attention_hd(int head_dim, int n_head, const AttnArgs& T, int chunks, int rows, bool direct,
                                      hipStream_t s, bool wide = false, bool kv_bf16 = false) {
    dim3 grid(chunks, n_head, rows);
    dim3 block(256);
    if (kv_bf16 && head_dim == 256) {
        if (direct && wide) hipLaunchKernelGGL((k_attention<256, true, 16, 1>), grid, dim3(1024), 0, s, T);
        else if (wide) hipLaunchKernelGGL((k_attention<256, false, 16, 1>), grid, dim3(1024), 0, s, T);
        else if (direct) hipLaunchKernelGGL((k_attention<256, true, 4, 1>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<256, false, 4, 1>), grid, block, 0, s, T);
    } else if (kv_bf16 && head_dim == 64) {
        if (direct) hipLaunchKernelGGL((k_attention<64, true, 4, 1>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<64, false, 4, 1>), grid, block, 0, s, T);
    } else if (head_dim == 256) {
        if (direct && wide) hipLaunchKernelGGL((k_attention<256, true, 16>), grid, dim3(1024), 0, s, T);
        else if (wide) hipLaunchKernelGGL((k_attention<256, false, 16>), grid, dim3(1024), 0, s, T);
        else if (direct) hipLaunchKernelGGL((k_attention<256, true>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<256, false>), grid, block, 0, s, T);
    } else if (head_dim == 64) {
        if (direct) hipLaunchKernelGGL((k_attention<64, true>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<64, false>), grid, block, 0, s, T);
    } else if (kv_bf16 && head_dim == 128) {
        if (direct) hipLaunchKernelGGL((k_attention<128, true, 4, 1>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<128, false, 4, 1>), grid, block, 0, s, T);
    } else if (head_dim == 128) {
        if (direct) hipLaunchKernelGGL((k_attention<128, true>), grid, block, 0, s, T);
        else hipLaunchKernelGGL((k_attention<128, false>), grid, block, 0, s, T);
    } else {
        set_error("attention: head_dim %d unsupported (64, 128 or 256)", head_dim);
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
