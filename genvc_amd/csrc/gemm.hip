// fp32 MFMA GEMM (see gemm.h)
#include "gemm.h"
#include <algorithm>

namespace gvc {

constexpr int BM = 64, BN = 64, BK = 32, LDL = 36;

__global__ __launch_bounds__(256) void k_gemm_f32(const GemmArgs G) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDL];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int batch = blockIdx.z / G.SK, sk = blockIdx.z - batch * G.SK;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int ntile_all = (G.K + BK - 1) / BK;
    const int tps = (ntile_all + G.SK - 1) / G.SK;
    const int kbeg = sk * tps * BK;
    const int kend = min(G.K, kbeg + tps * BK);
    const int nt = kend > kbeg ? (kend - kbeg + BK - 1) / BK : 0;
    const float* Ab = G.A + gemm_boff(G, batch, G.a_batch_stride, G.a_batch_stride2);
    const float* Wb = G.Wt + gemm_binner(G, batch) * G.w_batch_stride;

    float4 ra[2], rb[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            const int k = kbeg + kt * BK + c4;
            const int m = m0 + r, n = n0 + r;
            size_t aoff = (size_t)m * G.lda + k;
            if (G.conv_cin > 0) {       // a float4 never straddles a tap (conv_cin % 4 == 0)
                const int tap = k / G.conv_cin;
                aoff = (size_t)m * G.lda + (size_t)tap * G.conv_tap_stride + (k - tap * G.conv_cin);
            }
            ra[j] = (m < G.M && k < kend) ? *reinterpret_cast<const float4*>(Ab + aoff) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[j] = (n < G.N && k < kend) ? *reinterpret_cast<const float4*>(Wb + (size_t)n * G.ldw + k)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            if (G.a_act == AACT_LRELU) {        // (applied where the tile is staged, a k step after its request: next to the load it waited for it)
                ra[j].x = ra[j].x > 0.f ? ra[j].x : ra[j].x * G.a_slope; ra[j].y = ra[j].y > 0.f ? ra[j].y : ra[j].y * G.a_slope;
                ra[j].z = ra[j].z > 0.f ? ra[j].z : ra[j].z * G.a_slope; ra[j].w = ra[j].w > 0.f ? ra[j].w : ra[j].w * G.a_slope;
            }
            *reinterpret_cast<float4*>(&As[buf][r * LDL + c4]) = ra[j];
            *reinterpret_cast<float4*>(&Bs[buf][r * LDL + c4]) = rb[j];
        }
    };

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    if (nt > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    int buf = 0;
    const int arow = (wm * 32 + (lane & 31)) * LDL + (lane >> 5) * 4;
    const int brow = (wn * 32 + (lane & 31)) * LDL + (lane >> 5) * 4;
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) load_tile(kt + 1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const float4 a4 = *reinterpret_cast<const float4*>(&As[buf][arow + kk * 8]);
            const float4 b4 = *reinterpret_cast<const float4*>(&Bs[buf][brow + kk * 8]);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
        }
        if (kt + 1 < nt) store_tile(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    const int n = n0 + wn * 32 + (lane & 31);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < G.M && n < G.N) {
            if (G.SK > 1)
                G.work[(((size_t)batch * G.SK + sk) * G.M + m) * G.N + n] = acc[r];
            else
                gemm_store(G, batch, m, n, acc[r]);
        }
    }
}

__global__ void k_splitk_epilogue(const GemmArgs G) {
    const int batch = blockIdx.z;
    const size_t mn = (size_t)G.M * G.N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (size_t)gridDim.x * blockDim.x) {
        const float* w = G.work + (size_t)batch * G.SK * mn + i;
        // the SK <= 16 planes are requested together (clamped index; a loop with a run-time trip count was a round trip per plane)
        // and added in plane order
        float pv[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) pv[s] = w[(size_t)min(s, G.SK - 1) * mn];
        float v = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s)
            if (s < G.SK) v += pv[s];
        gemm_store(G, batch, (int)(i / G.N), (int)(i % G.N), v);
    }
}

int launch_gemm_cap(GemmArgs G, int batch, long long work_cap, hipStream_t s) {
    GVC_REQUIRE(G.K % 4 == 0 && G.lda % 4 == 0 && G.ldw % 4 == 0 && G.conv_cin % 4 == 0 && G.conv_tap_stride % 4 == 0,
                GVC_ERR_ARG, "gemm: K/lda/ldw/conv strides must be multiples of 4 (K=%d lda=%d ldw=%d)", G.K, G.lda, G.ldw);
    const int tm = cdiv(G.M, BM), tn = cdiv(G.N, BN);
    const long long tiles = (long long)tm * tn * batch;
    int SK = 1;
    constexpr int sk_tiles = 128, sk_target = 192;       // split K below this many tiles / towards this many workgroups
    if (tiles < sk_tiles && G.work) {
        SK = (int)((sk_target + tiles - 1) / tiles);
        const int max_by_k = G.K / (4 * BK) > 0 ? G.K / (4 * BK) : 1;
        if (SK > max_by_k) SK = max_by_k;
        if (SK > 16) SK = 16;
        while (SK > 1 && (long long)batch * SK * G.M * G.N > work_cap) --SK;
    } else if (tiles < 2 * sk_tiles && G.K >= 64 * BK && G.work) {
        // about one workgroup per CU and a long K (ContentVec's positional conv on an utterance: 128 tiles x 192 k steps): the
        // k-loop is latency-bound at one workgroup per CU, 183 us; three or four per CU overlap their round trips
        SK = (int)((4 * sk_tiles + tiles - 1) / tiles);
        if (SK > G.K / (16 * BK)) SK = G.K / (16 * BK);
        while (SK > 1 && (long long)batch * SK * G.M * G.N > work_cap) --SK;
    }
    G.SK = SK;
    dim3 grid(tn, tm, batch * SK);
    hipLaunchKernelGGL(k_gemm_f32, grid, dim3(256), 0, s, G);
    GVC_LAUNCH_CHECK();
    if (SK > 1) {
        const long long mn = (long long)G.M * G.N;
        int gx = (int)((mn + 255) / 256);
        if (gx > 2048) gx = 2048;
        hipLaunchKernelGGL(k_splitk_epilogue, dim3(gx, 1, batch), dim3(256), 0, s, G);
        GVC_LAUNCH_CHECK();
    }
    return GVC_OK;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// a lane's four weight elements of a 16x16 FM16 block: requested as stored (WB = 1: bf16, 8 bytes), widened where the MFMA
// consumes them (a conversion next to the load would serialise the requests)
template <int WB> struct WRaw { typedef float4 T; };
template <> struct WRaw<1> { typedef uint2 T; };
__device__ __forceinline__ float4 w_f4(const float4& r) { return r; }
__device__ __forceinline__ float4 w_f4(const uint2& u) {
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xffff0000u));
}

template <int MT, int NW, int WB = 0, int NC = 0>          // NC > 0: A is merged from NC attention chunk partials (G.att_part)
__global__ __launch_bounds__(NW * 64) void k_gemm_skinny(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) float red_raw[];     // [NW][MT][256]
    float (*red)[MT][256] = reinterpret_cast<float (*)[MT][256]>(red_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kslice = G.K / G.SK;                 // per workgroup
    const int kw = kslice / NW;                    // per wave, multiple of 16
    const int kb16 = (blockIdx.y * kslice + wave * kw) >> 4;       // first 16-wide k block of this wave
    const int K16 = G.K >> 4;
    typedef typename WRaw<WB>::T wraw_t;
    // FM16 operands: a k step = +256 elements
    const wraw_t* wp = reinterpret_cast<const wraw_t*>(G.Wt) + ((size_t)(n0 >> 4) * K16 + kb16) * 64 + lane;
    const float* ap[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) ap[t] = G.A + ((size_t)t * K16 + kb16) * 256 + lane * 4;
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
    // what the epilogue of THIS thread will need besides the sums (see gemm_prefetch): same element mapping as below
    EpiPre pre;
    {
        int pm, pn;
        bool four = false;
        if constexpr (MT <= 2) {
            pm = 16 * (tid >> 8) + 4 * (lane >> 4) + ((tid >> 6) & 3); pn = n0 + (lane & 15);
        } else {
            const bool pfm = G.SK == 1 && G.e.c_fm16;
            const int pt = tid >> 6, pj = tid & 63;
            pm = 16 * pt + (pfm ? (pj & 15) : (pj >> 2)); pn = n0 + 4 * (pfm ? (pj >> 4) : (pj & 3));
            four = true;
        }
        if (G.SK == 1) pre = gemm_prefetch(G, min(pm, G.M - 1), pn, four);
    }
    // all fragment loads of a batch are issued before the first MFMA; a batch is sized to ~32 float4 per lane
    // (the whole K range of a wave for the GenVC shapes), so a wave pays one memory round trip
    constexpr int U = NC > 0 ? (32 / (MT * NC) >= 8 ? 8 : 32 / (MT * NC)) : (32 / (1 + MT) >= 8 ? 8 : (32 / (1 + MT) >= 4 ? 4 : 2));
    // NC > 0: the wave's K range lies inside one head (att_hd % kw == 0), so the merge weights exp(m_c - M) / L of a row are
    // computed once per wave and M tile; a fragment is sum_c weight_c * o_c
    float wn[MT][NC > 0 ? NC : 1], mc[MT][NC > 0 ? NC : 1], lc[MT][NC > 0 ? NC : 1];
    const float* pp[MT];
    if constexpr (NC > 0) {
        const int kg = blockIdx.y * kslice + wave * kw;              // first k of this wave
        const int h = kg / G.att_hd, pstride = G.att_hd + 4;
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int row = min(t * 16 + r, G.M - 1);
            pp[t] = G.att_part + ((size_t)(row * G.att_heads + h) * NC) * pstride;
#pragma unroll
            for (int c = 0; c < NC; ++c) { mc[t][c] = pp[t][c * pstride + G.att_hd]; lc[t][c] = pp[t][c * pstride + G.att_hd + 1]; }
            pp[t] += (kg - h * G.att_hd) + 4 * g;                    // this lane's four dims of k-block 0
        }
    }
    if constexpr (NC == 0 && MT >= 4) {
        // 4+ M tiles: a whole K range does not fit one batch of requests.  Two-k-block batches, double-buffered: the next batch
        // is requested before the MFMAs of the current one are issued, so the memory round trips hide under the matrix work
        constexpr int UB = 2;
        wraw_t w0[UB], w1[UB];
        float4 a0[UB][MT], a1[UB][MT];
        auto request = [&](int ks, wraw_t (&wr)[UB], float4 (&a4)[UB][MT]) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int k = min(ks + 16 * u, kw - 16);           // (kw is a multiple of 32 on every caller's path; clamped anyway)
                wr[u] = wp[k * 4];
#pragma unroll
                for (int t = 0; t < MT; ++t) a4[u][t] = *reinterpret_cast<const float4*>(ap[t] + k * 16);
            }
        };
        auto multiply = [&](int ks, const wraw_t (&wr)[UB], const float4 (&a4)[UB][MT]) {
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                float4 w4 = w_f4(wr[u]);
                if (ks + 16 * u >= kw) w4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].x, w4.x, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].y, w4.y, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].z, w4.z, acc[t], 0, 0, 0);
#pragma unroll
                for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].w, w4.w, acc[t], 0, 0, 0);
            }
        };
        // (sched_barrier: left alone, the scheduler sinks every load to just above its first use -- least registers, no overlap)
        request(0, w0, a0);
        __builtin_amdgcn_sched_barrier(0);
        for (int ks = 0; ks < kw; ks += 32 * UB) {
            request(ks + 16 * UB, w1, a1);             // (past the end: the last block again, multiplied by zero weights)
            __builtin_amdgcn_sched_barrier(0);
            multiply(ks, w0, a0);
            __builtin_amdgcn_sched_barrier(0);
            request(ks + 32 * UB, w0, a0);
            __builtin_amdgcn_sched_barrier(0);
            multiply(ks + 16 * UB, w1, a1);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else
    for (int ks = 0; ks < kw; ks += 16 * U) {
        wraw_t wr[U];
        float4 a4[U][MT];
        float4 oc[NC > 0 ? U : 1][MT][NC > 0 ? NC : 1];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = ks + 16 * u;
            const bool kin = k < kw;                   // rows past M hold stale data: their results are never stored
            wr[u] = kin ? wp[k * 4] : wraw_t{};
            if constexpr (NC > 0) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        oc[u][t][c] = *reinterpret_cast<const float4*>(pp[t] + c * (G.att_hd + 4) + (kin ? k : 0));
            } else {
#pragma unroll
                for (int t = 0; t < MT; ++t)
                    a4[u][t] = kin ? *reinterpret_cast<const float4*>(ap[t] + k * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        if constexpr (NC > 0) {
            if (ks == 0) {           // (after the first batch of requests has gone out: the (m, l) pairs arrive ahead of it)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    float mx = mc[t][0];
#pragma unroll
                    for (int c = 1; c < NC; ++c) mx = fmaxf(mx, mc[t][c]);
                    float L = 0.f;
#pragma unroll
                    for (int c = 0; c < NC; ++c) { wn[t][c] = __expf(mc[t][c] - mx); L += wn[t][c] * lc[t][c]; }
                    const float inv = 1.0f / L;
#pragma unroll
                    for (int c = 0; c < NC; ++c) wn[t][c] *= inv;
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (ks + 16 * u < kw) {
#pragma unroll
                        for (int c = 0; c < NC; ++c) {
                            o.x += wn[t][c] * oc[u][t][c].x; o.y += wn[t][c] * oc[u][t][c].y;
                            o.z += wn[t][c] * oc[u][t][c].z; o.w += wn[t][c] * oc[u][t][c].w;
                        }
                    }
                    a4[u][t] = o;
                }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float4 w4 = w_f4(wr[u]);
            // consecutive MFMAs go to different accumulators (40-cycle dependent latency vs 32-cycle issue)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].x, w4.x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].y, w4.y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].z, w4.z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].w, w4.w, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) red[wave][t][q * 64 + lane] = acc[t][q];
    __syncthreads();
    // a thread finishes FOUR consecutive columns of one row (the four accumulator slots sit next to each other in `red`) and
    // stores them as one float4: a row-major destination gets the tile as 16 rows x 64 bytes, an FM16 destination as its
    // contiguous 1 KiB block (4-byte stores in 16-64 byte pieces are what the write path is slowest at).  Per element the waves'
    // partial sums are still added in wave order.
    if constexpr (MT <= 2) {    // one or two tiles: a thread per element (512 threads) finishes sooner than 64-128 threads with a float4 each
        const int q = (tid >> 6) & 3;
        const int n = n0 + (lane & 15);
        for (int t = tid >> 8; t < MT; t += NW / 4) {
            const int m = 16 * t + 4 * (lane >> 4) + q;
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red[w][t][q * 64 + lane];
            if (G.SK == 1 && G.e.geglu) {       // (x_j, gate_j) sit in neighbouring lanes: the even lane stores gelu_erf(gate_j) * x_j
                v += pre.bias.x;
                const float other = __shfl_xor(v, 1);
                if (m < G.M && !(lane & 1)) G.C[fm16_index(m, n >> 1, G.N >> 1)] = gelu_erf(other) * v;
                continue;
            }
            if (m < G.M) {
                if (G.SK > 1) G.work[((size_t)blockIdx.y * G.M + m) * G.N + n] = v;
                else gemm_store_pre(G, m, n, v, pre);
            }
        }
        return;
    }
    const bool fm = G.SK == 1 && G.e.c_fm16;
    for (int item = tid; item < MT * 64; item += NW * 64) {
        const int t = item >> 6, j = item & 63;
        const int rr = fm ? (j & 15) : (j >> 2), k4 = fm ? (j >> 4) : (j & 3);
        const int idx = (rr & 3) * 64 + (rr >> 2) * 16 + 4 * k4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float4 p = *reinterpret_cast<const float4*>(&red[w][t][idx]);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        const int m = 16 * t + rr, n = n0 + 4 * k4;
        if (m < G.M) {
            if (G.SK > 1) *reinterpret_cast<float4*>(G.work + ((size_t)blockIdx.y * G.M + m) * G.N + n) = v;
            else gemm_store4_pre(G, m, n, v, pre);
        }
    }
}

// one workgroup (256 threads) per row: every operand of the row is requested in one round trip
__global__ void k_to_fm16(const float* src, float* dst, int N, int K) {
    const size_t n4 = (size_t)N * K / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / (K / 4)), k = (int)(i % (K / 4)) * 4;
        *reinterpret_cast<float4*>(dst + fm16_index(n, k, K)) = *reinterpret_cast<const float4*>(src + (size_t)n * K + k);
    }
}

// the same permutation into bf16 storage; src holds values that are already bf16-representable (k_round_bf16 ran on it)
__global__ void k_to_fm16_bf16(const float* src, unsigned short* dst, int N, int K) {
    const size_t n4 = (size_t)N * K / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const int n = (int)(i / (K / 4)), k = (int)(i % (K / 4)) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + (size_t)n * K + k);
        ushort4 o;
        o.x = (unsigned short)(__float_as_uint(v.x) >> 16); o.y = (unsigned short)(__float_as_uint(v.y) >> 16);
        o.z = (unsigned short)(__float_as_uint(v.z) >> 16); o.w = (unsigned short)(__float_as_uint(v.w) >> 16);
        *reinterpret_cast<ushort4*>(dst + fm16_index(n, k, K)) = o;
    }
}

// ---- row completion + LayerNorm, ONE WAVE PER ROW (the order every path shares, fused or not) ----
// lane L owns the float4s at k = (i*64 + L)*4, i < NV (d = 256*NV).  v = x + bias + part[0] + ... + part[SK-1] (in that
// order); mean and variance by a per-lane sum over i followed by wave_sum.
template <int NV> struct RowOps { float4 x[NV], b[NV], p[NV][4]; };

// the loads of a row completion, and the sums over them.  PART = false: the row is x alone.  SK4 = true: the four partial
// planes are requested by row_issue; otherwise row_finish reads the SK planes itself.
template <int NV, bool PART, bool SK4>
__device__ __forceinline__ void row_issue(const float* xr, const float* part, size_t prow, size_t pstride, const float* bias, int lane,
                                          RowOps<NV>& r) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (i * 64 + lane) * 4;
        r.x[i] = *reinterpret_cast<const float4*>(xr + k);
        if (PART) {
            r.b[i] = *reinterpret_cast<const float4*>(bias + k);
            if (SK4) {
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) r.p[i][sidx] = *reinterpret_cast<const float4*>(part + sidx * pstride + prow + k);
            }
        }
    }
}

template <int NV, bool PART, bool SK4>
__device__ __forceinline__ void row_finish(const RowOps<NV>& r, const float* part, size_t prow, size_t pstride, int SK, int lane,
                                           float4 (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        v[i] = r.x[i];
        if (PART) {
            v[i].x += r.b[i].x; v[i].y += r.b[i].y; v[i].z += r.b[i].z; v[i].w += r.b[i].w;
            if (SK4) {
#pragma unroll
                for (int sidx = 0; sidx < 4; ++sidx) { v[i].x += r.p[i][sidx].x; v[i].y += r.p[i][sidx].y; v[i].z += r.p[i][sidx].z; v[i].w += r.p[i][sidx].w; }
            } else {
                // up to 8 planes (the strip GEMM's K split), requested together: a loop with a run-time trip count made every
                // plane a round trip of its own; added in plane order
                const int k = (i * 64 + lane) * 4;
                float4 q4[8];
#pragma unroll
                for (int sidx = 0; sidx < 8; ++sidx)
                    q4[sidx] = *reinterpret_cast<const float4*>(part + min(sidx, SK - 1) * pstride + prow + k);       // (clamped, not predicated: branches would serialise the requests)
#pragma unroll
                for (int sidx = 0; sidx < 8; ++sidx)
                    if (sidx < SK) { v[i].x += q4[sidx].x; v[i].y += q4[sidx].y; v[i].z += q4[sidx].z; v[i].w += q4[sidx].w; }
            }
        }
    }
}

// SK = 4: the skinny path's split (wave-uniform branches); other splits (<= 8 planes) come from the strip GEMM
template <int NV>
__device__ __forceinline__ void row_sum(const float* xr, const float* part, size_t prow, size_t pstride, int SK, const float* bias,
                                        int lane, float4 (&v)[NV]) {
    RowOps<NV> r;
    if (!part) {
        row_issue<NV, false, false>(xr, part, prow, pstride, bias, lane, r);
        row_finish<NV, false, false>(r, part, prow, pstride, SK, lane, v);
    } else if (SK == 4) {
        row_issue<NV, true, true>(xr, part, prow, pstride, bias, lane, r);
        row_finish<NV, true, true>(r, part, prow, pstride, SK, lane, v);
    } else {
        row_issue<NV, true, false>(xr, part, prow, pstride, bias, lane, r);
        row_finish<NV, true, false>(r, part, prow, pstride, SK, lane, v);
    }
}

template <int NV>
__device__ __forceinline__ void row_ln(const float4 (&v)[NV], const float* ln_w, const float* ln_b, int lane, float4 (&o)[NV]) {
    constexpr float inv_d = 1.0f / (float)(256 * NV);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    const float mean = wave_sum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
    const float rstd = 1.0f / sqrtf(wave_sum(q) * inv_d + 1e-5f);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (i * 64 + lane) * 4;
        const float4 gw = *reinterpret_cast<const float4*>(ln_w + k);
        const float4 gb = *reinterpret_cast<const float4*>(ln_b + k);
        o[i].x = (v[i].x - mean) * rstd * gw.x + gb.x; o[i].y = (v[i].y - mean) * rstd * gw.y + gb.y;
        o[i].z = (v[i].z - mean) * rstd * gw.z + gb.z; o[i].w = (v[i].w - mean) * rstd * gw.w + gb.w;
    }
}

// x_out[row] = x_in[row] + bias + sum_s part[s][row];  a[row] = LayerNorm(x_out[row]) (skipped when ln_w is null).
// grid = rows, 64 threads: one wave per row.  x_out may be x_in.
template <int NV>
__global__ __launch_bounds__(64) void k_ln_sum_rows_t(const float* x_in, float* x_out, float* a, const float* part, int SK,
                                                       const float* bias, int rows, const float* ln_w, const float* ln_b, int a_fm16) {
    constexpr int d = 256 * NV;
    const int lane = threadIdx.x, row = blockIdx.x;
    float4 v[NV], o[NV];
    row_sum<NV>(x_in + (size_t)row * d, part, (size_t)row * d, (size_t)rows * d, SK, bias, lane, v);
#pragma unroll
    for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(x_out + (size_t)row * d + (i * 64 + lane) * 4) = v[i];
    if (!ln_w) return;
    row_ln<NV>(v, ln_w, ln_b, lane, o);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (i * 64 + lane) * 4;
        if (a_fm16) *reinterpret_cast<float4*>(a + fm16_index(row, k, d)) = o[i];
        else *reinterpret_cast<float4*>(a + (size_t)row * d + k) = o[i];
    }
}

int launch_ln_sum_rows(const float* x_in, float* x_out, float* a, const float* part, int SK, const float* bias, int rows, int d,
                       const float* ln_w, const float* ln_b, int a_fm16, hipStream_t s) {
    GVC_REQUIRE(d % 256 == 0 && d >= 256 && d <= 2048 && SK >= 0 && SK <= 8, GVC_ERR_UNSUPPORTED, "ln_sum_rows: d=%d SK=%d unsupported", d, SK);
#define GVC_LN_SUM(nv)                                                                                                                      \
    case nv:                                                                                                                                \
        hipLaunchKernelGGL(k_ln_sum_rows_t<nv>, dim3(rows), dim3(64), 0, s, x_in, x_out, a, part, SK, bias, rows, ln_w, ln_b, a_fm16);       \
        break;
    switch (d / 256) {
        GVC_LN_SUM(1) GVC_LN_SUM(2) GVC_LN_SUM(3) GVC_LN_SUM(4) GVC_LN_SUM(5) GVC_LN_SUM(6) GVC_LN_SUM(7) GVC_LN_SUM(8)
    }
#undef GVC_LN_SUM
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

// ---- skinny GEMM for <= 16 rows with the row completion + LayerNorm in its prologue (K = d) ----
// A = LayerNorm(x_in + bias + sum part) is built by the workgroup itself (wave w: rows w and w + 8) and staged in LDS in
// the FM16 fragment order; workgroup 0 also writes the completed rows to x_out (a buffer other than x_in: the other
// workgroups are still reading x_in).  Everything else as k_gemm_skinny<1, 8>: same per-element summation order, so the
// result is bit-identical to k_ln_sum_rows_t followed by k_gemm_skinny.

// one completed row: kept in x_out by workgroup 0, normalised, staged in LDS in the FM16 fragment order (zeros past the last row)
template <int NV>
__device__ __forceinline__ void skinny_ln_row(const LnFuse& P, bool has, int row, const float4 (&v)[NV], bool keep, int lane, float* a_lds) {
    constexpr int K = 256 * NV;
    float4 o[NV];
    if (has) {
        if (P.x_out && keep) {
#pragma unroll
            for (int i = 0; i < NV; ++i) *reinterpret_cast<float4*>(P.x_out + (size_t)row * K + (i * 64 + lane) * 4) = v[i];
        }
        row_ln<NV>(v, P.ln_w, P.ln_b, lane, o);
    } else {
#pragma unroll
        for (int i = 0; i < NV; ++i) o[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int k = (i * 64 + lane) * 4;
        *reinterpret_cast<float4*>(a_lds + (size_t)(k >> 4) * 256 + (((row & 15) + 16 * ((k & 15) >> 2)) << 2)) = o[i];
    }
}

// request order: row `wave`'s operands (L2), the weight fragments (HBM), then row `wave + 8`'s operands while the first
// row is normalised: the weight fetch is in flight under the whole prologue.  The compiler barriers pin that order.
// (rows past the last one read the last row again and stage zeros: straight-line code, nothing lands in scratch)
template <int NV, bool PART, typename wraw_t, int U>
__device__ __forceinline__ void skinny_ln_prologue(const LnFuse& P, const wraw_t* wp, wraw_t (&w4)[U], int wave, int lane, bool keep, float* a_lds) {
    constexpr int K = 256 * NV;
    const bool has0 = wave < P.rows, has1 = wave + 8 < P.rows;
    const int r0 = has0 ? wave : P.rows - 1, r1 = has1 ? wave + 8 : r0;
    const size_t ps = (size_t)P.rows * K;
    RowOps<NV> ops0, ops1;
    float4 v0[NV], v1[NV];
    row_issue<NV, PART, true>(P.x_in + (size_t)r0 * K, P.part, (size_t)r0 * K, ps, P.pbias, lane, ops0);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < U; ++u) w4[u] = wp[u * 64];
    asm volatile("" ::: "memory");
    row_finish<NV, PART, true>(ops0, P.part, (size_t)r0 * K, ps, 4, lane, v0);
    if (P.rows > 8) row_issue<NV, PART, true>(P.x_in + (size_t)r1 * K, P.part, (size_t)r1 * K, ps, P.pbias, lane, ops1);
    asm volatile("" ::: "memory");
    skinny_ln_row<NV>(P, has0, wave, v0, keep, lane, a_lds);
    if (P.rows > 8) row_finish<NV, PART, true>(ops1, P.part, (size_t)r1 * K, ps, 4, lane, v1);
    skinny_ln_row<NV>(P, has1, wave + 8, v1, keep, lane, a_lds);
}

template <int NV, int WB = 0>
__global__ __launch_bounds__(512) void k_gemm_skinny_ln(const GemmArgs G, const LnFuse P) {
    constexpr int K = 256 * NV, K16 = K / 16, NW = 8, kw = K / NW;           // kw = 128 (d = 1024) or 32 (d = 256)
    constexpr int U = kw / 16;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* a_lds = sm;                                 // one 16-row M tile in FM16 order: [K16][256]
    float (*red)[256] = reinterpret_cast<float (*)[256]>(sm + 16 * K);       // [NW][256]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n0 = blockIdx.x * 16;
    typedef typename WRaw<WB>::T wraw_t;
    wraw_t w4[U];
    const wraw_t* wp = reinterpret_cast<const wraw_t*>(G.Wt) + ((size_t)(n0 >> 4) * K16 + (wave * kw >> 4)) * 64 + lane;
    EpiPre pre;                 // bias / slot / cached length of the element this thread stores at the end (gemm_prefetch)
    if (tid < 256) pre = gemm_prefetch(G, min(4 * (lane >> 4) + (tid >> 6), G.M - 1), n0 + (lane & 15), false);
    if (P.part) skinny_ln_prologue<NV, true, wraw_t, U>(P, wp, w4, wave, lane, blockIdx.x == 0, a_lds);
    else skinny_ln_prologue<NV, false, wraw_t, U>(P, wp, w4, wave, lane, blockIdx.x == 0, a_lds);
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const float* ap = a_lds + (size_t)(wave * kw >> 4) * 256 + lane * 4;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const float4 a4 = *reinterpret_cast<const float4*>(ap + u * 256);
        const float4 wv = w_f4(w4[u]);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.x, wv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.y, wv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.z, wv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a4.w, wv.w, acc, 0, 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[wave][q * 64 + lane] = acc[q];
    __syncthreads();
    if (tid < 256) {            // (one 16x16 tile: 256 threads with one element each finish sooner than 64 with a float4)
        const int q = tid >> 6;
        const int m = 4 * (lane >> 4) + q, n = n0 + (lane & 15);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][q * 64 + lane];
        if (m < G.M) gemm_store_pre(G, m, n, v, pre);
    }
}

int launch_gemm_skinny_ln(GemmArgs G, const LnFuse& P, hipStream_t s) {
    GVC_REQUIRE(G.M >= 1 && G.M <= 16 && P.rows == G.M && G.N % 16 == 0 && G.K % 256 == 0 && G.K >= 256 && G.K <= 1024 && (!P.part || P.SK == 4) &&
                    G.ldc % 4 == 0 && (!G.e.qkv || (G.e.d % 16 == 0 && G.e.head_dim % 4 == 0)), GVC_ERR_ARG,
                "skinny gemm + LN: unsupported shape M=%d N=%d K=%d", G.M, G.N, G.K);
    G.SK = 1;
    const size_t lds = ((size_t)16 * G.K + 8 * 256) * sizeof(float);
#define GVC_SKINNY_LN(nv)                                                                                                 \
    if (G.K == 256 * nv) {                                                                                                \
        if (G.w_bf16) hipLaunchKernelGGL((k_gemm_skinny_ln<nv, 1>), dim3(G.N / 16), dim3(512), lds, s, G, P);             \
        else hipLaunchKernelGGL((k_gemm_skinny_ln<nv, 0>), dim3(G.N / 16), dim3(512), lds, s, G, P);                      \
    }
    GVC_SKINNY_LN(1) GVC_SKINNY_LN(2) GVC_SKINNY_LN(3) GVC_SKINNY_LN(4)
#undef GVC_SKINNY_LN
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}


int launch_gemm_skinny(GemmArgs G, int SK, long long work_cap, hipStream_t s) {
    const int MT = cdiv(G.M, 16);
    GVC_REQUIRE(MT >= 1 && MT <= 8 && G.N % 16 == 0 && SK >= 1 && SK <= 8 && G.K % (SK * 64) == 0, GVC_ERR_ARG,
                "skinny gemm: unsupported shape M=%d N=%d K=%d SK=%d", G.M, G.N, G.K, SK);
    GVC_REQUIRE(SK == 1 || (G.work && (long long)SK * G.M * G.N <= work_cap), GVC_ERR_ARG, "skinny gemm: work buffer too small");
    GVC_REQUIRE(G.conv_cin == 0 && G.a_act == 0 && G.K % 16 == 0, GVC_ERR_ARG, "skinny gemm: FM16 operands only");
    GVC_REQUIRE(G.ldc % 4 == 0 && (!G.e.resid || G.e.ldr % 4 == 0) && (!G.e.qkv || (G.e.d % 16 == 0 && G.e.head_dim % 4 == 0)), GVC_ERR_ARG,
                "skinny gemm: rows must be 16-byte aligned (ldc=%d)", G.ldc);
    G.SK = SK;
    dim3 grid(G.N / 16, SK);
    const bool w8 = (G.K / SK) % 128 == 0;           // 8 waves when every wave still gets whole 16-wide k steps
    const size_t lds = (size_t)(w8 ? 8 : 4) * MT * 256 * sizeof(float);
    if (G.att_part) {            // A merged from the batched decode attention's chunk partials
        GVC_REQUIRE(w8 && MT <= 2 && (G.att_nc == 2 || G.att_nc == 4) && G.att_heads * G.att_hd == G.K && G.att_hd % (G.K / SK / 8) == 0 &&
                        G.att_hd % 4 == 0, GVC_ERR_ARG, "skinny gemm: attention-partial operand needs <= 32 rows, 2 or 4 chunks (M=%d chunks=%d)", G.M, G.att_nc);
#define GVC_SKINNY_ATT(mt, nc)                                                                                   \
        if (MT == mt && G.att_nc == nc) {                                                                          \
            if (G.w_bf16) hipLaunchKernelGGL((k_gemm_skinny<mt, 8, 1, nc>), grid, dim3(512), lds, s, G);           \
            else hipLaunchKernelGGL((k_gemm_skinny<mt, 8, 0, nc>), grid, dim3(512), lds, s, G);                    \
        }
        GVC_SKINNY_ATT(1, 2) GVC_SKINNY_ATT(1, 4) GVC_SKINNY_ATT(2, 2) GVC_SKINNY_ATT(2, 4)
#undef GVC_SKINNY_ATT
        GVC_LAUNCH_CHECK();
        return GVC_OK;
    }
#define GVC_SKINNY(mt)                                                                               \
    case mt:                                                                                         \
        if (w8 && G.w_bf16) hipLaunchKernelGGL((k_gemm_skinny<mt, 8, 1>), grid, dim3(512), lds, s, G); \
        else if (w8) hipLaunchKernelGGL((k_gemm_skinny<mt, 8, 0>), grid, dim3(512), lds, s, G);       \
        else if (G.w_bf16) hipLaunchKernelGGL((k_gemm_skinny<mt, 4, 1>), grid, dim3(256), lds, s, G); \
        else hipLaunchKernelGGL((k_gemm_skinny<mt, 4, 0>), grid, dim3(256), lds, s, G);               \
        break;
    switch (MT) { GVC_SKINNY(1) GVC_SKINNY(2) GVC_SKINNY(3) GVC_SKINNY(4) GVC_SKINNY(5) GVC_SKINNY(6) GVC_SKINNY(7) GVC_SKINNY(8) }
#undef GVC_SKINNY
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

// ---- strip GEMM for 17 .. a few thousand rows (batched prefill, latent re-pass): see gemm.h ----
// A workgroup = 4 waves = one (m group, 64-column n block, K split).  The m group's A tiles are staged ONCE per workgroup in
// LDS by LDS-DMA (a 16x16 FM16 block is 1 KiB contiguous = one global_load_lds of 16 bytes per lane), in stages of KC k-blocks,
// two stage buffers; every wave streams the weights of its own 16 columns straight from global memory into MFMA operands
// (one float4 per lane per k-block, requested a stage ahead) and keeps MTW independent 16x16 accumulators, so an A fragment
// read from LDS feeds 4 MFMAs and a weight fragment 4*MTW.  Per k-block and wave: 1 global float4, MTW ds_read_b128,
// 4*MTW v_mfma_f32_16x16x4_f32 (32 cycles each): the loop is bound by the MFMA pipe.
struct StripGeom { int MG, NB, SK, mt, kb, raw; };    // raw: partial planes go to G.work even when SK == 1    // m groups, n blocks, K splits, m tiles, k blocks of the whole problem

template <int MTW> struct StripCfg {
    static constexpr int KC = MTW >= 5 ? 4 : 8;                      // k-blocks per stage: <= 36 KiB of A per stage
    static constexpr size_t stage_bytes = (size_t)2 * KC * MTW * 1024, tile_bytes = (size_t)MTW * 16 * 68 * 4;
    static constexpr size_t lds_bytes = stage_bytes > tile_bytes ? stage_bytes : tile_bytes;
};

// ds_read_b128 of the MTW fragments of one k-block of a stage (tile t at byte t * 1024 from `addr`), and the wait that
// orders their consumers behind the data (the registers pass through the asm so that nothing is scheduled across it)
template <int MTW, int T = 0>
__device__ __forceinline__ void strip_read(f32x4 (&a)[MTW], unsigned addr) {
    if constexpr (T < MTW) {
        asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(a[T]) : "v"(addr), "n"(T * 1024));
        strip_read<MTW, T + 1>(a, addr);
    }
}
template <int MTW>
__device__ __forceinline__ void strip_wait(f32x4 (&a)[MTW]) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int t = 0; t < MTW; ++t) asm volatile("" : "+v"(a[t]));
}

// KH = 2: eight waves; waves 4..7 take the upper half of every stage's k-blocks for the same four column tiles, so each SIMD
// has two waves to issue MFMAs from (one wave alone reaches 89 % of the MFMA rate and stalls on its own LDS waits); the two
// halves' accumulators are added in the epilogue tile (lower half + upper half, a fixed order).
template <int MTW, int WB, int KH>
__global__ __launch_bounds__(256 * KH) void k_gemm_strip(const GemmArgs G, const StripGeom S) {
    constexpr int KC = StripCfg<MTW>::KC, KCW = KC / KH, NWV = 4 * KH;
    extern __shared__ __attribute__((aligned(16))) float strip_lds[];        // [2][KC][MTW][256]
    typedef typename WRaw<WB>::T wraw_t;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, kh = wave >> 2;
    int id = blockIdx.x;
    const int nb = id % S.NB;       // n block fastest: the m groups / K splits of one n block share an XCD (NB % 8 == 0) and its L2
    id /= S.NB;
    const int mg = id % S.MG, sk = id / S.MG;
    const int t_lo = (int)((long long)mg * S.mt / S.MG), nt = (int)((long long)(mg + 1) * S.mt / S.MG) - t_lo;     // <= MTW
    const int kb_lo = (int)((long long)sk * S.kb / S.SK), nkb = (int)((long long)(sk + 1) * S.kb / S.SK) - kb_lo;
    const int n_tile = nb * 4 + wn;
    const wraw_t* wp = reinterpret_cast<const wraw_t*>(G.Wt) + ((size_t)n_tile * S.kb + kb_lo) * 64 + lane;
    const float* abase = G.A + ((size_t)t_lo * S.kb + kb_lo) * 256 + lane * 4;
    const int nst = (nkb + KC - 1) / KC;
    const unsigned lds_base = (unsigned)(size_t)strip_lds + (unsigned)lane * 16u;       // LDS byte address of this lane's float4 in block 0

    // this wave's share of a stage's A blocks: pairs p = wave + NWV i -> (t = p / KC, kc = p % KC)
    auto stage_a = [&](int st, int buf) {
        constexpr int NP = (MTW * KC + NWV - 1) / NWV;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pidx = wave + NWV * i;
            const int t = pidx / KC, kc = pidx - t * KC;
            const int kbi = min(st * KC + kc, nkb - 1);            // (past the end: a valid block again, met by zero weights)
            if (pidx < MTW * KC && t < nt)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(abase + ((size_t)t * S.kb + kbi) * 256),
                                                 (__attribute__((address_space(3))) void*)(strip_lds + ((size_t)(buf * KC + kc) * MTW + t) * 256),
                                                 16, 0, 0);
        }
    };
    wraw_t wnext[KCW], wcur[KCW];
    auto stage_w = [&](int st) {
#pragma unroll
        for (int j = 0; j < KCW; ++j)
            wnext[j] = wp[(size_t)min(st * KC + kh * KCW + j, nkb - 1) * 64];        // (past the end: a valid block again, zeroed at use)
    };
    f32x4 acc[MTW];
#pragma unroll
    for (int t = 0; t < MTW; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};

    stage_a(0, 0);
    stage_w(0);
    for (int st = 0; st < nst; ++st) {
        const int buf = st & 1;
        // k-blocks past the end get zero weights, so that the k loop has no tail case (a branch around MFMAs makes the
        // accumulators commute between AGPRs and VGPRs); the select sits here, not at the load, which it would wait for
#pragma unroll
        for (int j = 0; j < KCW; ++j) wcur[j] = st * KC + kh * KCW + j < nkb ? wnext[j] : wraw_t{};
        // everything this wave requested has landed (stage st of A, in LDS, included); past the barrier that holds for all
        // waves, and every wave is done reading the other buffer
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // the A fragments are read with ds_read_b128 issued from inline asm: a ds_read the compiler can see makes it drain vmcnt
        // first (the LDS-DMA of the NEXT stage might alias), which serialises the stage's memory latency with its MFMAs.
        // Fragments of the next k-block are requested before the MFMAs of the current one are issued.
        const unsigned ab = lds_base + (unsigned)((buf * KC + kh * KCW) * MTW * 1024);
        f32x4 af[2][MTW];
        strip_read<MTW>(af[0], ab);
#pragma unroll
        for (int j = 0; j < KCW; ++j) {
            strip_wait<MTW>(af[j & 1]);
            if (j + 1 < KCW) strip_read<MTW>(af[(j + 1) & 1], ab + (unsigned)((j + 1) * MTW * 1024));
            const float4 wv = w_f4(wcur[j]);
#pragma unroll
            for (int t = 0; t < MTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j & 1][t][0], wv.x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j & 1][t][1], wv.y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j & 1][t][2], wv.z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j & 1][t][3], wv.w, acc[t], 0, 0, 0);
            // the next stage's requests go out behind the first k-block's MFMAs: their issue time is off the critical path
            if (j == 0 && st + 1 < nst) {
                stage_a(st + 1, buf ^ 1);
                stage_w(st + 1);
            }
        }
    }
    // Epilogue through LDS: a lane holds rows 4*(lane>>4) + q, column lane & 15 of each 16x16 tile -- 4-byte stores in 64-byte
    // row pieces, which the write path takes at under 1 TB/s.  The accumulators are laid out as the workgroup's
    // [16*MTW rows][64 columns] tile in LDS and go out as float4: rows of 256 contiguous bytes (four rows per wave store), or,
    // for a fragment-major destination, a wave's 16x16 tile as one contiguous 1 KiB block.
    constexpr int LDT = 68;
    __syncthreads();                    // every wave is done with the stage buffers
    {
        float* T = strip_lds;
        if (kh == 0) {
#pragma unroll
            for (int t = 0; t < MTW; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) T[(t * 16 + 4 * (lane >> 4) + q) * LDT + wn * 16 + (lane & 15)] = acc[t][q];
        }
        if (KH > 1) {
            __syncthreads();
            if (kh == 1) {
#pragma unroll
                for (int t = 0; t < MTW; ++t)
#pragma unroll
                    for (int q = 0; q < 4; ++q) T[(t * 16 + 4 * (lane >> 4) + q) * LDT + wn * 16 + (lane & 15)] += acc[t][q];
            }
        }
        __syncthreads();
        const bool part = S.SK > 1 || S.raw;
        const bool fm = !part && G.e.c_fm16;
        // row-major: wave w stores rows w, w + NWV/... of every tile; fragment-major: wave (wn, kh) stores column tile wn of the
        // tiles t with t % KH == kh
        const int row = fm ? (lane & 15) : (wave * 4 + (lane >> 4)) & 15;
        const int col = fm ? wn * 16 + 4 * (lane >> 4) : 4 * (lane & 15);
        const int n = nb * 64 + col;
        const int t0 = fm ? kh : (wave * 4) >> 4;               // first tile of this wave's share, stride KH
#pragma unroll
        for (int tt = 0; tt < (MTW + KH - 1) / KH; ++tt) {
            const int t = t0 + tt * KH;
            const int m = (t_lo + t) * 16 + row;
            if (t < nt && m < G.M) {
                const float4 v = *reinterpret_cast<const float4*>(T + (t * 16 + row) * LDT + col);
                if (part) *reinterpret_cast<float4*>(G.work + ((size_t)sk * G.M + m) * G.N + n) = v;
                else gemm_store4(G, m, n, v);
            }
        }
    }
}

// geometry: the (tiles per group, K split) pair with the shortest modelled critical path, in units of one (m tile, k block) =
// 4 MFMAs = 128 cycles: a CU runs its workgroups' MFMA work back to back, and every round of resident workgroups pays one
// pipeline fill.  GVC_STRIP="MTW,SK" overrides (measurement).
int launch_gemm_strip(GemmArgs G, int sk_max, long long work_cap, int raw_partials, int* sk_used, hipStream_t s) {
    GVC_REQUIRE(G.M >= 1 && G.N % 64 == 0 && G.K % 16 == 0 && G.conv_cin == 0 && G.a_act == 0 && G.ldc % 4 == 0 &&
                    (!G.e.resid || G.e.ldr % 4 == 0) && (!G.e.qkv || (G.e.d % 64 == 0 && G.e.head_dim % 4 == 0)), GVC_ERR_ARG,
                "strip gemm: unsupported shape M=%d N=%d K=%d (FM16 operands, N %% 64 == 0, 16-byte rows)", G.M, G.N, G.K);
    StripGeom S;
    S.mt = cdiv(G.M, 16); S.NB = G.N / 64; S.kb = G.K / 16;
    GVC_REQUIRE(!raw_partials || (G.work && (long long)G.M * G.N <= work_cap), GVC_ERR_ARG, "strip gemm: raw partials need a work buffer");
    if (!G.work) sk_max = 1;
    if (sk_max < 1) sk_max = 1;
    if (sk_max > 8) sk_max = 8;
    int best_w = 0, best_sk = 1;
    double best = 1e30;
    constexpr double fill = 40.0;                        // the cost model's pipeline-fill term
    for (int w = 1; w <= 9; ++w) {
        const int MG = cdiv(S.mt, w);
        if (cdiv(S.mt, MG) != w) continue;              // the balanced partition's largest group: only exact fits are candidates
        const int wpc = 2;                               // 72 KiB of LDS at most: two workgroups per CU
        for (int sk = 1; sk <= sk_max; ++sk) {
            if (S.kb / sk < 4) break;
            if (sk > 1 && (long long)sk * G.M * G.N > work_cap) break;
            const long long wgs = (long long)MG * S.NB * sk;
            const double serial = (double)cdiv((int)wgs, 256) * w * cdiv(S.kb, sk);
            const double cost = serial + fill * cdiv((int)wgs, 256 * wpc) + (sk > 1 && !raw_partials ? 30.0 : 0.0);
            if (cost < best) { best = cost; best_w = w; best_sk = sk; }
        }
    }
    GVC_REQUIRE(best_w > 0, GVC_ERR_ARG, "strip gemm: no geometry for M=%d N=%d K=%d", G.M, G.N, G.K);
    S.MG = cdiv(S.mt, best_w); S.SK = best_sk;
    S.raw = raw_partials && G.work ? 1 : 0;
    G.SK = best_sk;
    if (sk_used) *sk_used = best_sk;
    const dim3 grid(S.MG * S.NB * S.SK);
#define GVC_STRIP(w)                                                                                                         \
    case w:                                                                                                                  \
        if (G.w_bf16) hipLaunchKernelGGL((k_gemm_strip<w, 1, 2>), grid, dim3(512), StripCfg<w>::lds_bytes, s, G, S);              \
        else hipLaunchKernelGGL((k_gemm_strip<w, 0, 2>), grid, dim3(512), StripCfg<w>::lds_bytes, s, G, S);                       \
        break;
    switch (best_w) { GVC_STRIP(1) GVC_STRIP(2) GVC_STRIP(3) GVC_STRIP(4) GVC_STRIP(5) GVC_STRIP(6) GVC_STRIP(7) GVC_STRIP(8) GVC_STRIP(9) }
#undef GVC_STRIP
    GVC_LAUNCH_CHECK();
    if (best_sk > 1 && !raw_partials) {
        const long long mn = (long long)G.M * G.N;
        int gx = (int)((mn + 255) / 256);
        if (gx > 2048) gx = 2048;
        hipLaunchKernelGGL(k_splitk_epilogue, dim3(gx, 1, 1), dim3(256), 0, s, G);
        GVC_LAUNCH_CHECK();
    }
    return GVC_OK;
}

void gemm_init_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny_ln<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny_ln<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
#define GVC_STRIP_ATTR(w)                                                                                                                      \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_strip<w, 0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);     \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_strip<w, 1, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    GVC_STRIP_ATTR(1) GVC_STRIP_ATTR(2) GVC_STRIP_ATTR(3) GVC_STRIP_ATTR(4) GVC_STRIP_ATTR(5) GVC_STRIP_ATTR(6) GVC_STRIP_ATTR(7) GVC_STRIP_ATTR(8) GVC_STRIP_ATTR(9)
#undef GVC_STRIP_ATTR
}

}  // namespace gvc

// Measurement / test hook (include/genvc_hip.h): one GEMM C = A W^T (+ bias) through a chosen kernel on row-major operands.
extern "C" int gvc_gemm_probe(int32_t variant, const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N,
                              int32_t K, int32_t sk_max, int32_t iters, float* avg_us, gvc_stream sv) {
    using namespace gvc;
    GVC_REQUIRE(A && W && C && M >= 1 && N >= 16 && K >= 16 && K % 16 == 0 && N % 16 == 0 && variant >= 0 && variant <= 2, GVC_ERR_ARG,
                "gemm_probe: bad argument");
    hipStream_t s = (hipStream_t)sv;
    static bool attrs = false;
    if (!attrs) { gemm_init_attributes(); attrs = true; }
    const int Mp = (M + 15) & ~15;
    float *Af = nullptr, *Wf = nullptr, *work = nullptr;
    const long long work_cap = (long long)8 * M * N;
    struct Scratch {            // freed on every return path
        float **a, **b, **c;
        ~Scratch() { for (float** p : {a, b, c}) if (*p) (void)hipFree(*p); }
    } scratch{&Af, &Wf, &work};
    GVC_CHECK_HIP(hipMalloc((void**)&work, (size_t)work_cap * sizeof(float)));
    GemmArgs G;
    memset(&G, 0, sizeof(G));
    G.C = C; G.ldc = N; G.M = M; G.N = N; G.K = K; G.work = work; G.e.bias = bias;
    if (variant == 0) {
        G.A = A; G.lda = K; G.Wt = W; G.ldw = K;
    } else {
        GVC_CHECK_HIP(hipMalloc((void**)&Af, (size_t)Mp * K * sizeof(float)));
        GVC_CHECK_HIP(hipMalloc((void**)&Wf, (size_t)N * K * sizeof(float)));
        GVC_CHECK_HIP(hipMemsetAsync(Af, 0, (size_t)Mp * K * sizeof(float), s));
        hipLaunchKernelGGL(k_to_fm16, dim3(1024), dim3(256), 0, s, A, Af, M, K);
        hipLaunchKernelGGL(k_to_fm16, dim3(1024), dim3(256), 0, s, W, Wf, N, K);
        G.A = Af; G.lda = K; G.Wt = Wf; G.ldw = K;
    }
    auto once = [&]() -> int {
        if (variant == 0) return launch_gemm_cap(G, 1, work_cap, s);
        if (variant == 1) return launch_gemm_strip(G, sk_max, work_cap, 0, nullptr, s);
        return launch_gemm_skinny(G, 1, work_cap, s);
    };
    int rc = once();
    if (rc == GVC_OK && iters > 0 && avg_us) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, s);
        for (int i = 0; i < iters && rc == GVC_OK; ++i) rc = once();
        hipEventRecord(e1, s);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        *avg_us = ms * 1000.f / (float)iters;
        hipEventDestroy(e0); hipEventDestroy(e1);
    }
    (void)hipStreamSynchronize(s);
    return rc;
}
