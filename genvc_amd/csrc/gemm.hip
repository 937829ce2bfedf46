// fp32 MFMA GEMM (see gemm.h)
#include "gemm.h"

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
    const float* Ab = G.A + batch * G.a_batch_stride;
    const float* Wb = G.Wt + batch * G.w_batch_stride;

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
            if (G.a_act == AACT_LRELU) {
                ra[j].x = ra[j].x > 0.f ? ra[j].x : ra[j].x * G.a_slope; ra[j].y = ra[j].y > 0.f ? ra[j].y : ra[j].y * G.a_slope;
                ra[j].z = ra[j].z > 0.f ? ra[j].z : ra[j].z * G.a_slope; ra[j].w = ra[j].w > 0.f ? ra[j].w : ra[j].w * G.a_slope;
            }
            rb[j] = (n < G.N && k < kend) ? *reinterpret_cast<const float4*>(Wb + (size_t)n * G.ldw + k)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
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
        float v = 0.f;
        for (int s = 0; s < G.SK; ++s) v += w[(size_t)s * mn];
        gemm_store(G, batch, (int)(i / G.N), (int)(i % G.N), v);
    }
}

int launch_gemm_cap(GemmArgs G, int batch, long long work_cap, hipStream_t s) {
    GVC_REQUIRE(G.K % 4 == 0 && G.lda % 4 == 0 && G.ldw % 4 == 0 && G.conv_cin % 4 == 0 && G.conv_tap_stride % 4 == 0,
                GVC_ERR_ARG, "gemm: K/lda/ldw/conv strides must be multiples of 4 (K=%d lda=%d ldw=%d)", G.K, G.lda, G.ldw);
    const int tm = cdiv(G.M, BM), tn = cdiv(G.N, BN);
    const long long tiles = (long long)tm * tn * batch;
    int SK = 1;
    static const int sk_tiles = getenv("GVC_GEMM_SK_TILES") ? atoi(getenv("GVC_GEMM_SK_TILES")) : 128;
    static const int sk_target = getenv("GVC_GEMM_SK_TARGET") ? atoi(getenv("GVC_GEMM_SK_TARGET")) : 192;
    if (tiles < sk_tiles && G.work) {
        SK = (int)((sk_target + tiles - 1) / tiles);
        const int max_by_k = G.K / (4 * BK) > 0 ? G.K / (4 * BK) : 1;
        if (SK > max_by_k) SK = max_by_k;
        if (SK > 16) SK = 16;
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

template <int MT, int NW, int WB = 0>
__global__ __launch_bounds__(NW * 64) void k_gemm_skinny(const GemmArgs G) {
    extern __shared__ __attribute__((aligned(16))) float red_raw[];     // [NW][MT][256]
    float (*red)[MT][256] = reinterpret_cast<float (*)[MT][256]>(red_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int kslice = G.K / G.SK;                 // per workgroup
    const int kw = kslice / NW;                    // per wave, multiple of 16
    (void)r; (void)g;
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
    // all fragment loads of a batch are issued before the first MFMA; a batch is sized to ~32 float4 per lane
    // (the whole K range of a wave for the GenVC shapes), so a wave pays one memory round trip
    constexpr int U = 32 / (1 + MT) >= 8 ? 8 : (32 / (1 + MT) >= 4 ? 4 : 2);
    for (int ks = 0; ks < kw; ks += 16 * U) {
        wraw_t wr[U];
        float4 a4[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = ks + 16 * u;
            const bool kin = k < kw;                   // rows past M hold stale data: their results are never stored
            wr[u] = kin ? wp[k * 4] : wraw_t{};
#pragma unroll
            for (int t = 0; t < MT; ++t)
                a4[u][t] = kin ? *reinterpret_cast<const float4*>(ap[t] + k * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    // thread (q, lane) finishes element row 16t + 4*(lane>>4) + q, column n0 + (lane & 15); tiles are shared out
    // over the workgroup's thread quads
    const int q = (tid >> 6) & 3;
    const int n = n0 + (lane & 15);
    for (int t = tid >> 8; t < MT; t += NW / 4) {
        const int m = 16 * t + 4 * (lane >> 4) + q;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][t][q * 64 + lane];
        if (m < G.M) {
            if (G.SK > 1) G.work[((size_t)blockIdx.y * G.M + m) * G.N + n] = v;
            else gemm_store(G, 0, m, n, v);
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
                const int k = (i * 64 + lane) * 4;
                for (int sidx = 0; sidx < SK; ++sidx) {
                    const float4 q4 = *reinterpret_cast<const float4*>(part + sidx * pstride + prow + k);
                    v[i].x += q4.x; v[i].y += q4.y; v[i].z += q4.z; v[i].w += q4.w;
                }
            }
        }
    }
}

// SK is 4 on every caller's path (wave-uniform branches); the general case loops
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
    GVC_REQUIRE(d % 256 == 0 && d >= 256 && d <= 1024 && SK >= 0 && SK <= 8, GVC_ERR_UNSUPPORTED, "ln_sum_rows: d=%d SK=%d unsupported", d, SK);
    if (d == 1024) hipLaunchKernelGGL(k_ln_sum_rows_t<4>, dim3(rows), dim3(64), 0, s, x_in, x_out, a, part, SK, bias, rows, ln_w, ln_b, a_fm16);
    else if (d == 768) hipLaunchKernelGGL(k_ln_sum_rows_t<3>, dim3(rows), dim3(64), 0, s, x_in, x_out, a, part, SK, bias, rows, ln_w, ln_b, a_fm16);
    else if (d == 512) hipLaunchKernelGGL(k_ln_sum_rows_t<2>, dim3(rows), dim3(64), 0, s, x_in, x_out, a, part, SK, bias, rows, ln_w, ln_b, a_fm16);
    else hipLaunchKernelGGL(k_ln_sum_rows_t<1>, dim3(rows), dim3(64), 0, s, x_in, x_out, a, part, SK, bias, rows, ln_w, ln_b, a_fm16);
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
    if (tid < 256) {
        const int q = tid >> 6;
        const int m = 4 * (lane >> 4) + q, n = n0 + (lane & 15);
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][q * 64 + lane];
        if (m < G.M) gemm_store(G, 0, m, n, v);
    }
}

int launch_gemm_skinny_ln(GemmArgs G, const LnFuse& P, hipStream_t s) {
    GVC_REQUIRE(G.M >= 1 && G.M <= 16 && P.rows == G.M && G.N % 16 == 0 && G.K % 256 == 0 && G.K >= 256 && G.K <= 1024 && (!P.part || P.SK == 4), GVC_ERR_ARG,
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

void gemm_init_attributes() {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny_ln<4, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_gemm_skinny_ln<4, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
}

int launch_gemm_skinny(GemmArgs G, int SK, long long work_cap, hipStream_t s) {
    const int MT = cdiv(G.M, 16);
    GVC_REQUIRE(MT >= 1 && MT <= 8 && G.N % 16 == 0 && SK >= 1 && SK <= 8 && G.K % (SK * 64) == 0, GVC_ERR_ARG,
                "skinny gemm: unsupported shape M=%d N=%d K=%d SK=%d", G.M, G.N, G.K, SK);
    GVC_REQUIRE(SK == 1 || (G.work && (long long)SK * G.M * G.N <= work_cap), GVC_ERR_ARG, "skinny gemm: work buffer too small");
    GVC_REQUIRE(G.conv_cin == 0 && G.a_act == 0 && G.K % 16 == 0, GVC_ERR_ARG, "skinny gemm: FM16 operands only");
    G.SK = SK;
    dim3 grid(G.N / 16, SK);
    const bool w8 = (G.K / SK) % 128 == 0;           // 8 waves when every wave still gets whole 16-wide k steps
    const size_t lds = (size_t)(w8 ? 8 : 4) * MT * 256 * sizeof(float);
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

}  // namespace gvc
