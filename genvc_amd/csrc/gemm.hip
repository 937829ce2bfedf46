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

template <int MT, int NW>
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
    const float* wp = G.Wt + ((size_t)(n0 >> 4) * K16 + kb16) * 256 + lane * 4;      // FM16 operands: a k step = +256 floats
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
        float4 w4[U], a4[U][MT];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = ks + 16 * u;
            const bool kin = k < kw;                   // rows past M hold stale data: their results are never stored
            w4[u] = kin ? *reinterpret_cast<const float4*>(wp + k * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int t = 0; t < MT; ++t)
                a4[u][t] = kin ? *reinterpret_cast<const float4*>(ap[t] + k * 16) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            // consecutive MFMAs go to different accumulators (40-cycle dependent latency vs 32-cycle issue)
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].x, w4[u].x, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].y, w4[u].y, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].z, w4[u].z, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < MT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[u][t].w, w4[u].w, acc[t], 0, 0, 0);
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

__global__ __launch_bounds__(256) void k_ln_sum_rows(float* x, float* a, const float* part, int SK, const float* bias,
                                                     int rows, int d, const float* ln_w, const float* ln_b, int a_fm16) {
    __shared__ float red[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    float* xr = x + (size_t)row * d;
    const size_t pstride = (size_t)rows * d;
    constexpr int MAXV = 4;                          // d <= 4096
    float4 v[MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int k = (i * 256 + tid) * 4;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (k < d) {
            v[i] = *reinterpret_cast<const float4*>(xr + k);
            const float4 b4 = *reinterpret_cast<const float4*>(bias + k);
            float4 p4[8];
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx)
                p4[sidx] = sidx < SK ? *reinterpret_cast<const float4*>(part + sidx * pstride + (size_t)row * d + k)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
            v[i].x += b4.x; v[i].y += b4.y; v[i].z += b4.z; v[i].w += b4.w;
#pragma unroll
            for (int sidx = 0; sidx < 8; ++sidx) { v[i].x += p4[sidx].x; v[i].y += p4[sidx].y; v[i].z += p4[sidx].z; v[i].w += p4[sidx].w; }
            *reinterpret_cast<float4*>(xr + k) = v[i];
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        }
    }
    if (!ln_w) return;
    const float inv_d = 1.0f / (float)d;
    const float mean = block4_sum(s, red) * inv_d;
    float qv = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int k = (i * 256 + tid) * 4;
        if (k < d) {
            const float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
            qv += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
        }
    }
    const float rstd = 1.0f / sqrtf(block4_sum(qv, red) * inv_d + 1e-5f);
    float* ar = a + (size_t)row * d;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int k = (i * 256 + tid) * 4;
        if (k < d) {
            const float4 gw = *reinterpret_cast<const float4*>(ln_w + k);
            const float4 gb = *reinterpret_cast<const float4*>(ln_b + k);
            float4 o;
            o.x = (v[i].x - mean) * rstd * gw.x + gb.x; o.y = (v[i].y - mean) * rstd * gw.y + gb.y;
            o.z = (v[i].z - mean) * rstd * gw.z + gb.z; o.w = (v[i].w - mean) * rstd * gw.w + gb.w;
            if (a_fm16) *reinterpret_cast<float4*>(a + fm16_index(row, k, d)) = o;
            else *reinterpret_cast<float4*>(ar + k) = o;
        }
    }
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
        if (w8) hipLaunchKernelGGL((k_gemm_skinny<mt, 8>), grid, dim3(512), lds, s, G);              \
        else hipLaunchKernelGGL((k_gemm_skinny<mt, 4>), grid, dim3(256), lds, s, G);                 \
        break;
    switch (MT) { GVC_SKINNY(1) GVC_SKINNY(2) GVC_SKINNY(3) GVC_SKINNY(4) GVC_SKINNY(5) GVC_SKINNY(6) GVC_SKINNY(7) GVC_SKINNY(8) }
#undef GVC_SKINNY
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

}  // namespace gvc
