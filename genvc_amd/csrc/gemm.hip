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
            rb[j] = (n < G.N && k < kend) ? *reinterpret_cast<const float4*>(G.Wt + (size_t)n * G.ldw + k)
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
    if (tiles < 128 && G.work) {
        SK = (int)((192 + tiles - 1) / tiles);
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

}  // namespace gvc
