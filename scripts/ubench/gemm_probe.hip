// What bounds the 64x64x32 fp32-MFMA GEMM tile loop?  M=880 N=3072 K=1024 (the batched-prefill QKV shape), variants:
//   0 full kernel                          1 no global loads after the first tile (LDS tiles reused)
//   2 as 1, and no LDS stores / barriers   3 as 2, and no LDS reads (operands stay in registers): pure MFMA issue
//   4 full kernel with 16x16x4 MFMAs (4 independent accumulators per wave)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BM = 64, BN = 64, BK = 32, LDL = 36;

template <int V>
__global__ __launch_bounds__(256) void k_gemm(const float* A, const float* W, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDL];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nt = K / BK;
    float4 ra[2], rb[2];
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            const int k = kt * BK + c4;
            const int m = m0 + r, n = n0 + r;
            ra[j] = m < M ? *reinterpret_cast<const float4*>(A + (size_t)m * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            rb[j] = n < N ? *reinterpret_cast<const float4*>(W + (size_t)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
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
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 acc4[4];
    for (int i = 0; i < 4; ++i) acc4[i] = {0.f, 0.f, 0.f, 0.f};
    load_tile(0);
    store_tile(0);
    __syncthreads();
    int buf = 0;
    const int arow = (wm * 32 + (lane & 31)) * LDL + (lane >> 5) * 4;
    const int brow = (wn * 32 + (lane & 31)) * LDL + (lane >> 5) * 4;
    // 16x16x4 operand addressing: lane (r = lane % 16, g = lane / 16) reads float4 at column 4g of a 16-wide k block
    const int r16 = lane & 15, g16 = lane >> 4;
    float4 ka[4], kb[4];
    for (int kk = 0; kk < 4; ++kk) {
        ka[kk] = *reinterpret_cast<const float4*>(&As[0][arow + kk * 8]);
        kb[kk] = *reinterpret_cast<const float4*>(&Bs[0][brow + kk * 8]);
    }
    for (int kt = 0; kt < nt; ++kt) {
        if (V == 0 || V == 4) { if (kt + 1 < nt) load_tile(kt + 1); }
        if (V == 4) {
#pragma unroll
            for (int kb16 = 0; kb16 < 2; ++kb16) {
                float4 a0 = *reinterpret_cast<const float4*>(&As[buf][(wm * 32 + r16) * LDL + kb16 * 16 + 4 * g16]);
                float4 a1 = *reinterpret_cast<const float4*>(&As[buf][(wm * 32 + 16 + r16) * LDL + kb16 * 16 + 4 * g16]);
                float4 b0 = *reinterpret_cast<const float4*>(&Bs[buf][(wn * 32 + r16) * LDL + kb16 * 16 + 4 * g16]);
                float4 b1 = *reinterpret_cast<const float4*>(&Bs[buf][(wn * 32 + 16 + r16) * LDL + kb16 * 16 + 4 * g16]);
#define M4(c, x, y) acc4[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, acc4[c], 0, 0, 0)
                M4(0, a0.x, b0.x); M4(1, a0.x, b1.x); M4(2, a1.x, b0.x); M4(3, a1.x, b1.x);
                M4(0, a0.y, b0.y); M4(1, a0.y, b1.y); M4(2, a1.y, b0.y); M4(3, a1.y, b1.y);
                M4(0, a0.z, b0.z); M4(1, a0.z, b1.z); M4(2, a1.z, b0.z); M4(3, a1.z, b1.z);
                M4(0, a0.w, b0.w); M4(1, a0.w, b1.w); M4(2, a1.w, b0.w); M4(3, a1.w, b1.w);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                float4 a4 = ka[kk], b4 = kb[kk];
                if (V != 3) {
                    a4 = *reinterpret_cast<const float4*>(&As[buf][arow + kk * 8]);
                    b4 = *reinterpret_cast<const float4*>(&Bs[buf][brow + kk * 8]);
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
        }
        if (V == 0 || V == 4) { if (kt + 1 < nt) store_tile(buf ^ 1); }
        if (V == 1) store_tile(buf ^ 1);
        if (V <= 1 || V == 4) { __syncthreads(); buf ^= 1; }
    }
    const int n = n0 + wn * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = acc[r];
        if (V == 4) v = acc4[r & 3][r >> 2];
        if (m < M && n < N) C[(size_t)m * N + n] = v;
    }
}

// two register tiles in flight (loads issued two k-steps ahead), MFMA flavour selectable
template <int M16>
__global__ __launch_bounds__(256) void k_gemm_deep(const float* A, const float* W, float* C, int M, int N, int K) {
    __shared__ __attribute__((aligned(16))) float As[2][BM * LDL];
    __shared__ __attribute__((aligned(16))) float Bs[2][BN * LDL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
    const int nt = K / BK;
    float4 ra[2][2], rb[2][2];
    auto load_tile = [&](int kt, float4 (&qa)[2], float4 (&qb)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            const int k = kt * BK + c4;
            const int m = m0 + r, n = n0 + r;
            qa[j] = m < M ? *reinterpret_cast<const float4*>(A + (size_t)m * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
            qb[j] = n < N ? *reinterpret_cast<const float4*>(W + (size_t)n * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tile = [&](int buf, const float4 (&qa)[2], const float4 (&qb)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 256 * j;
            const int r = idx >> 3, c4 = (idx & 7) * 4;
            *reinterpret_cast<float4*>(&As[buf][r * LDL + c4]) = qa[j];
            *reinterpret_cast<float4*>(&Bs[buf][r * LDL + c4]) = qb[j];
        }
    };
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    f32x4 acc4[4];
    for (int i = 0; i < 4; ++i) acc4[i] = {0.f, 0.f, 0.f, 0.f};
    const int arow = (wm * 32 + (lane & 31)) * LDL + (lane >> 5) * 4;
    const int brow = (wn * 32 + (lane & 31)) * LDL + (lane >> 5) * 4;
    const int r16 = lane & 15, g16 = lane >> 4;
    auto compute = [&](int buf) {
        if (M16) {
#pragma unroll
            for (int kb16 = 0; kb16 < 2; ++kb16) {
                float4 a0 = *reinterpret_cast<const float4*>(&As[buf][(wm * 32 + r16) * LDL + kb16 * 16 + 4 * g16]);
                float4 a1 = *reinterpret_cast<const float4*>(&As[buf][(wm * 32 + 16 + r16) * LDL + kb16 * 16 + 4 * g16]);
                float4 b0 = *reinterpret_cast<const float4*>(&Bs[buf][(wn * 32 + r16) * LDL + kb16 * 16 + 4 * g16]);
                float4 b1 = *reinterpret_cast<const float4*>(&Bs[buf][(wn * 32 + 16 + r16) * LDL + kb16 * 16 + 4 * g16]);
                M4(0, a0.x, b0.x); M4(1, a0.x, b1.x); M4(2, a1.x, b0.x); M4(3, a1.x, b1.x);
                M4(0, a0.y, b0.y); M4(1, a0.y, b1.y); M4(2, a1.y, b0.y); M4(3, a1.y, b1.y);
                M4(0, a0.z, b0.z); M4(1, a0.z, b1.z); M4(2, a1.z, b0.z); M4(3, a1.z, b1.z);
                M4(0, a0.w, b0.w); M4(1, a0.w, b1.w); M4(2, a1.w, b0.w); M4(3, a1.w, b1.w);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const float4 a4 = *reinterpret_cast<const float4*>(&As[buf][arow + kk * 8]);
                const float4 b4 = *reinterpret_cast<const float4*>(&Bs[buf][brow + kk * 8]);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.x, b4.x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.y, b4.y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.z, b4.z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a4.w, b4.w, acc, 0, 0, 0);
            }
        }
    };
    load_tile(0, ra[0], rb[0]);
    if (nt > 1) load_tile(1, ra[1], rb[1]);
    store_tile(0, ra[0], rb[0]);
    __syncthreads();
    for (int kt = 0; kt < nt; kt += 2) {
        if (kt + 2 < nt) load_tile(kt + 2, ra[0], rb[0]);
        compute(0);
        if (kt + 1 < nt) store_tile(1, ra[1], rb[1]);
        __syncthreads();
        if (kt + 1 >= nt) break;
        if (kt + 3 < nt) load_tile(kt + 3, ra[1], rb[1]);
        compute(1);
        if (kt + 2 < nt) store_tile(0, ra[0], rb[0]);
        __syncthreads();
    }
    const int n = n0 + wn * 32 + (lane & 31);
    for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        float v = M16 ? acc4[r & 3][r >> 2] : acc[r];
        if (m < M && n < N) C[(size_t)m * N + n] = v;
    }
}

template <int V>
static void run(const char* name, const float* A, const float* W, float* C, int M, int N, int K, hipStream_t s) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(k_gemm<V>, grid, dim3(256), 0, s, A, W, C, M, N, K);
    CK(hipEventRecord(e0, s));
    const int reps = 20;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k_gemm<V>, grid, dim3(256), 0, s, A, W, C, M, N, K);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000 / reps, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
    printf("%-62s %7.1f us  %6.1f TFLOP/s (%4.1f %% of 157.3)\n", name, us, tf, tf / 157.3 * 100);
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int M : {880, 4096}) {
        const int N = 3072, K = 1024;
        float *A, *W, *C;
        CK(hipMalloc(&A, (size_t)M * K * 4)); CK(hipMalloc(&W, (size_t)N * K * 4)); CK(hipMalloc(&C, (size_t)M * N * 4));
        {   // pseudo-random operands (all-zero operands run the MFMA pipes cooler and faster than real data)
            size_t na = (size_t)M * K, nw = (size_t)N * K;
            float* h = (float*)malloc((na > nw ? na : nw) * 4);
            unsigned x = 12345u;
            for (size_t i = 0; i < na; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
            CK(hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice));
            for (size_t i = 0; i < nw; ++i) { x = x * 1664525u + 1013904223u; h[i] = ((x >> 8) & 0xffff) / 65536.0f - 0.5f; }
            CK(hipMemcpy(W, h, nw * 4, hipMemcpyHostToDevice));
            free(h);
        }
        printf("M=%d N=%d K=%d (%d workgroups)\n", M, N, K, ((N + 63) / 64) * ((M + 63) / 64));
        run<0>("0 full kernel (32x32x2)", A, W, C, M, N, K, s);
        run<1>("1 no global loads after tile 0", A, W, C, M, N, K, s);
        run<2>("2 + no LDS stores / barriers", A, W, C, M, N, K, s);
        run<3>("3 + no LDS reads: pure MFMA issue", A, W, C, M, N, K, s);
        run<4>("4 full kernel with 16x16x4 MFMAs, 4 accumulators", A, W, C, M, N, K, s);
        for (int m16 = 0; m16 < 2; ++m16) {
            hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
            dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM);
            for (int i = 0; i < 3; ++i) { if (m16) hipLaunchKernelGGL(k_gemm_deep<1>, grid, dim3(256), 0, s, A, W, C, M, N, K); else hipLaunchKernelGGL(k_gemm_deep<0>, grid, dim3(256), 0, s, A, W, C, M, N, K); }
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < 20; ++i) { if (m16) hipLaunchKernelGGL(k_gemm_deep<1>, grid, dim3(256), 0, s, A, W, C, M, N, K); else hipLaunchKernelGGL(k_gemm_deep<0>, grid, dim3(256), 0, s, A, W, C, M, N, K); }
            CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1000 / 20, tf = 2.0 * M * N * K / (us * 1e-6) / 1e12;
            printf("%-62s %7.1f us  %6.1f TFLOP/s (%4.1f %% of 157.3)\n", m16 ? "6 two register tiles in flight, 16x16x4" : "5 two register tiles in flight, 32x32x2", us, tf, tf / 157.3 * 100);
        }
        CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(C));
    }
    return 0;
}
