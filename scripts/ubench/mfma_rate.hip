// fp32 MFMA issue rate: NACC independent accumulators per wave, WPS waves per SIMD, every CU busy; no memory traffic.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ void k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = {0.f, 0.f, 0.f, 0.f};
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ void k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    float a = threadIdx.x * 0.001f, b = threadIdx.x * 0.002f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <typename F>
static void run(const char* name, F launch, double flops_per_wave_iter, int threads, int grid, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    launch(iters); hipDeviceSynchronize();
    hipEventRecord(e0); launch(iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves = (double)grid * threads / 64;
    printf("%-28s grid %4d x %3d threads: %8.1f us  %6.1f TFLOP/s\n", name, grid, threads, ms * 1e3, waves * iters * flops_per_wave_iter / (ms * 1e-3) / 1e12);
}
int main() {
    float* d; hipMalloc(&d, 1 << 24);
    const int it = 20000;
    for (int threads : {256, 512}) for (int grid : {256, 512}) {
        run("16x16x4 x1 acc", [&](int n) { hipLaunchKernelGGL(k16<1>, dim3(grid), dim3(threads), 0, 0, d, n); }, 1 * 2048.0, threads, grid, it);
        run("16x16x4 x4 acc", [&](int n) { hipLaunchKernelGGL(k16<4>, dim3(grid), dim3(threads), 0, 0, d, n); }, 4 * 2048.0, threads, grid, it);
        run("16x16x4 x9 acc", [&](int n) { hipLaunchKernelGGL(k16<9>, dim3(grid), dim3(threads), 0, 0, d, n); }, 9 * 2048.0, threads, grid, it);
        run("32x32x2 x1 acc", [&](int n) { hipLaunchKernelGGL(k32<1>, dim3(grid), dim3(threads), 0, 0, d, n); }, 1 * 4096.0, threads, grid, it);
        run("32x32x2 x4 acc", [&](int n) { hipLaunchKernelGGL(k32<4>, dim3(grid), dim3(threads), 0, 0, d, n); }, 4 * 4096.0, threads, grid, it);
    }
    return 0;
}
