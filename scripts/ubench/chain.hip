// Floor of a dependent launch chain replayed from a hipGraph on MI355X.
//  mode 0: empty kernels      mode 1: every wave reads the 4 KB vector written by the previous launch, reduces it,
//  and one lane per wave writes one float of the next vector (the all-to-all of a decode GEMV without the weights)
//  mode 2: mode 1 + each wave streams 4 KB (or 16 KB) of private weights (nt loads)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__device__ __forceinline__ float wsum(float v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

template <int MODE, int NI>
__global__ __launch_bounds__(1024) void k(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ w, int n) {
    if (MODE == 0) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * (blockDim.x >> 6) + wave;
    typedef float f4 __attribute__((ext_vector_type(4)));
    f4 wv[NI];
    if (MODE == 2) {
        const f4* wp = reinterpret_cast<const f4*>(w + (size_t)row * NI * 256) + lane;
#pragma unroll
        for (int i = 0; i < NI; ++i) wv[i] = __builtin_nontemporal_load(wp + i * 64);
    }
    float s = 0.f;
    for (int i = lane * 4; i < 1024; i += 256) {
        const float4 v = *reinterpret_cast<const float4*>(in + i);
        s += v.x + v.y + v.z + v.w;
    }
    if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < NI; ++i) s += wv[i].x + wv[i].w;
    }
    s = wsum(s);
    if (lane == 0 && row < n) out[row] = s * 1e-3f;
}

template <int MODE, int NI>
void run(const char* name, int grid, int block, int nk, float* a, float* b, float* w, size_t wstride, hipStream_t s) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < nk; ++i)
        hipLaunchKernelGGL((k<MODE, NI>), dim3(grid), dim3(block), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, w + (size_t)(i % 60) * wstride, 1024);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, s));
        for (int i = 0; i < 20; ++i) CK(hipGraphLaunch(ge, s));
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("%-44s grid %4d x %4d thr : %6.2f us per launch\n", name, grid, block, ms * 1000.0 / (20.0 * nk));
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    float *a, *b, *w;
    CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
    const size_t wstride = (size_t)4096 * 4096;           // floats per operand slot (64 MB)
    CK(hipMalloc(&w, 60 * wstride * sizeof(float))); CK(hipMemset(w, 0, 60 * wstride * sizeof(float)));
    run<0, 1>("empty", 256, 256, 150, a, b, w, wstride, s);
    run<0, 1>("empty", 256, 1024, 150, a, b, w, wstride, s);
    run<1, 1>("4KB all-to-all (read prev, write 1/wave)", 256, 256, 150, a, b, w, wstride, s);
    run<1, 1>("4KB all-to-all", 256, 768, 150, a, b, w, wstride, s);
    run<2, 4>("all-to-all + 4KB weights/wave  (4 MB)", 256, 256, 150, a, b, w, wstride, s);
    run<2, 4>("all-to-all + 4KB weights/wave (12 MB)", 256, 768, 150, a, b, w, wstride, s);
    run<2, 4>("all-to-all + 4KB weights/wave (16 MB)", 512, 512, 150, a, b, w, wstride, s);
    run<2, 16>("all-to-all + 16KB weights/wave (16 MB)", 256, 256, 150, a, b, w, wstride, s);
    run<2, 16>("all-to-all + 16KB weights/wave (16 MB)", 128, 512, 150, a, b, w, wstride, s);
    return 0;
}
