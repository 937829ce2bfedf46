// Does data fetched by launch N stay hot for launch N+1 (same workgroup -> chunk mapping)?
// Times a streaming-read kernel over a buffer of `mb` MiB, launched back to back:
//   mode 0: same buffer every launch (reuse possible)    mode 1: rotate over 64 buffers (no reuse)
//   shift s: launch i maps workgroup b to chunk (b + s*i) % nblocks  (breaks XCD affinity when s % 8 != 0)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ buf, size_t chunk4, int shift, float* out) {
    const int nb = gridDim.x;
    const int cb = (blockIdx.x + shift) % nb;
    const float4* p = buf + (size_t)cb * chunk4;
    float acc = 0.f;
    for (size_t i = threadIdx.x; i < chunk4; i += 256 * 4) {
        float4 a = p[i];
        float4 b = i + 256 < chunk4 ? p[i + 256] : make_float4(0, 0, 0, 0);
        float4 c = i + 512 < chunk4 ? p[i + 512] : make_float4(0, 0, 0, 0);
        float4 d = i + 768 < chunk4 ? p[i + 768] : make_float4(0, 0, 0, 0);
        acc += a.x + b.y + c.z + d.w;
    }
    if (acc == 123.456f) out[blockIdx.x] = acc;
}

int main(int argc, char** argv) {
    const int nblocks = 256 * 4;
    hipStream_t s; CK(hipStreamCreate(&s));
    float* out; CK(hipMalloc(&out, nblocks * 4));
    const int NB = 64;
    for (int mb : {4, 8, 16, 32, 64, 128}) {
        const size_t bytes = (size_t)mb << 20;
        std::vector<float4*> bufs(NB);
        for (auto& b : bufs) { CK(hipMalloc(&b, bytes)); CK(hipMemset(b, 1, bytes)); }
        const size_t chunk4 = bytes / 16 / nblocks;
        for (int mode = 0; mode < 2; ++mode)
            for (int shift : {0, 1, 8}) {
                if (mode == 1 && shift) continue;
                hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                const int iters = 200;
                for (int w = 0; w < 2; ++w) {
                    CK(hipEventRecord(e0, s));
                    for (int i = 0; i < iters; ++i)
                        hipLaunchKernelGGL(k_read, dim3(nblocks), dim3(256), 0, s, bufs[mode ? i % NB : 0], chunk4, shift * i, out);
                    CK(hipEventRecord(e1, s));
                    CK(hipStreamSynchronize(s));
                }
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                const double us = ms * 1000.0 / iters;
                printf("%4d MiB  %s shift %d : %7.2f us/launch  %7.2f TB/s\n", mb, mode ? "rotate64" : "same    ", shift, us, bytes / us / 1e6);
            }
        for (auto& b : bufs) CK(hipFree(b));
    }
    return 0;
}
