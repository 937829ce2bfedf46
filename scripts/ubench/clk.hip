#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(unsigned long long* out, int spin, int sleepy) {
    unsigned long long c0 = clock64(), w0 = wall_clock64();
    float a = threadIdx.x;
    if (sleepy) { for (int i = 0; i < spin; ++i) __builtin_amdgcn_s_sleep(8); }
    else { for (int i = 0; i < spin * 64; ++i) a = a * 1.0001f + 0.5f; }
    unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (unsigned long long)a; }
}
int main() {
    unsigned long long* d; hipMalloc(&d, 64); unsigned long long h[3];
    for (int sleepy = 0; sleepy < 2; ++sleepy) for (int grid : {1, 256}) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, 20000, sleepy); hipDeviceSynchronize();
        hipLaunchKernelGGL(k, dim3(grid), dim3(64), 0, 0, d, 20000, sleepy); hipDeviceSynchronize();
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("sleepy=%d grid=%d: clock64 %llu wall(100MHz) %llu -> clock64 rate %.1f MHz, %.1f us\n", sleepy, grid, h[0], h[1], 100.0 * h[0] / h[1], h[1] / 100.0);
    }
    return 0;
}
