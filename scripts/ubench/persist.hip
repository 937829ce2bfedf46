// Persistent-kernel seam cost on MI355X: 256 workgroups x 16 waves stay resident; per phase every wave
// streams NI*1 KiB... of private weights (prefetched one phase ahead), produces ONE output of an N-vector as an
// 8-byte {tag,value} granule (relaxed agent-scope atomic store), then the workgroup gathers the whole N-vector
// (each wave polls its share of the granules with relaxed agent-scope loads), stages it in LDS, and goes on.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

// N = 4096: 1 output per wave (256*16 waves).  granules[buf][N]
template <int NI>
__global__ __launch_bounds__(1024) void k_persist(u64* gran, const float* w, size_t wstride, int phases, float* out, int* timeout) {
    __shared__ float vec[4096];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 16 + wave;                 // 0..4095
    f4 wv[NI], wn[NI];
    const f4* wp = reinterpret_cast<const f4*>(w + (size_t)item * NI * 256) + lane;
#pragma unroll
    for (int i = 0; i < NI; ++i) wv[i] = __builtin_nontemporal_load(wp + i * 64);
    for (int i = threadIdx.x; i < 4096; i += 1024) vec[i] = 1.0f;
    __syncthreads();
    float last = 0.f;
    for (int p = 1; p <= phases; ++p) {
        // prefetch next phase's weights (different 64 MB slab)
        const f4* wq = reinterpret_cast<const f4*>(w + (size_t)(p % 30) * wstride + (size_t)item * NI * 256) + lane;
#pragma unroll
        for (int i = 0; i < NI; ++i) wn[i] = __builtin_nontemporal_load(wq + i * 64);
        // "GEMV": dot of this wave's weights with the staged vector segment
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(&vec[(i * 256 + lane * 4) & 4095]);
            s += wv[i].x * a.x + wv[i].y * a.y + wv[i].z * a.z + wv[i].w * a.w;
        }
        s = wsum(s) * 1e-3f + 1.0f;
        u64* g = gran + (size_t)(p & 1) * 4096;
        if (lane == 0) __hip_atomic_store(g + item, ((u64)(unsigned)p << 32) | (u64)__float_as_uint(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // gather: wave `wave` owns granules [wave*256, wave*256+256): 4 per lane
        int spins = 0;
        for (int j = 0; j < 4; ++j) {
            const int idx = wave * 256 + j * 64 + lane;
            u64 x;
            while (true) {
                x = __hip_atomic_load(g + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all((unsigned)(x >> 32) == (unsigned)p)) break;
                if (++spins > 2000000) { if (lane == 0) *timeout = p; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            vec[idx] = __uint_as_float((unsigned)x);
        }
        __syncthreads();
        last = s;
#pragma unroll
        for (int i = 0; i < NI; ++i) wv[i] = wn[i];
    }
    if (lane == 0) out[item] = last;
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    u64* gran; float *w, *out; int* tmo;
    CK(hipMalloc(&gran, 2 * 4096 * 8)); CK(hipMemset(gran, 0, 2 * 4096 * 8));
    const size_t wstride = (size_t)4096 * 4096;
    CK(hipMalloc(&w, 30 * wstride * sizeof(float))); CK(hipMemset(w, 0, 30 * wstride * sizeof(float)));
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&tmo, 4)); CK(hipMemset(tmo, 0, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int ni : {1, 4}) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(gran, 0, 2 * 4096 * 8, s));
            const int phases = 120;
            CK(hipEventRecord(e0, s));
            if (ni == 1) hipLaunchKernelGGL(k_persist<1>, dim3(256), dim3(1024), 0, s, gran, w, wstride, phases, out, tmo);
            else hipLaunchKernelGGL(k_persist<4>, dim3(256), dim3(1024), 0, s, gran, w, wstride, phases, out, tmo);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
            float o0; CK(hipMemcpy(&o0, out, 4, hipMemcpyDeviceToHost));
            printf("NI=%d (%2d MB weights/phase): %6.2f us per phase (4096-granule all-to-all)  timeout=%d out=%f\n", ni, ni * 4, ms * 1000 / phases, t, o0);
        }
    }
    return 0;
}
