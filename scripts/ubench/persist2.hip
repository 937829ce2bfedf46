// Persistent-kernel seam, second design: ONE arrival counter per phase instead of per-element tags.
// 256 workgroups x 512 threads stay resident.  Per phase a workgroup
//   1. issues the non-temporal loads of the NEXT phase's private weights (NI float4 per thread),
//   2. "GEMV": dots its current weights with the N-vector staged in LDS, producing N/256 outputs,
//   3. stores them (agent-scope write-through), fences, and adds 1 to the phase counter,
//   4. one lane polls the counter until all 256 workgroups arrived, then the workgroup re-reads the fresh N-vector
//      (agent-scope loads) into LDS.
// MODE 0: counter barrier.   MODE 1: {tag,value} granules polled by the first N/64/4 waves (no counter).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

template <int NI, int MODE, int N>
__global__ __launch_bounds__(512) void k_persist2(float* vecs, u64* gran, unsigned* ctr, const float* w, size_t wstride, int phases,
                                                  float* out, int* timeout) {
    __shared__ float vec[N];
    __shared__ float part[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int OUT_PER_WG = N / 256;
    f4 wv[NI > 0 ? NI : 1], wn[NI > 0 ? NI : 1];
    const size_t toff = (size_t)blockIdx.x * 512 + tid;
#pragma unroll
    for (int i = 0; i < NI; ++i) wv[i] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(w) + toff + (size_t)i * 131072);
    for (int i = tid; i < N; i += 512) vec[i] = 1.0f;
    __syncthreads();
    float last = 0.f;
    for (int p = 1; p <= phases; ++p) {
        const f4* wq = reinterpret_cast<const f4*>(w + (size_t)(p % 30) * wstride) + toff;
#pragma unroll
        for (int i = 0; i < NI; ++i) wn[i] = __builtin_nontemporal_load(wq + (size_t)i * 131072);
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(&vec[((i * 512 + tid) * 4) & (N - 1)]);
            s += wv[i].x * a.x + wv[i].y * a.y + wv[i].z * a.z + wv[i].w * a.w;
        }
        if (NI == 0) s = vec[tid & (N - 1)];
        s = wsum(s);
        if (lane == 0) part[wave] = s;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) tot += part[i];
        tot = tot * 1e-6f + 1.0f;
        last = tot;
        if (MODE == 0) {
            float* vb = vecs + (size_t)(p & 1) * N;
            if (tid < OUT_PER_WG) {
                __hip_atomic_store(vb + blockIdx.x * OUT_PER_WG + tid, tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __threadfence();
            }
            __syncthreads();
            if (tid == 0) {
                __hip_atomic_fetch_add(ctr + p, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                int spins = 0;
                while (__hip_atomic_load(ctr + p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < 256u) {
                    if (++spins > 4000000) { *timeout = p; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            for (int i = tid; i < N; i += 512) vec[i] = __hip_atomic_load(vb + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
        } else {
            u64* g = gran + (size_t)(p & 1) * N;
            if (tid < OUT_PER_WG)
                __hip_atomic_store(g + blockIdx.x * OUT_PER_WG + tid, ((u64)(unsigned)p << 32) | (u64)__float_as_uint(tot),
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            for (int idx = tid; idx < N; idx += 512) {
                u64 x;
                while (true) {
                    x = __hip_atomic_load(g + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (__all((unsigned)(x >> 32) == (unsigned)p)) break;
                    if (++spins > 4000000) { if (lane == 0) *timeout = p; break; }
                    __builtin_amdgcn_s_sleep(1);
                }
                vec[idx] = __uint_as_float((unsigned)x);
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < NI; ++i) wv[i] = wn[i];
    }
    if (tid == 0) out[blockIdx.x] = last;
}

template <int NI, int MODE, int N>
static void run(const char* name, hipStream_t s, float* vecs, u64* gran, unsigned* ctr, float* w, size_t wstride, float* out, int* tmo) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        const int phases = 120;
        CK(hipMemsetAsync(gran, 0, 2 * 4096 * 8, s));
        CK(hipMemsetAsync(ctr, 0, 256 * 4, s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL((k_persist2<NI, MODE, N>), dim3(256), dim3(512), 0, s, vecs, gran, ctr, w, wstride, phases, out, tmo);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
        float o0; CK(hipMemcpy(&o0, out, 4, hipMemcpyDeviceToHost));
        if (rep == 2) printf("%-28s N=%4d weights %2d MB/phase: %6.2f us per phase  timeout=%d out=%f\n", name, N, NI * 2, ms * 1000 / phases, t, o0);
    }
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    u64* gran; float *w, *out, *vecs; int* tmo; unsigned* ctr;
    CK(hipMalloc(&gran, 2 * 4096 * 8));
    CK(hipMalloc(&vecs, 2 * 4096 * 4)); CK(hipMemset(vecs, 0, 2 * 4096 * 4));
    CK(hipMalloc(&ctr, 256 * 4));
    const size_t wstride = (size_t)4096 * 4096;
    CK(hipMalloc(&w, 30 * wstride * sizeof(float))); CK(hipMemset(w, 0, 30 * wstride * sizeof(float)));
    CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&tmo, 4)); CK(hipMemset(tmo, 0, 4));
    run<0, 0, 1024>("counter", s, vecs, gran, ctr, w, wstride, out, tmo);
    run<2, 0, 1024>("counter", s, vecs, gran, ctr, w, wstride, out, tmo);
    run<6, 0, 1024>("counter", s, vecs, gran, ctr, w, wstride, out, tmo);
    run<8, 0, 1024>("counter", s, vecs, gran, ctr, w, wstride, out, tmo);
    run<8, 0, 4096>("counter", s, vecs, gran, ctr, w, wstride, out, tmo);
    run<0, 1, 1024>("tags", s, vecs, gran, ctr, w, wstride, out, tmo);
    run<6, 1, 1024>("tags", s, vecs, gran, ctr, w, wstride, out, tmo);
    run<8, 1, 1024>("tags", s, vecs, gran, ctr, w, wstride, out, tmo);
    run<8, 1, 4096>("tags", s, vecs, gran, ctr, w, wstride, out, tmo);
    return 0;
}
