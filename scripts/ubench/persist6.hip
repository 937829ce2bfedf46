// persist6: how much cheaper is a {tag,value} all-to-all that stays INSIDE one XCD (32 workgroups, the L2 they share)
// than one across all 8 XCDs?  256 resident workgroups x 512 threads, no weights.  Each group of 32 workgroups runs its own
// all-to-all over a 128-vector (4 outputs per workgroup).
//   GROUPING 0: group = blockIdx.x % 8   (the workgroups of one XCD, if the round-robin placement holds)
//   GROUPING 1: group = blockIdx.x / 32  (32 consecutive workgroups: spread over all 8 XCDs)
//   SCOPE 0: agent-scope stores and loads (sc1: what a cross-XCD hand-off needs)
//   SCOPE 1: workgroup-scope stores and loads (sc0)
//   SCOPE 2: agent-scope stores, workgroup-scope loads
//   SCOPE 3: workgroup-scope stores, agent-scope loads
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;

__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

template <int GROUPING, int SCOPE>
__global__ __launch_bounds__(512) void k_seam(u64* gran, int phases, float* out, int* timeout, int* xcc) {
    __shared__ float vec[128];
    __shared__ float part[8];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const int group = GROUPING == 0 ? (wg & 7) : (wg >> 5);
    const int member = GROUPING == 0 ? (wg >> 3) : (wg & 31);
    if (tid == 0) {
        unsigned id;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
        xcc[wg] = (int)(id & 0xf);
    }
    if (tid < 128) vec[tid] = 1.0f;
    __syncthreads();
    float last = 0.f;
    for (int p = 1; p <= phases; ++p) {
        float s = tid < 128 ? vec[tid] : 0.f;
        s = wsum(s);
        if (lane == 0) part[wave] = s;
        __syncthreads();
        float tot = (part[0] + part[1]) * 1e-6f + 1.0f;
        last = tot;
        u64* g = gran + ((size_t)(p & 1) * 8 + group) * 128;
        if (tid < 4) {
            const u64 v = ((u64)(unsigned)p << 32) | (u64)__float_as_uint(tot);
            if (SCOPE == 0 || SCOPE == 2) __hip_atomic_store(g + member * 4 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_store(g + member * 4 + tid, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (tid < 128) {
            int spins = 0;
            u64 x;
            while (true) {
                if (SCOPE == 0 || SCOPE == 3) x = __hip_atomic_load(g + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else x = __hip_atomic_load(g + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (__all((unsigned)(x >> 32) == (unsigned)p)) break;
                if (++spins > 300000) { if (lane == 0) *timeout = p; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            vec[tid] = __uint_as_float((unsigned)x);
        }
        __syncthreads();
        if (*timeout) break;
    }
    if (tid == 0) out[wg] = last;
}

template <int GROUPING, int SCOPE>
static void run(const char* name, hipStream_t s, u64* gran, float* out, int* tmo, int* xcc) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep) {
        const int phases = 200;
        CK(hipMemsetAsync(gran, 0, 2 * 8 * 128 * 8, s));
        CK(hipMemsetAsync(tmo, 0, 4, s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL((k_seam<GROUPING, SCOPE>), dim3(256), dim3(512), 0, s, gran, phases, out, tmo, xcc);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int t; CK(hipMemcpy(&t, tmo, 4, hipMemcpyDeviceToHost));
        if (rep == 2) printf("%-52s %6.2f us per seam  %s\n", name, ms * 1000 / phases, t ? "(TIMEOUT: stale reads)" : "");
    }
}

int main() {
    hipStream_t s; CK(hipStreamCreate(&s));
    u64* gran; float* out; int *tmo, *xcc;
    CK(hipMalloc(&gran, 2 * 8 * 128 * 8)); CK(hipMalloc(&out, 1024)); CK(hipMalloc(&tmo, 4)); CK(hipMalloc(&xcc, 1024));
    run<1, 0>("32 consecutive workgroups (all XCDs), agent scope", s, gran, out, tmo, xcc);
    run<0, 0>("workgroups b % 8 == x (one XCD), agent scope", s, gran, out, tmo, xcc);
    int h[256]; CK(hipMemcpy(h, xcc, sizeof(h), hipMemcpyDeviceToHost));
    int ok = 1; for (int i = 0; i < 256; ++i) if (h[i] != (h[i & 7])) ok = 0;
    printf("XCC_ID of workgroups 0..15:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("   (b %% 8 groups share an XCD: %s)\n", ok ? "yes" : "NO");
    run<0, 2>("one XCD, agent-scope stores, workgroup-scope loads", s, gran, out, tmo, xcc);
    run<0, 3>("one XCD, workgroup-scope stores, agent-scope loads", s, gran, out, tmo, xcc);
    run<0, 1>("one XCD, workgroup scope both", s, gran, out, tmo, xcc);
    return 0;
}
