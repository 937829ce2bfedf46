// persist5: is the seam chain of persist3/4 slowed by the weight stream because the stream is issued in bursts?
// Roles inside one resident workgroup (1024 threads): 4 compute waves, 8 polling waves, 4 streaming waves.
// No s_barrier at all: polling -> compute hand-off through LDS arrival counters, so the streaming waves can free-run.
// The streaming waves read the layer's 50.4 MB (197 KB per workgroup) at a PACED rate: each keeps at most
// DEPTH x 1 KiB loads in flight.  The compute waves do not consume the weights here (this measures interference only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int CW = 4, PW = 8, SW = 4, NT = (CW + PW + SW) * 64;

__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }

struct Bufs { u64 *x, *qkv, *o, *h; const float* w; float* out; int* timeout; int layers; int stream; int depth; int en; u64* clk; };

__device__ __forceinline__ void gather(const u64* g, int n, unsigned tag, float* vec, int ptid, int* timeout) {
    int spins = 0;
    for (int idx = ptid; idx < n; idx += PW * 64) {
        u64 x;
        while (true) {
            x = __hip_atomic_load(g + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(x >> 32) == tag)) break;
            if (++spins > 2000000) { *timeout = (int)tag; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        vec[idx] = __uint_as_float((unsigned)x);
    }
}

__device__ __forceinline__ void lds_wait_ge(unsigned* ctr, unsigned target) {
    while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
}

__global__ __launch_bounds__(NT) void k_layer(Bufs B) {
    __shared__ __attribute__((aligned(16))) float vec[2][4096];
    __shared__ float part[CW];
    __shared__ unsigned ready, consumed, arrive;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    if (tid == 0) { ready = 0; consumed = 0; arrive = 0; }
    __syncthreads();
    const int nphase = B.layers * 5;
    const u64 t_start = wall_clock64();
    if (wave >= CW + PW) {
        // ---- streaming waves: paced, free-running ----
        if (!B.stream) return;
        const int st = tid - (CW + PW) * 64;                       // 0..255
        const size_t per_layer_f4 = 50331648 / 16 / 256;           // float4 per workgroup per layer (48 MiB / 256)
        const f4* base = reinterpret_cast<const f4*>(B.w) + (size_t)wg * per_layer_f4;
        f4 acc = {0.f, 0.f, 0.f, 0.f};
        const int iters = (int)(per_layer_f4 / 256);               // float4 per thread per layer = 48
        for (int l = 0; l < B.layers; ++l) {
            const f4* p = base + (size_t)(l % 20) * (50331648 / 16) + st;
            for (int i = 0; i < iters; i += 4) {
                f4 a = __builtin_nontemporal_load(p + (size_t)(i + 0) * 256);
                f4 b = __builtin_nontemporal_load(p + (size_t)(i + 1) * 256);
                f4 c = __builtin_nontemporal_load(p + (size_t)(i + 2) * 256);
                f4 d = __builtin_nontemporal_load(p + (size_t)(i + 3) * 256);
                if (B.depth == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                acc += a + b + c + d;
                if (B.depth == 8 && (i & 4)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
        }
        if (acc.x == 12345.f) B.out[wg] = acc.y;
        if (tid == (CW + PW) * 64) B.clk[256 + wg] = wall_clock64() - t_start;
        return;
    }
    if (wave >= CW) {
        // ---- polling waves ----
        const int ptid = tid - CW * 64;
        unsigned tag = 1;
        for (int ph = 0; ph < nphase; ++ph) {
            const int k = ph % 5;
            const u64* gin = k == 0 ? B.x : k == 1 ? B.qkv : k == 2 ? B.o : k == 3 ? B.x : B.h;
            const int nin = k == 4 ? B.en : 1024;
            const bool producer = k != 1 || wg < 64;
            // the buffer being overwritten was read in phase ph-2
            if (ph >= 2) lds_wait_ge(&consumed, (unsigned)(ph - 1) * CW);
            if (producer) gather(gin, nin, tag, vec[ph & 1], ptid, B.timeout);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) __hip_atomic_fetch_add(&ready, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ++tag;
        }
        return;
    }
    // ---- compute waves ----
    unsigned tag = 1;
    float last = 0.f;
    for (int ph = 0; ph < nphase; ++ph) {
        const int k = ph % 5;
        u64* gout = k == 0 ? B.qkv : k == 1 ? B.o : k == 2 ? B.x : k == 3 ? B.h : B.x;
        const int nout = k == 0 ? 3072 : k == 3 ? B.en : 1024;
        const int producers = k == 1 ? 64 : 256;
        const int nin = k == 4 ? B.en : 1024;
        lds_wait_ge(&ready, (unsigned)(ph + 1) * PW);
        float s = 0.f;
        for (int i = tid * 4; i < nin; i += CW * 64 * 4) {
            const float4 a = *reinterpret_cast<const float4*>(&vec[ph & 1][i]);
            s += a.x + a.y + a.z + a.w;
        }
        s = wsum(s);
        unsigned prev = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) {
            part[wave] = s;
            __hip_atomic_fetch_add(&consumed, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            prev = __hip_atomic_fetch_add(&arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        prev = __builtin_amdgcn_readfirstlane(prev);
        if (prev == (unsigned)ph * CW + CW - 1) {
            float tot = 0.f;
            for (int i = 0; i < CW; ++i) tot += part[i];
            tot = tot * 1e-6f + 1.0f;
            last = tot;
            const int per = nout / producers;
            if (wg < producers && lane < per)
                __hip_atomic_store(gout + wg * per + lane, ((u64)(tag + 1) << 32) | (u64)__float_as_uint(tot), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
        }
        ++tag;
    }
    if (lane == 0) B.out[wg] = last;
    if (tid == 0) B.clk[wg] = wall_clock64() - t_start;
}

int main(int argc, char** argv) {
    hipStream_t s; CK(hipStreamCreate(&s));
    Bufs B;
    u64* g; CK(hipMalloc(&g, 4 * 4096 * 8));
    B.x = g; B.qkv = g + 4096; B.o = g + 2 * 4096; B.h = g + 3 * 4096;
    float* w; const size_t bytes = (size_t)20 * 50331648;
    CK(hipMalloc(&w, bytes)); CK(hipMemset(w, 0, bytes));
    B.w = w;
    CK(hipMalloc(&B.out, 4096)); CK(hipMalloc(&B.timeout, 4)); CK(hipMemset(B.timeout, 0, 4));
    B.layers = 30;
    CK(hipMalloc(&B.clk, 512 * 8)); CK(hipMemset(B.clk, 0, 512 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int en : {1024, 4096})
    for (int cfg = 0; cfg < 4; ++cfg) {
        B.en = en;
        B.stream = cfg > 0; B.depth = cfg == 1 ? 4 : cfg == 2 ? 8 : 48;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemsetAsync(g, 0, 4 * 4096 * 8, s));
            u64 init[1024];
            for (int i = 0; i < 1024; ++i) init[i] = ((u64)1 << 32) | 0x3f800000u;
            CK(hipMemcpyAsync(B.x, init, sizeof(init), hipMemcpyHostToDevice, s));
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            hipLaunchKernelGGL(k_layer, dim3(256), dim3(NT), 0, s, B);
            CK(hipEventRecord(e1, s));
            CK(hipStreamSynchronize(s));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            int t; CK(hipMemcpy(&t, B.timeout, 4, hipMemcpyDeviceToHost));
            u64 clk[512]; CK(hipMemcpy(clk, B.clk, sizeof(clk), hipMemcpyDeviceToHost));
            u64 cmax = 0, smax = 0; for (int i = 0; i < 256; ++i) { if (clk[i] > cmax) cmax = clk[i]; if (clk[256 + i] > smax) smax = clk[256 + i]; }
            if (rep == 2) printf("compute done %.1f us, stream done %.1f us | ", cmax / 100.0, B.stream ? smax / 100.0 : 0.0);
            if (rep == 2) printf("EN=%d stream=%d depth=%2d KiB/wave: %7.1f us total, %6.2f us per layer  timeout=%d\n", en, B.stream, B.depth,
                                 ms * 1000, ms * 1000 / B.layers, t);
        }
    }
    return 0;
}
