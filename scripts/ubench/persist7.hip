// (persist7: persist4 with each 64-byte group of granules on its own 4160-byte stride -> polling spread over channels)
// (persist4: one barrier per phase via an LDS arrival counter; optional 16-byte {tag,f,f,f} granules on the h seam)
// Persistent decode-layer skeleton: can a tag-based all-to-all with DEDICATED polling waves hide the weight stream?
// 256 workgroups x (8 compute waves + 4 polling waves).  Five phases per layer with the GenVC decode shapes:
//   A c_attn  12.6 MB weights, consumes x[1024]            -> qkv[3072]
//   B attn    no weights, 64 workgroups consume qkv, read 0.5 MB of "KV" -> o[1024]
//   C proj    4.2 MB, consumes o[1024]                     -> x[1024]
//   D c_fc    16.8 MB, consumes x[1024]                    -> h[4096]
//   E c_proj  16.8 MB, consumes h[4096]                    -> x[1024]
// Compute waves keep the NEXT phase's weights in flight (registers) while the polling waves (which never have a
// weight load outstanding: vmcnt retires in order, per wave) spin on {tag,value} granules and stage the vector in LDS.
// Workgroup barriers are bare s_barrier (+ lgkmcnt wait), not __syncthreads, so that they do not drain vmcnt.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned long long u64;
typedef float f4 __attribute__((ext_vector_type(4)));
#ifndef GSTRIDE
#define GSTRIDE 520
#endif
__device__ __forceinline__ size_t gidx(int i) { return (size_t)(i >> 3) * GSTRIDE + (i & 7); }

#ifndef PWAVES
#define PWAVES 8
#endif
#ifndef EN
#define EN 4096
#endif
constexpr int CW = 8, PW = PWAVES, NT = (CW + PW) * 64, CT = CW * 64;

__device__ __forceinline__ float wsum(float v) { for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o); return v; }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct Bufs { u64 *x, *qkv, *o, *h; const float* w; const float* kv; float* out; int* timeout; int layers; int sleep; int nopoll; int noweights; int g16; int* torn; f4* h16; };

template <int NI>
__device__ __forceinline__ void prefetch(f4 (&dst)[8], const float* w, size_t slab, int tid_c, int wg) {
    const f4* p = reinterpret_cast<const f4*>(w + slab) + (size_t)wg * CT + tid_c;
#pragma unroll
    for (int i = 0; i < NI; ++i) dst[i] = __builtin_nontemporal_load(p + (size_t)i * 256 * CT);
}

// polling waves: gather n granules tagged `tag` from g into vec
__device__ __forceinline__ void gather(const u64* g, int n, unsigned tag, float* vec, int ptid, int* timeout, int sleep, int nopoll) {
    int spins = 0;
    if (nopoll) return;
    for (int idx = ptid; idx < n; idx += PW * 64) {
        u64 x;
        while (true) {
            x = __hip_atomic_load(g + gidx(idx), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all((unsigned)(x >> 32) == tag)) break;
            if (++spins > 2000000) { *timeout = (int)tag; break; }
            for (int q = 0; q < sleep; ++q) __builtin_amdgcn_s_sleep(1);
        }
        vec[idx] = __uint_as_float((unsigned)x);
    }
}

// 16-byte granules {tag, f0, f1, f2}: n values -> ceil(n/3) granules
__device__ __forceinline__ void gather16(const f4* g, int n, unsigned tag, float* vec, int ptid, int* timeout, int* torn) {
    int spins = 0;
    const int ng = (n + 2) / 3;
    for (int idx = ptid; idx < ng; idx += PW * 64) {
        f4 x;
        while (true) {
            asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(x) : "v"(g + idx) : "memory");
            if (__all(__float_as_uint(x.x) == tag)) break;
            if (++spins > 2000000) { *timeout = (int)tag; break; }
            __builtin_amdgcn_s_sleep(1);
        }
        if (__float_as_uint(x.w) != tag * 3u + 7u) *torn = 1;
        vec[idx * 3] = x.y; vec[idx * 3 + 1] = x.z; vec[idx * 3 + 2] = 1.0f;
    }
}

// compute waves: dot NI float4 of weights with the staged vector, reduce over the workgroup, emit NOUT/producers outputs
template <int NI>
__device__ __forceinline__ float dot_phase(const f4 (&wv)[8], const float* vec, int nmask, int tid_c) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NI; ++i) {
        const float4 a = *reinterpret_cast<const float4*>(&vec[((i * CT + tid_c) * 4) & nmask]);
        s += wv[i].x * a.x + wv[i].y * a.y + wv[i].z * a.z + wv[i].w * a.w;
    }
    if (NI == 0) s = vec[tid_c & nmask];
    return wsum(s);
}

__global__ __launch_bounds__(NT) void k_layer(Bufs B) {
    __shared__ __attribute__((aligned(16))) float vec[2][4096];
    __shared__ float part[CW];
    __shared__ unsigned arrive;
    if (threadIdx.x == 0) arrive = 0;
    __syncthreads();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
    const size_t slab = (size_t)4096 * 4096;       // floats per phase slab
    if (wave >= CW) {
        // ---- polling waves: no weight load is ever outstanding here ----
        const int ptid = tid - CT;
        unsigned tag = 1;
        int vb = 0;
        for (int l = 0; l < B.layers; ++l) {
#define POLL(GIN, NIN, PRODUCERS, KVREAD)                                                                             \
            {                                                                                                         \
                if (wg < PRODUCERS) {                                                                                 \
                    gather(GIN, NIN, tag, vec[vb], ptid, B.timeout, B.sleep, B.nopoll);                                                 \
                    if (KVREAD) {                                                                                     \
                        const f4 kvv = *(reinterpret_cast<const f4*>(B.kv) + (size_t)wg * CT + (ptid & (CT - 1)));                 \
                        vec[vb][2048 + (ptid & 1023)] = kvv.x + kvv.y;                                                         \
                    }                                                                                                 \
                }                                                                                                     \
                lds_barrier();                                                                                        \
                ++tag; vb ^= 1;                                                                                       \
            }
            POLL(B.x, 1024, 256, 0)
            POLL(B.qkv, 1024, 64, 1)
            POLL(B.o, 1024, 256, 0)
            POLL(B.x, 1024, 256, 0)
            if (B.g16) { gather16(B.h16, EN, tag, vec[vb], ptid, B.timeout, B.torn); lds_barrier(); ++tag; vb ^= 1; }
            else POLL(B.h, EN, 256, 0)
#undef POLL
        }
        return;
    }
    // ---- compute waves: weight loads, LDS, granule stores only ----
    f4 w0[8], w1[8], w2[8];
    unsigned tag = 1;
    int vb = 0;
    float last = 0.f;
    prefetch<6>(w0, B.w, 0, tid, wg);                    // A of layer 0; B has no weights
    for (int l = 0; l < B.layers; l += 3) {
        const float* wl = B.w + (size_t)(l % 6) * 5 * slab;
#define PHASE(NI_CUR, WCUR, NI_NEXT, WNEXT, NEXT_SLAB, NIN, GOUT, NOUT, PRODUCERS, G16OUT)                                    \
        {                                                                                                             \
            if (!B.noweights) prefetch<NI_NEXT>(WNEXT, wl, NEXT_SLAB, tid, wg);                                                       \
            lds_barrier();                                                                                            \
            float s = dot_phase<NI_CUR>(WCUR, vec[vb], NIN - 1, tid);                                                 \
            unsigned prev = 0;                                                                                        \
            if (lane == 0) { part[wave] = s; prev = __hip_atomic_fetch_add(&arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); } \
            prev = __builtin_amdgcn_readfirstlane(prev);                                                              \
            if (prev == (tag - 1) * CW + CW - 1) {                                                                    \
                float tot = 0.f;                                                                                      \
                for (int i = 0; i < CW; ++i) tot += part[i];                                                          \
                tot = tot * 1e-6f + 1.0f;                                                                             \
                last = tot;                                                                                           \
                constexpr int per = NOUT / PRODUCERS;                                                                 \
                if (G16OUT && B.g16) {                                                                                \
                    constexpr int pg = (per + 2) / 3 + 0;                                                             \
                    if (wg < PRODUCERS && lane < pg) {                                                                \
                        f4 v = {__uint_as_float(tag + 1), tot, tot, __uint_as_float((tag + 1) * 3u + 7u)};            \
                        asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(B.h16 + wg * pg + lane), "v"(v) : "memory"); \
                    }                                                                                                 \
                } else if (wg < PRODUCERS && lane < per)                                                              \
                    __hip_atomic_store(GOUT + gidx(wg * per + lane), ((u64)(tag + 1) << 32) | (u64)__float_as_uint(tot),     \
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                                   \
            }                                                                                                         \
            ++tag; vb ^= 1;                                                                                           \
        }
        PHASE(6, w0, 2, w2, 1 * slab, 1024, B.qkv, 3072, 256, 0)   // A, prefetch C
        PHASE(0, w1, 8, w0, 2 * slab, 1024, B.o, 1024, 64, 0)   // B, prefetch D
        PHASE(2, w2, 8, w1, 3 * slab, 1024, B.x, 1024, 256, 0)   // C, prefetch E
        PHASE(8, w0, 6, w2, 4 * slab, 1024, B.h, EN, 256, 1)   // D, prefetch A
        PHASE(8, w1, 0, w0, 0 * slab, EN, B.x, 1024, 256, 0)   // E, prefetch B
        PHASE(6, w2, 2, w1, 1 * slab, 1024, B.qkv, 3072, 256, 0)   // A, prefetch C
        PHASE(0, w0, 8, w2, 2 * slab, 1024, B.o, 1024, 64, 0)   // B, prefetch D
        PHASE(2, w1, 8, w0, 3 * slab, 1024, B.x, 1024, 256, 0)   // C, prefetch E
        PHASE(8, w2, 6, w1, 4 * slab, 1024, B.h, EN, 256, 1)   // D, prefetch A
        PHASE(8, w0, 0, w2, 0 * slab, EN, B.x, 1024, 256, 0)   // E, prefetch B
        PHASE(6, w1, 2, w0, 1 * slab, 1024, B.qkv, 3072, 256, 0)   // A, prefetch C
        PHASE(0, w2, 8, w1, 2 * slab, 1024, B.o, 1024, 64, 0)   // B, prefetch D
        PHASE(2, w0, 8, w2, 3 * slab, 1024, B.x, 1024, 256, 0)   // C, prefetch E
        PHASE(8, w1, 6, w0, 4 * slab, 1024, B.h, EN, 256, 1)   // D, prefetch A
        PHASE(8, w2, 0, w1, 0 * slab, EN, B.x, 1024, 256, 0)   // E, prefetch B
#undef PHASE
    }
    if (tid == CT - 64 || tid == 0) B.out[wg] = last;
}

int main(int argc, char** argv) {
    hipStream_t s; CK(hipStreamCreate(&s));
    Bufs B;
    const size_t VB = (size_t)512 * GSTRIDE + 64;
    u64* g; CK(hipMalloc(&g, 4 * VB * 8));
    B.x = g; B.qkv = g + VB; B.o = g + 2 * VB; B.h = g + 3 * VB;
    float* w; const size_t slab = (size_t)4096 * 4096;
    CK(hipMalloc(&w, 30 * slab * sizeof(float))); CK(hipMemset(w, 0, 30 * slab * sizeof(float)));
    B.w = w;
    float* kv; CK(hipMalloc(&kv, 64 * CT * 16)); CK(hipMemset(kv, 0, 64 * CT * 16)); B.kv = kv;
    CK(hipMalloc(&B.out, 4096)); CK(hipMalloc(&B.timeout, 4)); CK(hipMemset(B.timeout, 0, 4));
    B.layers = 30;
    CK(hipMalloc(&B.torn, 4)); CK(hipMemset(B.torn, 0, 4));
    CK(hipMalloc(&B.h16, 2048 * 16)); CK(hipMemset(B.h16, 0, 2048 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int cfg : {0, 4}) {
    B.sleep = cfg == 1 ? 8 : (cfg == 2 ? 32 : 1); B.nopoll = cfg == 3; B.noweights = cfg == 4; B.g16 = cfg >= 6; if (cfg == 7) B.noweights = 1;
    printf("cfg %d: sleep=%d nopoll=%d noweights=%d g16=%d\n", cfg, B.sleep, B.nopoll, B.noweights, B.g16);
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemsetAsync(g, 0, 4 * VB * 8, s));
        CK(hipMemsetAsync(B.h16, 0, 2048 * 16, s));
        // the first phase consumes x tagged 1
        static u64 init[512 * GSTRIDE + 64];
        for (int i = 0; i < 1024; ++i) init[(size_t)(i >> 3) * GSTRIDE + (i & 7)] = ((u64)1 << 32) | 0x3f800000u;
        CK(hipMemcpyAsync(B.x, init, sizeof(init), hipMemcpyHostToDevice, s));
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(k_layer, dim3(256), dim3(NT), 0, s, B);
        CK(hipEventRecord(e1, s));
        CK(hipStreamSynchronize(s));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int t; CK(hipMemcpy(&t, B.timeout, 4, hipMemcpyDeviceToHost));
        int torn; CK(hipMemcpy(&torn, B.torn, 4, hipMemcpyDeviceToHost)); if (torn) printf("TORN 16-byte granule observed!\n");
        float o0; CK(hipMemcpy(&o0, B.out, 4, hipMemcpyDeviceToHost));
        printf("30 layers x 5 phases: %7.1f us total, %6.2f us per layer, %5.2f us per phase   timeout=%d out=%f\n", ms * 1000,
               ms * 1000 / B.layers, ms * 1000 / B.layers / 5, t, o0);
    }
    }
    return 0;
}
