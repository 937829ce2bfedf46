// seam_xcd: price of an all-to-all hand-off that stays INSIDE an XCD (round 4).
//   256 workgroups x (8 consumer waves + 1 loader wave), all resident, blockIdx % 8 == XCC_ID (checked).  Per phase every workgroup
//   publishes its 1/32 of its XCD's P-byte block (16-byte stores) and gathers the XCD's whole block; the eight XCDs run the same chain
//   side by side and never read each other's blocks.  Question: what does a hand-off cost when producers and consumers share an L2?
//   (the device-wide hand-off of the decode steps: 2.4 us idle, 3.7 us beside the weight stream, scripts/ubench/seam_rows.hip).
//   The cache policy of the stores / loads is a template parameter (aux bits: 1 = sc0, 2 = nt, 16 = sc1): device scope (sc1) is what
//   the decode steps use; sc0 / plain accesses are only coherent through the shared L2 if the per-CU vector cache is bypassed, which
//   the value check decides (every phase publishes different values; a stale line shows up as a time-out or a wrong value).
//   mode 1: every G-th phase is a DEVICE-wide hand-off of PG bytes (sc1), the others XCD-local: the mix a layer would have.
// hipcc --offload-arch=gfx950 -O3 -o seam_xcd seam_xcd.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int kG = 256, kCW = 8, kThreads = (kCW + 1) * 64, kSlot = 16384, kRing = 8;
constexpr unsigned kPoison = 0xffffffffu;

struct Args {
    unsigned* buf;          // [4 parities][8 XCDs][P / 4]   (local)  |  [4][PG / 4] (global) behind it
    unsigned* gbuf;
    const char* w;
    size_t w_bytes;
    int phases, pay_bytes, gpay_bytes, gevery, stream_kb, salt, inv;
    int* err;               // [0] timeouts, [1] wrong values, [2] xcc mismatches
};

__device__ __forceinline__ unsigned val_of(int p, int i) { return ((unsigned)p * 2654435761u + (unsigned)i * 40503u + 12345u) & 0x7fffffffu; }
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bool clean(u32x4 v) { return v.x != kPoison && v.y != kPoison && v.z != kPoison && v.w != kPoison; }

// gather `n4` dwords (a multiple of 4): piece j of (wave, lane) = dwords [(j * kCW + wave) * 256 + lane * 4, +4); waves / lanes past
// the end sit out
template <int NL, int LAUX>
__device__ __forceinline__ int gather_check(__amdgpu_buffer_rsrc_t rs, int n4, int wave, int lane, int p, int base_i, int* err, int inv) {
    u32x4 v[NL];
    const int d0 = wave * 256 + lane * 4;
    const bool in0 = d0 < n4;
    if (wave * 256 >= n4) return 0;
    unsigned spins = 0;
    while (true) {
        bool again = false;
#pragma unroll
        for (int j = 0; j < NL; ++j) {
            const bool in = d0 + j * kCW * 256 < n4;
            if (in) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, d0 * 4, j * kCW * 1024, LAUX);
            else v[j] = u32x4{0u, 0u, 0u, 0u};
        }
#pragma unroll
        for (int j = 0; j < NL; ++j) again = again || __any(!clean(v[j]));
        if (!again) break;
        if (inv) asm volatile("buffer_inv sc0" ::: "memory");
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 300000u) { if (lane == 0) atomicAdd(err, 1); return 1; }
    }
    (void)in0;
    int bad = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = d0 + j * kCW * 256;
        if (i < n4) bad += (v[j].x != val_of(p, base_i + i)) + (v[j].y != val_of(p, base_i + i + 1)) + (v[j].z != val_of(p, base_i + i + 2)) +
                           (v[j].w != val_of(p, base_i + i + 3));
    }
    if (bad) atomicAdd(err + 1, bad);
    return 0;
}

template <int NL, int NLG, int SAUX, int LAUX>
__global__ __launch_bounds__(kThreads) void k_seam(const Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    unsigned* ctl = reinterpret_cast<unsigned*>(smem + kRing * kSlot);      // [0] arrive, [1] consumer phase, [2] abort
    if (threadIdx.x < 16) ctl[threadIdx.x] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wg = blockIdx.x;
    if (wave == kCW) {
        if (A.stream_kb == 0) return;
        unsigned fill = 0;
        size_t off = (size_t)wg * A.stream_kb * 1024;
        for (int p = 0; p < A.phases; ++p) {
            unsigned spins = 0;
            while ((int)lds_ld(ctl + 1) + 1 < p) { __builtin_amdgcn_s_sleep(2); if (++spins > 4000000u || lds_ld(ctl + 2)) return; }
            for (int kb = 0; kb < A.stream_kb; kb += 16) {
                const int n = min(16, A.stream_kb - kb);
                char* dst = ring + (fill & (kRing - 1)) * kSlot;
                const char* src = A.w + (off % (A.w_bytes - (size_t)kG * A.stream_kb * 1024 - 65536)) + (size_t)kb * 1024 + lane * 16;
                for (int i = 0; i < n; ++i)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + i * 1024),
                                                     (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 2);
                asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                ++fill;
            }
            off += (size_t)kG * A.stream_kb * 1024;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    unsigned bar = 0;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (threadIdx.x == 0 && (int)xcc != (wg & 7)) atomicAdd(A.err + 2, 1);
    const int x = wg & 7, rank = wg >> 3;            // XCD, rank inside it (0..31)
    const int P4 = A.pay_bytes / 4, S4 = P4 / 32;    // dwords of an XCD's block / of a workgroup's slice
    const int G4 = A.gpay_bytes / 4, GS4 = G4 / kG;
    for (int p = 0; p < A.phases; ++p) {
        const int par = p & 3, par2 = (p + 2) & 3, ps = p + A.salt;
        const bool global = A.gevery > 0 && (p % A.gevery) == A.gevery - 1;
        int dead = 0;
        if (!global) {
            unsigned* blk = A.buf + ((size_t)par * 8 + x) * P4;
            unsigned* oth = A.buf + ((size_t)par2 * 8 + x) * P4;
            const __amdgpu_buffer_rsrc_t wr = rsrc(blk, (unsigned)P4 * 4u), po = rsrc(oth, (unsigned)P4 * 4u);
            if (wave == 0 && lane * 4 < S4) {
                const int i = rank * S4 + lane * 4;
                u32x4 v = {val_of(ps, x * P4 + i), val_of(ps, x * P4 + i + 1), val_of(ps, x * P4 + i + 2), val_of(ps, x * P4 + i + 3)};
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_raw_buffer_store_b128(v, wr, i * 4, 0, SAUX);
                u32x4 q = {kPoison, kPoison, kPoison, kPoison};
                __builtin_amdgcn_raw_buffer_store_b128(q, po, i * 4, 0, SAUX);
            }
            dead = gather_check<NL, LAUX>(wr, P4, wave, lane, ps, x * P4, A.err, A.inv);
        } else {
            unsigned* blk = A.gbuf + (size_t)par * G4;
            unsigned* oth = A.gbuf + (size_t)par2 * G4;
            const __amdgpu_buffer_rsrc_t wr = rsrc(blk, (unsigned)G4 * 4u), po = rsrc(oth, (unsigned)G4 * 4u);
            if (wave == 0 && lane * 4 < GS4) {
                const int i = wg * GS4 + lane * 4;
                u32x4 v = {val_of(ps, i), val_of(ps, i + 1), val_of(ps, i + 2), val_of(ps, i + 3)};
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_raw_buffer_store_b128(v, wr, i * 4, 0, 16);
                u32x4 q = {kPoison, kPoison, kPoison, kPoison};
                __builtin_amdgcn_raw_buffer_store_b128(q, po, i * 4, 0, 16);
            }
            dead = gather_check<NLG, 16>(wr, G4, wave, lane, ps, 0, A.err, 0);
        }
        if (dead) lds_st(ctl + 2, 1u);
        bar += kCW;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned spins = 0;
        while (lds_ld(ctl) < bar && !lds_ld(ctl + 2)) { if (++spins > 4000000u) break; }
        if (lds_ld(ctl + 2)) return;
        if (wave == 0 && lane == 0) lds_st(ctl + 1, (unsigned)p + 1u);
    }
}

template <int NL, int NLG, int SAUX, int LAUX>
static float run(const Args& A, size_t lds) {
    CK(hipFuncSetAttribute((const void*)k_seam<NL, NLG, SAUX, LAUX>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Args B = A;
    hipLaunchKernelGGL((k_seam<NL, NLG, SAUX, LAUX>), dim3(kG), dim3(kThreads), lds, 0, B);
    CK(hipDeviceSynchronize());
    B.salt = 7777;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_seam<NL, NLG, SAUX, LAUX>), dim3(kG), dim3(kThreads), lds, 0, B);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / A.phases;
}

int main() {
    const size_t wbytes = (size_t)2 << 30;
    char* w;
    CK(hipMalloc(&w, wbytes));
    CK(hipMemset(w, 1, wbytes));
    const int maxP = 64 * 1024, maxG = 128 * 1024;
    unsigned *buf, *gbuf;
    int* err;
    CK(hipMalloc(&buf, (size_t)4 * 8 * maxP)); CK(hipMalloc(&gbuf, (size_t)4 * maxG)); CK(hipMalloc(&err, 16));
    const size_t lds = kRing * kSlot + 64;
    printf("%-22s %-9s %-9s %-7s %-10s %10s   %s\n", "policy(store/load)", "local_KB", "global_KB", "gevery", "stream_kb", "us/phase", "timeouts wrong xcc-mismatch");
    struct Cfg { const char* name; int id; };
    for (int stream_kb : {0, 48}) {
        for (int gevery : {0, 2}) {      // (every 2nd phase device-wide; other periods would need their own buffer rotation)
            for (int pk : {2, 4, 16}) {
                for (int pol = 0; pol < 5; ++pol) {
                    Args A;
                    A.buf = buf; A.gbuf = gbuf; A.w = w; A.w_bytes = wbytes; A.phases = 400; A.pay_bytes = pk * 1024; A.gpay_bytes = 64 * 1024;
                    A.gevery = gevery; A.stream_kb = stream_kb; A.err = err; A.salt = 0; A.inv = 0;
                    CK(hipMemset(buf, 0xff, (size_t)4 * 8 * maxP)); CK(hipMemset(gbuf, 0xff, (size_t)4 * maxG)); CK(hipMemset(err, 0, 16));
                    float us = 0.f;
                    const char* name = "";
                    // local pieces per lane: 2 KB / 4 KB -> 1 (partly idle waves), 16 KB -> 2; global 64 KB -> 8
#define RUN(SA, LA) (pk <= 8 ? run<1, 8, SA, LA>(A, lds) : run<2, 8, SA, LA>(A, lds))
                    switch (pol) {
                        case 0: name = "sc1 / sc1"; us = RUN(16, 16); break;
                        case 1: name = "plain / sc1"; us = RUN(0, 16); break;
                        case 2: name = "sc0 / sc0"; us = RUN(1, 1); break;
                        case 3: name = "plain / sc0"; us = RUN(0, 1); break;
                        default: name = "plain / plain+inv"; A.inv = 1; us = RUN(0, 0); break;
                    }
                    int he[4];
                    CK(hipMemcpy(he, err, 16, hipMemcpyDeviceToHost));
                    printf("%-22s %-9d %-9d %-7d %-10d %10.2f   %d %d %d\n", name, pk, gevery ? 64 : 0, gevery, stream_kb, us, he[0], he[1], he[2]);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
