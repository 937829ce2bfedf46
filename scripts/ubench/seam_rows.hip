// seam_rows: price of the all-to-all hand-off a MULTI-ROW one-launch decode step needs (round 3).
//   256 workgroups x (8 consumer waves + 1 loader wave), all resident.  Per phase every workgroup publishes its 1/256 of a
//   P-byte activation block with 16-byte write-through (sc1) stores, then gathers the WHOLE block (each wave 1/8 of it) with
//   16-byte sc1 loads.  No tags and no flags: the buffer of the other parity is poisoned (0xffffffff in every dword) by its
//   producers one phase ahead, a consumer re-reads a 16-byte piece until none of its dwords is the poison pattern.
//   mode 0: flat (every workgroup reads the device-wide buffer)
//   mode 1: two-level (NLD leader workgroups per XCD copy the block into a per-XCD buffer with plain stores, the XCD's
//           workgroups gather that copy with sc1 loads)
//   stream_kb: KiB of weights per phase each loader wave pulls into an LDS ring with LDS-DMA beside the hand-offs.
// Also: layout check of v_mfma_f32_4x4x1_16b_f32 (the matrix instruction of the planned 4-weight-row x R-activation-row step).
// hipcc --offload-arch=gfx950 -O3 -o seam_rows seam_rows.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kG = 256, kCW = 8, kThreads = (kCW + 1) * 64, kSlot = 16384, kRing = 8;
constexpr unsigned kPoison = 0xffffffffu;

struct Args {
    unsigned* buf;          // [4][P / 4]: phase p publishes into buffer p % 4 and poisons buffer (p + 2) % 4
    unsigned* loc;          // [8][4][P / 4]
    const char* w;          // weight pool
    size_t w_bytes;
    int phases, pay_bytes, mode, nld, stream_kb, salt;
    int* err;               // [0] timeouts, [1] wrong values, [2] xcc mismatches
};

__device__ __forceinline__ unsigned val_of(int p, int i) { return ((unsigned)p * 2654435761u + (unsigned)i * 40503u + 12345u) & 0x7fffffffu; }
__device__ __forceinline__ unsigned lds_ld(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_st(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ bool clean(u32x4 v) { return v.x != kPoison && v.y != kPoison && v.z != kPoison && v.w != kPoison; }

// gather NL 16-byte pieces per lane: piece j of lane `lane` of wave `wave` = dwords [(j * kCW + wave) * 256 + lane * 4, +4)
template <int NL>
__device__ __forceinline__ int gather_check(__amdgpu_buffer_rsrc_t rs, int wave, int lane, int p, int* err, float* stage) {
    u32x4 v[NL];
    const int voff = (wave * 256 + lane * 4) * 4;
    unsigned spins = 0;
    while (true) {
        v[0] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, 0, 16);
        if (__all(clean(v[0]))) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000u) { if (lane == 0) atomicAdd(err, 1); return 1; }
    }
#pragma unroll
    for (int j = 1; j < NL; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * kCW * 1024, 16);
    while (true) {
        bool again = false;
#pragma unroll
        for (int j = 1; j < NL; ++j) {
            if (__any(!clean(v[j]))) { v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, j * kCW * 1024, 16); again = true; }
        }
        if (!again) break;
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 2000000u) { if (lane == 0) atomicAdd(err, 1); return 1; }
    }
    int bad = 0;
#pragma unroll
    for (int j = 0; j < NL; ++j) {
        const int i = (j * kCW + wave) * 256 + lane * 4;
        bad += (v[j].x != val_of(p, i)) + (v[j].y != val_of(p, i + 1)) + (v[j].z != val_of(p, i + 2)) + (v[j].w != val_of(p, i + 3));
        if (stage) *reinterpret_cast<u32x4*>(stage + (j * 64 + lane) * 4) = v[j];
    }
    if (bad) atomicAdd(err + 1, bad);
    return 0;
}

template <int NL>
__global__ __launch_bounds__(kThreads) void k_seam(const Args A) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;
    unsigned* ctl = reinterpret_cast<unsigned*>(smem + kRing * kSlot);      // [0] arrive, [1] consumer phase, [2] abort
    float* stage = reinterpret_cast<float*>(ctl + 16);                      // [kCW][1024] floats (first pieces only)
    if (threadIdx.x < 16) ctl[threadIdx.x] = 0u;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), wg = blockIdx.x;
    const int P4 = A.pay_bytes / 4;                  // dwords
    const int S4 = P4 / kG;                          // dwords a workgroup publishes
    if (wave == kCW) {
        // loader: stream_kb KiB per phase, at most one phase ahead of the consumers
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
    const unsigned pbytes = (unsigned)A.pay_bytes;
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 0xf;
    if (threadIdx.x == 0 && (int)xcc != (wg & 7)) atomicAdd(A.err + 2, 1);
    const int x = wg & 7, rank = wg >> 3;            // assumed XCD, rank inside it
    for (int p = 0; p < A.phases; ++p) {
        // Rotation of four buffers: the buffer poisoned at phase p held phase p - 2 (every workgroup has published p - 1, so it has
        // finished reading p - 2); the poison is drained before this workgroup's NEXT publish, and a reader of phase p + 2 polls only
        // after it has seen that publish.
        const int par = p & 3, par2 = (p + 2) & 3, ps = p + A.salt;
        const __amdgpu_buffer_rsrc_t wr = rsrc(A.buf + (size_t)par * P4, pbytes);
        const __amdgpu_buffer_rsrc_t po = rsrc(A.buf + (size_t)par2 * P4, pbytes);
        // ---- publish this workgroup's slice, poison the slice of the other parity ----
        if (wave == 0 && lane * 4 < S4) {
            const int i = wg * S4 + lane * 4;
            u32x4 v = {val_of(ps, i), val_of(ps, i + 1), val_of(ps, i + 2), val_of(ps, i + 3)};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the poison of the previous phase has landed
            __builtin_amdgcn_raw_buffer_store_b128(v, wr, i * 4, 0, 16);
            u32x4 q = {kPoison, kPoison, kPoison, kPoison};
            __builtin_amdgcn_raw_buffer_store_b128(q, po, i * 4, 0, 16);
        }
        int dead = 0;
        if (A.mode == 0) {
            dead = gather_check<NL>(wr, wave, lane, ps, A.err, nullptr);
        } else {
            unsigned* lbase = A.loc + ((size_t)x * 4 + par) * P4;
            unsigned* lother = A.loc + ((size_t)x * 4 + par2) * P4;
            if (rank < A.nld) {
                // leader `rank` of this XCD copies dwords [rank * P4 / nld, +P4 / nld): each wave 1/8 of that, NL / nld pieces per lane
                const int L4 = P4 / A.nld;
                const __amdgpu_buffer_rsrc_t src = rsrc(A.buf + (size_t)par * P4 + (size_t)rank * L4, (unsigned)L4 * 4u);
                for (int d4 = (wave * 64 + lane) * 4; d4 < L4; d4 += kCW * 256) {
                    u32x4 v;
                    unsigned spins = 0;
                    while (true) {
                        v = __builtin_amdgcn_raw_buffer_load_b128(src, d4 * 4, 0, 16);
                        if (clean(v)) break;                 // (per lane: a leader's part may be smaller than a wave)
                        __builtin_amdgcn_s_sleep(1);
                        if (++spins > 2000000u) { atomicAdd(A.err, 1); dead = 1; break; }
                    }
                    // (this leader has gathered phase p - 1 => everyone is done with the local copy of phase p - 2: poison it)
                    u32x4 q = {kPoison, kPoison, kPoison, kPoison};
                    *reinterpret_cast<u32x4*>(lother + (size_t)rank * L4 + d4) = q;
                    *reinterpret_cast<u32x4*>(lbase + (size_t)rank * L4 + d4) = v;
                }
            }
            if (!dead) dead = gather_check<NL>(rsrc(lbase, pbytes), wave, lane, ps, A.err, nullptr);
        }
        if (dead) lds_st(ctl + 2, 1u);
        // consumer barrier (LDS arrival counter), as in the real kernel
        bar += kCW;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(ctl, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        unsigned spins = 0;
        while (lds_ld(ctl) < bar && !lds_ld(ctl + 2)) { if (++spins > 4000000u) break; }
        if (lds_ld(ctl + 2)) return;
        if (wave == 0 && lane == 0) lds_st(ctl + 1, (unsigned)p + 1u);
    }
}

__global__ void k_mfma_layout(const float* a, const float* b, float* d) {
    const int l = threadIdx.x;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}

template <int NL>
static float run(const Args& A, size_t lds) {
    CK(hipFuncSetAttribute((const void*)k_seam<NL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    Args B = A;
    hipLaunchKernelGGL(k_seam<NL>, dim3(kG), dim3(kThreads), lds, 0, B);     // warm-up (phases is a multiple of 4: the rotation carries over)
    CK(hipDeviceSynchronize());
    B.salt = 7777;
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k_seam<NL>, dim3(kG), dim3(kThreads), lds, 0, B);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.f / A.phases;
}

int main() {
    // ---- MFMA 4x4x1 16-block layout: expect d[lane 4b + j][v] = a[4b + v] * b[4b + j] ----
    {
        float *a, *b, *d;
        CK(hipMalloc(&a, 256)); CK(hipMalloc(&b, 256)); CK(hipMalloc(&d, 1024));
        float ha[64], hb[64], hd[256];
        for (int i = 0; i < 64; ++i) { ha[i] = 1.0f + i; hb[i] = 100.0f + 3 * i; }
        CK(hipMemcpy(a, ha, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(b, hb, 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_mfma_layout, dim3(1), dim3(64), 0, 0, a, b, d);
        CK(hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) {
            const int blk = l / 4, j = l % 4;
            if (hd[l * 4 + v] != ha[4 * blk + v] * hb[4 * blk + j]) ++bad;
        }
        printf("mfma_f32_4x4x1_16b layout: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as assumed: D_b[v][j] in lane 4b+j vgpr v; A_b[i] lane 4b+i; B_b[j] lane 4b+j", bad);
        if (bad) for (int l = 0; l < 8; ++l) printf("  lane %d: %g %g %g %g\n", l, hd[l * 4], hd[l * 4 + 1], hd[l * 4 + 2], hd[l * 4 + 3]);
    }
    const size_t wbytes = (size_t)2 << 30;
    char* w;
    CK(hipMalloc(&w, wbytes));
    CK(hipMemset(w, 1, wbytes));
    const int maxP = 256 * 1024;
    unsigned *buf, *loc;
    int* err;
    CK(hipMalloc(&buf, 4 * maxP)); CK(hipMalloc(&loc, 32 * maxP)); CK(hipMalloc(&err, 16));
    const size_t lds = kRing * kSlot + 64 + 16;
    printf("%-8s %-6s %-5s %-10s %10s   %s\n", "payload", "mode", "nld", "stream_kb", "us/phase", "timeouts wrong xcc-mismatch");
    const int pays[] = {16, 32, 64, 128, 256};
    for (int stream_kb : {0, 48}) {
        for (int mode : {0, 1}) {
            for (int nld : {4, 32}) {
                if (mode == 0 && nld != 4) continue;
                for (int pk : pays) {
                    Args A;
                    A.buf = buf; A.loc = loc; A.w = w; A.w_bytes = wbytes; A.phases = 400; A.pay_bytes = pk * 1024; A.mode = mode;
                    A.nld = nld; A.stream_kb = stream_kb; A.err = err; A.salt = 0;
                    CK(hipMemset(buf, 0xff, 4 * maxP)); CK(hipMemset(loc, 0xff, 32 * maxP)); CK(hipMemset(err, 0, 16));
                    float us = 0.f;
                    switch (pk) {
                        case 16: us = run<2>(A, lds); break;
                        case 32: us = run<4>(A, lds); break;
                        case 64: us = run<8>(A, lds); break;
                        case 128: us = run<16>(A, lds); break;
                        default: us = run<32>(A, lds); break;
                    }
                    int he[4];
                    CK(hipMemcpy(he, err, 16, hipMemcpyDeviceToHost));
                    printf("%-8d %-6d %-5d %-10d %10.2f   %d %d %d\n", pk, mode, mode ? nld : 0, stream_kb, us, he[0], he[1], he[2]);
                    fflush(stdout);
                }
            }
        }
    }
    return 0;
}
