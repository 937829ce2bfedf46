// MFMA flash attention for head_dim 64 shared by ContentVec (csrc/hubert.hip) and the Perceiver (csrc/perceiver.hip).
#pragma once
#include "gemm.h"

namespace gvc {

typedef float hb_f32x4 __attribute__((ext_vector_type(4)));

// Non-causal multi-head attention, head_dim 64, fp32, on v_mfma_f32_16x16x4_f32.  q / k / v: row-major matrices with a common row
// stride `ld` and batch stride `bs` (head h at columns h*64 of each); query rows 0 .. Tq-1, key rows 0 .. Tk-1 of a batch element;
// out [B * out_rows][E] row-major or FM16 (row b * out_rows + query).  Users: ContentVec's self-attention (q | k | v side by side in
// one [T][3E] buffer, Tq = Tk = T, padding-frame mask) and the Perceiver's cross-attention (32 latent queries over latents + context).
// grid (ceil(Tq/16), H, B); the NW waves of a workgroup (4: ContentVec, hundreds of workgroups; 16: the Perceiver, whose 2 query
// tiles x 8 heads are all the workgroups there are) share 16 queries and take every NW-th 16-key tile, merged
// through LDS at the end.  Per tile:   S^T[key][q] = sum_d K[key][d] Q[q][d]      (A = K fragment, B = Q fragment)
//                                      O^T[d][q] += sum_key V[key][d] P[key][q]   (A = V fragment, B = P = the S^T registers)
// lane (r = lane%16, g = lane/16) holds S^T rows key0 + 4g + i (i = 0..3) for query q0 + r, which is exactly the
// B-operand layout of the second product when its k-step i is mapped to keys {key0 + 4g + i}.  The d index of the
// first product is permuted (d = 16s + 4g + comp) and the row index of O^T is permuted (row m of tile mt <-> d = 4m + mt)
// so that every fragment load is a float4.
// MASK: fmask [B][Tk], keys of padding frames are excluded (fairseq MultiheadAttention key_padding_mask: scores -> -inf)
template <bool MASK, int NW = 4>
__global__ __launch_bounds__(NW * 64) void k_attn64_mfma(const float* qb, const float* kb, const float* vb, long long ld, long long bs, int Tq,
                                                     int Tk, float* out, int out_rows, int E, float scale, int out_fm16, const int32_t* fmask) {
    __shared__ float sm[NW][16], sl[NW][16];
    __shared__ __attribute__((aligned(16))) float so[NW][16][68];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int q0 = blockIdx.x * 16, h = blockIdx.y, b = blockIdx.z;
    const float* base = qb + (size_t)b * bs + h * 64;
    const float* kbase = kb + (size_t)b * bs + h * 64;
    const float* vbase = vb + (size_t)b * bs + h * 64;
    const int T = Tk;
    const int qi = min(q0 + r, Tq - 1);
    float4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        qf[s] = *reinterpret_cast<const float4*>(base + (size_t)qi * ld + 16 * s + 4 * g);
        qf[s].x *= scale; qf[s].y *= scale; qf[s].z *= scale; qf[s].w *= scale;
    }
    hb_f32x4 o[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) o[mt] = {0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const int ntile = (T + 15) >> 4;
    for (int kt = wave; kt < ntile; kt += NW) {
        const int key0 = kt * 16;
        const int kr = min(key0 + r, T - 1);
        float4 kf[4], vf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) kf[s] = *reinterpret_cast<const float4*>(kbase + (size_t)kr * ld + 16 * s + 4 * g);
        int kmask[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vr = min(key0 + 4 * g + i, T - 1);
            vf[i] = *reinterpret_cast<const float4*>(vbase + (size_t)vr * ld + 4 * r);
            kmask[i] = MASK ? fmask[b * T + vr] : 0;
        }
        hb_f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].x, qf[s].x, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].y, qf[s].y, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].z, qf[s].z, st, 0, 0, 0);
            st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[s].w, qf[s].w, st, 0, 0, 0);
        }
        float p[4];
        float tmax = -INFINITY;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            p[i] = (key0 + 4 * g + i < T && !kmask[i]) ? st[i] : -INFINITY;
            tmax = fmaxf(tmax, p[i]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 16));
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32));
        const float mnew = fmaxf(m, tmax);
        // a tile (or everything so far) made of padding keys only leaves mnew at -inf: nothing to add, nothing to rescale
        const bool none = mnew == -INFINITY;
        const float alpha = none ? 1.0f : expf(m - mnew);
        float psum = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) { p[i] = none ? 0.f : expf(p[i] - mnew); psum += p[i]; }
        l = l * alpha + psum;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) o[mt] *= alpha;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            o[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[i].x, p[i], o[0], 0, 0, 0);
            o[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[i].y, p[i], o[1], 0, 0, 0);
            o[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[i].z, p[i], o[2], 0, 0, 0);
            o[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[i].w, p[i], o[3], 0, 0, 0);
        }
        m = mnew;
    }
    // per-lane l covers this lane's keys only
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    if (g == 0) { sm[wave][r] = m; sl[wave][r] = l; }
#pragma unroll
    for (int i = 0; i < 4; ++i)      // o[mt][i] = O[q0 + r][16g + 4i + mt]
        *reinterpret_cast<float4*>(&so[wave][r][16 * g + 4 * i]) = make_float4(o[0][i], o[1][i], o[2][i], o[3][i]);
    __syncthreads();
    const int qr = tid >> 4, dc = (tid & 15) * 4;
    if (tid < 256 && q0 + qr < Tq) {
        float M = sm[0][qr];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, sm[w][qr]);
        float L = 0.f;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            const float f = sm[w][qr] == -INFINITY ? 0.f : expf(sm[w][qr] - M);
            L += sl[w][qr] * f;
            const float4 ov = *reinterpret_cast<const float4*>(&so[w][qr][dc]);
            acc.x += ov.x * f; acc.y += ov.y * f; acc.z += ov.z * f; acc.w += ov.w * f;
        }
        const float inv = 1.0f / L;
        acc.x *= inv; acc.y *= inv; acc.z *= inv; acc.w *= inv;
        if (out_fm16) *reinterpret_cast<float4*>(out + fm16_index(b * out_rows + q0 + qr, h * 64 + dc, E)) = acc;
        else *reinterpret_cast<float4*>(out + ((size_t)b * out_rows + q0 + qr) * E + h * 64 + dc) = acc;
    }
}

}  // namespace gvc
