#include "sampler.h"

namespace gvc {

constexpr int kSortN = 2048;          // vocab (1026) padded to a power of two
constexpr int kSampThreads = 1024;

__device__ __forceinline__ float rng_uniform(uint64_t seed, uint64_t step, uint64_t row) {
    // same integer hash as oracle/genvc_oracle.py:rng_uniform
    uint64_t x = seed * 0x9E3779B97F4A7C15ull + step * 0xBF58476D1CE4E5B9ull + row * 0x94D049BB133111EBull +
                 0x2545F4914F6CDD1Dull;
    x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
    x ^= x >> 27; x *= 0x94D049BB133111EBull;
    x ^= x >> 31;
    return (float)(x >> 40) * (1.0f / 16777216.0f);
}

__global__ __launch_bounds__(kSampThreads) void k_sample(SampleCall cv, const SampleCall* cp) {
    __shared__ float sc[kSortN];        // processed scores in vocabulary order
    __shared__ float srt[kSortN];       // descending sort of the scores
    __shared__ unsigned char seen[kSortN];
    __shared__ float red_v[16];
    __shared__ int red_i[16];
    __shared__ int s_tok;
    const SampleCall& C = cp ? *cp : cv;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int V = C.p.vocab;
    const int step = C.step_ptr ? *C.step_ptr : C.step;
    const float* lg = C.logits + (size_t)b * V;
    int32_t* ids = C.ids + (size_t)b * C.ids_stride;
    const int len = C.ids_len[b];

    for (int i = tid; i < kSortN; i += kSampThreads) seen[i] = 0;
    __syncthreads();
    for (int i = tid; i < len; i += kSampThreads) {
        const int id = ids[i];
        if (id >= 0 && id < V) seen[id] = 1;
    }
    __syncthreads();
    // RepetitionPenalty (every id of input_ids incl. the fake prefix, once) then Temperature
    for (int i = tid; i < kSortN; i += kSampThreads) {
        float v = -INFINITY;
        if (i < V) {
            v = lg[i];
            if (seen[i]) v = v < 0.f ? v * C.p.repetition_penalty : v / C.p.repetition_penalty;
            v = v / C.p.temperature;
        }
        sc[i] = v;
        srt[i] = v;
    }
    __syncthreads();

    int tok;
    if (C.p.top_k == 1) {
        // exactly one candidate survives TopK(1): argmax of the penalised scores, first index on ties
        float bv = -INFINITY; int bi = 0x7fffffff;
        for (int i = tid; i < V; i += kSampThreads)
            if (sc[i] > bv || (sc[i] == bv && i < bi)) { bv = sc[i]; bi = i; }
        for (int off = 32; off > 0; off >>= 1) {
            const float ov = __shfl_xor(bv, off);
            const int oi = __shfl_xor(bi, off);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
        }
        if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < kSampThreads / 64; ++w)
                if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
            s_tok = bi;
        }
        __syncthreads();
        tok = s_tok;
    } else {
        // bitonic sort, descending
        for (int k = 2; k <= kSortN; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < kSortN; i += kSampThreads) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const float a = srt[i], c = srt[ixj];
                        const bool desc = (i & k) == 0;
                        if (desc ? (a < c) : (a > c)) { srt[i] = c; srt[ixj] = a; }
                    }
                }
                __syncthreads();
            }
        }
        if (tid == 0) {
            // TopK: keep scores >= k-th largest (ties kept); TopP over the ascending cumulative softmax
            float thresh = -INFINITY;
            if (C.p.top_k > 0 && C.p.top_k < V) thresh = srt[C.p.top_k - 1];
            int nk = 0;
            while (nk < V && srt[nk] >= thresh && srt[nk] > -INFINITY) ++nk;
            const float mx = srt[0];
            if (C.p.top_p < 1.0f && nk > 1) {
                float Z = 0.f;
                for (int i = nk - 1; i >= 0; --i) Z += expf(srt[i] - mx);
                // ascending cumsum: element i (descending index) is removed when the mass of all
                // elements <= it, itself included, is <= 1 - top_p; the largest is always kept
                float cum = 0.f;
                int keep = nk;
                for (int i = nk - 1; i >= 1; --i) {
                    cum += expf(srt[i] - mx) / Z;
                    if (cum <= 1.0f - C.p.top_p) keep = i; else break;
                }
                thresh = srt[keep - 1];
            }
            // draw: first kept vocabulary index whose running mass reaches u * total
            double total = 0.0;
            for (int i = 0; i < V; ++i)
                if (sc[i] >= thresh) total += (double)expf(sc[i] - mx);
            // the RNG counter is the position of the step in the whole run (i0 + step), not in this call
            const double target = (double)rng_uniform(C.p.seed, (uint64_t)(C.i0 + step), (uint64_t)b) * total;
            double acc = 0.0;
            int pick = -1, lastk = 0;
            for (int i = 0; i < V; ++i) {
                if (sc[i] >= thresh) {
                    acc += (double)expf(sc[i] - mx);
                    lastk = i;
                    if (acc >= target) { pick = i; break; }
                }
            }
            s_tok = pick >= 0 ? pick : lastk;
        }
        __syncthreads();
        tok = s_tok;
    }

    // finished rows emit the pad (= eos) token (stream_generator.py:861-864, 872-874); a row whose ids buffer is
    // full is finished too (the caller sized it for the whole run: nothing past it can be accounted for)
    if (C.finished[b] || len >= C.ids_stride) tok = C.p.eos_token;
    __syncthreads();
    if (tid == 0) {
        if (len < C.ids_stride) { ids[len] = tok; C.ids_len[b] = len + 1; }
        if (tok == C.p.eos_token) C.finished[b] = 1;
        C.tok_out[b] = tok;
        if (C.tokens_out) C.tokens_out[(size_t)b * C.tok_stride + C.i0 + step] = tok;
    }
    if (C.latents_out && C.latent_src) {
        const float* src = C.latent_src + (size_t)b * C.d;
        float* dst = C.latents_out + ((size_t)b * C.lat_stride + C.i0 + step) * C.d;
        for (int k = tid; k < C.d; k += kSampThreads) dst[k] = src[k];
    }
}

int launch_sample(const SampleCall& sc, hipStream_t s) {
    GVC_REQUIRE(sc.p.vocab > 0 && sc.p.vocab <= kSortN, GVC_ERR_UNSUPPORTED, "sample: vocab %d > %d", sc.p.vocab, kSortN);
    hipLaunchKernelGGL(k_sample, dim3(sc.B), dim3(kSampThreads), 0, s, sc, (const SampleCall*)nullptr);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

int launch_sample_indirect(const SampleCall* sc_dev, int B, hipStream_t s) {
    SampleCall dummy;
    memset(&dummy, 0, sizeof(dummy));
    hipLaunchKernelGGL(k_sample, dim3(B), dim3(kSampThreads), 0, s, dummy, sc_dev);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

}  // namespace gvc

extern "C" int gvc_sample(const float* logits, int32_t B, int32_t* ids, int32_t ids_stride, int32_t* ids_len,
                          int32_t* finished, const gvc_sample_params* p, int32_t step, int32_t* tok_out,
                          gvc_stream s) {
    GVC_REQUIRE(logits && ids && ids_len && finished && p && tok_out && B >= 1, GVC_ERR_ARG, "gvc_sample: bad argument");
    gvc::SampleCall sc;
    memset(&sc, 0, sizeof(sc));
    sc.logits = logits; sc.B = B; sc.ids = ids; sc.ids_stride = ids_stride; sc.ids_len = ids_len;
    sc.finished = finished; sc.p = *p; sc.step = step; sc.tok_out = tok_out;
    return gvc::launch_sample(sc, (hipStream_t)s);
}
