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

// exclusive prefix sum of one value per thread over the workgroup (kSampThreads = 16 waves) in thread order; *total = sum
template <typename T>
__device__ __forceinline__ T block_scan_excl(T v, T* scr /*[17]*/, T* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    T inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const T o = __shfl_up(inc, off);
        if (lane >= off) inc += o;
    }
    __syncthreads();                         // scr may still be read from a previous scan
    if (lane == 63) scr[wave] = inc;
    __syncthreads();
    if (wave == 0) {
        T w = lane < kSampThreads / 64 ? scr[lane] : T(0);
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            const T o = __shfl_up(w, off);
            if (lane >= off) w += o;
        }
        if (lane < kSampThreads / 64) scr[lane] = w;
    }
    __syncthreads();
    const T base = wave > 0 ? scr[wave - 1] : T(0);
    *total = scr[kSampThreads / 64 - 1];
    return base + inc - v;
}

__global__ __launch_bounds__(kSampThreads) void k_sample(SampleCall cv, const SampleCall* cp) {
    __shared__ float sc[kSortN];        // processed scores in vocabulary order
    __shared__ float srt[kSortN];       // descending sort of the scores
    __shared__ unsigned char seen[kSortN];
    __shared__ float red_v[16];
    __shared__ int red_i[16];
    __shared__ int s_tok, s_nk, s_pick, s_last;
    __shared__ float fscr[17];
    __shared__ int iscr[17];
    __shared__ double dscr[17];
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
            s_tok = bi < V ? bi : 0;          // (all scores NaN -- the step before produced garbage: any valid id, never an out-of-range one)
        }
        __syncthreads();
        tok = s_tok;
    } else {
        // bitonic sort, descending.  Thread t keeps elements t and t + 1024 in registers; a partner at distance j < 64 is a
        // lane of the same wave (shuffle, no barrier), j = 1024 is the thread's own second element, and only the 14 stages
        // with 64 <= j <= 512 go through LDS (the 66 LDS + barrier stages of the plain version cost ~30 us per step).
        {
            float v0 = srt[tid], v1 = srt[tid + kSampThreads];
            for (int k = 2; k <= kSortN; k <<= 1) {
                for (int j = k >> 1; j > 0; j >>= 1) {
                    const bool d0 = (tid & k) == 0, d1 = ((tid + kSampThreads) & k) == 0;
                    if (j == kSampThreads) {                       // only k == 2048: i = tid is the low index, descending
                        const float hi = fmaxf(v0, v1), lo = fminf(v0, v1);
                        v0 = hi; v1 = lo;
                    } else if (j >= 64) {
                        __syncthreads();
                        srt[tid] = v0; srt[tid + kSampThreads] = v1;
                        __syncthreads();
                        const float p0 = srt[tid ^ j], p1 = srt[(tid ^ j) + kSampThreads];
                        const bool low = (tid & j) == 0;
                        v0 = (low == d0) ? fmaxf(v0, p0) : fminf(v0, p0);
                        v1 = (low == d1) ? fmaxf(v1, p1) : fminf(v1, p1);
                    } else {
                        const float p0 = __shfl_xor(v0, j), p1 = __shfl_xor(v1, j);
                        const bool low = (tid & j) == 0;
                        v0 = (low == d0) ? fmaxf(v0, p0) : fminf(v0, p0);
                        v1 = (low == d1) ? fmaxf(v1, p1) : fminf(v1, p1);
                    }
                }
            }
            __syncthreads();
            srt[tid] = v0; srt[tid + kSampThreads] = v1;
            __syncthreads();
        }
        // Everything below is workgroup-parallel (a single lane doing the top-p / inverse-CDF loops over the vocabulary cost
        // ~120 us per step at top_k = 15 and ~330 us without top-k).  Thread t owns the element pair (2t, 2t+1).
        const float mx = srt[0];
        // TopK: keep scores >= k-th largest (ties kept); nk = how many lead the descending order
        float thresh = -INFINITY;
        if (C.p.top_k > 0 && C.p.top_k < V) thresh = srt[C.p.top_k - 1];
        if (tid == 0) s_nk = 0;
        __syncthreads();
        for (int i = tid; i < kSortN; i += kSampThreads) {
            const bool in = srt[i] >= thresh && srt[i] > -INFINITY;
            const bool nxt = i + 1 < kSortN && srt[i + 1] >= thresh && srt[i + 1] > -INFINITY;
            if (in && !nxt) s_nk = i + 1;                       // exactly one boundary in a sorted array
        }
        __syncthreads();
        const int nk = s_nk;
        if (C.p.top_p < 1.0f && nk > 1) {
            // ascending order j = 0..nk-1 <-> descending index nk-1-j; p_j = exp(s - max) / Z; drop the leading run with
            // cumulative mass <= 1 - top_p, always keeping the largest
            const int j0 = 2 * tid, j1 = 2 * tid + 1;
            const float e0 = j0 < nk ? expf(srt[nk - 1 - j0] - mx) : 0.f;
            const float e1 = j1 < nk ? expf(srt[nk - 1 - j1] - mx) : 0.f;
            float Z;
            (void)block_scan_excl<float>(e0 + e1, fscr, &Z);
            const float q0 = e0 / Z, q1 = e1 / Z;
            float tot;
            const float ex = block_scan_excl<float>(q0 + q1, fscr, &tot);
            const float c0 = ex + q0, c1 = c0 + q1;
            int removed = 0;
            if (j0 < nk - 1 && c0 <= 1.0f - C.p.top_p) ++removed;
            if (j1 < nk - 1 && c1 <= 1.0f - C.p.top_p) ++removed;
            int nrem;
            (void)block_scan_excl<int>(removed, iscr, &nrem);
            thresh = srt[nk - nrem - 1];
        }
        // draw: first kept vocabulary index whose running mass (double, vocabulary order) reaches u * total
        {
            const int i0v = 2 * tid, i1v = 2 * tid + 1;
            const bool k0 = i0v < V && sc[i0v] >= thresh, k1 = i1v < V && sc[i1v] >= thresh;
            const double w0 = k0 ? (double)expf(sc[i0v] - mx) : 0.0, w1 = k1 ? (double)expf(sc[i1v] - mx) : 0.0;
            double total;
            const double ex = block_scan_excl<double>(w0 + w1, dscr, &total);
            // the RNG counter is the position of the step in the whole run (i0 + step), not in this call
            const double target = (double)rng_uniform(C.p.seed, (uint64_t)(C.i0 + step), (uint64_t)b) * total;
            const double a0 = ex + w0, a1 = a0 + w1;
            int pick = 0x7fffffff, lastk = -1;
            if (k0) { lastk = i0v; if (a0 >= target) pick = i0v; }
            if (k1) { lastk = i1v; if (a1 >= target && pick == 0x7fffffff) pick = i1v; }
            if (tid == 0) { s_pick = 0x7fffffff; s_last = -1; }
            __syncthreads();
            if (pick != 0x7fffffff) atomicMin(&s_pick, pick);
            if (lastk >= 0) atomicMax(&s_last, lastk);
            __syncthreads();
            if (tid == 0) s_tok = s_pick != 0x7fffffff ? s_pick : (s_last >= 0 ? s_last : 0);
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

// top_k = 1 (the configuration of every BASELINE workload that fixes top_k: TopK(1) leaves one candidate, so top-p and the draw are
// no-ops): RepetitionPenalty -> Temperature -> argmax with the arithmetic of k_sample, on 256 threads -- four waves meet at four
// barriers instead of sixteen at six, and nothing is sorted (7.7 -> ~5 us per decode step)
constexpr int kGreedyThreads = 256;
__global__ __launch_bounds__(kGreedyThreads) void k_sample_greedy(SampleCall cv, const SampleCall* cp) {
    __shared__ unsigned seen_w[kSortN / 4];          // one byte per vocabulary entry
    __shared__ float red_v[kGreedyThreads / 64];
    __shared__ int red_i[kGreedyThreads / 64];
    __shared__ int s_tok;
    unsigned char* seen = reinterpret_cast<unsigned char*>(seen_w);
    const SampleCall& C = cp ? *cp : cv;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int V = C.p.vocab;
    const int step = C.step_ptr ? *C.step_ptr : C.step;
    const float* lg = C.logits + (size_t)b * V;
    int32_t* ids = C.ids + (size_t)b * C.ids_stride;
    const int len = C.ids_len[b];
    // this thread's logits are requested before the id pass (they do not depend on it)
    constexpr int PER = kSortN / kGreedyThreads;
    float v[PER];
#pragma unroll
    for (int u = 0; u < PER; ++u) { const int i = tid + u * kGreedyThreads; v[u] = i < V ? lg[i] : -INFINITY; }
    for (int i = tid; i < kSortN / 4; i += kGreedyThreads) seen_w[i] = 0u;
    __syncthreads();
    for (int i = tid; i < len; i += kGreedyThreads) {
        const int id = ids[i];
        if (id >= 0 && id < V) seen[id] = 1;
    }
    __syncthreads();
    float bv = -INFINITY; int bi = 0x7fffffff;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = tid + u * kGreedyThreads;
        if (i < V) {
            float x = v[u];
            if (seen[i]) x = x < 0.f ? x * C.p.repetition_penalty : x / C.p.repetition_penalty;
            x = x / C.p.temperature;
            if (x > bv || (x == bv && i < bi)) { bv = x; bi = i; }
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    if ((tid & 63) == 0) { red_v[tid >> 6] = bv; red_i[tid >> 6] = bi; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < kGreedyThreads / 64; ++w)
            if (red_v[w] > bv || (red_v[w] == bv && red_i[w] < bi)) { bv = red_v[w]; bi = red_i[w]; }
        int tok = bi < V ? bi : 0;            // (all scores NaN -- the step before produced garbage: any valid id, never an out-of-range one)
        // finished rows emit the pad (= eos) token; a row whose ids buffer is full is finished too (see k_sample)
        if (C.finished[b] || len >= C.ids_stride) tok = C.p.eos_token;
        if (len < C.ids_stride) { ids[len] = tok; C.ids_len[b] = len + 1; }
        if (tok == C.p.eos_token) C.finished[b] = 1;
        C.tok_out[b] = tok;
        if (C.tokens_out) C.tokens_out[(size_t)b * C.tok_stride + C.i0 + step] = tok;
    }
    if (C.latents_out && C.latent_src) {
        const float* src = C.latent_src + (size_t)b * C.d;
        float* dst = C.latents_out + ((size_t)b * C.lat_stride + C.i0 + step) * C.d;
        for (int k = tid * 4; k < C.d; k += kGreedyThreads * 4) *reinterpret_cast<float4*>(dst + k) = *reinterpret_cast<const float4*>(src + k);
    }
}
int launch_sample(const SampleCall& sc, hipStream_t s) {
    GVC_REQUIRE(sc.p.vocab > 0 && sc.p.vocab <= kSortN, GVC_ERR_UNSUPPORTED, "sample: vocab %d > %d", sc.p.vocab, kSortN);
    if (sample_greedy_ok(sc.p.top_k, sc.latents_out ? sc.d : 0)) hipLaunchKernelGGL(k_sample_greedy, dim3(sc.B), dim3(kGreedyThreads), 0, s, sc, (const SampleCall*)nullptr);
    else hipLaunchKernelGGL(k_sample, dim3(sc.B), dim3(kSampThreads), 0, s, sc, (const SampleCall*)nullptr);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}

int launch_sample_indirect(const SampleCall* sc_dev, int B, bool greedy, hipStream_t s) {
    SampleCall dummy;
    memset(&dummy, 0, sizeof(dummy));
    if (greedy) hipLaunchKernelGGL(k_sample_greedy, dim3(B), dim3(kGreedyThreads), 0, s, dummy, sc_dev);
    else hipLaunchKernelGGL(k_sample, dim3(B), dim3(kSampThreads), 0, s, dummy, sc_dev);
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
