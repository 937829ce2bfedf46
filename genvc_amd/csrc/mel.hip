// Log-mel front end (reference utils.py:97-162 -> torchaudio.transforms.MelSpectrogram with
// n_fft 2048, hop 256, win 1024 (periodic Hann, zero-padded and centred in the FFT frame), centre=True
// with reflect padding, power 2, slaney-normalised HTK mel filterbank, then log(clamp(., 1e-5)) / norms).
//
// One workgroup per (frame, batch element): the windowed frame is written bit-reversed into LDS, an
// in-LDS radix-2 FFT (11 stages for 2048 points) produces the spectrum, |X|^2 is reduced against the
// triangular filters (each mel bin touches a short contiguous bin range, stored compactly), and the
// log-mel column is written in both the reference layout [B,80,F] and frames-major [B,F,80].
// Bytes: 4*T in, 2*4*80*F out per batch element; launch/latency-bound at GenVC sizes (282 frames for 3 s).
#include <math.h>

#include <vector>

#include "common.h"

namespace gvc {

struct MelDev {
    int n_fft, log2n, hop, win, n_mels, n_bins;
    const float* window;      // [win]
    const float2* twiddle;    // [n_fft/2]  exp(-2 pi i k / n_fft)
    const int* fb_lo;         // [n_mels] first bin of the filter
    const int* fb_len;        // [n_mels]
    const int* fb_off;        // [n_mels] offset into fb_w
    const float* fb_w;        // concatenated non-zero filter weights
    const float* norms;       // [n_mels]
};

__global__ __launch_bounds__(256) void k_mel(const MelDev M, const float* wav, int T, int n_frames, float* out,
                                             float* out_fm) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* re = lds;
    float* im = lds + M.n_fft;
    const int frame = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float* x = wav + (size_t)b * T;
    const int pad = M.n_fft / 2;
    const int woff = (M.n_fft - M.win) / 2;

    // windowed, reflect-padded frame, stored at the bit-reversed index
    for (int n = tid; n < M.n_fft; n += 256) {
        float v = 0.f;
        const int wn = n - woff;
        if (wn >= 0 && wn < M.win) {
            int j = frame * M.hop + n - pad;
            if (j < 0) j = -j;
            if (j >= T) j = 2 * (T - 1) - j;
            v = x[j] * M.window[wn];
        }
        const int r = __brev((unsigned)n) >> (32 - M.log2n);
        re[r] = v;
        im[r] = 0.f;
    }
    __syncthreads();
    // decimation-in-time butterflies
    const int half_n = M.n_fft >> 1;
    for (int s = 0; s < M.log2n; ++s) {
        const int half = 1 << s;
        const int tw_stride = half_n >> s;
        for (int j = tid; j < half_n; j += 256) {
            const int pos = j & (half - 1);
            const int i0 = ((j >> s) << (s + 1)) + pos;
            const int i1 = i0 + half;
            const float2 w = M.twiddle[pos * tw_stride];
            const float br = re[i1], bi = im[i1];
            const float tr = br * w.x - bi * w.y;
            const float ti = br * w.y + bi * w.x;
            const float ar = re[i0], ai = im[i0];
            re[i0] = ar + tr; im[i0] = ai + ti;
            re[i1] = ar - tr; im[i1] = ai - ti;
        }
        __syncthreads();
    }
    // power spectrum of the one-sided bins, in place (re[k] <- |X_k|^2)
    for (int k = tid; k < M.n_bins; k += 256) {
        const float a = re[k], c = im[k];
        re[k] = a * a + c * c;
    }
    __syncthreads();
    if (tid < M.n_mels) {
        const int lo = M.fb_lo[tid], len = M.fb_len[tid];
        const float* w = M.fb_w + M.fb_off[tid];
        float acc = 0.f;
        for (int i = 0; i < len; ++i) acc = fmaf(re[lo + i], w[i], acc);
        const float v = logf(fmaxf(acc, 1e-5f)) / M.norms[tid];
        out[((size_t)b * M.n_mels + tid) * n_frames + frame] = v;
        if (out_fm) out_fm[((size_t)b * n_frames + frame) * M.n_mels + tid] = v;
    }
}

}  // namespace gvc

using namespace gvc;

struct gvc_mel {
    MelDev dev;
    void* blob = nullptr;
};

extern "C" int gvc_mel_create(int32_t n_fft, int32_t hop, int32_t win, int32_t sample_rate, float f_min, float f_max,
                              int32_t n_mels, const float* mel_norms_host, gvc_mel** out) {
    GVC_REQUIRE(out && mel_norms_host, GVC_ERR_ARG, "gvc_mel_create: null argument");
    int log2n = 0;
    while ((1 << log2n) < n_fft) ++log2n;
    GVC_REQUIRE((1 << log2n) == n_fft && n_fft >= 64 && n_fft <= 8192 && win <= n_fft && hop >= 1 && n_mels <= 256,
                GVC_ERR_UNSUPPORTED, "mel: n_fft must be a power of two in [64, 8192], win <= n_fft, n_mels <= 256");
    const int n_bins = n_fft / 2 + 1;
    // periodic Hann window, twiddles (double precision on the host)
    std::vector<float> window(win);
    for (int n = 0; n < win; ++n) window[n] = (float)(0.5 - 0.5 * cos(2.0 * M_PI * n / win));
    std::vector<float> tw(n_fft);
    for (int k = 0; k < n_fft / 2; ++k) {
        tw[2 * k] = (float)cos(-2.0 * M_PI * k / n_fft);
        tw[2 * k + 1] = (float)sin(-2.0 * M_PI * k / n_fft);
    }
    // torchaudio.functional.melscale_fbanks(norm="slaney", mel_scale="htk")
    auto hz2mel = [](double f) { return 2595.0 * log10(1.0 + f / 700.0); };
    auto mel2hz = [](double m) { return 700.0 * (pow(10.0, m / 2595.0) - 1.0); };
    std::vector<double> fpts(n_mels + 2);
    const double m_lo = hz2mel(f_min), m_hi = hz2mel(f_max);
    for (int i = 0; i < n_mels + 2; ++i) fpts[i] = mel2hz(m_lo + (m_hi - m_lo) * i / (n_mels + 1));
    std::vector<int> lo(n_mels), len(n_mels), off(n_mels);
    std::vector<float> wts;
    for (int m = 0; m < n_mels; ++m) {
        const double enorm = 2.0 / (fpts[m + 2] - fpts[m]);
        int first = -1, last = -1;
        std::vector<float> row(n_bins);
        for (int k = 0; k < n_bins; ++k) {
            const double f = (double)(sample_rate / 2) * k / (n_bins - 1);
            const double down = (f - fpts[m]) / (fpts[m + 1] - fpts[m]);
            const double up = (fpts[m + 2] - f) / (fpts[m + 2] - fpts[m + 1]);
            const double v = fmax(0.0, fmin(down, up)) * enorm;
            row[k] = (float)v;
            if (v > 0.0) { if (first < 0) first = k; last = k; }
        }
        lo[m] = first < 0 ? 0 : first;
        len[m] = first < 0 ? 0 : last - first + 1;
        off[m] = (int)wts.size();
        for (int k = 0; k < len[m]; ++k) wts.push_back(row[lo[m] + k]);
    }
    // one device blob
    auto al = [](size_t n) { return (n + 15) & ~(size_t)15; };
    const size_t o_win = 0, o_tw = al(o_win + win * 4), o_lo = al(o_tw + n_fft * 4), o_len = al(o_lo + n_mels * 4),
                 o_off = al(o_len + n_mels * 4), o_w = al(o_off + n_mels * 4), o_nrm = al(o_w + wts.size() * 4),
                 total = al(o_nrm + n_mels * 4);
    std::vector<char> host(total, 0);
    memcpy(&host[o_win], window.data(), win * 4);
    memcpy(&host[o_tw], tw.data(), n_fft * 4);
    memcpy(&host[o_lo], lo.data(), n_mels * 4);
    memcpy(&host[o_len], len.data(), n_mels * 4);
    memcpy(&host[o_off], off.data(), n_mels * 4);
    memcpy(&host[o_w], wts.data(), wts.size() * 4);
    memcpy(&host[o_nrm], mel_norms_host, n_mels * 4);
    auto* c = new gvc_mel();
    GVC_CHECK_HIP(hipMalloc(&c->blob, total));
    GVC_CHECK_HIP(hipMemcpy(c->blob, host.data(), total, hipMemcpyHostToDevice));
    char* base = (char*)c->blob;
    c->dev.n_fft = n_fft; c->dev.log2n = log2n; c->dev.hop = hop; c->dev.win = win; c->dev.n_mels = n_mels;
    c->dev.n_bins = n_bins;
    c->dev.window = (const float*)(base + o_win); c->dev.twiddle = (const float2*)(base + o_tw);
    c->dev.fb_lo = (const int*)(base + o_lo); c->dev.fb_len = (const int*)(base + o_len);
    c->dev.fb_off = (const int*)(base + o_off); c->dev.fb_w = (const float*)(base + o_w);
    c->dev.norms = (const float*)(base + o_nrm);
    *out = c;
    return GVC_OK;
}

extern "C" int gvc_mel_destroy(gvc_mel* c) {
    if (!c) return GVC_OK;
    if (c->blob) hipFree(c->blob);
    delete c;
    return GVC_OK;
}

extern "C" int gvc_mel_forward(gvc_mel* c, const float* wav, int32_t B, int32_t T, float* out, float* out_fm,
                               gvc_stream sv) {
    GVC_REQUIRE(c && wav && out && B >= 1, GVC_ERR_ARG, "gvc_mel_forward: bad argument");
    GVC_REQUIRE(T > c->dev.n_fft / 2, GVC_ERR_ARG, "mel: %d samples are too few for reflect padding of %d", T,
                c->dev.n_fft / 2);
    const int n_frames = 1 + T / c->dev.hop;
    const size_t lds = 2 * (size_t)c->dev.n_fft * sizeof(float);
    hipLaunchKernelGGL(k_mel, dim3(n_frames, B), dim3(256), lds, (hipStream_t)sv, c->dev, wav, T, n_frames, out, out_fm);
    GVC_LAUNCH_CHECK();
    return GVC_OK;
}
