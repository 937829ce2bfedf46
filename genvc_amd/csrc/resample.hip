// Polyphase windowed-sinc resampler: torchaudio.functional.resample with its defaults
// (sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99) as called by utils.load_audio (reference utils.py:58-62).
// out[i*n + p] = sum_j x[i*o + j - width] * h[p][j]; the n x (2*width + o) filter bank is built on the host in
// double precision, one thread per output sample.  Bytes: 4*(T_in + T_out); latency-bound at utterance sizes.
#include <math.h>

#include <vector>

#include "common.h"

namespace gvc {
__global__ void k_resample(const float* x, int T, const float* h, int o, int n, int width, int klen, float* out, int T_out) {
    const int b = blockIdx.y;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T_out) return;
    const int i = idx / n, p = idx - i * n;
    const float* xb = x + (size_t)b * T;
    const float* hp = h + (size_t)p * klen;
    const int base = i * o - width;
    float acc = 0.f;
    for (int j = 0; j < klen; ++j) {
        const int t = base + j;
        if (t >= 0 && t < T) acc = fmaf(xb[t], hp[j], acc);
    }
    out[(size_t)b * T_out + idx] = acc;
}
}  // namespace gvc

static int gcd_i(int a, int b) { while (b) { int t = a % b; a = b; b = t; } return a; }

extern "C" int gvc_resample_length(int32_t T, int32_t orig_sr, int32_t new_sr) {
    const int g = gcd_i(orig_sr, new_sr);
    const long long o = orig_sr / g, n = new_sr / g;
    return (int)((n * (long long)T + o - 1) / o);
}

// x [B,T] at orig_sr -> out [B, gvc_resample_length(T)] at new_sr.  Builds its filter bank per call (small) and
// frees it after the stream has consumed it: this entry point synchronises the stream (file loading path).
extern "C" int gvc_resample(const float* x, int32_t B, int32_t T, int32_t orig_sr, int32_t new_sr, float* out, gvc_stream sv) {
    GVC_REQUIRE(x && out && B >= 1 && T >= 1 && orig_sr > 0 && new_sr > 0, GVC_ERR_ARG, "gvc_resample: bad argument");
    hipStream_t s = (hipStream_t)sv;
    const int g = gcd_i(orig_sr, new_sr);
    const int o = orig_sr / g, n = new_sr / g;
    const int T_out = gvc_resample_length(T, orig_sr, new_sr);
    if (o == n) {
        GVC_CHECK_HIP(hipMemcpyAsync(out, x, (size_t)B * T * sizeof(float), hipMemcpyDeviceToDevice, s));
        return GVC_OK;
    }
    const double lpw = 6.0, rolloff = 0.99;
    const double base = (o < n ? o : n) * rolloff;
    const int width = (int)ceil(lpw * o / base);
    const int klen = 2 * width + o;
    std::vector<float> h((size_t)n * klen);
    for (int p = 0; p < n; ++p)
        for (int j = 0; j < klen; ++j) {
            double t = (-(double)p / n + (double)(j - width) / o) * base;
            if (t < -lpw) t = -lpw;
            if (t > lpw) t = lpw;
            const double win = cos(t * M_PI / lpw / 2.0);
            const double tp = t * M_PI;
            const double sinc = tp == 0.0 ? 1.0 : sin(tp) / tp;
            h[(size_t)p * klen + j] = (float)(sinc * win * win * (base / o));
        }
    float* hd = nullptr;
    GVC_CHECK_HIP(hipMalloc((void**)&hd, h.size() * sizeof(float)));
    GVC_CHECK_HIP(hipMemcpyAsync(hd, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(gvc::k_resample, dim3(gvc::cdiv(T_out, 256), B), dim3(256), 0, s, x, T, hd, o, n, width, klen, out, T_out);
    hipError_t e = hipGetLastError();
    hipError_t e2 = hipStreamSynchronize(s);
    (void)hipFree(hd);
    GVC_CHECK_HIP(e);
    GVC_CHECK_HIP(e2);
    return GVC_OK;
}
