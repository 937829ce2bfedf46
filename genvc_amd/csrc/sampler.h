// Per-step sampling: the body of NewGenerationMixin.sample_stream (reference
// layers/stream_generator.py:834-874) with HF's RepetitionPenalty / Temperature / TopK / TopP
// processors restated on the device.  One workgroup per stream, logits staged in LDS.
#pragma once
#include <stdlib.h>

#include "common.h"

namespace gvc {

struct SampleCall {
    const float* logits;     // [B][vocab]
    int B;
    int32_t* ids;            // [B][ids_stride] input_ids rows of the reference loop
    int ids_stride;
    int32_t* ids_len;        // [B]
    int32_t* finished;       // [B]
    gvc_sample_params p;
    int step;                // RNG counter (used when step_ptr is null)
    const int32_t* step_ptr; // device step counter of the generation loop (nullable)
    int32_t* tok_out;        // [B] token of this step (input of the next decode step)
    int32_t* tokens_out;     // nullable: [B][tok_stride], written at i0 + step
    int tok_stride, i0;
    const float* latent_src; // nullable: [B][d] latent that predicted this token
    float* latents_out;      // nullable: [B][lat_stride][d]
    int lat_stride, d;
};

int launch_sample(const SampleCall& sc, hipStream_t s);
// the call parameters live in device memory (graph replay)
// greedy: the call has top_k == 1 (sample_greedy_ok): the argmax-only kernel
int launch_sample_indirect(const SampleCall* sc_dev, int B, bool greedy, hipStream_t s);
static inline bool sample_greedy_ok(int top_k, int d) {
    return top_k == 1 && d % 4 == 0;
}

}  // namespace gvc
