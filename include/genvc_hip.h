/*
 * genvc_hip.h -- C ABI of libgenvc_hip.so: the MI355X (gfx950) implementation of GenVC's
 * autoregressive codec-token generation hot path.
 *
 * The reference (caizexin/GenVC) is pure Python and has no FFI layer of its own; the seam this
 * library sits behind is the duck-typed object API that inference/inference_utils.py consumes
 * (SURVEY.md section 8b).  Each entry point below names the reference interface it replaces
 * (paths relative to the reference repo).  INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - plain C symbols, no C++ types or exceptions across the boundary;
 *   - every function returns 0 on success or a negative GVC_ERR_* code; the message of the
 *     last failure on the calling thread is available from gvc_last_error();
 *   - every launch is asynchronous on the caller's stream (a hipStream_t passed as void*);
 *     no hidden synchronisation or allocation after *_create / *_bind_*;
 *   - tensor arguments are raw DEVICE pointers borrowed from the caller (fp32 unless stated,
 *     row-major, contiguous); index arrays are DEVICE int32; the library frees only what it
 *     allocated in *_create;
 *   - a context is bound to the device that was current at *_create and must not be used
 *     from two host threads at once; contexts are independent of each other.
 *
 * Environment switches read by the library (ten; everything else that used to be a run-time knob is a constant now, and the
 * experiments that lost are gone from the kernels: profiles/r06_removed_experiments.patch).  All default to the shipped configuration.
 *   GVC_PERSIST=0            never use the one-launch steps (one stream and 2..16 rows then take the launch-per-phase paths)
 *   GVC_PERSIST_ROWS=0       2..16 rows keep the launch-per-phase rows path (the one-stream one-launch step stays on)
 *   GVC_PERSIST_XCD=0        one-stream step: device-wide hand-off of the MLP's hidden units instead of the XCD-local layout
 *                            (for partition modes in which a 256-workgroup grid is not dealt 8 x 32 over the XCDs; the topology
 *                            probe switches it off by itself when it sees such a deal)
 *   GVC_PERSIST_TEST_GRID=n  test hook: launch the one-launch steps with n < 256 workgroups (every hand-off then times out)
 *   GVC_PERSIST_STAMPS=1     in-kernel wall-clock stamps of the one-launch steps (scripts/stamps_persist.py, scripts/stamps_rows.py)
 *   GVC_DEBUG_STAMPS=1       in-kernel stamps of the launch-per-phase decode kernels (scripts/stamps.py)
 *   GVC_ROWS_DECODE_MIN=n    smallest batch that decodes on the MFMA rows path when the one-launch rows step does not serve it
 *                            (default 5; 0: never -- the 8-stream GEMV groups; tests use it to reach both paths)
 *   GVC_GRAPHS=0             ContentVec and Perceiver: eager launches instead of one hipGraph per shape
 *   GVC_VOCODER_GRAPH=0|2    HiFi-GAN: eager launches / only the conv chain as a graph (default 1: the whole call)
 *   GVC_VOCODER_SMALL_CONV=0|2  HiFi-GAN: every conv on the tiled GEMM / only the ResBlocks on k_conv_lds (the paths other shapes take)
 * Outside the library: GENVC_HIP_LIB (genvc_amd/_lib.py: another build of the library, for A/B runs), GVC_BENCH_* (bench.py dry runs).
 */
#ifndef GENVC_HIP_H
#define GENVC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GVC_OK 0
#define GVC_ERR_ARG (-1)          /* bad argument / shape */
#define GVC_ERR_HIP (-2)          /* a HIP runtime call failed */
#define GVC_ERR_STATE (-3)        /* weights missing, cache overflow, wrong call order */
#define GVC_ERR_UNSUPPORTED (-4)  /* dimension not supported by the kernels */
#define GVC_ERR_TIMEOUT (-5)      /* a hand-off of a one-launch step timed out (not all 256 workgroups resident): the previous decode / generate /
                                     cached-prefill call produced garbage; the context has switched to the launch-per-phase paths and stays usable --
                                     reset the affected slots and repeat the call (gvc_gpt_health).  Distinct from GVC_ERR_STATE (a full KV cache,
                                     missing weights), which repeating does not cure */

typedef void* gvc_stream;         /* hipStream_t */

int gvc_version(void);
const char* gvc_last_error(void);

/* ------------------------------------------------------------------------------------------
 * GPT: prefill, KV-cached decode step, head, latent re-pass.
 * Replaces layers/gpt.py:GPT.{init_gpt_for_inference,compute_embeddings(prefix store),forward
 * (return_latent=True)} (gpt.py:197-218, 375-508, 572-592), layers/gpt_inference.py:
 * GPT2InferenceModel.forward (:55-124) and the HF GPT2Model block stack it drives
 * (gpt.py:42-84, gpt_inference.py:97-110).
 * ------------------------------------------------------------------------------------------ */
typedef struct gvc_gpt gvc_gpt;

typedef struct gvc_gpt_dims {
    int32_t n_layer;      /* gpt_layers */
    int32_t d_model;      /* gpt_n_model_channels; multiple of 256 */
    int32_t n_head;       /* gpt_n_heads; head_dim = d_model / n_head must be 64, 128 or 256 */
    int32_t vocab;        /* gpt_num_audio_tokens (1026) */
    int32_t max_mel_pos;  /* rows of mel_pos_embedding (608) */
    int32_t max_text_pos; /* rows of text_pos_embedding (404) */
    int32_t n_text;       /* gpt_number_text_tokens (258) */
    int32_t max_seq;      /* KV-cache positions per slot (>= 1083, gpt.py:198) */
    int32_t max_slots;    /* concurrent streams whose KV cache is resident */
    int32_t max_rows;     /* capacity (rows) of one prefill / latent re-pass call, all slots together */
    int32_t weight_dtype; /* 0: fp32 (reference numerics). 1: the c_attn/c_proj/c_fc/mlp c_proj/mel_head matrices are
                             rounded to bf16 at bind time; the decode step and the skinny MFMA path stream bf16 copies (half the HBM bytes),
                             every product and accumulation stays fp32; KV cache and activations stay fp32.
                             2: as 1, and the KV cache holds bf16 too (k / v rounded to nearest even where they enter the cache,
                             widened to fp32 where the attention kernels use them; half the cache bytes).
                             3: as 2, and the one-launch rows step (2..16 rows that continue cached sequences: batched decode steps, cached
                             chunk prefills; csrc/persist_rows_b16.h) rounds the four activations that cross its hand-offs -- x into LN1, the
                             attention output, x' into LN2, the gelu output -- to bf16 where they are published and multiplies on bf16 MFMAs with
                             fp32 accumulation (LayerNorm gain folded into the packed bf16 weights); residual stream, softmax, statistics and q stay
                             fp32.  The other paths of such a context (one stream, full prefills, 17+ streams) compute as mode 2.  Not bit-exact
                             against any reference by construction: oracle rounding points `dims["act_bf16"]` */
} gvc_gpt_dims;

int gvc_gpt_create(const gvc_gpt_dims* dims, gvc_gpt** out);
int gvc_gpt_destroy(gvc_gpt* ctx);

/* One-time repack of a tensor named as in the reference GPT state dict, e.g.
 * "gpt.h.3.attn.c_attn.weight" (HF Conv1D [in,out] -> row-per-output layout), "mel_head.weight",
 * "mel_embedding.weight", "final_norm.bias".  `src` is a device pointer with `numel` floats.
 * Unknown names (text_head.*, attn.bias buffers, conditioning_perceiver.*) return GVC_OK and are
 * ignored, mirroring load_state_dict(strict=False) at inference/model_init.py:22. */
int gvc_gpt_bind_weight(gvc_gpt* ctx, const char* name, const float* src, int64_t numel, gvc_stream s);
/* number of GPT tensors still unbound (0 = ready) */
int gvc_gpt_missing_weights(gvc_gpt* ctx);

/* Build the prefix embeddings of GPT.compute_embeddings (gpt.py:572-581):
 * prefix[b] = [cond_latents[b] (n_cond rows) | text_embedding(<s> codes </s>) + text_pos[0..Tc+1]].
 * codes: int32 [B,Tc] content codes; out: [B, n_cond+Tc+2, d]. */
int gvc_gpt_prefix_embeddings(gvc_gpt* ctx, const float* cond_latents, int32_t n_cond,
                              const int32_t* codes, int32_t B, int32_t Tc, int32_t start_text,
                              int32_t stop_text, float* out, gvc_stream s);

/* Prefill (gpt_inference.py:81-91): rows = [prefix_emb[b] (P rows) | mel_embedding[start_tok] +
 * mel_pos[0]]; fills the KV cache of slots[b] (positions 0..P), sets its length to P+1 and its next
 * mel position to 1; writes the last row's latent = final_norm(ln_f(h)) [B,d] and logits [B,vocab].
 * logits_out / latent_out may be NULL: the results then stay in the context's staging buffers, which is
 * where gvc_gpt_generate reads the first step's logits from. */
int gvc_gpt_prefill(gvc_gpt* ctx, const int32_t* slots, int32_t B, const float* prefix_emb, int32_t P,
                    int32_t start_tok, float* logits_out, float* latent_out, gvc_stream s);
/* Prefix caching: the same call when the first n_cached rows of prefix_emb (the 32 conditioning latents of the reference
 * speaker, identical for every segment of an utterance: inference_utils.py:43-66 rebuilds them per segment) are ALREADY in
 * the slots' KV cache from an earlier gvc_gpt_prefill[_cached] with the same leading rows.  Only rows n_cached..P are
 * computed and appended; results are bit-identical to the full prefill while the uncached rows fit the skinny path (<= 128
 * rows per call) and equal to rounding on the tiled path (its split-K depends on the row count).  The caller vouches for
 * the cache contents. */
int gvc_gpt_prefill_cached(gvc_gpt* ctx, const int32_t* slots, int32_t B, const float* prefix_emb, int32_t P,
                           int32_t n_cached, int32_t start_tok, float* logits_out, float* latent_out, gvc_stream s);
/* The conditioning rows alone: K/V of cond_latents[b] (n_cond rows, [B, n_cond, d]) into the caches of slots[b], length n_cond; no text
 * rows, no start token, no outputs.  For a caller that knows the target speaker before the first source segment arrives (a streaming
 * session; the reference rebuilds these rows inside every segment's prefill, inference_utils.py:43-66): every segment, the first one too,
 * then calls gvc_gpt_prefill_cached with n_cached = n_cond and its first audio chunk no longer waits for the 32 conditioning rows. */
int gvc_gpt_prefill_cond(gvc_gpt* ctx, const int32_t* slots, int32_t B, const float* cond_latents, int32_t n_cond, gvc_stream s);

/* One KV-cached decode step for B streams (gpt_inference.py:92-112):
 * x = mel_embedding[tok_in[b]] + mel_pos[pos(slot)], 30 blocks against the cache, double
 * LayerNorm, mel_head.  Appends K/V, advances the slot's length and mel position.  For callers that sample themselves: the
 * logits go to the caller only (gvc_gpt_generate continues a slot from its last prefill / generate, not from here). */
int gvc_gpt_decode_step(gvc_gpt* ctx, const int32_t* slots, int32_t B, const int32_t* tok_in,
                        float* logits_out, float* latent_out, gvc_stream s);

/* Reset a slot (length 0, mel position 0) so that decode steps can also build a context from
 * nothing (used by tests). */
int gvc_gpt_reset_slots(gvc_gpt* ctx, const int32_t* slots, int32_t B, gvc_stream s);

/* Teacher-forced latent re-pass, GPT.forward(..., cond_latents=, return_latent=True)
 * (gpt.py:375-508, call site inference_utils.py:71-76): rows = [prefix_emb (P) | start, codes(n),
 * stop x4]; out[b] = final_norm(ln_f(h)) of the first n mel rows -> [B,n,d].  Uses scratch slot
 * `slots[b]` for K/V (its previous content is overwritten).  gen_codes: int32 [B,n]. */
int gvc_gpt_latents(gvc_gpt* ctx, const int32_t* slots, int32_t B, const float* prefix_emb, int32_t P,
                    const int32_t* gen_codes, int32_t n, int32_t start_tok, int32_t stop_tok,
                    float* out, gvc_stream s);

/* ------------------------------------------------------------------------------------------
 * Sampling.  Replaces the per-step body of NewGenerationMixin.sample_stream
 * (layers/stream_generator.py:834-874): RepetitionPenalty -> Temperature -> TopK -> TopP ->
 * softmax -> draw, finished rows emit the pad token.
 * ------------------------------------------------------------------------------------------ */
typedef struct gvc_sample_params {
    float repetition_penalty;   /* 2.0 */
    float temperature;          /* 0.85 */
    float top_p;                /* 0.85; >= 1 disables */
    int32_t top_k;              /* 15; <= 0 disables; 1 = greedy (argmax of penalised logits) */
    int32_t eos_token;          /* stop_audio_token 1025 (also the pad) */
    int32_t vocab;              /* 1026 */
    uint64_t seed;              /* counter-based RNG key (torch.multinomial is not reproducible) */
} gvc_sample_params;

/* ids: int32 [B, ids_stride] row b holds the input_ids of the reference loop (fake prefix ids
 * 1..1,1024 followed by the generated tokens); ids_len[b] = current length (updated +1);
 * finished[b] in/out (1 once eos was emitted); step = RNG counter.  tok_out[b] receives the token. */
int gvc_sample(const float* logits, int32_t B, int32_t* ids, int32_t ids_stride, int32_t* ids_len,
               int32_t* finished, const gvc_sample_params* p, int32_t step, int32_t* tok_out,
               gvc_stream s);

/* ------------------------------------------------------------------------------------------
 * Fused generation loop: prefill state -> n_steps x (sample, decode step) replayed from one
 * captured hipGraph with all step state on the device (no host sync per token; the reference
 * syncs at stream_generator.py:877).  Replaces GPT.generate / GPT.get_generator (gpt.py:594-621).
 *
 * Call after gvc_gpt_prefill[_cached] on the same slots (its logits / latent are parked per slot, as are those of the last
 * gvc_gpt_generate call: consecutive calls may batch different sets of slots).  Per step i in [0,n_steps): samples token i from
 * the current logits, stores it at tokens_out[b*tok_stride + i0 + i] and the latent that predicted
 * it at latents_out[(b*lat_stride + i0 + i)*d], then runs the decode step that consumes it.
 * ids / ids_len / finished as in gvc_sample (the caller initialises them from compute_embeddings).
 * max_keys = cached positions of the longest of these streams once the call has run (prefix + 1 + steps so far + n_steps);
 * the library picks the kernel variant for that context length and returns GVC_ERR_STATE when it would overflow the KV
 * cache (max_seq).  0 = unknown: ids_stride (which must then cover the whole run of the stream) is taken as the bound.
 * On a full MI355X (256 CUs) the decode step of ONE stream is ONE launch (csrc/persist_kernel.h; fp32, bf16 weights, bf16 weights
 * + bf16 KV cache; d_model 1024 or 512 for the bf16 modes) and the step of 2..16 streams runs its whole block stack in ONE launch
 * (csrc/persist_rows.h; d_model 1024, 4 heads of 256, an even layer count); other shapes and batch sizes take the
 * launch-per-phase paths.  A hand-off of a one-launch step that times out (not all of its 256 workgroups resident: another
 * process or stream holds CUs) is reported ONCE -- by gvc_gpt_health, else by the next call -- as GVC_ERR_TIMEOUT, and the context
 * continues on the launch-per-phase paths (see gvc_gpt_health).  Eight consecutive steps are captured per graph.
 * ------------------------------------------------------------------------------------------ */
int gvc_gpt_generate(gvc_gpt* ctx, const int32_t* slots, int32_t B, int32_t* ids, int32_t ids_stride,
                     int32_t* ids_len, int32_t* finished, const gvc_sample_params* p, int32_t i0,
                     int32_t n_steps, int32_t max_keys, int32_t* tokens_out, int32_t tok_stride, float* latents_out,
                     int32_t lat_stride, gvc_stream s);

/* Which decode step the last gvc_gpt_generate call replayed (diagnostic): 0 none yet, 1 launch-per-phase with split-key attention,
 * 2 launch-per-phase with the fused short-context attention launch, 3 the one-launch step (one stream), 4 the MFMA rows path
 * (launch per phase: 17+ streams, or shapes the one-launch rows step does not serve), 5 the one-launch rows step (2..16 streams,
 * csrc/persist_rows.h). */
int gvc_gpt_decode_variant(gvc_gpt* ctx);
/* Diagnostic: how many one-launch rows steps (csrc/persist_rows.h: 2..16 rows that continue cached sequences -- batched decode
 * steps, the uncached rows of a streaming chunk's prefill) this context has issued; a step captured into the generation loop's
 * graph counts once.  Tests use it to prove which path served a call. */
long long gvc_gpt_rows_step_launches(gvc_gpt* ctx);
/* Health of the work a caller has just synchronised (no reference counterpart: the reference has no in-kernel hand-offs).  The
 * one-launch steps need all 256 workgroups co-resident; when another process or stream holds CUs a hand-off times out (~0.2 s,
 * bounded spins), the step's outputs are garbage and a device-visible word records it.  This call -- and, failing that, the next
 * library call -- then returns GVC_ERR_TIMEOUT ONCE, having switched the context to the launch-per-phase paths (captured graphs
 * dropped, hand-off buffers re-initialised): the context stays usable, the caller resets the affected slots and repeats the work.
 * Also reports a full KV cache / mel position table (see gvc_gpt_reset_slots).  GVC_OK otherwise. */
int gvc_gpt_health(gvc_gpt* ctx);
/* Warm-up (reference: the user's own warm-up conversions, /root/reference/infer.py:27-30; SURVEY.md 8(b)(iii): no hidden
 * synchronisation or allocation after create).  Does, for a generation over B streams that reaches max_keys cached positions
 * (0: up to max_seq) with this top_k, everything the first data-path call of that shape would otherwise do itself: the one-launch
 * steps' hand-off buffers, packed weight copy, XCD-topology probe and LDS opt-in, and the capture of the step graphs (eight steps /
 * one step).  Synchronous; call it after the weights are bound, once per (B, context class, greedy or sampled) a deployment uses.
 * Afterwards gvc_gpt_generate / gvc_gpt_decode_step / gvc_gpt_prefill_cached calls of that shape do no hipMalloc, no
 * hipDeviceSynchronize and no graph capture. */
int gvc_gpt_warmup(gvc_gpt* ctx, int32_t B, int32_t max_keys, int32_t top_k);
/* The same for EVERY context class a generation over B streams passes through while its longest stream grows from min_keys to max_keys
 * cached positions (the library keys its step graphs by classes of the context length -- fused / split-key attention, 1 / 2 / 4 key chunks
 * of the rows steps -- whose thresholds are its own business: callers name the range, not the classes). */
int gvc_gpt_warmup_range(gvc_gpt* ctx, int32_t B, int32_t min_keys, int32_t max_keys, int32_t top_k);
/* Diagnostic: allocations / device-wide synchronisations / graph captures this context has done INSIDE data-path calls (first use
 * of a path that gvc_gpt_warmup had not prepared; a rebind after the weight pack was built; the fallback after a hand-off
 * time-out).  gvc_gpt_warmup's own work does not count.  Tests assert it stays put across warmed-up calls. */
long long gvc_gpt_lazy_inits(gvc_gpt* ctx);
/* Re-arm the one-launch steps after a hand-off time-out has switched the context to the launch-per-phase paths (gvc_gpt_health): to be called
 * when the caller knows the GPU is its own again (no reference counterpart).  Reports pending errors like gvc_gpt_health; synchronises the
 * device, drops the captured step graphs and resets the hand-off state.  A later time-out falls back again. */
int gvc_gpt_rearm(gvc_gpt* ctx);

/* Measurement hook used by bench.py (not a reference interface): launches ONLY one kernel class of the
 * decode step (0 c_attn GEMV, 1 attention, 2 attn c_proj GEMV, 3 mlp c_fc GEMV, 4 mlp c_proj GEMV, 5 head
 * GEMV) for every layer, n_steps times back to back on the stream between two hipEvents, and returns the
 * mean microseconds per launch (same-stream launch boundary included) and the number of launches.
 * which = 6: the whole decode step (sampler excluded); which = 16 + X: the whole step WITHOUT class X, so that
 * (whole - without) / launches_per_step is the IN-SITU cost of class X (behind its real predecessor, as rocprofv3
 * sees it); both return microseconds per step.  which = 7: the one-launch decode step (csrc/persist_kernel.h), microseconds per
 * launch.  0..6 and 16+ always time the launch-per-phase step (the fallback of bf16 / multi-stream / partial-GPU contexts).
 * Synchronises the stream; leaves the slots' caches in an undefined state (reset or prefill afterwards). */
int gvc_gpt_time_kernel(gvc_gpt* ctx, int32_t which, const int32_t* slots, int32_t B, const int32_t* tok_in,
                        int32_t n_steps, float* avg_us, int32_t* n_launches, gvc_stream s);

/* Measurement / test hook for the fp32 MFMA GEMMs under the prefill (not a reference interface): C[M][N] = A[M][K] W[N][K]^T
 * (+ bias[N], nullable), all row-major device buffers, through one kernel: variant 0 the 64x64x32 tiled kernel, 1 the strip
 * kernel (operands converted to the fragment-major layout first; K split up to sk_max), 2 the skinny kernel (M <= 128).  With
 * iters > 0 the GEMM is then launched iters more times between two hipEvents and *avg_us is the mean microseconds per GEMM
 * (split-K epilogue included).  Synchronises the stream. */
int gvc_gemm_probe(int32_t variant, const float* A, const float* W, const float* bias, float* C, int32_t M, int32_t N, int32_t K,
                   int32_t sk_max, int32_t iters, float* avg_us, gvc_stream s);

/* ------------------------------------------------------------------------------------------
 * Perceiver resampler.  Replaces layers/perceiver_encoder.py:PerceiverResampler.forward (:265-276)
 * as called by GPT.get_style_emb (gpt.py:351-373) with mask=None.
 * ------------------------------------------------------------------------------------------ */
typedef struct gvc_perceiver gvc_perceiver;
typedef struct gvc_perceiver_dims {
    int32_t dim, depth, dim_context, num_latents, dim_head, heads, ff_mult;
    int32_t max_batch, max_frames;
} gvc_perceiver_dims;

int gvc_perceiver_create(const gvc_perceiver_dims* dims, gvc_perceiver** out);
int gvc_perceiver_destroy(gvc_perceiver* ctx);
/* names relative to the module: "latents", "proj_context.weight", "layers.0.0.to_q.weight",
 * "layers.0.1.0.weight", "norm.gamma", ... */
int gvc_perceiver_bind_weight(gvc_perceiver* ctx, const char* name, const float* src, int64_t numel,
                              gvc_stream s);
int gvc_perceiver_missing_weights(gvc_perceiver* ctx);
/* x: [B,F,dim_context] -> out [B,num_latents,dim] */
int gvc_perceiver_forward(gvc_perceiver* ctx, const float* x, int32_t B, int32_t F, float* out,
                          gvc_stream s);

/* ------------------------------------------------------------------------------------------
 * Log-mel front end.  Replaces utils.py:TorchMelSpectrogram.forward (:150-162) as instantiated at
 * trainers/hifigan_trainer.py:105-115 (n_fft 2048, hop 256, win 1024, 24 kHz, 0-8 kHz, 80 mels,
 * slaney norm, htk scale, centre + reflect pad).
 * ------------------------------------------------------------------------------------------ */
typedef struct gvc_mel gvc_mel;
int gvc_mel_create(int32_t n_fft, int32_t hop, int32_t win, int32_t sample_rate, float f_min,
                   float f_max, int32_t n_mels, const float* mel_norms_host, gvc_mel** out);
int gvc_mel_destroy(gvc_mel* ctx);
/* wav [B,T] -> out [B,n_mels,1+T/hop] (reference layout); if out_frames_major != NULL it also
 * receives [B,1+T/hop,n_mels] (the layout the Perceiver consumes). */
int gvc_mel_forward(gvc_mel* ctx, const float* wav, int32_t B, int32_t T, float* out,
                    float* out_frames_major, gvc_stream s);

/* ------------------------------------------------------------------------------------------
 * Content tokenizer.  Replaces layers/dvae.py:DiscreteVAE.get_codebook_indices (:324-331) with
 * the 1-D encoder of :252-291 and Quantize.forward (:87-93).
 * ------------------------------------------------------------------------------------------ */
typedef struct gvc_dvae gvc_dvae;
typedef struct gvc_dvae_dims {
    int32_t channels, hidden_dim, num_layers, num_resnet_blocks, kernel_size, codebook_dim,
        num_tokens;
    int32_t max_batch, max_frames;
} gvc_dvae_dims;
int gvc_dvae_create(const gvc_dvae_dims* dims, gvc_dvae** out);
int gvc_dvae_destroy(gvc_dvae* ctx);
/* names relative to the module: "encoder.0.0.weight", "encoder.2.net.0.bias", "codebook.embed" */
int gvc_dvae_bind_weight(gvc_dvae* ctx, const char* name, const float* src, int64_t numel, gvc_stream s);
int gvc_dvae_missing_weights(gvc_dvae* ctx);
/* feat [B,channels,T] (reference layout) -> codes int32 [B,Tc], Tc = T halved num_layers times
 * (ceil); enc_out (optional, may be NULL) receives the encoder output [B,Tc,codebook_dim]. */
int gvc_dvae_encode(gvc_dvae* ctx, const float* feat, int32_t B, int32_t T, int32_t* codes_out,
                    float* enc_out, gvc_stream s);
/* same, from frame-major features [B,T,channels] -- the layout ContentVec emits, i.e. the harness's
 * `get_codebook_indices(content_feat.transpose(1, 2))` (inference/inference_utils.py:53,167) without the transpose */
int gvc_dvae_encode_frames(gvc_dvae* ctx, const float* feat, int32_t B, int32_t T, int32_t* codes_out,
                           float* enc_out, gvc_stream s);
/* standalone VQ: x [N,dim], embed [dim,n_embed] (reference layout) -> idx int32 [N] */
int gvc_vq_argmin(const float* x, const float* embed, int32_t N, int32_t dim, int32_t n_embed,
                  int32_t* idx, float* work /* N*n_embed floats */, gvc_stream s);

/* ------------------------------------------------------------------------------------------
 * HiFi-GAN generator.  Replaces layers/hifigan.py:HiFiGAN.forward (:218-233, ResBlock2
 * :119-157) and, for the latent entry point, the x4 linear interpolation in front of it
 * (inference/inference_utils.py:81-85, 196-202).
 * ------------------------------------------------------------------------------------------ */
typedef struct gvc_hifigan gvc_hifigan;
typedef struct gvc_hifigan_dims {
    int32_t in_dim;               /* vocoder_config.input_feat_dim (1024) */
    int32_t up_init_ch;           /* upsample_initial_channel (256) */
    int32_t n_ups;                /* len(upsample_rates) <= 4 */
    int32_t up_rates[4];          /* 8, 8, 4 */
    int32_t up_kernels[4];        /* 16, 16, 8 */
    int32_t n_kernels;            /* len(resblock_kernel_sizes) <= 4 */
    int32_t res_kernels[4];       /* 3, 5, 7 */
    int32_t res_dilations[4][2];  /* ResBlock2: [[1,2],[2,6],[3,12]] */
    int32_t max_batch, max_frames; /* capacity: input frames AFTER the interpolation */
} gvc_hifigan_dims;
int gvc_hifigan_create(const gvc_hifigan_dims* dims, gvc_hifigan** out);
int gvc_hifigan_destroy(gvc_hifigan* ctx);
/* names of the reference state dict with weight-norm folded (w = g * v / |v|, as remove_weight_norm leaves
 * them): "conv_pre.weight", "ups.1.bias", "resblocks.4.convs.0.weight", "conv_post.weight", ... */
int gvc_hifigan_bind_weight(gvc_hifigan* ctx, const char* name, const float* src, int64_t numel, gvc_stream s);
int gvc_hifigan_missing_weights(gvc_hifigan* ctx);
/* x [B,in_dim,T] (reference layout) -> wav [B, T * prod(up_rates)] */
int gvc_hifigan_forward(gvc_hifigan* ctx, const float* x, int32_t B, int32_t T, float* wav, gvc_stream s);
/* latents [B,n,in_dim] (as yielded by the generation loop) -> F.interpolate(scale, linear) -> wav
 * [B, n * scale * prod(up_rates)] */
int gvc_hifigan_forward_latents(gvc_hifigan* ctx, const float* latents, int32_t B, int32_t n, int32_t scale,
                                float* wav, gvc_stream s);

/* ------------------------------------------------------------------------------------------
 * Resampler.  Replaces torchaudio.functional.resample(audio, lsr, sampling_rate) as called by
 * utils.load_audio (utils.py:58-62): sinc_interp_hann, lowpass_filter_width 6, rolloff 0.99.
 * ------------------------------------------------------------------------------------------ */
int gvc_resample_length(int32_t T, int32_t orig_sr, int32_t new_sr);
/* x [B,T] -> out [B, gvc_resample_length(T, orig_sr, new_sr)]; synchronises the stream (file-loading path) */
int gvc_resample(const float* x, int32_t B, int32_t T, int32_t orig_sr, int32_t new_sr, float* out, gvc_stream s);

/* ------------------------------------------------------------------------------------------
 * ContentVec / HuBERT-base feature extractor.  Replaces
 * ContentVecExtractor.extract_content_features (layers/content_processor.py:17-31):
 *   fairseq HubertModel.extract_features(source=wav, output_layer=n_layers)[0] -> final_proj
 * i.e. 7-layer conv extractor (GroupNorm + GELU on layer 0) -> LayerNorm -> post_extract_proj -> x + GELU(grouped
 * positional conv) -> LayerNorm -> n_layers post-LN transformer layers -> final_proj.  The reference's
 * `padding_mask = (wav == 0)` (content_processor.py:24) is derived from `wav` by the library itself, with fairseq's
 * reduction to frames (trailing samples % frames dropped, a frame is padding when every sample of its chunk is exactly
 * zero): padding frames are zeroed ahead of the positional conv and their keys are excluded from every attention.
 * Weight names are fairseq's (the keys under `content_extractor.model.` in a GenVC checkpoint), except the
 * weight-normed positional conv, bound folded as "encoder.pos_conv.0.weight" [E, E/groups, k] (g * v / |v|, dim=2).
 * ------------------------------------------------------------------------------------------ */
typedef struct gvc_hubert gvc_hubert;
typedef struct {
    int32_t n_conv;                 /* <= 8 */
    int32_t conv_dim[8], conv_kernel[8], conv_stride[8];
    int32_t embed_dim, n_layers, n_heads, ffn_dim, pos_conv_kernel, pos_conv_groups, final_dim;
    int32_t max_batch, max_samples;
} gvc_hubert_dims;
int gvc_hubert_create(const gvc_hubert_dims* dims, gvc_hubert** out);
int gvc_hubert_destroy(gvc_hubert* ctx);
int gvc_hubert_bind_weight(gvc_hubert* ctx, const char* name, const float* src, int64_t numel, gvc_stream s);
int gvc_hubert_missing_weights(gvc_hubert* ctx);
/* frames the conv stack yields for n_samples input samples (<= 0: too short) */
int gvc_hubert_frames(gvc_hubert* ctx, int32_t n_samples);
/* wav [B,T] (16 kHz) -> out [B, gvc_hubert_frames(T), final_dim] */
int gvc_hubert_forward(gvc_hubert* ctx, const float* wav, int32_t B, int32_t T, float* out, gvc_stream s);

#ifdef __cplusplus
}
#endif
#endif /* GENVC_HIP_H */
