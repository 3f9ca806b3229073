/*
 * amphion_hip.h -- C ABI of libamphion_hip.so: the MI355X (gfx950) vocoder-inference hot path.
 *
 * The reference (open-mmlab/Amphion) has NO native code on this path: its "kernels" are
 * torch.nn.functional calls.  The entry points below are what a ctypes binding of the reference's
 * generator / front-end would call instead of those torch ops; each one cites the reference
 * interface it replaces (paths relative to the reference tree).  INTEGRATION.md shows the
 * reference-side binding.
 *
 * Conventions
 *   - plain C, plain pointers and sizes; no torch types.
 *   - every function returns amp_status (0 = OK, <0 = error); amp_last_error() gives the text
 *     (thread-local).  Nothing calls abort().
 *   - `*_dev` pointers are device (HBM) pointers on the current HIP device; `*_host` are host
 *     pointers.  The caller owns inputs, outputs and the workspace; a handle owns only its packed
 *     weights.  No allocation and no host synchronisation inside *_forward: all work is enqueued on
 *     `stream` (a hipStream_t passed as void*; NULL = the default stream).
 *   - all tensors are fp32, contiguous, layout [B, C, T] (time fastest), as in the reference.
 */
#ifndef AMPHION_HIP_H
#define AMPHION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum amp_status {
    AMP_OK = 0,
    AMP_ERR_INVALID = -1,        /* bad argument / shape */
    AMP_ERR_MISSING_WEIGHT = -2, /* finalize(): a tensor the architecture needs was never set */
    AMP_ERR_HIP = -3,            /* a HIP runtime call failed */
    AMP_ERR_UNSUPPORTED = -4,    /* configuration outside what the kernels cover */
    AMP_ERR_STATE = -5,          /* call order violated (e.g. forward before finalize) */
    AMP_ERR_RANGE = -6           /* an activation left the split-f16 operand range (see amp_range_check) */
} amp_status;

typedef enum amp_arch {
    AMP_ARCH_HIFIGAN = 0,      /* models/vocoders/gan/generator/hifigan.py:151-219  HiFiGAN       */
    AMP_ARCH_BIGVGAN = 1,      /* models/vocoders/gan/generator/bigvgan.py:232-331  BigVGAN       */
    AMP_ARCH_HIFIGAN_VITS = 2  /* models/vocoders/gan/generator/hifigan.py:376-449  HiFiGAN_vits  */
} amp_arch;

typedef enum amp_activation {
    AMP_ACT_LRELU = 0,     /* F.leaky_relu(x, 0.1)            hifigan.py:14,95-97          */
    AMP_ACT_SNAKE = 1,     /* Activation1d(Snake)             bigvgan.py:85-102, snake.py:51-61   */
    AMP_ACT_SNAKEBETA = 2  /* Activation1d(SnakeBeta)         bigvgan.py:104-124, snake.py:110-122 */
} amp_activation;

/* Arithmetic of the conv contractions (every Conv1d / ConvTranspose1d of the path; the reference runs
 * them as fp32 torch ops).  Inputs, outputs, accumulators and everything stored in HBM are fp32 in both
 * modes; results of both modes meet the same 1e-4 max-abs parity bound against the fp32 reference. */
typedef enum amp_precision {
    AMP_PRECISION_F32 = 0,   /* v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fmaf chain (157 TFLOP/s peak)       */
    AMP_PRECISION_F16X3 = 1  /* operands split hi+lo in f16 (22 mantissa bits), 3 x v_mfma_f32_32x32x16_f16 per
                                term, fp32 accumulate (838 TFLOP/s effective peak) -- the default                */
} amp_precision;

#define AMP_MAX_STAGES 8
#define AMP_MAX_KERNELS 8
#define AMP_MAX_DILATIONS 8

/* Mirrors cfg.model.hifigan.* / cfg.model.bigvgan.* (hifigan.py:155-199, bigvgan.py:237-306) and the
 * explicit HiFiGAN_vits constructor arguments (hifigan.py:377-387). */
typedef struct amp_gen_desc {
    int32_t arch;                     /* amp_arch */
    int32_t n_in;                     /* cfg.preprocess.n_mel, or initial_channel for HiFiGAN_vits */
    int32_t upsample_initial_channel;
    int32_t n_stages;                 /* len(upsample_rates) */
    int32_t upsample_rates[AMP_MAX_STAGES];
    int32_t upsample_kernel_sizes[AMP_MAX_STAGES];
    int32_t n_kernels;                /* len(resblock_kernel_sizes) */
    int32_t resblock_kernel_sizes[AMP_MAX_KERNELS];
    int32_t n_dilations[AMP_MAX_KERNELS];
    int32_t resblock_dilation_sizes[AMP_MAX_KERNELS][AMP_MAX_DILATIONS];
    int32_t resblock_type;            /* 1 = ResBlock1/AMPBlock1, 2 = ResBlock2/AMPBlock2 */
    int32_t activation;               /* amp_activation (BigVGAN: cfg.model.bigvgan.activation) */
    int32_t snake_logscale;           /* cfg.model.bigvgan.snake_logscale */
    int32_t gin_channels;             /* HiFiGAN_vits only; 0 = no `cond` conv */
} amp_gen_desc;

typedef struct amp_gen amp_gen;

/* Library / device probes.  amp_version: 100 = round 1's surface; 120 adds the fused-WN and conv + activation entry points, 122
 * amp_set_conv_blk / amp_set_conv_rg_fast / amp_set_pingpong; 130 (round 3) appended the four range_* fields to amp_mel_desc and
 * added amp_resblock_forward / amp_set_resblock_fusion / amp_gen_kernel_name; 140 (round 4): amp_mel_desc starts with struct_size
 * (an ABI break for every earlier consumer of that struct -- re-compile against this header), + amp_mel_init,
 * amp_ampblock_forward, amp_set_ampblock_fusion; 141 (additive): amp_conv_forward_ragged, amp_layer_norm_c_ragged,
 * amp_dwconv_layer_norm_c, amp_rel_attention_strided, amp_set_rel_attention_tiled, amp_expand_path_strided; 142 (round 5) REMOVES
 * amp_conv_act_forward, amp_set_fuse_act and amp_set_wn_layer_fusion together with the kernels behind them (bit-identical forms the launch
 * policy never chose), refuses amp_set_pair_strips(1), and lets amp_mel_forward / amp_mel_backward / amp_istft_forward / amp_istft_same take
 * any n_fft in [64, 4096] (mixed-radix kernels: compile-time butterflies for the primes 2 .. 13, a run-time radix pass for larger prime factors;
 * powers of two keep their kernels). */
int amp_version(void);
const char* amp_last_error(void);
/* Number of HIP devices visible (0 when there is no GPU); never fails. */
int amp_device_count(void);

/* Process-wide default for handles created AFTER the call (also settable with the environment variable
 * AMP_PRECISION=f32|f16x3 before the first handle is created).  A handle keeps the precision it was
 * built with. */
int amp_set_precision(int precision /* amp_precision */);
int amp_get_precision(void);

/* ---- Generator handle: replaces nn.Module construction + forward of HiFiGAN / BigVGAN / HiFiGAN_vits ---- */

/* Replaces HiFiGAN.__init__ (hifigan.py:151-201) / BigVGAN.__init__ (bigvgan.py:232-311) /
 * HiFiGAN_vits.__init__ (hifigan.py:376-422): validates the architecture, allocates nothing on the device. */
int amp_gen_create(const amp_gen_desc* desc, amp_gen** out);

/* Replaces load_state_dict for one tensor (vocoder_inference.py:270-332).  `ref_key` is the
 * reference state_dict key: "...weight_g"/"...weight_v" pairs or folded
 * "...weight", "...bias", Snake "...act.alpha"/"...act.beta", "...filter" buffers.  The data is
 * copied; `data_host` may be a host pointer or a device pointer of the current context (fp32, contiguous).
 * Unknown keys are rejected (AMP_ERR_INVALID). */
int amp_gen_set_weight(amp_gen* g, const char* ref_key, const float* data_host, const int64_t* shape, int ndim);

/* Folds weight-norm (w = g*v/||v||, torch.nn.utils.weight_norm; hifigan.py:23,157,176,199),
 * packs the weights into MFMA fragment order and uploads them to the current device. */
int amp_gen_finalize(amp_gen* g);

/* Output samples per input frame (prod(upsample_rates)) and scratch size for a (B, T) forward. */
int amp_gen_hop(const amp_gen* g);
size_t amp_gen_workspace_bytes(const amp_gen* g, int B, int T);

/* amp_gen_forward can run the batch depth-first in groups of items (each group through the whole
 * generator) to bound the workspace: target working set in MiB, 0 = whole batch per layer (default; the
 * faster setting on MI355X).
 * Results do not depend on it. */
int amp_set_group_mb(int megabytes);

/* Operand-range guard of the f16x3 arithmetic.  Activations are staged as hi + lo f16 pairs after an exact x16, so
 * |x| up to 4094 is representable; the fp32 reference (F.conv1d) has no such limit.  Every f16x3 kernel raises a flag
 * when a staged value is larger (or infinite) -- the output of that launch is then NOT the reference's (inf / NaN).
 * A generator handle owns its flag: amp_gen_range_check synchronises `stream` and returns AMP_ERR_RANGE if a forward of
 * `g` since the last check was affected (and clears the flag); amp_gen_forward reports the same condition WITHOUT
 * synchronising: a call that finds the flag of an earlier, finished forward of the same handle set returns
 * AMP_ERR_RANGE instead of running.  amp_range_check does the synchronising check for the op-level entry points
 * (amp_conv_forward, amp_pair_forward, ...), which share one flag per device.  Remedy: amp_set_precision(
 * AMP_PRECISION_F32) and rebuild the handle (the exact fp32 MFMA kernels have the reference's range).  A NaN input is
 * not flagged: it propagates to the output as it does through the reference. */
int amp_range_check(void* stream);
int amp_gen_range_check(amp_gen* g, void* stream);

/* Fused ResBlock pairs (hifigan.py:93-100) have two kernels with bit-identical results (tests/test_gpu_pair.py): the
 * per-tile kernel and the strip-mined kernel (a workgroup walks a strip of one utterance and carries conv2's halo in
 * LDS; C = 128, k >= 7, launches that fill the chip).  -1 (default): the measured per-shape policy; 0: per-tile
 * everywhere (the cross-check).  Mode 1 (four-wave / C = 256 strips) left with amp_version 142. */
int amp_set_pair_strips(int on);

/* Replaces HiFiGAN.forward (hifigan.py:203-219), BigVGAN.forward (bigvgan.py:313-331) and
 * HiFiGAN_vits.forward (hifigan.py:424-443):  mel_dev [B, n_in, T]  ->  wav_dev [B, 1, T*hop].
 * cond_dev: optional speaker embedding g [B, gin_channels, 1] (HiFiGAN_vits), else NULL. */
int amp_gen_forward(amp_gen* g, const float* mel_dev, const float* cond_dev, int B, int T, float* wav_dev,
                    void* workspace_dev, size_t workspace_bytes, void* stream);

/* Ragged batch: item b holds lens_dev[b] <= T valid mel frames (int32, device) inside the zero-padded
 * [B, n_in, T] input.  Every layer then pads at the utterance's OWN end (zero padding of the convs,
 * replicate padding of the anti-aliased activations), so wav[b, : lens[b] * hop] is bit-identical to
 * running that utterance alone (the reference's per-utterance loop, gan_vocoder_inference.py:74-96);
 * samples beyond it are unspecified.  lens_dev == NULL is amp_gen_forward. */
int amp_gen_forward_ragged(amp_gen* g, const float* mel_dev, const float* cond_dev, const int32_t* lens_dev, int B,
                           int T, float* wav_dev, void* workspace_dev, size_t workspace_bytes, void* stream);

/* Kernel timing with HIP events recorded on the launch stream.  amp_gen_set_profiling(g, slots) keeps a ring of
 * `slots` event sets (0 = off): every amp_gen_forward records into the next one without synchronising, so a whole
 * timed region of `slots` forwards can be read back afterwards.  amp_gen_timing_ms(g, back, which, &ms): `back` = 0
 * is the most recent forward, 1 the one before, ...; which: 0 = whole forward, 1 = MRF conv stack (all
 * ResBlock/AMPBlock convs), 2 + i = the MRF convs of upsampling stage i, 100 + 16*i + j = resblock j of stage i
 * (its back-to-back conv / fused-pair launches).  Synchronises on that forward's events.  Returns <0 on error.
 * amp_gen_last_timing_ms(g, which, &ms) == amp_gen_timing_ms(g, 0, which, &ms). */
int amp_gen_set_profiling(amp_gen* g, int slots);
int amp_gen_timing_ms(amp_gen* g, int back, int which, float* ms_out);
int amp_gen_last_timing_ms(amp_gen* g, int which, float* ms_out);
/* Which kernels the launches of resblock j of stage i (which = 100 + 16*i + j, as above) were in that profiled forward:
 * the distinct kernel names as rocprofv3 prints them (template arguments included), " | "-joined, written to buf[n].
 * This is what the launch policy actually picked for the shape -- bench.py reports it instead of assuming. */
int amp_gen_kernel_name(amp_gen* g, int back, int which, char* buf, size_t n);

void amp_gen_destroy(amp_gen* g);

/* ---- Op level (used by tests and by callers that want one fused op) ---- */

typedef struct amp_conv amp_conv;

/* One Conv1d (nn.Conv1d(cin, cout, k, 1, dilation=d, padding=get_padding(k, d)), gan_utils.py:12) or
 * ConvTranspose1d(cin, cout, k, stride, padding) (hifigan.py:176-186).  weight_host is the FOLDED
 * weight: [cout, cin, k] for a conv, [cin, cout, k] for a transposed conv.  bias_host may be NULL. */
int amp_conv_create(int transposed, int cin, int cout, int k, int stride, int dilation, int padding,
                    const float* weight_host, const float* bias_host, amp_conv** out);
int amp_conv_out_len(const amp_conv* c, int T);
/* y = lrelu_out( conv( lrelu_in(x) ) + bias + res ).  slope == 1.0f disables an activation;
 * res_dev may be NULL or alias y_dev (in-place residual).  x_dev [B, cin, T] -> y_dev [B, cout, T_out]. */
int amp_conv_forward(const amp_conv* c, const float* x_dev, int B, int T, float slope_in, const float* res_dev,
                     float slope_out, float* y_dev, void* stream);
/* Same, reading x from a channel slice of a wider tensor: batch item b starts at x_dev + b*x_batch_stride
 * (elements).  Used by ResidualCouplingLayer.pre on x0 = x[:, :half] (modules/flow/modules.py:380-381). */
int amp_conv_forward_strided(const amp_conv* c, const float* x_dev, long long x_batch_stride, int B, int T,
                             float slope_in, const float* res_dev, float slope_out, float* y_dev, void* stream);
/* conv(x * mask) with the sequence mask taken by the kernel (`conv_1(x * x_mask)`, modules/transformer/attentions.py:392-400;
 * `pre(x0) * x_mask`, modules/flow/modules.py:381): lens_dev int32 [B] valid lengths or NULL; input columns t >= lens[b] count
 * as zero whatever the buffer holds (a select, NaN-safe); output tiles wholly beyond an utterance's end are skipped, so columns
 * t >= lens[b] of y_dev are UNSPECIFIED (assign, do not multiply).  'same' zero-padded Conv1d only when lens_dev is given.
 * x_batch_stride 0 = cin*T. */
int amp_conv_forward_ragged(const amp_conv* c, const float* x_dev, long long x_batch_stride, int B, int T, const int32_t* lens_dev,
                            float slope_in, const float* res_dev, float slope_out, float* y_dev, void* stream);
/* amp_conv_forward with the MRF accumulation of hifigan.py:208-214 / apnet.py:355-363 exposed:
 *   v = conv(lrelu_in(x)) + bias (+ res);   mode 0: y = v   1: y = y + v   2: y = (y + v) / div. */
int amp_conv_forward_mrf(const amp_conv* c, const float* x_dev, int B, int T, float slope_in, const float* res_dev,
                         float* y_dev, int mode, float div, void* stream);

/* APNet head, element-wise over n values (models/vocoders/gan/generator/apnet.py:379-383):
 * pha = atan2(I, R); rea = exp(logamp) * cos(pha); imag = exp(logamp) * sin(pha). */
int amp_apnet_polar(const float* logamp_dev, const float* r_dev, const float* i_dev, size_t n, float* pha_dev,
                    float* rea_dev, float* imag_dev, void* stream);

/* UpSample1d.forward (modules/anti_aliasing/resample.py:36-45) as a stand-alone op: replicate-pad by
 * K/ratio - 1, depthwise ConvTranspose1d with the [K] filter at stride = ratio, x ratio, crop -> y [B, C, ratio*T].
 * filt_host: K <= AMP_FIR_MAX_TAPS floats on the host (the module's `filter` buffer). */
#define AMP_FIR_MAX_TAPS 64
int amp_fir_upsample(const float* x_dev, int B, int C, int T, const float* filt_host, int K, int ratio, float* y_dev,
                     void* stream);
/* LowPassFilter1d.forward (modules/anti_aliasing/filter.py:92-99; DownSample1d, resample.py:62-65, is this with
 * stride = ratio): pad (pad_left, pad_right) in `pad_mode`, depthwise Conv1d with the [K] filter at `stride`
 * -> y [B, C, (T + pad_left + pad_right - K) / stride + 1].  padding=False is pad_left = pad_right = 0. */
typedef enum amp_pad_mode { AMP_PAD_REPLICATE = 0, AMP_PAD_ZEROS = 1, AMP_PAD_REFLECT = 2 } amp_pad_mode;
int amp_fir_filter(const float* x_dev, int B, int C, int T, const float* filt_host, int K, int stride, int pad_left,
                   int pad_right, int pad_mode, float* y_dev, void* stream);

/* ---- frame-rate ops of the text -> duration -> alignment front of VITS
 * inference, SynthesizerTrn.infer models/tts/vits/vits.py:320-369.  All tensors [B, C, T] fp32 on the
 * device; lens_dev = int32 [B] valid lengths (NULL = all T). ---- */
/* LayerNorm over channels (modules/base/base_module.py:20-23) of x (+ res if not NULL: Encoder's norm(x + y),
 * modules/transformer/attentions.py:69,73), optionally followed by GELU and by "+ post" (DDSConv,
 * modules/flow/modules.py:64-70: x = x + gelu(norm(y))). */
int amp_layer_norm_c(const float* x_dev, const float* res_dev, const float* gamma_dev, const float* beta_dev,
                     const float* post_dev, int B, int C, int T, float eps, int gelu, float* y_dev, void* stream);
/* amp_layer_norm_c whose result is ZERO in the columns t >= lens[b] (a select: x / res may hold anything there) -- the `* x_mask`
 * that follows the Encoder's norms (attentions.py:64-76) without its own launch. */
int amp_layer_norm_c_ragged(const float* x_dev, const float* res_dev, const float* gamma_dev, const float* beta_dev,
                            const float* post_dev, const int* lens_dev, int B, int C, int T, float eps, int gelu, float* y_dev,
                            void* stream);
/* act(LN(dwconv(x * mask))) in one launch: DDSConv's convs_sep[i] -> norms_1[i] -> gelu (modules/flow/modules.py:63-65), the
 * depthwise taps evaluated on load in amp_dwconv's order (same bits as the two launches).  K = 3 (AMP_ERR_INVALID otherwise: run
 * amp_dwconv + amp_layer_norm_c); dw_weight [C, 1, K], dw_bias [C] or NULL; columns t >= lens[b] of y are zero; y != x. */
int amp_dwconv_layer_norm_c(const float* x_dev, const float* dw_weight_dev, const float* dw_bias_dev, int K, int dilation,
                            const float* gamma_dev, const float* beta_dev, const int* lens_dev, int B, int C, int T, float eps,
                            int gelu, float* y_dev, void* stream);
/* The seam between DDSConv layers i and i + 1 (modules/flow/modules.py:63-70) in one launch:
 *     x_out = x + gelu(LN(y; gamma2, beta2))                    norms_2[i] on the 1 x 1 conv's output y, residual x
 *     z_out = gelu(LN(dwconv(x_out * mask); gamma1, beta1))     convs_sep[i+1] -> norms_1[i+1]
 * Same bits as amp_layer_norm_c (gelu, post = x) followed by amp_dwconv_layer_norm_c.  K = 3 and C <= 192, AMP_ERR_UNSUPPORTED
 * otherwise; z_out is zero beyond lens[b], x_out is not masked; the outputs must not alias the inputs. */
int amp_dds_seam(const float* y_dev, const float* x_dev, const float* gamma2_dev, const float* beta2_dev, float eps2,
                 const float* dw_weight_dev, const float* dw_bias_dev, int K, int dilation, const float* gamma1_dev, const float* beta1_dev,
                 float eps1, const int* lens_dev, int B, int C, int T, float* x_out_dev, float* z_out_dev, void* stream);
/* x[b, c, :] += cb[b, c]: the broadcast add of a length-1 condition, x + cond(g) (hifigan.py:426-427,
 * stochastic_duration_predictor.py:64-66). */
int amp_add_channel_bias(float* x_dev, const float* cb_dev, int B, int C, int T, void* stream);
/* MultiHeadAttention.attention for self-attention with windowed relative-position embeddings shared by the heads
 * (modules/transformer/attentions.py:232-272): q, k, v, out [B, H*dk, T]; emb_k, emb_v [2*window+1, dk]. */
int amp_rel_attention(const float* q_dev, const float* k_dev, const float* v_dev, const float* emb_k_dev,
                      const float* emb_v_dev, const int* lens_dev, int B, int H, int dk, int T, int window, float* out_dev,
                      void* stream);
/* The same with q / k / v given as slices of ONE tensor (the merged q|k|v projection [B, 3*H*dk, T]): element (b, c, t) of each
 * at ptr[b*qkv_batch_stride + c*T + t]; out stays [B, H*dk, T] dense. */
int amp_rel_attention_strided(const float* q_dev, const float* k_dev, const float* v_dev, long long qkv_batch_stride,
                              const float* emb_k_dev, const float* emb_v_dev, const int* lens_dev, int B, int H, int dk, int T,
                              int window, float* out_dev, void* stream);
/* 1 (default): blocks of 16 queries share the keys / values staged in LDS; 0: one workgroup per query (rounds 2-3).  Same bits. */
int amp_set_rel_attention_tiled(int on);
/* Depthwise dilated Conv1d of x * mask, padding (K*d - d)/2 (DDSConv.convs_sep, modules/flow/modules.py:46-56,63);
 * w [C, 1, K], bias [C] or NULL. */
int amp_dwconv(const float* x_dev, const float* w_dev, const float* bias_dev, const int* lens_dev, int B, int C, int T, int K,
               int dilation, float* y_dev, void* stream);
/* ConvFlow's spline step (modules/flow/modules.py:435-458 + modules/transformer/transforms.py:56-215): the piecewise
 * rational-quadratic transform with linear tails of one channel of z [B, 2, T] given h [B, 3*num_bins - 1, T] (masked
 * here), both channels * mask; flip_in / flip_out fold the neighbouring Flip layers (:315-321) in.  z_out != z. */
int amp_spline_flow(const float* z_dev, const float* h_dev, const int* lens_dev, int B, int T, int num_bins,
                    int filter_channels, float tail_bound, int inverse, int flip_in, int flip_out, float* z_out_dev,
                    void* stream);
/* amp_spline_flow with ConvFlow's `proj` (modules/flow/modules.py:418,427: Conv1d(C, 3*num_bins - 1, 1)) evaluated inside: hc_dev [B, C, T] is
 * the DDSConv output, proj_w_dev [3*num_bins - 1, C] / proj_b_dev [3*num_bins - 1] (or NULL) the conv's parameters as stored, on the
 * device.  The projection is a plain fp32 dot product (not the f16x3 conv arithmetic).  C <= 256; AMP_ERR_UNSUPPORTED otherwise. */
int amp_spline_flow_proj(const float* z_dev, const float* hc_dev, const float* proj_w_dev, const float* proj_b_dev, const int* lens_dev,
                         int B, int C, int T, int num_bins, int filter_channels, float tail_bound, int inverse, int flip_in, int flip_out,
                         float* z_out_dev, void* stream);
/* ElementwiseAffine reverse: (x - m) * exp(-logs) * mask (modules/flow/modules.py:338-340); m, logs [C]. */
int amp_affine_reverse(const float* x_dev, const float* m_dev, const float* logs_dev, const int* lens_dev, int B, int C, int T,
                       float* y_dev, void* stream);
/* emb(tokens) * scale, transposed to [B, hidden, T] and masked (TextEncoder.forward vits.py:58-62); tokens int64 [B, T],
 * weight [n_vocab, hidden]. */
int amp_embed_tokens(const long long* tokens_dev, const float* weight_dev, const int* lens_dev, int B, int T, int hidden,
                     int n_vocab, float scale, float* y_dev, void* stream);
/* w_ceil = ceil(exp(logw) * mask * length_scale) [B, T], its running sum (int32 [B, T]) and y_len = max(sum, 1)
 * (vits.py:341-343, utils/util.py:633). */
int amp_durations(const float* logw_dev, const int* lens_dev, int B, int T, float length_scale, float* w_ceil_dev, int* cum_dev,
                  int* ylen_dev, void* stream);
/* out[b, :, y] = src[b, :, x(y)] along the monotonic path cum[x-1] <= y < cum[x] ( = generate_path(...) @ src,
 * utils/util.py:625-640, vits.py:345-353); attn_dev [B, 1, Ty, Tx] receives the path itself when not NULL. */
int amp_expand_path(const float* src_dev, const int* cum_dev, const int* xlens_dev, const int* ylens_dev, int B, int D, int Tx,
                    int Ty, float* out_dev, float* attn_dev, void* stream);
/* The same with src a channel slice of a wider tensor (m / logs = the halves of the text encoder's stats, vits.py:65): element
 * (b, d, x) at src_dev[b*src_batch_stride + d*Tx + x]. */
int amp_expand_path_strided(const float* src_dev, long long src_batch_stride, const int* cum_dev, const int* xlens_dev,
                            const int* ylens_dev, int B, int D, int Tx, int Ty, float* out_dev, float* attn_dev, void* stream);
/* z_p = m + noise * exp(logs) * noise_scale over n elements (vits.py:355). */
int amp_gauss_sample(const float* m_dev, const float* logs_dev, const float* noise_dev, size_t n, float noise_scale,
                     float* out_dev, void* stream);

/* Snake.forward / SnakeBeta.forward as a stand-alone element-wise op (modules/activation_functions/snake.py:51-61,
 * 110-122): y = x + sin(a x)^2 / (b + 1e-9) over [B, C, T]; alpha_dev / beta_dev: [C] on the device, beta_dev NULL
 * for Snake (b = a); logscale: a = exp(alpha), b = exp(beta). */
int amp_snake(const float* x_dev, int B, int C, int T, const float* alpha_dev, const float* beta_dev, int logscale,
              float* y_dev, void* stream);

/* fp32 waveform [B, L] (row stride wav_stride elements) -> signed 16-bit PCM [B, L] (row stride pcm_stride), on
 * the device, so the D2H copy and any gather move 2 bytes per sample instead of 4.  Replaces the host conversion
 * inside the reference's save_audio (utils/io.py:68-76 -> torchaudio.save(encoding="PCM_S", bits_per_sample=16);
 * torchaudio 2.0.2 + libsox 14.4.2 semantics: sample = x * 2^31 clamped and truncated to int32, then rounded
 * half-up to 16 bits with saturation).  lens_dev (samples per row, may be NULL) zeroes the tail of each row: the
 * crop `[: l * hop_size]` of models/vocoders/vocoder_inference.py:359.  Bit-exact against oracle/pcm16.py. */
int amp_wav_to_pcm16(const float* wav_dev, int B, int L, long long wav_stride, const int* lens_dev, int16_t* pcm_dev,
                     long long pcm_stride, void* stream);

/* Per-conv options (default 0).  PAD_REFLECT: columns outside the input mirror instead of reading zero, i.e.
 * nn.ReflectionPad1d(p) followed by an unpadded conv == this conv created with padding = p (MelGAN,
 * models/vocoders/gan/generator/melgan.py:39,56,92); needs p < T.  TANH: tanh on store (melgan.py:94). */
typedef enum amp_conv_option { AMP_CONV_OPT_PAD_REFLECT = 1, AMP_CONV_OPT_TANH = 2 } amp_conv_option;
int amp_conv_set_option(amp_conv* c, int option, int value);

/* One ResBlock1 iteration fused in a single kernel (hifigan.py:93-100):
 *     y = x + c2( leaky_relu( c1( leaky_relu(x, slope) ), slope ) )
 * c1: Conv1d(C, C, k, dilation d, 'same' padding), c2: Conv1d(C, C, k, dilation 1), both with bias, built
 * with amp_conv_create under AMP_PRECISION_F16X3.  Covered: C in {32, 64, 128}, k in {3, 5, 7, 11} and
 * (k-1)*d <= 64; anything else returns AMP_ERR_UNSUPPORTED (amp_gen_forward then runs the two convs).
 * y_dev must not alias x_dev. */
int amp_pair_forward(const amp_conv* c1, const amp_conv* c2, const float* x_dev, int B, int T, float slope,
                     float* y_dev, void* stream);

/* ResBlock1.forward (hifigan.py:93-100) in ONE launch:  for p < n_pairs:  x = x + c2[p](lrelu(c1[p](lrelu(x)))), on the
 * whole-resblock kernel (csrc/rb_f16x3.hip: x read once, y written once, the residual carried in registers).  Handles from
 * amp_conv_create as for amp_pair_forward (same C and k for all pairs, c1[p] dilated, c2[p] dilation 1, 'same' padding);
 * covered: C in {32, 64} with k in {3, 5, 7, 11}, C = 128 with k in {3, 5}, (k-1)/2 * dilation within the tile's guard columns (32;
 * 16 at C = 128), n_pairs <= 3, f16x3 arithmetic, under the shapes the current
 * amp_set_resblock_fusion mode admits -- otherwise AMP_ERR_UNSUPPORTED (run amp_pair_forward n_pairs times: the same bits).
 * y_dev must not alias x_dev. */
int amp_resblock_forward(const amp_conv* const* c1, const amp_conv* const* c2, int n_pairs, const float* x_dev, int B, int T,
                         float slope, float* y_dev, void* stream);
/* 0: generators run every ResBlock1 as fused pairs; 1 (default): the measured policy (whole-resblock kernel for the narrow
 * late stages when the launch fills the chip); 2: wherever the kernel is built, any grid; 3: as 2 with the four-wave
 * tiles (two workgroups per CU) at C = 32 (512 columns) and at C = 64, k <= 5 (256 columns).  Bit-identical results in every mode (tests/test_gpu_resblock.py); env AMP_RB_FUSION. */
int amp_set_resblock_fusion(int mode);
/* The n_kernels resblocks of a generator stage on CONCURRENT streams (they read the same stage tensor and only meet in the MRF mean,
 * hifigan.py:208-214 / bigvgan.py:320-327): -1 (default) while B * T <= 4096 mel frames -- small batches, whose launches do not
 * fill the chip and whose forward is a chain of ~50 dependent launches; 0 never; 1 always.  Only the launch of each resblock that
 * accumulates into the mean waits for the previous resblock's (an event) and keeps its `=` / `+=` / `(y + v) / n` form: bit-identical
 * results in every mode.  amp_gen_workspace_bytes accounts for the extra R / TMP buffers; the side streams fork from and join
 * `stream` by events (legal under stream capture).  They are created by the first uncaptured forward or by amp_gen_prepare_streams
 * -- a forward captured before either runs the sequential chain.  Not used while amp_gen_set_profiling is on.  Env AMP_RB_STREAMS. */
int amp_set_resblock_streams(int mode);
int amp_gen_prepare_streams(amp_gen* g);

/* AMPBlock1.forward of BigVGAN (bigvgan.py:137-146) in ONE launch:
 *     for p < n_pairs:  x = x + c2[p]( a[2p+1]( c1[p]( a[2p](x) ) ) ),   a[i] = Activation1d(Snake | SnakeBeta)
 * on the whole-AMPBlock kernel (csrc/ampb_f16x3.hip: x read once, y written once, the six anti-aliased activations
 * (act.py:31-36, resample.py:36-65, filter.py:92-99, snake.py:51-61) evaluated in registers between the convs).  Bit-identical
 * to running amp_antialias_snake / amp_conv_forward[_mrf] one by one (tests/test_gpu_ampblock.py).
 * Handles from amp_conv_create under AMP_PRECISION_F16X3: same C and k for all convs, c1[p] dilated, c2[p] dilation 1, 'same'
 * zero padding, with bias.  alpha_dev / beta_dev: [2 * n_pairs, C] per-channel parameters AS STORED (exp() applied when
 * logscale; beta_dev NULL -> Snake); filt_up_host / filt_down_host: the 12 filter taps shared by all activations.
 * mode 0: y = v;  1: y = y + v;  2: y = (y + v) / div  (the MRF accumulation of the generator, bigvgan.py:320-327).
 * Covered: C in {32, 64}, k in {3, 5, 7, 11}, (k-1)/2 * dilation <= 32, n_pairs <= 3, T % 4 == 0, under
 * the shapes the current amp_set_ampblock_fusion mode admits -- otherwise AMP_ERR_UNSUPPORTED.  y_dev must not alias x_dev.
 * Op-level convenience: synchronises the stream. */
int amp_ampblock_forward(const amp_conv* const* c1, const amp_conv* const* c2, int n_pairs, const float* alpha_dev,
                         const float* beta_dev, int logscale, const float* filt_up_host, const float* filt_down_host,
                         const float* x_dev, int B, int T, float* y_dev, int mode, float div, void* stream);
/* 0: BigVGAN generators run every AMPBlock as separate conv / activation launches; 1 (default): the whole-AMPBlock kernel
 * where it is built and the launch fills the chip; 2: wherever it is built, any grid; 3: as 2 with the four-wave 512-column
 * tiles at C = 32.  Bit-identical results in every mode; env AMP_AMPB_FUSION.  -1: back to the default. */
int amp_set_ampblock_fusion(int mode);

void amp_conv_destroy(amp_conv* c);

/* Frame-rate convs (short contraction, small grid: the convs around the VITS decoder) run on a kernel that stages the
 * whole K extent of its input tile at once (csrc/conv_small_f16x3.hip; same bits as the pipelined kernel).  0 keeps
 * them on the pipelined kernel -- an A/B and cross-check switch (no environment form since round 3). */
int amp_set_small_conv(int on);

/* Transposed convs and k = 3 / 7 / 11 convs whose GEMM rows are a multiple of 256 (ConvTranspose1d: Cout * stride) run, on grids
 * of 512+ workgroups, on the row-blocked kernel (csrc/conv_blk_f16x3.hip: 64 rows per wave, x staged once per 256 rows;
 * same bits as the pipelined kernel).  mode 0 keeps them on the pipelined kernel, 1 = one 16-channel chunk per staging
 * round, 2 = two where available, 3 = 2 + k = 7 / 11 on the A-fragment-ring form (default), -1 = back to
 * the default -- an A/B and cross-check switch. */
int amp_set_conv_blk(int mode);
/* Conv1d with 128 output rows (BigVGAN's unpaired AMPBlock convs at C = 128) on the row-blocked kernel, two waves along the columns: 1
 * (default) the measured policy (k = 7 / 11), 2 every tap count the kernel is built for, 0 all on the pipelined kernel.  Same bits in every
 * mode (an A/B and cross-check switch); -1: default. */
int amp_set_conv_blk_narrow(int on);

/* Convs with several row groups (more GEMM rows than one workgroup holds) launch with the row group as the fastest grid
 * index: the row groups of one x tile run back to back on one XCD and share its L2 copy of x (same bits; 1 = default).
 * 0 = the 2-D grid with the row group in blockIdx.y, -1 = back to the default (on) -- an A/B switch. */
int amp_set_conv_rg_fast(int on);

/* Ping-pong tile order: every other conv / fused-pair launch walks its tiles in descending order, starting on the part of its
 * input that the previous launch wrote last (same bits).  -1 = back to the default (on) -- an A/B switch. */
int amp_set_pingpong(int on);

/* ---- WN (modules/flow/modules.py:74-151), fused: two launches per layer ---- */

/* WN.in_layers[i] = Conv1d(H, 2H, k, dilation, padding) (modules.py:106-114) built for the gate epilogue: the kernel
 * forms fused_add_tanh_sigmoid_multiply (utils/util.py:602-609) in registers and writes [B, H, T].  weight_host
 * [2H, H, k] FOLDED (weight_norm applied), bias_host [2H].  Covered: H a multiple of 32 and <= 256, k in {1, 3, 5},
 * 'same' padding, (k-1)*dilation <= 64, f16x3 arithmetic; otherwise AMP_ERR_UNSUPPORTED (run the unfused ops).
 * The handle only runs inside amp_wn_forward. */
int amp_conv_create_gated(int hidden, int k, int dilation, int padding, const float* weight_host, const float* bias_host,
                          amp_conv** out);

/* WN.forward (modules/flow/modules.py:126-151) without the final `* x_mask`:
 *   for i: acts = tanh((in_i(x) + g_i)[:H]) * sigmoid((in_i(x) + g_i)[H:]);  rs = res_skip_i(acts)
 *          i < n-1: x = (x + rs[:H]) * mask, output += rs[H:];   i == n-1: output += rs
 * in_layers[i]: amp_conv_create_gated handles; res_skip_layers[i]: amp_conv_create 1x1 convs (H -> 2H, last H -> H).
 * x_dev [B, H, T] is the caller's working copy and is MODIFIED; cond_dev = cond_layer(g) [B, 2H*n_layers] (element
 * (b, r) at cond_dev[b*cond_batch_stride + r]) or NULL; lens_dev int32 [B] or NULL; acts_ws_dev scratch [B, H, T];
 * out_dev [B, H, T] (written, not read).  With lens_dev, columns t >= lens[b] of out_dev (and of x_dev) are UNSPECIFIED --
 * whole tiles beyond an utterance's end are skipped and never stored: ASSIGN zero there (amp_sequence_mask does, as
 * WN.forward's final `* x_mask` would), do not multiply what is left in them by a mask (0 * NaN). */
int amp_wn_forward(const amp_conv* const* in_layers, const amp_conv* const* res_skip_layers, int n_layers, float* x_dev,
                   const float* cond_dev, long long cond_batch_stride, const int32_t* lens_dev, int B, int T,
                   float* acts_ws_dev, float* out_dev, void* stream);

/* ---- VITS posterior encoder + flow (config 5): element-wise pieces between the convs ---- */

/* fused_add_tanh_sigmoid_multiply (utils/util.py:602-609) as called by WN.forward
 * (modules/flow/modules.py:141): out[b,c,t] = tanh(a[b,c,t] + g[b,c]) * sigmoid(a[b,c+H,t] + g[b,c+H]).
 * a_dev [B, 2H, T]; cond_dev: this layer's slice of cond_layer(g) (time-constant), element (b, c) at
 * cond_dev[b*cond_batch_stride + c], or NULL when g is None; out_dev [B, H, T]. */
int amp_wn_gate(const float* a_dev, const float* cond_dev, long long cond_batch_stride, float* out_dev, int B, int H,
                int T, void* stream);
/* WN residual/skip update (modules/flow/modules.py:144-151): not last: x = (x + rs[:, :H]) * mask,
 * out += rs[:, H:]; last: out += rs.  first != 0 starts `out` from zero (torch.zeros_like, :127).
 * lens_dev: int32 [B] valid lengths (sequence_mask, utils/util.py:618-622) or NULL for no mask. */
int amp_wn_accumulate(float* x_dev, float* out_dev, const float* rs_dev, const int32_t* lens_dev, int B, int H, int T,
                      int first, int last, void* stream);
/* x[b, :, t >= lens[b]] = 0   (`* x_mask`, vits.py:147,149; modules/flow/modules.py:152,381,383) */
int amp_sequence_mask(float* x_dev, const int32_t* lens_dev, int B, int C, int T, void* stream);
/* Mean-only ResidualCouplingLayer update on the second half of x [B, 2h, T] in place
 * (modules/flow/modules.py:390-397): forward x1 = m + x1*mask; reverse x1 = (x1 - m)*mask. */
int amp_coupling_apply(float* x_dev, const float* m_dev, const int32_t* lens_dev, int B, int half_channels, int T,
                       int reverse, void* stream);
/* Flip.forward: torch.flip(x, [1]) (modules/flow/modules.py:314-321); y must not alias x. */
int amp_flip_channels(const float* x_dev, float* y_dev, int B, int C, int T, void* stream);
/* PosteriorEncoder sampling (models/tts/vits/vits.py:150-151): stats = [m ; logs] [B, 2C, T],
 * z = (m + eps * exp(logs)) * mask. */
int amp_posterior_sample(const float* stats_dev, const float* eps_dev, const int32_t* lens_dev, float* z_dev, int B,
                         int C, int T, void* stream);

/* Activation1d(Snake|SnakeBeta) (modules/anti_aliasing/act.py:31-36; resample.py:36-45,62-65;
 * filter.py:92-99; snake.py:51-61,110-122), ratio 2, 12-tap filters.  alpha_dev/beta_dev: per-channel
 * parameters AS STORED (the kernel applies exp() when logscale); beta_dev NULL -> Snake.
 * filt_up_host/filt_down_host: the 12 filter taps.  x_dev [B, C, T] -> y_dev [B, C, T]. */
int amp_antialias_snake(const float* x_dev, int B, int C, int T, const float* alpha_dev, const float* beta_dev,
                        int logscale, const float* filt_up_host, const float* filt_down_host, float* y_dev,
                        void* stream);

/* Mel / STFT front end descriptor: cfg.preprocess.{sample_rate,n_fft,win_size,hop_size,n_mel,fmin,fmax}. */
typedef struct amp_mel_desc {
    uint32_t struct_size; /* sizeof(amp_mel_desc) AS THE CALLER WAS COMPILED (since amp_version 140): the library reads nothing
                            beyond it, so fields appended by later versions default to 0 / NULL for older consumers; a value
                            below the round-2 fields (up to mel_bands_dev) is refused with AMP_ERR_INVALID */
    int32_t n_fft;
    int32_t win_size;
    int32_t hop_size;
    int32_t n_mel;       /* 0 = linear spectrogram only */
    int32_t pad_mode;    /* 0: reflect-pad (n_fft-hop)/2, center=False (utils/mel.py:145-164);
                            1: reflect-pad n_fft/2 (utils/stft.py:152-165, TacotronSTFT) */
    float mag_eps;       /* added under the sqrt: 1e-9 (mel.py:166), 1e-6 (mel.py:99), 0 (stft.py:177) */
    float log_clip;      /* clamp before log: 1e-5 (mel.py:10-12); <=0 -> no log (raw mel / magnitude) */
    const int32_t* mel_bands_dev; /* optional, device: [n_mel][2] = first and one-past-last FFT bin with a non-zero
                            weight in each row of melbasis (librosa's triangular filters touch 2..40 of the 513 bins);
                            NULL = every row is summed over all bins.  Must cover every non-zero of the basis. */
    /* Optional sample-range report of utils/mel.py:21-24 ("min value is" / "max value is" when the audio leaves [-1, 1])
     * without a reduction pass of its own.  range_dev: device int32[3] = { bits of the smallest sample < -1 seen
     * (initialised to the bits of -1.0f = nothing seen), bits of the largest sample > 1 (initialised to +1.0f), range_seq };
     * the kernel folds the samples it reads anyway into the first two (atomic max of the int bits: monotone for both) and
     * stores range_seq into the third.  range_host (pinned host memory, 3 x int32): the call enqueues a copy of the three
     * words behind the kernel; the caller knows it has landed when word 2 equals range_seq.  range_reset_dev: another
     * int32[3] the kernel re-initialises for a LATER call (a ring of slots then needs no reset launch).  NULL = off. */
    int32_t* range_dev;
    int32_t* range_host;
    int32_t* range_reset_dev;
    int32_t range_seq;
} amp_mel_desc;

/* One-time set-up of the n_fft = 1024 front-end kernel on the CURRENT device (thread-safe, idempotent, blocking): call it
 * before capturing a stream that contains amp_mel_forward.  Without it the first forward does the same lazily. */
int amp_mel_init(void);
/* Number of frames produced for L samples. */
int amp_mel_num_frames(const amp_mel_desc* d, int L);
/* Replaces extract_mel_features / mel_spectrogram_torch / extract_linear_features (utils/mel.py:20-170)
 * and TacotronSTFT.mel_spectrogram (utils/stft.py:259-278).
 * wav_dev [B, L]; window_dev [n_fft] (already centre-padded to n_fft); melbasis_dev [n_mel, n_fft/2+1]
 * (may be NULL when n_mel == 0).  Outputs (any may be NULL): mel_dev [B, n_mel, F] (log-mel),
 * mag_dev [B, n_fft/2+1, F] (magnitude), re_dev/im_dev [B, n_fft/2+1, F].
 * The first n_fft = 1024 call on a device sets the kernel up (a 4.5-KB twiddle table, uploaded with a blocking copy under a
 * lock): inside a stream capture that first call is refused with AMP_ERR_STATE -- call amp_mel_init() (or one forward) first.
 * COST by transform length (64 utterances of 65 536 samples, one MI355X): n_fft 1024 / 2048 / 512 / 1920 run one WAVE per frame (register butterflies,
 * 0.040 / 0.051 / 0.053 / 0.103 ms); other powers of two and lengths whose prime factors are <= 13 run one WORKGROUP per frame (about 0.4-1 ms); a
 * prime factor p > 13 adds a run-time radix pass of n_fft * p complex multiply-adds per frame, so a PRIME n_fft is a direct O(n_fft^2) DFT (4093: 16.7 M
 * per frame, three orders of magnitude slower per sample than 1024) -- accepted because torch.stft accepts it, not because it is a sensible choice. */
int amp_mel_forward(const amp_mel_desc* d, const float* wav_dev, int B, int L, const float* window_dev,
                    const float* melbasis_dev, float* mel_dev, float* mag_dev, float* re_dev, float* im_dev,
                    void* stream);

/* Ragged batch of the same: row b of wav_dev [B, L] holds lens_dev[b] <= L samples (int32, device), zero-padded.
 * Every utterance is reflect-padded at ITS OWN end and fills frames [0, amp_mel_num_frames(d, lens[b])) of its
 * output rows -- identical to running it alone (what the reference's one-file-at-a-time feature extraction,
 * processors/acoustic_extractor.py:376-403, computes); later frames of a row are left unwritten. */
int amp_mel_forward_ragged(const amp_mel_desc* d, const float* wav_dev, const int32_t* lens_dev, int B, int L,
                           const float* window_dev, const float* melbasis_dev, float* mel_dev, float* mag_dev,
                           float* re_dev, float* im_dev, void* stream);

/* Backward of amp_mel_forward for the training-time mel loss (models/vocoders/gan/gan_vocoder_trainer.py:387-392:
 * 45 * L1(extract_mel_features(y_gt), extract_mel_features(y_pred)) differentiated w.r.t. y_pred).  Inputs are what a
 * forward with log_clip <= 0 returns for the same audio -- mel_linear_dev [B, n_mel, F] (mel energies BEFORE the log),
 * mag_dev / re_dev / im_dev [B, n_fft/2+1, F] -- plus grad_logmel_dev [B, n_mel, F] = d loss / d log(max(mel,
 * d->log_clip)) (d->log_clip <= 0: the gradient w.r.t. the linear mel).  Output grad_wav_dev [B, L].
 * spec_ws_dev: scratch of 2 * B * (n_fft/2+1) * F floats, frames_ws_dev: B * F * n_fft floats.  lens_dev as in
 * amp_mel_forward_ragged (NULL = full rows). */
int amp_mel_backward(const amp_mel_desc* d, const int32_t* lens_dev, int B, int L, const float* window_dev,
                     const float* melbasis_dev, const float* mel_linear_dev, const float* mag_dev, const float* re_dev,
                     const float* im_dev, const float* grad_logmel_dev, float* spec_ws_dev, float* frames_ws_dev,
                     float* grad_wav_dev, void* stream);

/* Replaces STFT.inverse (utils/stft.py:183-222; used by STFT.forward and griffin_lim :78-95): magnitude and
 * phase [B, n_fft/2+1, F] -> waveform [B, hop*(F-1) + (n_fft & 1)] (overlap-add of the windowed inverse FFTs, divided by the
 * window-sum-square envelope where it exceeds float32 tiny, times n_fft/hop, floor(n_fft/2) cropped per side).
 * window_dev [n_fft]; wss_dev [n_fft + hop*(F-1)] = window_sumsquare (stft.py:19-75) as float32;
 * frames_ws_dev: scratch of B*F*n_fft floats. */
int amp_istft_forward(const amp_mel_desc* d, const float* mag_dev, const float* phase_dev, int B, int F,
                      const float* window_dev, const float* wss_dev, float* frames_ws_dev, float* wav_dev, void* stream);

/* APNet's ISTFT with "same" padding (apnet.py:16-101): complex spectrogram (re, im) [B, n_fft/2+1, F] ->
 * waveform [B, F * hop]: overlap-add of irfft(spec) * window, cropped by (win - hop)/2 per side, divided by the
 * window envelope.  envelope_dev [hop*(F-1) + win] = overlap-added window^2 (uncropped); needs win_size == n_fft. */
int amp_istft_same(const amp_mel_desc* d, const float* re_dev, const float* im_dev, int B, int F, const float* window_dev,
                   const float* envelope_dev, float* frames_ws_dev, float* wav_dev, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* AMPHION_HIP_H */
