// Internal declarations shared by the translation units of libamphion_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include "amphion_hip.h"

namespace amp {

constexpr int KC = 8;     // f32 kernel: input channels staged per K-chunk (2 per MFMA k-step, 4 k-steps per tap)
constexpr int KC16 = 16;  // f16x3 kernel: input channels per K-chunk (one 32x32x16 MFMA k-extent per tap)

// Arithmetic of the conv contraction (amp_set_precision / AMP_PRECISION).
enum { PREC_F32 = 0, PREC_F16X3 = 1 };

// Arguments of the implicit-GEMM conv kernel (conv_mfma.hip).
//   Y'[m, q] = sum_i sum_j  W'[m, i, j] * act_in(X[i, q + off0 + j*dstep])          (zero outside [0, Tin))
// For a Conv1d:          m = output channel, off_j = (j - (k-1)/2) * dilation, up = 1.
// For a ConvTranspose1d: m = o*up + r  (polyphase row), taps s: X[i, q - s], W' = W[i, o, r + s*up];
//                        the result is scattered to n = q*up + r - up_pad.
struct ConvArgs {
    const float* x;      // [B, Cin, Tin]
    const void* wp;      // packed A fragments (f32, or f16 hi/lo planes for f16x3), see conv_build()
    const float* bias;   // [Cout] or nullptr
    const float* res;    // [B, Cout, Tout] or nullptr (may alias y)
    float* y;            // [B, Cout, Tout]
    int B, Cin, Tin;
    long long xbs;       // batch stride of x in elements (Cin*Tin when dense; larger for a channel slice)
    int nchunks;         // ceil(Cin / KC)
    int M;               // GEMM rows = Cout * up
    int Tq;              // GEMM columns per batch item
    int tiles_per_item;  // ceil(Tq / NT)
    int rev;             // 1: walk the tiles in descending order.  Successive launches alternate (conv_run / pair_run): a launch's
                         //    input is mostly the previous launch's output, whose LAST-written part is what the 256-MB Infinity
                         //    Cache still holds when the next launch starts
    int row_groups;      // > 0: 1-D grid of B * tiles_per_item * row_groups workgroups with the row group as the FASTEST index
                         //      (the row groups of one x tile run back to back on one XCD: x is fetched into ONE L2 once
                         //      instead of once per row group from HBM); 0: 2-D grid, row group = blockIdx.y
    int off0, dstep;     // tap offsets
    int halo_left;       // max(0, -min_j off_j)
    int wd;              // staged columns actually needed: NT + halo_left + halo_right
    int Cout, Tout;
    int up, up_pad;
    float slope_in, slope_out;  // 1.0f = identity
    int mode;                   // 0: y = v   1: y = y + v   2: y = (y + v) / div
    float div;
    float acc_scale, inv_scale; // f16x3 only: 16 * 2^s (operand scaling) and its reciprocal
    // ragged batches: item b's input is valid on [0, lens[b] * len_mul) and reads as zero beyond (every
    // layer zero-pads at the utterance's OWN end, so a padded batch equals the per-utterance results)
    const int* lens;            // [B] on the device, or nullptr (all items valid on [0, Tin))
    int len_mul;
    unsigned* range_flag;       // f16x3: set to 1 when a staged operand leaves the f16 range (|x| * 16 > 65504) or is not finite
    int pad_reflect;            // 1: out-of-range columns mirror (nn.ReflectionPad1d + unpadded conv, melgan.py:39,56,92)
    int tanh_out;               // 1: tanh on store (melgan.py:94)
    // ---- conv_small_f16x3.hip only (frame-rate convs; the other kernels ignore these) ----
    int Mpad;                   // rows of the packed weight (M rounded up to the pack's row group)
    const float* gate_cond;     // EPI_GATE: additive condition per (item, ORIGINAL row) or nullptr (WN's g_l, modules.py:135-139)
    long long gate_cond_bs;     //           its batch stride in elements
    int wn_H;                   // EPI_GATE / EPI_WNACC: hidden channels H
    float* wn_x;                // EPI_WNACC: x [B, H, T], updated in place: x = (x + rs[:H]) * mask
    float* wn_out;              //            output [B, H, T]: (+)= rs[H:] (last layer: (+)= rs)
    int wn_first, wn_last;      //            first layer starts `output` from zero; last layer has H rows only
};

// Arguments of the fused ResBlock-pair kernel (pair_f16x3.hip):
//   y = x + c2(lrelu(c1(lrelu(x))))   [mode 1: y += ..., mode 2: y = (y + ...) / div]
struct PairArgs {
    const float* x;            // [B, C, T] input and residual
    float* y;                  // [B, C, T] output; must NOT alias x (other tiles read x's halo)
    const void* wp1;           // conv1 (kernel k, dilation dil): packed f16x3 A fragments
    const float* bias1;
    const void* wp2;           // conv2 (kernel k, dilation 1)
    const float* bias2;
    int B, C, T;
    int tiles_per_item;        // per-tile kernel (pair_f16x3.hip): ceil(T / NT)
    int strip_len;             // strip kernel (pair_strip_f16x3.hip): output columns per workgroup ...
    int strips_per_item;       // ... and workgroups per batch item, ceil(T / strip_len)
    int wide;                  // strip kernel: 0 = four-wave strips, 3 = the 2 x 2-blocked A-ring form (one workgroup per CU)
    int rev;                   // 1: descending tile / strip order, see ConvArgs::rev
    int dil;
    float slope;               // leaky_relu slope (on x while staging, on conv1's output at the seam)
    float sc1, isc1, sc2, isc2;  // 16 * 2^s operand scaling of each conv and its reciprocal
    int mode;
    float div;
    const int* lens;           // ragged batches, see ConvArgs
    int len_mul;
    unsigned* range_flag;      // see ConvArgs
};

// Three whole-K convs in one grid (conv_small3_f16x3.hip): a[0] / a[1] / a[2] = the k = 11 / 7 / 3 conv (standard epilogue) of a stage's three
// resblocks, nx[j] = B * tiles_per_item column tiles x ny[j] = ceil(M / 128) row groups each
struct ConvSmall3Args {
    ConvArgs a[3];
    int nx[3], ny[3];
};
hipError_t launch_conv_small3(const ConvSmall3Args& p, const int ni[3], hipStream_t stream);

// Three fused pairs in one grid (pair3_f16x3.hip): a[0] / a[1] / a[2] = the k = 11 / 7 / 3 pair of a stage's three resblocks (per-tile form:
// tiles_per_item set), n[j] = B * a[j].tiles_per_item workgroups each
struct Pair3Args {
    PairArgs a[3];
    int n[3];
};
hipError_t launch_pair3(const Pair3Args& p, hipStream_t stream);

// Arguments of the whole-ResBlock kernel (rb_f16x3.hip): for p < np:  x = x + c2_p(lrelu(c1_p(lrelu(x))))   [then the MRF modes]
constexpr int AMP_RB_MAX_PAIRS = 3;
struct RbArgs {
    const float* x;            // [B, C, T] input of the block
    float* y;                  // [B, C, T] output; must NOT alias x (other tiles read x's halo)
    const void* wp1[AMP_RB_MAX_PAIRS];     // c1_p (kernel k, dilation dil[p]): packed f16x3 A fragments
    const float* bias1[AMP_RB_MAX_PAIRS];
    const void* wp2[AMP_RB_MAX_PAIRS];     // c2_p (kernel k, dilation 1)
    const float* bias2[AMP_RB_MAX_PAIRS];
    float sc1[AMP_RB_MAX_PAIRS], isc1[AMP_RB_MAX_PAIRS], sc2[AMP_RB_MAX_PAIRS], isc2[AMP_RB_MAX_PAIRS];   // 16 * 2^s and reciprocal
    int dil[AMP_RB_MAX_PAIRS];
    int np;                    // pairs in the block (1 .. AMP_RB_MAX_PAIRS)
    int rh;                    // one-sided receptive field of the block: sum_p (k-1)/2 * (dil[p] + 1) columns
    int B, C, T;
    int tiles_per_item;        // ceil(T / (W - 2 * rh))
    int rev;                   // 1: descending tile order, see ConvArgs::rev
    float slope;               // leaky_relu slope
    int mode;                  // 0: y = v   1: y = y + v   2: y = (y + v) / div   (the LAST pair's MRF mode)
    float div;
    const int* lens;           // ragged batches, see ConvArgs
    int len_mul;
    unsigned* range_flag;      // see ConvArgs
};

// Arguments of the whole-AMPBlock kernel (ampb_f16x3.hip), BigVGAN's AMPBlock1.forward (bigvgan.py:137-146):
//   for p < np:  x = x + c2_p( a_{2p+1}( c1_p( a_{2p}(x) ) ) )     [then the MRF modes]
// as ns = 2 * np STEPS  "Activation1d, then conv":  step s = 2p is (a_{2p}, c1_p), step s = 2p + 1 is (a_{2p+1}, c2_p) + residual.
constexpr int AMP_AMPB_MAX_STEPS = 6;
struct AmpbArgs {
    const float* x;            // [B, C, T] input of the block
    float* y;                  // [B, C, T] output (mode != 0: also the running MRF sum); must NOT alias x
    const void* wp[AMP_AMPB_MAX_STEPS];        // conv of step s: packed f16x3 A fragments (conv_build)
    const float* bias[AMP_AMPB_MAX_STEPS];
    float sc[AMP_AMPB_MAX_STEPS], isc[AMP_AMPB_MAX_STEPS];   // 16 * 2^s operand scaling of each conv and its reciprocal
    int dil[AMP_AMPB_MAX_STEPS];
    const float* act_a[AMP_AMPB_MAX_STEPS];    // Activation1d of step s: alpha (exp'ed when logscale), [C]
    const float* act_invb[AMP_AMPB_MAX_STEPS]; // 1 / (beta + 1e-9), [C]
    const float* act_fu[AMP_AMPB_MAX_STEPS];   // 12 up-sampling taps, each x 2 (the gain of UpSample1d folded in on the host: exact)
    const float* act_fd[AMP_AMPB_MAX_STEPS];   // 12 down-sampling taps
    int ns;                    // steps (2 per pair)
    int rh;                    // tile halo either side: the block's one-sided receptive field rounded up to a multiple of 4
    int B, C, T;               // T % 4 == 0 (16-B rows)
    int tiles_per_item;        // ceil(T / (W - 2 * rh))
    int rev;                   // 1: descending tile order, see ConvArgs::rev
    int mode;                  // 0: y = v   1: y = y + v   2: y = (y + v) / div   (the LAST conv's MRF mode)
    float div;
    const int* lens;           // ragged batches, see ConvArgs
    int len_mul;
    unsigned* range_flag;      // see ConvArgs
};

struct ConvPlan {
    int KT;      // taps compiled into the kernel (1,2,3,5,7,11)
    int WM, WN;  // waves along M / N (WM*WN == 4)
    int NI;      // 32-column MFMA tiles per wave
    int HALO;    // staged halo capacity (64 or 128)
    int NT() const { return 32 * NI * WN; }
    int Mgroup() const { return 32 * WM; }
};

// Chooses the kernel variant for a conv with `ntaps` taps, GEMM rows M, halo_total columns and Tq
// columns per item.  Returns false if unsupported.
bool choose_plan(int ntaps, int M, int halo_total, int Tq, ConvPlan* plan);
hipError_t launch_conv(const ConvPlan& plan, const ConvArgs& a, hipStream_t stream);        // exact f32 MFMA
hipError_t launch_conv_f16x3(const ConvPlan& plan, const ConvArgs& a, hipStream_t stream);  // split-f16 MFMA
// frame-rate convs (conv_small_f16x3.hip): whole-K staging; epi 0 standard, 1 gate, 2 WN accumulate; ni 2 | 4
constexpr int kSmallConvMaxChunks = 16;
hipError_t launch_conv_small(int KT, int ni, int epi, const ConvArgs& a, hipStream_t stream);
// fused pair: output columns per workgroup for (C, k, dilation), 0 = not covered; launch
int pair_tile(int k, int C, int dil);
hipError_t launch_pair(int k, const PairArgs& a, hipStream_t stream);
// strip-mined fused pair (pair_strip_f16x3.hip): columns per step for (C, k, dilation), 0 = not covered
int strip_step(int k, int C, int dil, int wide, int* wg_per_cu);
hipError_t launch_strip(int k, const PairArgs& a, hipStream_t stream);

// whole ResBlock1 (rb_f16x3.hip): tile width for (C, k, max dilation) in form `wide`, 0 = not covered; launch
int rb_tile(int k, int C, int max_dil, int wide);
hipError_t launch_rb(int k, const RbArgs& a, int wide, hipStream_t stream);

// whole AMPBlock1 (ampb_f16x3.hip): tile width for (C, k, max dilation) in form `wide`, 0 = not covered; launch
int ampb_tile(int k, int C, int max_dil, int wide);
hipError_t launch_ampb(int k, const AmpbArgs& a, int wide, hipStream_t stream);

// conv_post: y[b,0,t] = tanh( bias + sum_i sum_j w[i][j] * act_in(x[b,i,t+j-pad]) )   (Cout == 1)
hipError_t launch_conv_post(const float* x, const float* w_dev /*[Cin*K]*/, const float* bias_dev /*[1] or null*/,
                            float* y, int B, int Cin, int T, int K, float slope_in, int apply_tanh,
                            const int* lens, int len_mul, hipStream_t stream);

// Activation1d (anti-aliased Snake); a_dev = alpha (already exp'ed if logscale), invb_dev = 1/(beta+1e-9)
hipError_t launch_act1d(const float* x, float* y, int B, int C, int T, const float* a_dev, const float* invb_dev,
                        const float* filt_up12, const float* filt_dn12, const int* lens, int len_mul,
                        hipStream_t stream, int rev = 0 /* 1: descending workgroup order, ConvArgs::rev */);

// y[b,c,t] += cond[b,c]   (HiFiGAN_vits `x + self.cond(g)` with g of length 1)
hipError_t launch_add_channel_bias(float* y, const float* cb, int B, int C, int T, hipStream_t stream);
// y = ((y + p[0]) + p[1] ...) / div over `count` floats (16-byte aligned): the MRF mean of resblocks that stored their results separately
constexpr int AMP_MRF_MAX_PARTS = 3;
struct MrfSumArgs {
    float* y;
    const float* p[AMP_MRF_MAX_PARTS];
    int n;
    float div;
    size_t count;
};
hipError_t launch_mrf_sum(const MrfSumArgs& a, hipStream_t stream);

// VITS posterior-encoder / flow element-wise kernels (small_kernels.hip)
hipError_t launch_wn_gate(const float* a, const float* cond, long long cond_bs, float* out, int B, int H, int T,
                          hipStream_t stream);
hipError_t launch_wn_accumulate(float* x, float* out, const float* rs, const int* lens, int B, int H, int T, int last,
                                int first, hipStream_t stream);
hipError_t launch_mask(float* x, const int* lens, int B, int C, int T, hipStream_t stream);
hipError_t launch_coupling(float* x, const float* m, const int* lens, int B, int h, int T, int reverse,
                           hipStream_t stream);
hipError_t launch_flip_channels(const float* x, float* y, int B, int C, int T, hipStream_t stream);
hipError_t launch_posterior_sample(const float* stats, const float* eps, const int* lens, float* z, int B, int C, int T,
                                   hipStream_t stream);

hipError_t launch_fir_up(const float* x, float* y, int rows, int T, const float* taps_host, int K, int ratio, int pad,
                         int pad_left, hipStream_t stream);
hipError_t launch_fir_filter(const float* x, float* y, int rows, int T, int Tout, const float* taps_host, int K,
                             int stride, int pad_left, int mode, hipStream_t stream);
hipError_t launch_snake(const float* x, float* y, int B, int C, int T, const float* alpha, const float* beta, int logscale,
                        hipStream_t stream);
hipError_t launch_pcm16(const float* x, short* y, int B, int L, long long x_stride, long long y_stride, const int* lens,
                        hipStream_t stream);
hipError_t launch_apnet_polar(const float* logamp, const float* R, const float* I, size_t n, float* pha, float* rea,
                              float* imag, hipStream_t stream);

// mel front end (mel.hip)
hipError_t launch_mel(const amp_mel_desc& d, const float* wav, const int* lens, int B, int L, int F, const float* window,
                      const float* melbasis, float* mel, float* mag, float* re, float* im, hipStream_t stream);

void set_error(const char* fmt, ...);

// Kernels that ask for more than 64 KB of dynamic LDS need hipFuncAttributeMaxDynamicSharedMemorySize, and the attribute belongs to
// ONE device's copy of the function: set it once per (kernel, device), under a lock -- two handles may launch from two host threads
// (the per-launcher copies of this bookkeeping had no lock and mapped devices >= 64 onto bit 0 without setting anything).
// The kernel is a template ARGUMENT: one mask per kernel instantiation.  Devices beyond 63 set the attribute on every launch.
template <auto Kernel>
inline hipError_t ensure_dynamic_lds(size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    static std::atomic<unsigned long long> done{0};
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
    if (dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    std::lock_guard<std::mutex> lock(mu);
    if (dev < 64 && ((done.load(std::memory_order_acquire) >> dev) & 1ull)) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return e;
    if (dev < 64) done.fetch_or(1ull << dev, std::memory_order_release);
    return hipSuccess;
}

// Which kernels a stretch of launch_* calls issued, as rocprofv3 prints them ("pair_strip_kernel<11, 2, 2, 4, 320, 2, 4, 1>"):
// while the calling thread points tl_kernel_log at a string (a profiled amp_gen_forward does, per resblock), every launch
// appends its kernel's name once (" | "-joined) -- bench.py reports the variant the policy actually picked, not a literal.
extern thread_local std::string* tl_kernel_log;
// AMP_LAUNCH_MANIFEST=<file> (read once per process): every launch appends one line
//     <kernel name with its template arguments>\t<workgroups>\t<algorithmic GFLOP>\t<algorithmic MB>\t<what it computes>
// -- the launcher's OWN statement of the work a launch does, which tools/roofline_table.py joins with rocprofv3's per-dispatch durations
// by (name, workgroups) instead of guessing shapes from template arguments.  Off: one predictable branch per launch.
bool manifest_on();
void manifest_add(const char* name, unsigned long long workgroups, double gflop, double mb, const char* what);
extern thread_local char tl_last_kernel[160];
template <typename... A>
inline void note_kernel(const char* base, A... args) {
    if (!tl_kernel_log && !manifest_on()) return;
    char* buf = tl_last_kernel;
    constexpr int cap = (int)sizeof(tl_last_kernel);
    const int v[] = {static_cast<int>(args)...};
    int n = snprintf(buf, cap, "%s<", base);
    for (size_t i = 0; i < sizeof...(args) && n < cap - 16; ++i)
        n += snprintf(buf + n, cap - n, i ? ", %d" : "%d", v[i]);
    snprintf(buf + n, cap - n, ">");
    if (!tl_kernel_log) return;
    if (tl_kernel_log->find(buf) != std::string::npos) return;
    if (!tl_kernel_log->empty()) *tl_kernel_log += " | ";
    *tl_kernel_log += buf;
}
// after note_kernel(): the launch's grid and algorithmic work (manifest only)
inline void note_work(unsigned long long workgroups, double gflop, double mb, const char* fmt, ...) __attribute__((format(printf, 4, 5)));
inline void note_work(unsigned long long workgroups, double gflop, double mb, const char* fmt, ...) {
    if (!manifest_on()) return;
    char what[160];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(what, sizeof(what), fmt, ap);
    va_end(ap);
    manifest_add(tl_last_kernel, workgroups, gflop, mb, what);
}
// a Conv1d / ConvTranspose1d launch: GEMM M = Cout * up rows, K = Cin * KT, N = B * Tq columns; bytes = x read once + y written once
// (+ the residual / running sum read)
inline void note_conv_work(const ConvArgs& a, int KT, dim3 grid) {
    if (!manifest_on()) return;
    const double gf = 2.0 * a.M * a.Cin * KT * (double)a.Tq * a.B / 1e9;
    const double mb = 4.0 * a.B * ((double)a.Cin * a.Tin + (double)a.Cout * a.Tout * (1 + (a.res ? 1 : 0) + (a.mode ? 1 : 0))) / 1e6;
    note_work((unsigned long long)grid.x * grid.y, gf, mb, "%s %d->%d k=%d d=%d T=%d->%d B=%d%s%s", a.up > 1 ? "ConvT" : "conv", a.Cin, a.Cout,
              a.up > 1 ? KT * a.up : KT, a.dstep, a.Tin, a.Tout, a.B, a.res ? " +res" : "", a.mode ? " +sum" : "");
}

// The f16x3 kernels stage fp32 activations as hi + lo f16 pairs after an exact x16: anything beyond |x| = 4094 (or
// non-finite) cannot be represented.  They OR 1 into this per-device word when that happens (range_guard.hip).
unsigned* range_flag_for_current_device();
// (the kernels keep the running maximum of |staged value|: one v_max_f32 per element; a NaN operand is not flagged --
// it propagates to the output exactly as it does through the fp32 reference)

// v = hi + lo with hi = f16(v), lo = f16(v - hi): the split-f16 operand form of the f16x3 kernels.
// The empty asm makes `v` opaque: under HIP's default -ffp-contract=fast hipcc otherwise folds the
// multiply that produced v into the conversions (v_fma_mix*: f16(x*k) rounded ONCE from the exact
// product) for `lo` but not for the stored `hi` (v_cvt_pk_f16_f32 of the fp32-rounded product): in the
// rare double-rounding cases the two disagree by one f16 ulp and the pair (hi, lo) is off by 2^-11.
__device__ __forceinline__ void split_f16(float v, _Float16& h, _Float16& l) {
    asm("" : "+v"(v));
    h = (_Float16)v;
    l = (_Float16)(v - (float)h);
}

// ---- the same split on FOUR operands with gfx950's packed conversions ------------------------------------------------
// Staging (and the fused pair's seam) is VALU work on a chip that runs these kernels at its power limit: per element the
// scalar form costs select + compare + select + multiply + 3 conversions + subtract + half a pack (9.5 instructions),
// the packed form 2 multiplies + max for the leaky ReLU and v_cvt_pk_f16_f32 / v_pk_add_f32 for the split (about 6).
// Every value is the one split_f16 / `v * (v > 0 ? kpos : kneg)` produce:
//   max(kpos * x, kneg * x) == x * (x > 0 ? kpos : kneg) for kneg <= kpos (slopes <= 1; the host refuses larger ones),
//   hi = RNE f16(v), v - hi is exact in fp32, lo = RNE f16(v - hi).
typedef float amp_f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 amp_f16x2 __attribute__((ext_vector_type(2)));

// lo = f16(v - hi) is ONE v_fma_mixlo_f16 / v_fma_mixhi_f16 per element (fp32 v, constant 1.0, the f16 half of hi negated: the
// difference is exact in fp32, so the single rounding of the mix form is the rounding of the conversion): 3 VALU instructions
// per operand pair instead of 5 (v_cvt_pk_f16_f32, 2 x v_cvt_f32_f16, v_pk_add_f32, v_cvt_pk_f16_f32).  hipcc does not select
// the mix form for a subtraction (it vectorises it); tests/experiments/split_mix.hip compares both forms on the hardware over
// 2^25 operand pairs covering every sign / exponent / upper-mantissa pattern.
// CAUTION (round 5, profiles/r5_b_fir_mfma.txt): send the results to LDS (as every caller does: the store interlocks), never straight into an
// MFMA -- on gfx950 an MFMA that reads a VGPR needs two wait states after the VALU instruction that wrote it, hipcc inserts them only for
// instructions it can see, and these are inline asm: fragments fed directly came out with a stale half in ~3 % of the waves.  An MFMA consumer
// needs an `s_nop 1` at the end of the asm (profiles/negative_kernels/act1d_mfma.h: act_split4).
__device__ __forceinline__ void split4_f16(amp_f32x2 v01, amp_f32x2 v23, uint2& h, uint2& l) {
    asm("" : "+v"(v01));      // opaque, as in split_f16: no folding of the producing multiply into ONE of the conversions
    asm("" : "+v"(v23));
    const amp_f16x2 h01 = __builtin_convertvector(v01, amp_f16x2), h23 = __builtin_convertvector(v23, amp_f16x2);
    h.x = __builtin_bit_cast(unsigned, h01); h.y = __builtin_bit_cast(unsigned, h23);
#ifndef AMP_SPLIT_PACKED
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l.x) : "v"(v01.x), "v"(h.x));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l.x) : "v"(v01.y), "v"(h.x));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(l.y) : "v"(v23.x), "v"(h.y));
    asm("v_fma_mixhi_f16 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l.y) : "v"(v23.y), "v"(h.y));
#else
    const amp_f32x2 d01 = v01 - __builtin_convertvector(h01, amp_f32x2), d23 = v23 - __builtin_convertvector(h23, amp_f32x2);
    const amp_f16x2 l01 = __builtin_convertvector(d01, amp_f16x2), l23 = __builtin_convertvector(d23, amp_f16x2);
    l.x = __builtin_bit_cast(unsigned, l01); l.y = __builtin_bit_cast(unsigned, l23);
#endif
}

// leaky-ReLU-on-load + x16 of four staged values (x already zeroed where the conv pads) -> hi / lo planes; keeps the
// running maximum of |staged operand| for the range guard
__device__ __forceinline__ void stage4_f16(float x0, float x1, float x2, float x3, float kpos, float kneg, float& range_max,
                                           uint2& h, uint2& l) {
    const amp_f32x2 x01 = {x0, x1}, x23 = {x2, x3};
    const amp_f32x2 a01 = x01 * kpos, a23 = x23 * kpos, b01 = x01 * kneg, b23 = x23 * kneg;
    const amp_f32x2 v01 = {__builtin_fmaxf(a01.x, b01.x), __builtin_fmaxf(a01.y, b01.y)};
    const amp_f32x2 v23 = {__builtin_fmaxf(a23.x, b23.x), __builtin_fmaxf(a23.y, b23.y)};
    range_max = __builtin_fmaxf(range_max, __builtin_fmaxf(__builtin_fabsf(v01.x), __builtin_fabsf(v01.y)));
    range_max = __builtin_fmaxf(range_max, __builtin_fmaxf(__builtin_fabsf(v23.x), __builtin_fabsf(v23.y)));
    split4_f16(v01, v23, h, l);
}

// the fused pair's seam: four conv1 accumulators -> leaky ReLU -> conv2's zero padding (`qok`) -> x16 -> hi / lo planes.
// Same values as the scalar form (v = acc * isc; v = v > 0 ? v : v * slope; v = qok ? v * 16 : 0): max(v, v * slope) is that leaky
// ReLU for slope <= 1, and the x16 and the padding are folded into the un-scaling factor -- isc * 16 is a power of two, so
// acc * (isc * 16) == (acc * isc) * 16 and ((acc * isc) * 16) * slope == ((acc * isc) * slope) * 16 bit for bit, and a factor 0 gives
// the padding's zero (round 4: 4 packed multiplies + 4 max instead of 6 + 4 + 4 selects per four values; the seams are a fifth of
// the whole-resblock launches, all VALU).
__device__ __forceinline__ void seam4_f16(float c0, float c1, float c2, float c3, float isc, float slope, bool qok,
                                          float& range_max, uint2& h, uint2& l) {
    const float k = qok ? isc * 16.f : 0.f;
    const amp_f32x2 c01 = {c0, c1}, c23 = {c2, c3};
    const amp_f32x2 w01 = c01 * k, w23 = c23 * k;
    const amp_f32x2 n01 = w01 * slope, n23 = w23 * slope;
    const amp_f32x2 v01 = {__builtin_fmaxf(w01.x, n01.x), __builtin_fmaxf(w01.y, n01.y)};
    const amp_f32x2 v23 = {__builtin_fmaxf(w23.x, n23.x), __builtin_fmaxf(w23.y, n23.y)};
    range_max = __builtin_fmaxf(range_max, __builtin_fmaxf(__builtin_fabsf(v01.x), __builtin_fabsf(v01.y)));
    range_max = __builtin_fmaxf(range_max, __builtin_fmaxf(__builtin_fabsf(v23.x), __builtin_fabsf(v23.y)));
    split4_f16(v01, v23, h, l);
}

}  // namespace amp
