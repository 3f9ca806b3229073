// First run on MI355X in round 2: tests/test_gpu_vits_infer.py green (every op against oracle/vits_infer_oracle.py,
// SynthesizerTrn.infer against the reference's golden vectors: durations / path exact, waveform <= 1e-4).
//
// Frame-rate kernels of the text -> duration -> alignment front of VITS inference (SURVEY.md §8 f.4,
// SynthesizerTrn.infer models/tts/vits/vits.py:320-369).  T_text is 10^2, channels ~200: everything here is
// latency-bound; the kernels are written for clarity (one thread per output, coalesced along time), the dense
// convolutions between them run on conv_f16x3.hip.  Layout [B, C, T], fp32.
#include "amp_internal.h"

#include <atomic>
#include <mutex>
#include <type_traits>

namespace amp {

// GELU and the two places of a LayerNorm where hipcc is free to contract a multiply into a following add -- or not, depending on the
// code around them (dds_seam_kernel fused gelu's last product into its caller's `+ x`, layer_norm_c_kernel, where that add sits behind
// a run-time `if (post)`, did not; hipcc's __fmul_rn is a plain multiply and does not stop it).  Contraction is switched off inside
// these helpers and the fused operations are written out, so every kernel that normalises rounds alike and the fused forms can be
// tested bit for bit; the opaque move keeps a CALLER from contracting across the return value.
__device__ __forceinline__ float fp_opaque(float v) {
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ float gelu_erf(float v) {
#pragma clang fp contract(off)
    const float e = 1.0f + fp_opaque(erff(v * 0.70710678118654752f));
    return fp_opaque((0.5f * v) * e);
}
__device__ __forceinline__ float ln_sq_acc(float part, float d, bool on) {
    const float dd = on ? d : 0.f;
    return fmaf(dd, dd, part);
}
__device__ __forceinline__ float ln_affine(float v, float mu, float rstd, float gamma, float beta) {
#pragma clang fp contract(off)
    const float t = fp_opaque((v - mu) * rstd);
    return fp_opaque(fmaf(t, gamma, beta));
}

// LayerNorm over the CHANNEL axis (modules/base/base_module.py:20-23), optionally of x + res (Encoder's
// norm(x + y), modules/transformer/attentions.py:69,73) and optionally followed by GELU (DDSConv,
// modules/flow/modules.py:64-68) and by "+ post" (DDSConv's x = x + y, :70).
// A workgroup = 32 time columns x 8 channel groups: thread (tx, g) holds channels g, g + 8, ... of column tx in registers (up to
// 32 of them = C <= 256; more are re-read), rows are read as 128-B segments, and the two reductions over C (mean, then the biased
// variance of the centred values -- two passes, like the op it replaces) go through LDS in a fixed order.  Round 3: the first
// version ran ONE thread per (b, t) with three serial loops over C -- 120 us per call at B = 16, C = 192, T = 150 (16 workgroups
// of 150 live threads, 576 dependent loads each), a third of VITS text -> wave.
//
// Round 4, two fusions that remove launches around it (the text side of VITS is ~250 launches of 5-15 us):
//   lens  the valid lengths: output columns t >= lens[b] are written as ZERO (a select, whatever x / res hold there), which is
//         what lets the Encoder drop the `* x_mask` launches around its FFN (attentions.py:392-400): the convs take the lengths
//         themselves (amp_conv_forward_ragged) and whatever they leave beyond an utterance's end stops here;
//   DWK   the depthwise dilated conv of DDSConv (modules/flow/modules.py:63-64: norm_1(conv_sep(x * mask))) evaluated on load:
//         value(c, t) = bias[c] + sum_j w[c, j] * xm[c, t - pad + j * dil] in the tap order of dwconv_kernel (same bits).
constexpr int LN_TT = 32, LN_G = 8, LN_DW_MAXC = 1024;   // NC = channels per thread held in registers: 24 (C <= 192) or 32
#define NTAP_(k) ((k) > 0 ? (k) : 1)
template <int DWK, int NC>
__global__ __launch_bounds__(256) void layer_norm_c_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ post,
                                                           const int* __restrict__ lens, const float* __restrict__ dw_w,
                                                           const float* __restrict__ dw_b, int dil,
                                                           float* __restrict__ y, int C, int T, float eps, int gelu) {
    __shared__ float red[LN_G][LN_TT + 1];
    __shared__ float dww[DWK > 0 ? LN_DW_MAXC * (DWK + 1) : 1];   // depthwise taps and bias of every channel: [C][DWK] | [C]
    const int tx = threadIdx.x & (LN_TT - 1), g = threadIdx.x / LN_TT;
    const int t = blockIdx.x * LN_TT + tx;
    const int b = blockIdx.y;
    const bool ok = t < T;
    const int len = lens ? lens[b] : T;
    const size_t ibase = (size_t)b * C * T;
    const size_t base = ibase + (ok ? t : 0);
    if (blockIdx.x * LN_TT >= len) {                     // block-uniform: a tile beyond the utterance's end is zeros
        if (ok)
            for (int c = g; c < C; c += LN_G) y[base + (size_t)c * T] = 0.f;
        return;
    }
    const float* xb = x + base;
    const float* rb = res ? res + base : nullptr;
    if (DWK > 0) {                                       // (kept out of the registers: 32 channels x 3 taps of weights next to as many
        for (int i = threadIdx.x; i < C * NTAP_(DWK); i += 256) dww[i] = dw_w[i];          //  input values is the whole VGPR file)
        for (int i = threadIdx.x; i < C; i += 256) dww[C * NTAP_(DWK) + i] = dw_b ? dw_b[i] : 0.f;
        __syncthreads();
    }
    // tap columns of the depthwise prologue (clamped for the load, selected afterwards)
    constexpr int NTAP = DWK > 0 ? DWK : 1;
    int tu[NTAP];
    bool tv[NTAP];
#pragma unroll
    for (int j = 0; j < NTAP; ++j) {
        const int u = DWK > 0 ? t - (DWK * dil - dil) / 2 + j * dil : t;
        tv[j] = u >= 0 && u < (DWK > 0 ? len : T);
        tu[j] = u < 0 ? 0 : (u > T - 1 ? T - 1 : u);
    }
    auto value = [&](int c) {                            // channels beyond the register file (C > 256): the plain form
        if (DWK > 0) {
            float acc = dww[C * NTAP + c];
            for (int j = 0; j < NTAP; ++j) acc = fmaf(dww[c * NTAP + j], tv[j] ? x[ibase + (size_t)c * T + tu[j]] : 0.f, acc);
            return acc + (rb ? rb[(size_t)c * T] : 0.f);
        }
        return xb[(size_t)c * T] + (rb ? rb[(size_t)c * T] : 0.f);
    };
    auto reduce = [&](float part) {                      // sum over the 8 channel groups of column tx, same order in every thread
        red[g][tx] = part;
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_G; ++k) s += red[k][tx];
        __syncthreads();
        return s;
    };
    float v[NC];
    float part = 0.f;
    {
        // Round 4: every load of the thread's 24-32 channels UNCONDITIONAL and issued before the first use (clamped channel index, the
        // select applied afterwards).  The round-3 form loaded inside `(ok && c < C) ? value(c) : 0.f`: hipcc sinks such a load into the
        // branch and waits for it with vmcnt(0) -- 24 dependent HBM round trips per thread, 19 us per call for 1.2 MB (now 13).
        float xr[NC][NTAP], rr[NC];
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = g + LN_G * i;
            const int cc = c < C ? c : C - 1;
#pragma unroll
            for (int j = 0; j < NTAP; ++j) xr[i][j] = x[ibase + (size_t)cc * T + (DWK > 0 ? tu[j] : (ok ? t : 0))];
        }
        if (rb) {                                        // block-uniform
#pragma unroll
            for (int i = 0; i < NC; ++i) {
                const int c = g + LN_G * i;
                rr[i] = rb[(size_t)(c < C ? c : C - 1) * T];
            }
        } else {
#pragma unroll
            for (int i = 0; i < NC; ++i) rr[i] = 0.f;
        }
        asm volatile("" ::: "memory");                   // keep the requests together (mel.hip: the same fence)
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = g + LN_G * i;
            float xv;
            if (DWK > 0) {
                const int cc = c < C ? c : C - 1;
                xv = dww[C * NTAP + cc];
#pragma unroll
                for (int j = 0; j < NTAP; ++j) xv = fmaf(dww[cc * NTAP + j], tv[j] ? xr[i][j] : 0.f, xv);
            } else {
                xv = xr[i][0];
            }
            v[i] = (ok && c < C) ? xv + rr[i] : 0.f;
            part += v[i];
        }
    }
    for (int c = g + LN_G * NC; c < C; c += LN_G) part += ok ? value(c) : 0.f;
    // what the output pass needs -- gamma, beta and `post` of the thread's channels -- requested here, unconditionally, so that it
    // arrives behind the two reductions (inside the guarded output pass each of the 72 loads was waited for alone: the GELU + post
    // form of DDSConv ran 13.8 us against 8.7 us for the plain one)
    float gr[NC], br[NC], po[NC];
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = g + LN_G * i;
        const int cc = c < C ? c : C - 1;
        gr[i] = gamma[cc];
        br[i] = beta[cc];
    }
    if (post) {                                          // block-uniform
#pragma unroll
        for (int i = 0; i < NC; ++i) {
            const int c = g + LN_G * i;
            po[i] = post[base + (size_t)(c < C ? c : C - 1) * T];
        }
    } else {
#pragma unroll
        for (int i = 0; i < NC; ++i) po[i] = 0.f;
    }
    const float mu = reduce(part) / (float)C;
    part = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) part = ln_sq_acc(part, v[i] - mu, g + LN_G * i < C);
    for (int c = g + LN_G * NC; c < C; c += LN_G) part = ln_sq_acc(part, (ok ? value(c) : 0.f) - mu, true);
    const float rstd = 1.0f / sqrtf(reduce(part) / (float)C + eps);
    if (!ok) return;
    float* yb = y + base;
    const bool live = t < len;
    auto emit = [&](int c, float xv) {
        float o = ln_affine(xv, mu, rstd, gamma[c], beta[c]);
        if (gelu) o = gelu_erf(o);
        if (post) o += post[base + (size_t)c * T];
        yb[(size_t)c * T] = live ? o : 0.f;
    };
#pragma unroll
    for (int i = 0; i < NC; ++i) {
        const int c = g + LN_G * i;
        float o = ln_affine(v[i], mu, rstd, gr[i], br[i]);
        if (gelu) o = gelu_erf(o);
        if (post) o += po[i];
        if (c < C) yb[(size_t)c * T] = live ? o : 0.f;
    }
    for (int c = g + LN_G * NC; c < C; c += LN_G) emit(c, value(c));
}

// The seam between two DDSConv layers (modules/flow/modules.py:63-70) in ONE launch instead of two LayerNorm launches:
//     x_new = x + gelu(LN_2(y))                        (layer i:   norms_2[i] on the 1 x 1 conv's output, residual)
//     z     = gelu(LN_1'(dwconv'(x_new * mask)))        (layer i+1: convs_sep[i+1] -> norms_1[i+1])
// The depthwise conv needs x_new at t - d, t, t + d: a workgroup (32 columns x 8 channel groups, 24 channels per thread: C <= 192)
// evaluates LN_2 at those three column sets -- the two shifted ones are recomputed, not exchanged: LN is ~1 us of arithmetic, a
// launch 14 -- with all 144 loads of a thread in flight at once, the three pairs of reductions batched through LDS, then forms the
// depthwise taps and LN_1' exactly as layer_norm_c_kernel<3> does.  Per element every operation and every summation order is that of
// the two launches it replaces: same bits (tests/test_gpu_vits_text_kernels.py::test_dds_seam_bitwise).  z is zero beyond the lengths
// (as amp_dwconv_layer_norm_c), x_new is stored unmasked (as amp_layer_norm_c).
constexpr int DS_NC = 24;
__global__ __launch_bounds__(256) void dds_seam_kernel(const float* __restrict__ y, const float* __restrict__ x,
                                                       const float* __restrict__ g2, const float* __restrict__ b2,
                                                       const float* __restrict__ dw_w, const float* __restrict__ dw_b,
                                                       const float* __restrict__ g1, const float* __restrict__ b1,
                                                       const int* __restrict__ lens, float* __restrict__ xo, float* __restrict__ zo,
                                                       int C, int T, int dil, float eps2, float eps1) {
    __shared__ float red[3][LN_G][LN_TT + 1];
    __shared__ float par[8 * LN_G * DS_NC];              // per channel: g2 | b2 | g1 | b1 | w0 | w1 | w2 | bias
    const int tx = threadIdx.x & (LN_TT - 1), g = threadIdx.x / LN_TT;
    const int t = blockIdx.x * LN_TT + tx;
    const int b = blockIdx.y;
    const bool ok = t < T;
    const int len = lens ? lens[b] : T;
    const size_t ibase = (size_t)b * C * T;
    constexpr int CP = LN_G * DS_NC;                     // 192
    for (int i = threadIdx.x; i < CP; i += 256) {
        const int c = i < C ? i : C - 1;
        par[i] = g2[c];
        par[CP + i] = b2[c];
        par[2 * CP + i] = g1[c];
        par[3 * CP + i] = b1[c];
        par[4 * CP + i] = dw_w[c * 3];
        par[5 * CP + i] = dw_w[c * 3 + 1];
        par[6 * CP + i] = dw_w[c * 3 + 2];
        par[7 * CP + i] = dw_b ? dw_b[c] : 0.f;
    }
    int tu[3];
    bool tv[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int u = t - dil + j * dil;
        tv[j] = u >= 0 && u < len;
        tu[j] = u < 0 ? 0 : (u > T - 1 ? T - 1 : u);
    }
    float yv[3][DS_NC], xv[3][DS_NC];
#pragma unroll
    for (int i = 0; i < DS_NC; ++i) {
        const int c = g + LN_G * i;
        const size_t row = ibase + (size_t)(c < C ? c : C - 1) * T;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            yv[j][i] = y[row + tu[j]];
            xv[j][i] = x[row + tu[j]];
        }
    }
    asm volatile("" ::: "memory");
    auto reduce3 = [&](float (&p)[3]) {                  // the three column sets at once; per set the order of layer_norm_c_kernel
#pragma unroll
        for (int j = 0; j < 3; ++j) red[j][g][tx] = p[j];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s_ = 0.f;
#pragma unroll
            for (int k = 0; k < LN_G; ++k) s_ += red[j][k][tx];
            p[j] = s_;
        }
        __syncthreads();
    };
    // ---- LN_2 + gelu + residual at the three column sets (columns beyond T are clamped duplicates: computed, never used) ----
    float part[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        part[j] = 0.f;
#pragma unroll
        for (int i = 0; i < DS_NC; ++i) {
            yv[j][i] = (g + LN_G * i < C) ? yv[j][i] : 0.f;
            part[j] += yv[j][i];
        }
    }
    reduce3(part);                                       // (also orders the parameter table before its first use)
    float mu[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        mu[j] = part[j] / (float)C;
        part[j] = 0.f;
#pragma unroll
        for (int i = 0; i < DS_NC; ++i) part[j] = ln_sq_acc(part[j], yv[j][i] - mu[j], g + LN_G * i < C);
    }
    reduce3(part);
    float xn[3][DS_NC];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const float rstd = 1.0f / sqrtf(part[j] / (float)C + eps2);
#pragma unroll
        for (int i = 0; i < DS_NC; ++i) {
            const int c = g + LN_G * i;
            float o = ln_affine(yv[j][i], mu[j], rstd, par[c], par[CP + c]);
            o = gelu_erf(o);
            o += xv[j][i];
            xn[j][i] = o;
        }
    }
    if (ok) {
#pragma unroll
        for (int i = 0; i < DS_NC; ++i) {
            const int c = g + LN_G * i;
            if (c < C) xo[ibase + (size_t)c * T + t] = xn[1][i];
        }
    }
    // ---- depthwise taps on x_new * mask, LN_1', gelu ----
    float v[DS_NC];
    float p1 = 0.f;
#pragma unroll
    for (int i = 0; i < DS_NC; ++i) {
        const int c = g + LN_G * i;
        float a = par[7 * CP + c];
#pragma unroll
        for (int j = 0; j < 3; ++j) a = fmaf(par[(4 + j) * CP + c], tv[j] ? xn[j][i] : 0.f, a);
        v[i] = (ok && c < C) ? a : 0.f;
        p1 += v[i];
    }
    float pr[3] = {p1, 0.f, 0.f};
    reduce3(pr);
    const float mu1 = pr[0] / (float)C;
    p1 = 0.f;
#pragma unroll
    for (int i = 0; i < DS_NC; ++i) p1 = ln_sq_acc(p1, v[i] - mu1, g + LN_G * i < C);
    pr[0] = p1; pr[1] = 0.f; pr[2] = 0.f;
    reduce3(pr);
    const float rstd1 = 1.0f / sqrtf(pr[0] / (float)C + eps1);
    if (!ok) return;
    const bool live = t < len;
#pragma unroll
    for (int i = 0; i < DS_NC; ++i) {
        const int c = g + LN_G * i;
        float o = ln_affine(v[i], mu1, rstd1, par[2 * CP + c], par[3 * CP + c]);
        o = gelu_erf(o);
        if (c < C) zo[ibase + (size_t)c * T + t] = live ? o : 0.f;
    }
}

// Self-attention with windowed relative-position embeddings, heads_share = True
// (MultiHeadAttention.attention, modules/transformer/attentions.py:232-272).  One 64-lane workgroup per (b, h, i):
//   s[j]   = (q_i . k_j + [|j - i| <= w] q_i . Ek[j - i + w]) / sqrt(dk);   -1e4 where query or key is padding
//   p      = softmax_j(s)
//   out[d] = sum_j p[j] v[d, j] + sum_{|j - i| <= w} p[j] Ev[j - i + w][d]
// q, k, v, out: [B, H*dk, T]; Ek, Ev: [2w+1, dk].  LDS: q_i (dk floats) + p (T floats).
// Round 4: the same arithmetic with the keys and values of a (b, h) pair staged in LDS ONCE per block of 16 queries instead of being
// re-read from L2 by every query (3200 workgroups x 77 KB = 245 MB of L2 reads per call at B = 16, T = 100: 49 us; now 224 workgroups
// x 77 KB).  A workgroup = 256 threads = (b, h, 16 queries):
//   pass 1  scores: wave w owns queries 4w .. 4w+3, lane l the keys l and l + 64 of the staged 128-key tile; the dot products run
//           over d in ascending order as fmaf chains (the bits of the one-query kernel below), the relative-key term comes from
//           qe[i][r] = q_i . Ek[r] (same chain), the masking is the same select;
//   pass 2  softmax: one wave per query row, the lane-strided partial sums and xor butterflies of the one-query kernel;
//   pass 3  values: thread (d, half) accumulates 8 queries over ascending j (the V tile replaces the K tile in LDS, row stride 129
//           words so that lanes = consecutive d hit distinct banks), adds the relative-value term and scales.
// Every sum is formed in the order of rel_attention_row_kernel, so the two kernels give identical bits
// (tests/test_gpu_vits_infer.py::test_rel_attention_tiled_bitwise); the row kernel remains for shapes outside this one's LDS
// budget (T > 1024, dk > 128 or not a multiple of 4, window > 7).
// q / k / v may be slices of one [B, 3*H*dk, T] tensor (the merged q|k|v projection): `bs` is their batch stride in elements.
constexpr int RA_QB = 16, RA_KT = 128, RA_KS = RA_KT + 1;
// ONE: T <= 128 -- keys AND values are staged together before pass 1 (two LDS buffers, their global loads in flight at once);
// otherwise the value tiles replace the key tiles in one buffer.  Every staging load is unconditional from a clamped address and
// a batch of them is issued before the first is used: written as `(j < T) ? src[..] : 0` hipcc sinks each load into its branch
// and waits for it alone -- 96 dependent L2 round trips per thread, 53 us per call (slower than the kernel it replaces).
template <bool ONE>
__global__ __launch_bounds__(256) void rel_attention_tile_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                 const float* __restrict__ v, const float* __restrict__ ek,
                                                                 const float* __restrict__ ev, const int* __restrict__ lens,
                                                                 float* __restrict__ out, long long bs, int H, int dk, int T,
                                                                 int window, int Tp) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* qs = sm;                          // [RA_QB][dk]   query / sqrt(dk)
    float* qe = qs + RA_QB * dk;             // [RA_QB][16]   q_i . Ek[r]
    float* sums = qe + RA_QB * 16;           // [RA_QB]
    float* p = sums + RA_QB;                 // [RA_QB][Tp]   scores, then exp(score - max)
    float* eks = p + RA_QB * Tp;             // [2w+1][dk]    relative-key embeddings
    float* evs = eks + 16 * dk;              // [2w+1][dk]    relative-value embeddings
    float* kv = evs + 16 * dk;               // [dk][RA_KS]   key tile, then value tile
    float* vb = ONE ? kv + dk * RA_KS : kv;  // [dk][RA_KS]   value tile (ONE: its own buffer)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i0 = blockIdx.x * RA_QB, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? lens[b] : T;
    const size_t ibase = (size_t)b * (size_t)bs + (size_t)h * dk * T;          // q / k / v
    const size_t obase = ((size_t)b * H + h) * (size_t)dk * T;                 // out [B, H*dk, T]
    const float scale = 1.0f / sqrtf((float)dk);
    const int nrel = 2 * window + 1;
    // Staging: everything a phase needs is requested in ONE batch of unconditional loads before the first value is used.  Rows /
    // elements beyond the end are clamped on the load AND on the store: such a thread re-writes the last row with the value that
    // row holds anyway, which keeps the batch free of branches (a guarded store drags its load into the branch with it).
    constexpr int QU = 8, EU = 8, SU = 48;                                     // q rows (x16), Ek / Ev elements (x256), k / v rows (x2)
    const int qi = tid & (RA_QB - 1);                                          // consecutive threads = consecutive queries of a row
    const int qic = i0 + qi < T ? i0 + qi : T - 1;
    const int ne = nrel * dk;
    const int jj = tid & (RA_KT - 1);                                          // k / v: thread -> column jj, rows (tid >> 7) + 2n
    auto kv_load = [&](auto both, const float* s0, const float* s1, int j0, int d0, float (&t0)[SU], float (&t1)[SU]) {
        constexpr bool BOTH = decltype(both)::value;
        const size_t col = ibase + (size_t)(j0 + jj < T ? j0 + jj : T - 1);
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int d = d0 + 2 * u;
            const size_t a = col + (size_t)(d < dk ? d : dk - 1) * T;
            t0[u] = s0[a];
            if (BOTH) t1[u] = s1[a];
        }
    };
    auto kv_store = [&](auto both, float* b0, float* b1, int j0, int d0, const float (&t0)[SU], const float (&t1)[SU]) {
        constexpr bool BOTH = decltype(both)::value;
        const bool jok = j0 + jj < T;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
            const int d = d0 + 2 * u < dk ? d0 + 2 * u : dk - 1;
            b0[d * RA_KS + jj] = jok ? t0[u] : 0.f;
            if (BOTH) b1[d * RA_KS + jj] = jok ? t1[u] : 0.f;
        }
    };
    // rows of 128 keys (coalesced) of one matrix, or of both: SU rows per thread and batch
    auto stage = [&](auto both, const float* s0, float* b0, const float* s1, float* b1, int j0, int dfirst) {
        for (int d0 = dfirst + (tid >> 7); d0 < dk; d0 += 2 * SU) {
            float t0[SU], t1[SU];
            kv_load(both, s0, s1, j0, d0, t0, t1);
            asm volatile("" ::: "memory");
            kv_store(both, b0, b1, j0, d0, t0, t1);
        }
    };
    {
        // the queries, the two embedding tables and the first 2 * SU rows of the first key (and, ONE, value) tile: one round trip
        float tq[QU], tk[EU], tv[EU], t0[SU], t1[SU];
#pragma unroll
        for (int u = 0; u < QU; ++u) {
            const int d = (tid >> 4) + 16 * u;
            tq[u] = q[ibase + (size_t)(d < dk ? d : dk - 1) * T + qic];
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = tid + 256 * u < ne ? tid + 256 * u : ne - 1;
            tk[u] = ek[e];
            tv[u] = ev[e];
        }
        if (ONE) kv_load(std::true_type{}, k, v, 0, tid >> 7, t0, t1); else kv_load(std::false_type{}, k, nullptr, 0, tid >> 7, t0, t1);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < QU; ++u) {
            const int d = (tid >> 4) + 16 * u;
            qs[qi * dk + (d < dk ? d : dk - 1)] = (i0 + qi < T) ? tq[u] * scale : 0.f;
        }
#pragma unroll
        for (int u = 0; u < EU; ++u) {
            const int e = tid + 256 * u < ne ? tid + 256 * u : ne - 1;
            eks[e] = tk[u];
            evs[e] = tv[u];
        }
        if (ONE) kv_store(std::true_type{}, kv, vb, 0, tid >> 7, t0, t1); else kv_store(std::false_type{}, kv, nullptr, 0, tid >> 7, t0, t1);
    }
    // ---- pass 1 ----
    for (int j0 = 0; j0 < T; j0 += RA_KT) {
        if (j0 > 0) __syncthreads();                                           // the previous tile consumed
        // the rest of this tile's rows
        if (ONE) stage(std::true_type{}, k, kv, v, vb, j0, 2 * SU); else stage(std::false_type{}, k, kv, nullptr, nullptr, j0, j0 == 0 ? 2 * SU : 0);
        __syncthreads();
        if (j0 == 0) {
            if (tid < RA_QB * nrel) {                                          // qe[i][r] = q_i . Ek[r], d ascending
                const int i = tid / nrel, r = tid - i * nrel;
                float sr = 0.f;
                for (int d = 0; d < dk; d += 4) {
                    const float4 a = *reinterpret_cast<const float4*>(&qs[i * dk + d]);
                    const float4 e = *reinterpret_cast<const float4*>(&eks[r * dk + d]);
                    sr = fmaf(a.x, e.x, sr);
                    sr = fmaf(a.y, e.y, sr);
                    sr = fmaf(a.z, e.z, sr);
                    sr = fmaf(a.w, e.w, sr);
                }
                qe[i * 16 + r] = sr;
            }
        }
        float s[4][2];
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e][0] = s[e][1] = 0.f;
#pragma unroll 2
        for (int d = 0; d < dk; d += 4) {
            float4 qv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) qv[e] = *reinterpret_cast<const float4*>(&qs[(4 * wave + e) * dk + d]);
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const float k0 = kv[(d + dd) * RA_KS + lane], k1 = kv[(d + dd) * RA_KS + lane + 64];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float qd = dd == 0 ? qv[e].x : dd == 1 ? qv[e].y : dd == 2 ? qv[e].z : qv[e].w;
                    s[e][0] = fmaf(qd, k0, s[e][0]);
                    s[e][1] = fmaf(qd, k1, s[e][1]);
                }
            }
        }
        if (j0 == 0) __syncthreads();                                          // qe visible
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int il = 4 * wave + e, i = i0 + il;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = j0 + lane + 64 * u;
                if (j < T) {
                    float sc = s[e][u];
                    const int r = j - i + window;
                    if (r >= 0 && r <= 2 * window) sc += qe[il * 16 + r];
                    if (i >= len || j >= len) sc = -1.0e4f;
                    p[il * Tp + j] = sc;
                }
            }
        }
    }
    __syncthreads();
    // ---- pass 2 ----
    for (int e = 0; e < 4; ++e) {
        float* pr = p + (4 * wave + e) * Tp;
        float mx = -3.0e38f;
        for (int j = lane; j < T; j += 64) mx = fmaxf(mx, pr[j]);
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int j = lane; j < T; j += 64) {
            const float ex = expf(pr[j] - mx);
            pr[j] = ex;
            sum += ex;
        }
        for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
        if (lane == 0) sums[4 * wave + e] = sum;
    }
    // ---- pass 3 ----
    const int d = tid & 127, qh = tid >> 7;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int j0 = 0; j0 < T; j0 += RA_KT) {
        __syncthreads();                                                       // p / sums complete, the K tile consumed
        if (!ONE) {
            stage(std::false_type{}, v, vb, nullptr, nullptr, j0, 0);
            __syncthreads();
        }
        if (d < dk) {
            const int jn = T - j0 < RA_KT ? T - j0 : RA_KT;
#pragma unroll 4
            for (int jx = 0; jx < jn; ++jx) {
                const float vv = vb[d * RA_KS + jx];
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = fmaf(p[(8 * qh + e) * Tp + j0 + jx], vv, acc[e]);
            }
        }
    }
    if (d < dk) {
        // relative-value term: ar_e = sum over j = i - w .. i + w (inside [0, T)) in ascending j; the 8 chains run side by side
        float ar[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) ar[e] = 0.f;
        for (int r = 0; r < nrel; ++r) {
            const float evd = evs[r * dk + d];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int il = 8 * qh + e;
                const int j = i0 + il - window + r;
                const float pj = p[il * Tp + (j < 0 ? 0 : (j > T - 1 ? T - 1 : j))];
                ar[e] = (j >= 0 && j < T) ? fmaf(pj, evd, ar[e]) : ar[e];
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int il = 8 * qh + e, i = i0 + il;
            if (i < T) out[obase + (size_t)d * T + i] = (acc[e] + ar[e]) * (1.0f / sums[il]);
        }
    }
}


// The one-query form (rounds 2-3; now the fallback for shapes outside the tiled kernel): one 64-lane workgroup per (b, h, i).
__global__ __launch_bounds__(64) void rel_attention_row_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ ek,
                                                               const float* __restrict__ ev, const int* __restrict__ lens,
                                                               float* __restrict__ out, long long bs, int H, int dk, int T, int window) {
    extern __shared__ float sm[];
    float* qs = sm;          // [dk]
    float* p = sm + dk;      // [T]
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? lens[b] : T;
    const size_t base = (size_t)b * (size_t)bs + (size_t)h * dk * T;      // q / k / v (batch stride bs)
    const size_t obase = ((size_t)b * H + h) * dk * T;                     // out
    const float scale = 1.0f / sqrtf((float)dk);
    for (int d = lane; d < dk; d += 64) qs[d] = q[base + (size_t)d * T + i] * scale;   // query / sqrt(dk), :239
    __syncthreads();
    float mx = -3.0e38f;
    for (int j = lane; j < T; j += 64) {
        float s = 0.f;
        for (int d = 0; d < dk; ++d) s = fmaf(qs[d], k[base + (size_t)d * T + j], s);
        const int r = j - i + window;
        if (r >= 0 && r <= 2 * window) {
            float sr = 0.f;
            for (int d = 0; d < dk; ++d) sr = fmaf(qs[d], ek[(size_t)r * dk + d], sr);
            s += sr;
        }
        if (i >= len || j >= len) s = -1.0e4f;                                          // masked_fill(mask == 0, -1e4)
        p[j] = s;
        mx = fmaxf(mx, s);
    }
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int j = lane; j < T; j += 64) {
        const float e = expf(p[j] - mx);
        p[j] = e;
        sum += e;
    }
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int d = lane; d < dk; d += 64) {
        float acc = 0.f;
        const float* vd = v + base + (size_t)d * T;
        for (int j = 0; j < T; ++j) acc = fmaf(p[j], vd[j], acc);
        const int jlo = i - window < 0 ? 0 : i - window, jhi = i + window > T - 1 ? T - 1 : i + window;
        float ar = 0.f;
        for (int j = jlo; j <= jhi; ++j) ar = fmaf(p[j], ev[(size_t)(j - i + window) * dk + d], ar);
        out[obase + (size_t)d * T + i] = (acc + ar) * inv;
    }
}

// Depthwise dilated Conv1d (nn.Conv1d(C, C, K, groups=C, dilation=d, padding=(K*d - d)/2), DDSConv.convs_sep,
// modules/flow/modules.py:46-56) applied to x * mask (:63): zero outside [0, len).
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                     const float* __restrict__ bias, const int* __restrict__ lens,
                                                     float* __restrict__ y, int C, int T, int K, int dil) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int bc = blockIdx.y;
    if (t >= T) return;
    const int c = bc % C;
    const int len = lens ? lens[bc / C] : T;
    const float* xr = x + (size_t)bc * T;
    const int pad = (K * dil - dil) / 2;
    float acc = bias ? bias[c] : 0.f;
    for (int j = 0; j < K; ++j) {
        const int u = t - pad + j * dil;
        if (u >= 0 && u < len) acc = fmaf(w[c * K + j], xr[u], acc);
    }
    y[(size_t)bc * T + t] = acc;
}

// Piecewise rational-quadratic spline with linear tails on ONE channel of z [B, 2, T] (ConvFlow.forward,
// modules/flow/modules.py:424-458; transforms.py:56-215), parameters h [B, 3K-1, T] (widths | heights | derivatives),
// then `* mask` on both channels.  The channel Flips around a ConvFlow (:315-321) are folded in: flip_in swaps the
// two channels on load, flip_out on store (x0 = the conditioning channel, x1 = the transformed one).
constexpr int SPL_MAXK = 16;
// the transform of one element: `hv(i)` = parameter row i of this column ALREADY scaled and masked as the reference does
// (widths / heights: h * filter_channels^-0.5 * mask, derivatives: h * mask)
template <class HV>
__device__ __forceinline__ float spline_eval(float x1, HV hv, int K, float tail, int inverse) {
    float out = x1;
    if (x1 >= -tail && x1 <= tail) {
        const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
        float xk[SPL_MAXK + 1], yk[SPL_MAXK + 1], dd[SPL_MAXK + 1];
        // softmax -> bin fractions -> knots in [-tail, tail] with exact end points
        for (int pass = 0; pass < 2; ++pass) {
            float* kn = pass == 0 ? xk : yk;
            const float lo = pass == 0 ? min_w : min_h;
            float mxv = -3.0e38f;
            for (int i = 0; i < K; ++i) mxv = fmaxf(mxv, hv(pass * K + i));
            float se = 0.f;
            for (int i = 0; i < K; ++i) se += expf(hv(pass * K + i) - mxv);
            float cum = 0.f;
            kn[0] = -tail;
            for (int i = 0; i < K; ++i) {
                const float frac = lo + (1.f - lo * K) * (expf(hv(pass * K + i) - mxv) / se);
                cum += frac;
                kn[i + 1] = 2.f * tail * cum - tail;
            }
            kn[K] = tail;
        }
        dd[0] = 1.f;                                     // min_d + softplus(log(exp(1 - min_d) - 1)) = 1 at both ends
        dd[K] = 1.f;
        for (int i = 1; i < K; ++i) {
            const float u = hv(2 * K + i - 1);
            dd[i] = min_d + (u > 20.f ? u : log1pf(expf(u)));   // F.softplus (threshold 20)
        }
        const float* kn = inverse ? yk : xk;
        int bin = 0;
        for (int i = 1; i < K; ++i) bin += (x1 >= kn[i]) ? 1 : 0;   // searchsorted over the interior knots (last += eps)
        const float xa = xk[bin], wb = xk[bin + 1] - xk[bin];
        const float ya = yk[bin], hbin = yk[bin + 1] - yk[bin];
        const float d0 = dd[bin], d1 = dd[bin + 1];
        const float sl = hbin / wb;
        if (inverse) {
            const float dy = x1 - ya;
            const float e = d0 + d1 - 2.f * sl;
            const float qa = dy * e + hbin * (sl - d0);
            const float qb = hbin * d0 - dy * e;
            const float qc = -sl * dy;
            const float root = (2.f * qc) / (-qb - sqrtf(qb * qb - 4.f * qa * qc));
            out = root * wb + xa;
        } else {
            const float th = (x1 - xa) / wb;
            const float tt = th * (1.f - th);
            out = ya + hbin * (sl * th * th + d0 * tt) / (sl + (d0 + d1 - 2.f * sl) * tt);
        }
    }
    return out;
}

__global__ __launch_bounds__(256) void spline_flow_kernel(const float* __restrict__ z, const float* __restrict__ h,
                                                          const int* __restrict__ lens, float* __restrict__ zo, int T,
                                                          int K, float inv_sqrt_fc, float tail, int inverse,
                                                          int flip_in, int flip_out) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (t >= T) return;
    const float m = (lens ? t < lens[b] : true) ? 1.f : 0.f;
    const float x0 = z[((size_t)b * 2 + (flip_in ? 1 : 0)) * T + t];
    const float x1 = z[((size_t)b * 2 + (flip_in ? 0 : 1)) * T + t];
    const float* hb = h + (size_t)b * (3 * K - 1) * T + t;
    const float out = spline_eval(x1, [&](int i) { return i < 2 * K ? hb[(size_t)i * T] * inv_sqrt_fc * m : hb[(size_t)i * T] * m; }, K, tail,
                                  inverse);
    const float o0 = x0 * m, o1 = out * m;
    zo[((size_t)b * 2 + (flip_out ? 1 : 0)) * T + t] = o0;
    zo[((size_t)b * 2 + (flip_out ? 0 : 1)) * T + t] = o1;
}

// The same with ConvFlow's `proj` (a 1 x 1 conv C -> 3K - 1, modules/flow/modules.py:418,427) evaluated here instead of by a conv launch of
// 29 output rows (16 workgroups, 25 us) followed by this kernel on 16 workgroups of 100 live threads (20 us, its 49 parameter loads sunk
// into the branches one by one).  A workgroup = 32 columns of one item: the [C, 32] tile of the DDSConv output and the [3K - 1, C] weights
// go to LDS in one batch of loads, thread (column, g) forms rows g, g + 8, ... as plain fp32 dot products over ascending c (more exact
// than the f16x3 conv it replaces, not bit-equal to it), and the first 32 threads run the transform from LDS.
constexpr int SPP_TT = 32, SPP_MAXC = 256, SPP_MAXR = 3 * SPL_MAXK - 1;
__global__ __launch_bounds__(256) void spline_flow_proj_kernel(const float* __restrict__ z, const float* __restrict__ hc,
                                                               const float* __restrict__ pw, const float* __restrict__ pb,
                                                               const int* __restrict__ lens, float* __restrict__ zo, int C, int T,
                                                               int K, float inv_sqrt_fc, float tail, int inverse, int flip_in,
                                                               int flip_out) {
    extern __shared__ float sm[];
    const int R = 3 * K - 1;
    float* xs = sm;                          // [C][SPP_TT + 1]
    float* ws = xs + C * (SPP_TT + 1);       // [R][C + 1]
    float* hs = ws + R * (C + 1);            // [R][SPP_TT + 1]
    const int tid = threadIdx.x, tx = tid & (SPP_TT - 1), g = tid >> 5;
    const int t0 = blockIdx.x * SPP_TT, b = blockIdx.y;
    const int t = t0 + tx;
    const int tc = t < T ? t : T - 1;
    {
        constexpr int XU = SPP_MAXC / 8, WU = (SPP_MAXR * SPP_MAXC + 255) / 256;   // 32 rows of x, 47 weights per thread at the limits
        float tx_[XU], tw[WU];
        const float* xb = hc + (size_t)b * C * T + tc;
        const int nw = R * C;
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int c = g + 8 * u;
            tx_[u] = xb[(size_t)(c < C ? c : C - 1) * T];
        }
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int e = tid + 256 * u;
            tw[u] = pw[e < nw ? e : nw - 1];
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u = 0; u < XU; ++u) {
            const int c = g + 8 * u < C ? g + 8 * u : C - 1;       // (clamped like the load: the same value to the same place)
            xs[c * (SPP_TT + 1) + tx] = tx_[u];
        }
#pragma unroll
        for (int u = 0; u < WU; ++u) {
            const int e = tid + 256 * u < nw ? tid + 256 * u : nw - 1;
            const int r = e / C, c = e - r * C;
            ws[r * (C + 1) + c] = tw[u];
        }
    }
    __syncthreads();
    {
        float acc[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const int r = g + 8 * q; acc[q] = pb ? pb[r < R ? r : R - 1] : 0.f; }
#pragma unroll 4
        for (int c = 0; c < C; ++c) {
            const float xv = xs[c * (SPP_TT + 1) + tx];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = g + 8 * q;
                acc[q] = fmaf(ws[(r < R ? r : R - 1) * (C + 1) + c], xv, acc[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int r = g + 8 * q;
            if (r < R) hs[r * (SPP_TT + 1) + tx] = acc[q];
        }
    }
    __syncthreads();
    if (g != 0 || t >= T) return;
    const bool live = lens ? t < lens[b] : true;
    const float m = live ? 1.f : 0.f;
    const float x0 = z[((size_t)b * 2 + (flip_in ? 1 : 0)) * T + t];
    const float x1 = z[((size_t)b * 2 + (flip_in ? 0 : 1)) * T + t];
    // (a select where the unfused kernel multiplies by the mask: whatever the conditioning tensor holds beyond the length stays out)
    const float out = spline_eval(x1, [&](int i) { const float v = live ? hs[i * (SPP_TT + 1) + tx] : 0.f; return i < 2 * K ? v * inv_sqrt_fc * m : v * m; },
                                  K, tail, inverse);
    zo[((size_t)b * 2 + (flip_out ? 1 : 0)) * T + t] = x0 * m;
    zo[((size_t)b * 2 + (flip_out ? 0 : 1)) * T + t] = out * m;
}

// ElementwiseAffine reverse: (x - m[c]) * exp(-logs[c]) * mask    (modules/flow/modules.py:338-340)
__global__ __launch_bounds__(256) void affine_reverse_kernel(const float* __restrict__ x, const float* __restrict__ mm,
                                                             const float* __restrict__ logs,
                                                             const int* __restrict__ lens, float* __restrict__ y, int C,
                                                             int T) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int bc = blockIdx.y;
    if (t >= T) return;
    const int c = bc % C;
    const float mk = (lens ? t < lens[bc / C] : true) ? 1.f : 0.f;
    y[(size_t)bc * T + t] = (x[(size_t)bc * T + t] - mm[c]) * expf(-logs[c]) * mk;
}

// x = emb(tokens) * sqrt(hidden), transposed to [B, H, T] and masked   (TextEncoder.forward vits.py:58-62)
__global__ __launch_bounds__(256) void embed_kernel(const long long* __restrict__ tok, const float* __restrict__ w,
                                                    const int* __restrict__ lens, float* __restrict__ y, int Hd, int T,
                                                    int n_vocab, float scale) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int bh = blockIdx.y;
    if (t >= T) return;
    const int b = bh / Hd, hh = bh - b * Hd;
    long long id = tok[(size_t)b * T + t];
    id = id < 0 ? 0 : (id > n_vocab - 1 ? n_vocab - 1 : id);
    const float mk = (lens ? t < lens[b] : true) ? 1.f : 0.f;
    y[(size_t)bh * T + t] = w[(size_t)id * Hd + hh] * scale * mk;
}

// Durations: w_ceil = ceil(exp(logw) * mask * length_scale), its running sum and y_len = max(sum, 1)
// (SynthesizerTrn.infer vits.py:341-343; the cumsum of generate_path utils/util.py:633).  One thread per item:
// T_text is ~10^2 and the sum must be sequential anyway.
__global__ void durations_kernel(const float* __restrict__ logw, const int* __restrict__ lens, float length_scale,
                                 float* __restrict__ w_ceil, int* __restrict__ cum, int* __restrict__ ylen, int B,
                                 int T) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    const int len = lens ? lens[b] : T;
    float run = 0.f;
    for (int t = 0; t < T; ++t) {
        const float wv = t < len ? ceilf(expf(logw[(size_t)b * T + t]) * length_scale) : 0.f;
        w_ceil[(size_t)b * T + t] = wv;
        run += wv;
        cum[(size_t)b * T + t] = (int)run;
    }
    ylen[b] = run < 1.f ? 1 : (int)run;
}

// Expansion along the alignment path: out[b, :, y] = src[b, :, x(y)] where cum[x-1] <= y < cum[x], 0 when no token
// owns frame y or y >= y_len  ( = attn @ src with attn = generate_path(w_ceil, mask), vits.py:345-353 );
// optionally writes attn [B, 1, Ty, Tx] itself.
// A workgroup = 64 frames of one item x one slice of EP_DS channels (blockIdx.z): the owner of each frame is found once (cum is
// non-decreasing, so it is the first x < x_len with cum[x] > y: a binary search), the channel rows are written as 256-B segments and
// the 64 x Tx block of attn -- contiguous in memory -- as one coalesced run by the z = 0 slice.  (Rounds 2-3: one THREAD per frame
// looping over all D channels and writing its own attn row, 32 workgroups: 64 us per call at B = 16, D = 192, Ty = 400.)
constexpr int EP_FR = 64, EP_DS = 32;
__global__ __launch_bounds__(256) void expand_path_kernel(const float* __restrict__ src, long long sbs, const int* __restrict__ cum,
                                                          const int* __restrict__ xlens, const int* __restrict__ ylens,
                                                          float* __restrict__ out, float* __restrict__ attn, int D,
                                                          int Tx, int Ty) {
    __shared__ int toks[EP_FR];
    const int tid = threadIdx.x;
    const int y0 = blockIdx.x * EP_FR, b = blockIdx.y;
    if (tid < EP_FR) {
        const int y = y0 + tid;
        const int xl = xlens ? xlens[b] : Tx;
        int tok = -1;
        if (y < Ty && y < ylens[b]) {
            const int* cr = cum + (size_t)b * Tx;
            int lo = 0, hi = xl;                         // first x in [0, xl) with cr[x] > y
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cr[mid] > y) hi = mid; else lo = mid + 1;
            }
            tok = lo < xl ? lo : -1;
        }
        toks[tid] = tok;
    }
    __syncthreads();
    const int yy = tid & (EP_FR - 1), dg = tid >> 6;
    const int y = y0 + yy;
    const int tok = toks[yy];
    const int d0 = blockIdx.z * EP_DS;
    if (y < Ty) {
        const float* sb = src + (size_t)b * (size_t)sbs;
#pragma unroll 4
        for (int d = d0 + dg; d < d0 + EP_DS && d < D; d += 4)
            out[((size_t)b * D + d) * Ty + y] = tok >= 0 ? sb[(size_t)d * Tx + tok] : 0.f;
    }
    if (attn && blockIdx.z == 0) {
        const int rows = Ty - y0 < EP_FR ? Ty - y0 : EP_FR;
        float* ab = attn + ((size_t)b * Ty + y0) * Tx;
        for (int idx = tid; idx < rows * Tx; idx += 256) {
            const int r = idx / Tx, x = idx - r * Tx;
            ab[idx] = x == toks[r] ? 1.f : 0.f;
        }
    }
}

// z_p = m + noise * exp(logs) * noise_scale      (vits.py:355; NOT masked there)
__global__ __launch_bounds__(256) void gauss_sample_kernel(const float* __restrict__ m, const float* __restrict__ logs,
                                                           const float* __restrict__ noise, float scale,
                                                           float* __restrict__ out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = m[i] + noise[i] * expf(logs[i]) * scale;
}

}  // namespace amp

using namespace amp;

// A/B switch of the tiled attention kernel (amp_set_rel_attention_tiled; both forms give the same bits)
static std::atomic<int> g_rel_attention_tiled{1};
static bool rel_attention_tiled() { return g_rel_attention_tiled.load() != 0; }

#define VT_CHECK(cond, ...)                                   \
    do {                                                      \
        if (!(cond)) { set_error(__VA_ARGS__); return AMP_ERR_INVALID; } \
    } while (0)
#define VT_LAUNCHED(name)                                                                         \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess) { set_error(name ": %s", hipGetErrorString(e__)); return AMP_ERR_HIP; } \
    } while (0)

// (ensure_dynamic_lds<Kernel>(): amp_internal.h -- the attribute is per device, set once per (kernel, device) under a lock)

extern "C" {

static int layer_norm_run(const char* who, const float* x_dev, const float* res_dev, const float* gamma_dev, const float* beta_dev,
                          const float* post_dev, const int* lens_dev, const float* dw_w, const float* dw_b, int K, int dil, int B, int C,
                          int T, float eps, int gelu, float* y_dev, void* stream) {
    VT_CHECK(x_dev && gamma_dev && beta_dev && y_dev && B > 0 && C > 0 && T > 0 && B <= 65535, "%s: bad argument", who);
    VT_CHECK(x_dev != y_dev || !dw_w, "%s: the depthwise prologue reads neighbouring columns: y must not alias x", who);
    const dim3 grid((T + LN_TT - 1) / LN_TT, B);
    const bool nc24 = C <= 24 * LN_G;
    if (!dw_w) {
        auto kern = nc24 ? layer_norm_c_kernel<0, 24> : layer_norm_c_kernel<0, 32>;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, x_dev,
                           res_dev, gamma_dev, beta_dev, post_dev, lens_dev, nullptr, nullptr, 1, y_dev, C, T, eps, gelu);
    } else {
        VT_CHECK(K == 3 && dil > 0 && C <= LN_DW_MAXC, "%s: the fused depthwise prologue covers K = 3, C <= %d (got K=%d dilation=%d C=%d): run amp_dwconv first", who, LN_DW_MAXC, K, dil, C);
        auto kern = nc24 ? layer_norm_c_kernel<3, 24> : layer_norm_c_kernel<3, 32>;
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, x_dev,
                           res_dev, gamma_dev, beta_dev, post_dev, lens_dev, dw_w, dw_b, dil, y_dev, C, T, eps, gelu);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: %s", who, hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_layer_norm_c(const float* x_dev, const float* res_dev, const float* gamma_dev, const float* beta_dev,
                     const float* post_dev, int B, int C, int T, float eps, int gelu, float* y_dev, void* stream) {
    return layer_norm_run("amp_layer_norm_c", x_dev, res_dev, gamma_dev, beta_dev, post_dev, nullptr, nullptr, nullptr, 0, 1, B, C, T, eps,
                          gelu, y_dev, stream);
}

int amp_layer_norm_c_ragged(const float* x_dev, const float* res_dev, const float* gamma_dev, const float* beta_dev,
                            const float* post_dev, const int* lens_dev, int B, int C, int T, float eps, int gelu, float* y_dev,
                            void* stream) {
    return layer_norm_run("amp_layer_norm_c_ragged", x_dev, res_dev, gamma_dev, beta_dev, post_dev, lens_dev, nullptr, nullptr, 0, 1, B,
                          C, T, eps, gelu, y_dev, stream);
}

int amp_dds_seam(const float* y_dev, const float* x_dev, const float* gamma2_dev, const float* beta2_dev, float eps2,
                 const float* dw_weight_dev, const float* dw_bias_dev, int K, int dilation, const float* gamma1_dev, const float* beta1_dev,
                 float eps1, const int* lens_dev, int B, int C, int T, float* x_out_dev, float* z_out_dev, void* stream) {
    VT_CHECK(y_dev && x_dev && gamma2_dev && beta2_dev && dw_weight_dev && gamma1_dev && beta1_dev && x_out_dev && z_out_dev && B > 0 &&
                 C > 0 && T > 0 && B <= 65535 && dilation > 0, "amp_dds_seam: bad argument");
    VT_CHECK(x_out_dev != y_dev && z_out_dev != y_dev && z_out_dev != x_dev && x_out_dev != x_dev && x_out_dev != z_out_dev,
             "amp_dds_seam: the outputs must not alias the inputs (neighbouring columns are re-read)");
    if (K != 3 || C > LN_G * DS_NC) {
        set_error("amp_dds_seam: K = %d, C = %d outside the fused form (K = 3, C <= %d): run amp_layer_norm_c + amp_dwconv_layer_norm_c", K, C, LN_G * DS_NC);
        return AMP_ERR_UNSUPPORTED;
    }
    hipLaunchKernelGGL(dds_seam_kernel, dim3((T + LN_TT - 1) / LN_TT, B), dim3(256), 0, (hipStream_t)stream, y_dev, x_dev, gamma2_dev,
                       beta2_dev, dw_weight_dev, dw_bias_dev, gamma1_dev, beta1_dev, lens_dev, x_out_dev, z_out_dev, C, T, dilation, eps2, eps1);
    VT_LAUNCHED("amp_dds_seam");
    return AMP_OK;
}

int amp_dwconv_layer_norm_c(const float* x_dev, const float* dw_weight_dev, const float* dw_bias_dev, int K, int dilation,
                            const float* gamma_dev, const float* beta_dev, const int* lens_dev, int B, int C, int T, float eps,
                            int gelu, float* y_dev, void* stream) {
    VT_CHECK(dw_weight_dev, "amp_dwconv_layer_norm_c: null depthwise weight");
    return layer_norm_run("amp_dwconv_layer_norm_c", x_dev, nullptr, gamma_dev, beta_dev, nullptr, lens_dev, dw_weight_dev, dw_bias_dev, K,
                          dilation, B, C, T, eps, gelu, y_dev, stream);
}

static int rel_attention_run(const char* who, const float* q_dev, const float* k_dev, const float* v_dev, long long bs, const float* emb_k_dev,
                             const float* emb_v_dev, const int* lens_dev, int B, int H, int dk, int T, int window, float* out_dev,
                             void* stream) {
    VT_CHECK(q_dev && k_dev && v_dev && emb_k_dev && emb_v_dev && out_dev && B > 0 && H > 0 && dk > 0 && T > 0 && window >= 0,
             "%s: bad argument", who);
    VT_CHECK(B <= 65535 && H <= 65535, "%s: B=%d H=%d exceed the grid", who, B, H);
    VT_CHECK(bs >= (long long)H * dk * T, "%s: batch stride %lld < H*dk*T", who, bs);
    const int Tp = (T + 3) & ~3;
    const size_t lds_fixed = (size_t)(RA_QB * dk + RA_QB * 16 + RA_QB + RA_QB * Tp + 32 * dk) * sizeof(float);
    const size_t lds_kv = (size_t)dk * RA_KS * sizeof(float);
    const bool one = T <= RA_KT && lds_fixed + 2 * lds_kv <= 150 * 1024;          // keys and values staged together
    const size_t lds_tile = lds_fixed + (one ? 2 : 1) * lds_kv;
    if (rel_attention_tiled() && dk % 4 == 0 && dk <= 128 && 2 * window + 1 <= 16 && lds_tile <= 150 * 1024) {
        auto kern = one ? rel_attention_tile_kernel<true> : rel_attention_tile_kernel<false>;
        {
            const hipError_t e = one ? ensure_dynamic_lds<&rel_attention_tile_kernel<true>>(150 * 1024) : ensure_dynamic_lds<&rel_attention_tile_kernel<false>>(150 * 1024);
            if (e != hipSuccess) { set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e)); return AMP_ERR_HIP; }
        }
        hipLaunchKernelGGL(kern, dim3((T + RA_QB - 1) / RA_QB, H, B), dim3(256), lds_tile, (hipStream_t)stream, q_dev,
                           k_dev, v_dev, emb_k_dev, emb_v_dev, lens_dev, out_dev, bs, H, dk, T, window, Tp);
    } else {
        const size_t lds = (size_t)(dk + T) * sizeof(float);
        if (lds > 60 * 1024) { set_error("%s: T=%d needs %zu B of LDS", who, T, lds); return AMP_ERR_UNSUPPORTED; }
        hipLaunchKernelGGL(rel_attention_row_kernel, dim3(T, H, B), dim3(64), lds, (hipStream_t)stream, q_dev, k_dev, v_dev, emb_k_dev,
                           emb_v_dev, lens_dev, out_dev, bs, H, dk, T, window);
    }
    {
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("%s: %s", who, hipGetErrorString(e)); return AMP_ERR_HIP; }
    }
    return AMP_OK;
}

int amp_rel_attention(const float* q_dev, const float* k_dev, const float* v_dev, const float* emb_k_dev,
                      const float* emb_v_dev, const int* lens_dev, int B, int H, int dk, int T, int window, float* out_dev,
                      void* stream) {
    return rel_attention_run("amp_rel_attention", q_dev, k_dev, v_dev, (long long)H * dk * T, emb_k_dev, emb_v_dev, lens_dev, B, H, dk, T,
                             window, out_dev, stream);
}

int amp_rel_attention_strided(const float* q_dev, const float* k_dev, const float* v_dev, long long qkv_batch_stride,
                              const float* emb_k_dev, const float* emb_v_dev, const int* lens_dev, int B, int H, int dk, int T,
                              int window, float* out_dev, void* stream) {
    return rel_attention_run("amp_rel_attention_strided", q_dev, k_dev, v_dev, qkv_batch_stride, emb_k_dev, emb_v_dev, lens_dev, B, H, dk,
                             T, window, out_dev, stream);
}

int amp_set_rel_attention_tiled(int on) {
    g_rel_attention_tiled.store(on ? 1 : 0);
    return AMP_OK;
}

int amp_dwconv(const float* x_dev, const float* w_dev, const float* bias_dev, const int* lens_dev, int B, int C, int T, int K,
               int dilation, float* y_dev, void* stream) {
    VT_CHECK(x_dev && w_dev && y_dev && B > 0 && C > 0 && T > 0 && K > 0 && dilation > 0 && (size_t)B * C <= 65535,
             "amp_dwconv: bad argument");
    hipLaunchKernelGGL(dwconv_kernel, dim3((T + 255) / 256, B * C), dim3(256), 0, (hipStream_t)stream, x_dev, w_dev, bias_dev,
                       lens_dev, y_dev, C, T, K, dilation);
    VT_LAUNCHED("amp_dwconv");
    return AMP_OK;
}

int amp_spline_flow(const float* z_dev, const float* h_dev, const int* lens_dev, int B, int T, int num_bins,
                    int filter_channels, float tail_bound, int inverse, int flip_in, int flip_out, float* z_out_dev,
                    void* stream) {
    VT_CHECK(z_dev && h_dev && z_out_dev && z_dev != z_out_dev && B > 0 && T > 0 && filter_channels > 0 && tail_bound > 0.f &&
                 B <= 65535, "amp_spline_flow: bad argument");
    if (num_bins < 2 || num_bins > SPL_MAXK) { set_error("amp_spline_flow: %d bins (2..%d)", num_bins, SPL_MAXK); return AMP_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(spline_flow_kernel, dim3((T + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, z_dev, h_dev, lens_dev,
                       z_out_dev, T, num_bins, 1.0f / sqrtf((float)filter_channels), tail_bound, inverse, flip_in, flip_out);
    VT_LAUNCHED("amp_spline_flow");
    return AMP_OK;
}

int amp_spline_flow_proj(const float* z_dev, const float* hc_dev, const float* proj_w_dev, const float* proj_b_dev, const int* lens_dev,
                         int B, int C, int T, int num_bins, int filter_channels, float tail_bound, int inverse, int flip_in, int flip_out,
                         float* z_out_dev, void* stream) {
    VT_CHECK(z_dev && hc_dev && proj_w_dev && z_out_dev && z_dev != z_out_dev && B > 0 && C > 0 && T > 0 && filter_channels > 0 &&
                 tail_bound > 0.f && B <= 65535, "amp_spline_flow_proj: bad argument");
    if (num_bins < 2 || num_bins > SPL_MAXK || C > SPP_MAXC) {
        set_error("amp_spline_flow_proj: %d bins (2..%d), %d channels (<= %d): run the 1x1 conv and amp_spline_flow", num_bins, SPL_MAXK, C, SPP_MAXC);
        return AMP_ERR_UNSUPPORTED;
    }
    const int R = 3 * num_bins - 1;
    const size_t lds = (size_t)(C * (SPP_TT + 1) + R * (C + 1) + R * (SPP_TT + 1)) * sizeof(float);
    {
        const hipError_t e = ensure_dynamic_lds<&spline_flow_proj_kernel>(100 * 1024);
        if (e != hipSuccess) { set_error("amp_spline_flow_proj: hipFuncSetAttribute: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    }
    hipLaunchKernelGGL(spline_flow_proj_kernel, dim3((T + SPP_TT - 1) / SPP_TT, B), dim3(256), lds, (hipStream_t)stream, z_dev, hc_dev,
                       proj_w_dev, proj_b_dev, lens_dev, z_out_dev, C, T, num_bins, 1.0f / sqrtf((float)filter_channels), tail_bound, inverse,
                       flip_in, flip_out);
    VT_LAUNCHED("amp_spline_flow_proj");
    return AMP_OK;
}

int amp_affine_reverse(const float* x_dev, const float* m_dev, const float* logs_dev, const int* lens_dev, int B, int C, int T,
                       float* y_dev, void* stream) {
    VT_CHECK(x_dev && m_dev && logs_dev && y_dev && B > 0 && C > 0 && T > 0 && (size_t)B * C <= 65535, "amp_affine_reverse: bad argument");
    hipLaunchKernelGGL(affine_reverse_kernel, dim3((T + 255) / 256, B * C), dim3(256), 0, (hipStream_t)stream, x_dev, m_dev,
                       logs_dev, lens_dev, y_dev, C, T);
    VT_LAUNCHED("amp_affine_reverse");
    return AMP_OK;
}

int amp_embed_tokens(const long long* tokens_dev, const float* weight_dev, const int* lens_dev, int B, int T, int hidden,
                     int n_vocab, float scale, float* y_dev, void* stream) {
    VT_CHECK(tokens_dev && weight_dev && y_dev && B > 0 && T > 0 && hidden > 0 && n_vocab > 0 && (size_t)B * hidden <= 65535,
             "amp_embed_tokens: bad argument");
    hipLaunchKernelGGL(embed_kernel, dim3((T + 255) / 256, B * hidden), dim3(256), 0, (hipStream_t)stream, tokens_dev, weight_dev,
                       lens_dev, y_dev, hidden, T, n_vocab, scale);
    VT_LAUNCHED("amp_embed_tokens");
    return AMP_OK;
}

int amp_durations(const float* logw_dev, const int* lens_dev, int B, int T, float length_scale, float* w_ceil_dev, int* cum_dev,
                  int* ylen_dev, void* stream) {
    VT_CHECK(logw_dev && w_ceil_dev && cum_dev && ylen_dev && B > 0 && T > 0, "amp_durations: bad argument");
    hipLaunchKernelGGL(durations_kernel, dim3((B + 63) / 64), dim3(64), 0, (hipStream_t)stream, logw_dev, lens_dev, length_scale,
                       w_ceil_dev, cum_dev, ylen_dev, B, T);
    VT_LAUNCHED("amp_durations");
    return AMP_OK;
}

static int expand_path_run(const char* who, const float* src_dev, long long sbs, const int* cum_dev, const int* xlens_dev,
                           const int* ylens_dev, int B, int D, int Tx, int Ty, float* out_dev, float* attn_dev, void* stream) {
    VT_CHECK(src_dev && cum_dev && ylens_dev && out_dev && B > 0 && D > 0 && Tx > 0 && Ty > 0 && B <= 65535 && D <= 65535 * EP_DS,
             "%s: bad argument", who);
    VT_CHECK(sbs >= (long long)D * Tx, "%s: batch stride %lld < D*Tx", who, sbs);
    hipLaunchKernelGGL(expand_path_kernel, dim3((Ty + EP_FR - 1) / EP_FR, B, (D + EP_DS - 1) / EP_DS), dim3(256), 0, (hipStream_t)stream,
                       src_dev, sbs, cum_dev, xlens_dev, ylens_dev, out_dev, attn_dev, D, Tx, Ty);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s: %s", who, hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_expand_path(const float* src_dev, const int* cum_dev, const int* xlens_dev, const int* ylens_dev, int B, int D, int Tx,
                    int Ty, float* out_dev, float* attn_dev, void* stream) {
    return expand_path_run("amp_expand_path", src_dev, (long long)D * Tx, cum_dev, xlens_dev, ylens_dev, B, D, Tx, Ty, out_dev, attn_dev,
                           stream);
}

int amp_expand_path_strided(const float* src_dev, long long src_batch_stride, const int* cum_dev, const int* xlens_dev,
                            const int* ylens_dev, int B, int D, int Tx, int Ty, float* out_dev, float* attn_dev, void* stream) {
    return expand_path_run("amp_expand_path_strided", src_dev, src_batch_stride, cum_dev, xlens_dev, ylens_dev, B, D, Tx, Ty, out_dev,
                           attn_dev, stream);
}

int amp_add_channel_bias(float* x_dev, const float* cb_dev, int B, int C, int T, void* stream) {
    VT_CHECK(x_dev && cb_dev && B > 0 && C > 0 && T > 0, "amp_add_channel_bias: bad argument");
    hipError_t e = launch_add_channel_bias(x_dev, cb_dev, B, C, T, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("amp_add_channel_bias: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_gauss_sample(const float* m_dev, const float* logs_dev, const float* noise_dev, size_t n, float noise_scale,
                     float* out_dev, void* stream) {
    VT_CHECK(m_dev && logs_dev && noise_dev && out_dev && n > 0, "amp_gauss_sample: bad argument");
    size_t blocks = (n + 255) / 256;
    if (blocks > 65535u * 16u) blocks = 65535u * 16u;
    hipLaunchKernelGGL(gauss_sample_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, m_dev, logs_dev, noise_dev,
                       noise_scale, out_dev, n);
    VT_LAUNCHED("amp_gauss_sample");
    return AMP_OK;
}

}  // extern "C"
