// First run on MI355X in round 2: tests/test_gpu_vits_infer.py green (every op against oracle/vits_infer_oracle.py,
// SynthesizerTrn.infer against the reference's golden vectors: durations / path exact, waveform <= 1e-4).
//
// Frame-rate kernels of the text -> duration -> alignment front of VITS inference (SURVEY.md §8 f.4,
// SynthesizerTrn.infer models/tts/vits/vits.py:320-369).  T_text is 10^2, channels ~200: everything here is
// latency-bound; the kernels are written for clarity (one thread per output, coalesced along time), the dense
// convolutions between them run on conv_f16x3.hip.  Layout [B, C, T], fp32.
#include "amp_internal.h"

namespace amp {

__device__ __forceinline__ float gelu_erf(float v) { return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); }

// LayerNorm over the CHANNEL axis (modules/base/base_module.py:20-23), optionally of x + res (Encoder's
// norm(x + y), modules/transformer/attentions.py:69,73) and optionally followed by GELU (DDSConv,
// modules/flow/modules.py:64-68) and by "+ post" (DDSConv's x = x + y, :70).
// A workgroup = 32 time columns x 8 channel groups: thread (tx, g) holds channels g, g + 8, ... of column tx in registers (up to
// 32 of them = C <= 256; more are re-read), rows are read as 128-B segments, and the two reductions over C (mean, then the biased
// variance of the centred values -- two passes, like the op it replaces) go through LDS in a fixed order.  Round 3: the first
// version ran ONE thread per (b, t) with three serial loops over C -- 120 us per call at B = 16, C = 192, T = 150 (16 workgroups
// of 150 live threads, 576 dependent loads each), a third of VITS text -> wave.
constexpr int LN_TT = 32, LN_G = 8, LN_NC = 32;
__global__ __launch_bounds__(256) void layer_norm_c_kernel(const float* __restrict__ x, const float* __restrict__ res,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ post,
                                                           float* __restrict__ y, int C, int T, float eps, int gelu) {
    __shared__ float red[LN_G][LN_TT + 1];
    const int tx = threadIdx.x & (LN_TT - 1), g = threadIdx.x / LN_TT;
    const int t = blockIdx.x * LN_TT + tx;
    const int b = blockIdx.y;
    const bool ok = t < T;
    const size_t base = (size_t)b * C * T + (ok ? t : 0);
    const float* xb = x + base;
    const float* rb = res ? res + base : nullptr;
    auto value = [&](int c) { return xb[(size_t)c * T] + (rb ? rb[(size_t)c * T] : 0.f); };
    auto reduce = [&](float part) {                      // sum over the 8 channel groups of column tx, same order in every thread
        red[g][tx] = part;
        __syncthreads();
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < LN_G; ++k) s += red[k][tx];
        __syncthreads();
        return s;
    };
    float v[LN_NC];
    float part = 0.f;
    {
        // Round 4: every load of the thread's 24-32 channels UNCONDITIONAL and issued before the first use (clamped channel index, the
        // select applied afterwards).  The round-3 form loaded inside `(ok && c < C) ? value(c) : 0.f`: hipcc sinks such a load into the
        // branch and waits for it with vmcnt(0) -- 24 dependent HBM round trips per thread, 19 us per call for 1.2 MB.
        float xr[LN_NC], rr[LN_NC];
#pragma unroll
        for (int i = 0; i < LN_NC; ++i) {
            const int c = g + LN_G * i;
            xr[i] = xb[(size_t)(c < C ? c : C - 1) * T];
        }
        if (rb) {                                        // block-uniform
#pragma unroll
            for (int i = 0; i < LN_NC; ++i) {
                const int c = g + LN_G * i;
                rr[i] = rb[(size_t)(c < C ? c : C - 1) * T];
            }
        } else {
#pragma unroll
            for (int i = 0; i < LN_NC; ++i) rr[i] = 0.f;
        }
        asm volatile("" ::: "memory");                   // keep the requests together (mel.hip: the same fence)
#pragma unroll
        for (int i = 0; i < LN_NC; ++i) {
            const int c = g + LN_G * i;
            v[i] = (ok && c < C) ? xr[i] + rr[i] : 0.f;
            part += v[i];
        }
    }
    for (int c = g + LN_G * LN_NC; c < C; c += LN_G) part += ok ? value(c) : 0.f;
    const float mu = reduce(part) / (float)C;
    part = 0.f;
#pragma unroll
    for (int i = 0; i < LN_NC; ++i) {
        const float d = v[i] - mu;
        part += (g + LN_G * i < C) ? d * d : 0.f;
    }
    for (int c = g + LN_G * LN_NC; c < C; c += LN_G) { const float d = (ok ? value(c) : 0.f) - mu; part += d * d; }
    const float rstd = 1.0f / sqrtf(reduce(part) / (float)C + eps);
    if (!ok) return;
    float* yb = y + base;
    auto emit = [&](int c, float xv) {
        float o = (xv - mu) * rstd * gamma[c] + beta[c];
        if (gelu) o = gelu_erf(o);
        if (post) o += post[base + (size_t)c * T];
        yb[(size_t)c * T] = o;
    };
#pragma unroll
    for (int i = 0; i < LN_NC; ++i) {
        const int c = g + LN_G * i;
        if (c < C) emit(c, v[i]);
    }
    for (int c = g + LN_G * LN_NC; c < C; c += LN_G) emit(c, value(c));
}

// Self-attention with windowed relative-position embeddings, heads_share = True
// (MultiHeadAttention.attention, modules/transformer/attentions.py:232-272).  One 64-lane workgroup per (b, h, i):
//   s[j]   = (q_i . k_j + [|j - i| <= w] q_i . Ek[j - i + w]) / sqrt(dk);   -1e4 where query or key is padding
//   p      = softmax_j(s)
//   out[d] = sum_j p[j] v[d, j] + sum_{|j - i| <= w} p[j] Ev[j - i + w][d]
// q, k, v, out: [B, H*dk, T]; Ek, Ev: [2w+1, dk].  LDS: q_i (dk floats) + p (T floats).
__global__ __launch_bounds__(64) void rel_attention_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                           const float* __restrict__ v, const float* __restrict__ ek,
                                                           const float* __restrict__ ev, const int* __restrict__ lens,
                                                           float* __restrict__ out, int H, int dk, int T, int window) {
    extern __shared__ float sm[];
    float* qs = sm;          // [dk]
    float* p = sm + dk;      // [T]
    const int lane = threadIdx.x;
    const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
    const int len = lens ? lens[b] : T;
    const size_t base = ((size_t)b * H + h) * dk * T;
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
        out[base + (size_t)d * T + i] = (acc + ar) * inv;
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
    float out = x1;
    if (x1 >= -tail && x1 <= tail) {
        const float min_w = 1e-3f, min_h = 1e-3f, min_d = 1e-3f;
        float xk[SPL_MAXK + 1], yk[SPL_MAXK + 1], dd[SPL_MAXK + 1];
        // softmax -> bin fractions -> knots in [-tail, tail] with exact end points
        for (int pass = 0; pass < 2; ++pass) {
            float* kn = pass == 0 ? xk : yk;
            const float lo = pass == 0 ? min_w : min_h;
            float mxv = -3.0e38f;
            for (int i = 0; i < K; ++i) mxv = fmaxf(mxv, hb[(size_t)(pass * K + i) * T] * inv_sqrt_fc * m);
            float se = 0.f;
            for (int i = 0; i < K; ++i) se += expf(hb[(size_t)(pass * K + i) * T] * inv_sqrt_fc * m - mxv);
            float cum = 0.f;
            kn[0] = -tail;
            for (int i = 0; i < K; ++i) {
                const float frac = lo + (1.f - lo * K) * (expf(hb[(size_t)(pass * K + i) * T] * inv_sqrt_fc * m - mxv) / se);
                cum += frac;
                kn[i + 1] = 2.f * tail * cum - tail;
            }
            kn[K] = tail;
        }
        dd[0] = 1.f;                                     // min_d + softplus(log(exp(1 - min_d) - 1)) = 1 at both ends
        dd[K] = 1.f;
        for (int i = 1; i < K; ++i) {
            const float u = hb[(size_t)(2 * K + i - 1) * T] * m;
            dd[i] = min_d + (u > 20.f ? u : log1pf(expf(u)));   // F.softplus (threshold 20)
        }
        const float* kn = inverse ? yk : xk;
        int bin = 0;
        for (int i = 1; i < K; ++i) bin += (x1 >= kn[i]) ? 1 : 0;   // searchsorted over the interior knots (last += eps)
        const float xa = xk[bin], wb = xk[bin + 1] - xk[bin];
        const float ya = yk[bin], hbin = yk[bin + 1] - yk[bin];
        const float d0 = dd[bin], d1 = dd[bin + 1];
        const float s = hbin / wb;
        if (inverse) {
            const float dy = x1 - ya;
            const float e = d0 + d1 - 2.f * s;
            const float qa = dy * e + hbin * (s - d0);
            const float qb = hbin * d0 - dy * e;
            const float qc = -s * dy;
            const float root = (2.f * qc) / (-qb - sqrtf(qb * qb - 4.f * qa * qc));
            out = root * wb + xa;
        } else {
            const float th = (x1 - xa) / wb;
            const float tt = th * (1.f - th);
            out = ya + hbin * (s * th * th + d0 * tt) / (s + (d0 + d1 - 2.f * s) * tt);
        }
    }
    const float o0 = x0 * m, o1 = out * m;
    zo[((size_t)b * 2 + (flip_out ? 1 : 0)) * T + t] = o0;
    zo[((size_t)b * 2 + (flip_out ? 0 : 1)) * T + t] = o1;
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
__global__ __launch_bounds__(256) void expand_path_kernel(const float* __restrict__ src, const int* __restrict__ cum,
                                                          const int* __restrict__ xlens, const int* __restrict__ ylens,
                                                          float* __restrict__ out, float* __restrict__ attn, int D,
                                                          int Tx, int Ty) {
    const int y = blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (y >= Ty) return;
    const int xl = xlens ? xlens[b] : Tx;
    int tok = -1;
    if (y < ylens[b]) {
        for (int x = 0; x < xl; ++x) {
            const int lo = x == 0 ? 0 : cum[(size_t)b * Tx + x - 1];
            if (y >= lo && y < cum[(size_t)b * Tx + x]) { tok = x; break; }
        }
    }
    for (int d = 0; d < D; ++d) out[((size_t)b * D + d) * Ty + y] = tok >= 0 ? src[((size_t)b * D + d) * Tx + tok] : 0.f;
    if (attn)
        for (int x = 0; x < Tx; ++x) attn[((size_t)b * Ty + y) * Tx + x] = x == tok ? 1.f : 0.f;
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

#define VT_CHECK(cond, ...)                                   \
    do {                                                      \
        if (!(cond)) { set_error(__VA_ARGS__); return AMP_ERR_INVALID; } \
    } while (0)
#define VT_LAUNCHED(name)                                                                         \
    do {                                                                                          \
        hipError_t e__ = hipGetLastError();                                                       \
        if (e__ != hipSuccess) { set_error(name ": %s", hipGetErrorString(e__)); return AMP_ERR_HIP; } \
    } while (0)

extern "C" {

int amp_layer_norm_c(const float* x_dev, const float* res_dev, const float* gamma_dev, const float* beta_dev,
                     const float* post_dev, int B, int C, int T, float eps, int gelu, float* y_dev, void* stream) {
    VT_CHECK(x_dev && gamma_dev && beta_dev && y_dev && B > 0 && C > 0 && T > 0 && B <= 65535, "amp_layer_norm_c: bad argument");
    hipLaunchKernelGGL(layer_norm_c_kernel, dim3((T + LN_TT - 1) / LN_TT, B), dim3(256), 0, (hipStream_t)stream, x_dev, res_dev,
                       gamma_dev, beta_dev, post_dev, y_dev, C, T, eps, gelu);
    VT_LAUNCHED("amp_layer_norm_c");
    return AMP_OK;
}

int amp_rel_attention(const float* q_dev, const float* k_dev, const float* v_dev, const float* emb_k_dev,
                      const float* emb_v_dev, const int* lens_dev, int B, int H, int dk, int T, int window, float* out_dev,
                      void* stream) {
    VT_CHECK(q_dev && k_dev && v_dev && emb_k_dev && emb_v_dev && out_dev && B > 0 && H > 0 && dk > 0 && T > 0 && window >= 0,
             "amp_rel_attention: bad argument");
    VT_CHECK(B <= 65535 && H <= 65535, "amp_rel_attention: B=%d H=%d exceed the grid", B, H);
    const size_t lds = (size_t)(dk + T) * sizeof(float);
    if (lds > 60 * 1024) { set_error("amp_rel_attention: T=%d needs %zu B of LDS", T, lds); return AMP_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(rel_attention_kernel, dim3(T, H, B), dim3(64), lds, (hipStream_t)stream, q_dev, k_dev, v_dev, emb_k_dev,
                       emb_v_dev, lens_dev, out_dev, H, dk, T, window);
    VT_LAUNCHED("amp_rel_attention");
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

int amp_expand_path(const float* src_dev, const int* cum_dev, const int* xlens_dev, const int* ylens_dev, int B, int D, int Tx,
                    int Ty, float* out_dev, float* attn_dev, void* stream) {
    VT_CHECK(src_dev && cum_dev && ylens_dev && out_dev && B > 0 && D > 0 && Tx > 0 && Ty > 0 && B <= 65535, "amp_expand_path: bad argument");
    hipLaunchKernelGGL(expand_path_kernel, dim3((Ty + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, src_dev, cum_dev, xlens_dev,
                       ylens_dev, out_dev, attn_dev, D, Tx, Ty);
    VT_LAUNCHED("amp_expand_path");
    return AMP_OK;
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
