// Mel / STFT front end on gfx950: framing (reflect padding) * window -> FFT -> |X| -> mel filterbank -> log in one
// kernel; nothing but the final features leaves the CU.  n_fft = 1024 runs mel1024_kernel (one wave per frame,
// radix-8 real FFT, lane-constant twiddles, 128-B row stores); any other power of two the generic mel_kernel below
// (radix-2 Stockham FFT in LDS, one workgroup per frame).
//
// Replaces utils/mel.py:20-170 (torch.stft + sqrt + matmul + log as 5 separate tensor ops) and the
// conv1d-with-Fourier-basis STFT of utils/stft.py:152-181,259-278 (TacotronSTFT).
#include <atomic>
#include <mutex>
#include <stddef.h>
#include <string.h>

#include "amp_internal.h"

namespace amp {

__device__ __forceinline__ int reflect_index(int s, int L) {
    // torch reflect padding (no edge repeat); pad < L is checked on the host
    if (s < 0) s = -s;
    if (s >= L) s = 2 * (L - 1) - s;
    return s;
}

// utils/mel.py:21-24 wants the extreme samples when the audio leaves [-1, 1].  The front-end kernels fold what they read anyway:
// `lo` / `hi` = the lane's running min / max; out-of-range lanes (rare) merge into range[0] / range[1] with an atomic max of the
// int bits -- monotone in the value for floats below -1 (more negative = larger int) and for floats above +1.
struct MelRange { int* range; int* reset; int seq; };
__device__ __forceinline__ void mel_range_begin(const MelRange& r) {
    if (r.range && blockIdx.x == 0 && threadIdx.x == 0) {
        r.range[2] = r.seq;
        if (r.reset) { r.reset[0] = __float_as_int(-1.0f); r.reset[1] = __float_as_int(1.0f); }
    }
}
__device__ __forceinline__ void mel_range_end(const MelRange& r, float lo, float hi) {
    if (!r.range) return;
    if (lo < -1.0f) atomicMax(&r.range[0], __float_as_int(lo));
    if (hi > 1.0f) atomicMax(&r.range[1], __float_as_int(hi));
}

__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ wav, const int* __restrict__ lens, int L, int F,
                                                  int n_fft, int log2n, int hop, int pad, int n_mel, float mag_eps,
                                                  float log_clip,
                                                  const float* __restrict__ window, const float* __restrict__ melbasis,
                                                  float* __restrict__ mel, float* __restrict__ mag,
                                                  float* __restrict__ re_out, float* __restrict__ im_out, const MelRange rng) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    mel_range_begin(rng);
    float2* buf0 = reinterpret_cast<float2*>(smem);   // [n_fft]
    float2* buf1 = buf0 + n_fft;                      // [n_fft]
    float2* tw = buf1 + n_fft;                        // [n_fft/2]  exp(-2*pi*i*m/n_fft)
    float* magl = reinterpret_cast<float*>(tw + n_fft / 2);  // [n_fft/2+1]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / F;
    const int f = blockIdx.x - b * F;
    const int half = n_fft >> 1;
    const int bins = half + 1;
    const float* wb = wav + (size_t)b * L;
    // ragged batch: item b holds lens[b] samples of the zero-padded row; its reflection padding mirrors at ITS
    // end and it has (lens[b] + 2*pad - n_fft) / hop + 1 frames -- later frames of the row are left untouched
    int Li = L;
    if (lens) { Li = lens[b] < L ? lens[b] : L; if (f > (Li + 2 * pad - n_fft) / hop || Li <= pad) return; }

    float rlo = 0.f, rhi = 0.f;
    for (int n = tid; n < n_fft; n += 256) {
        const int s = reflect_index(f * hop + n - pad, Li);
        const float xv = wb[s];
        rlo = fminf(rlo, xv); rhi = fmaxf(rhi, xv);
        buf0[n] = make_float2(xv * window[n], 0.f);
    }
    mel_range_end(rng, rlo, rhi);
    for (int m = tid; m < half; m += 256) {
        float sn, cs;
        sincospif(-2.0f * (float)m / (float)n_fft, &sn, &cs);
        tw[m] = make_float2(cs, sn);
    }
    __syncthreads();

    float2* in = buf0;
    float2* out = buf1;
    for (int s = 0; s < log2n; ++s) {
        const int Ns = 1 << s;
        const int tstride = half >> s;  // twiddle index step: (n_fft/2) / Ns
        for (int j = tid; j < half; j += 256) {
            const int k = j & (Ns - 1);
            const float2 w = tw[k * tstride];
            const float2 v0 = in[j];
            const float2 x1 = in[j + half];
            const float2 v1 = make_float2(x1.x * w.x - x1.y * w.y, x1.x * w.y + x1.y * w.x);
            const int j0 = ((j - k) << 1) + k;
            out[j0] = make_float2(v0.x + v1.x, v0.y + v1.y);
            out[j0 + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
        }
        __syncthreads();
        float2* t = in; in = out; out = t;
    }
    // `in` now holds the spectrum in natural order
    for (int k = tid; k < bins; k += 256) {
        const float2 v = in[k];
        const float m = sqrtf(v.x * v.x + v.y * v.y + mag_eps);
        magl[k] = m;
        const size_t o = ((size_t)b * bins + k) * F + f;
        if (mag) mag[o] = m;
        if (re_out) re_out[o] = v.x;
        if (im_out) im_out[o] = v.y;
    }
    if (!mel) return;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int m = wave; m < n_mel; m += 4) {
        const float* row = melbasis + (size_t)m * bins;
        float acc = 0.f;
        for (int k = lane; k < bins; k += 64) acc = fmaf(row[k], magl[k], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) {
            float v = acc;
            if (log_clip > 0.f) v = logf(fmaxf(v, log_clip));
            mel[((size_t)b * n_mel + m) * F + f] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Any other n_fft the reference accepts (round 5): torch.stft takes every length (utils/mel.py:145-169) and the reference ships
// n_fft = 1920 = 2^7 * 3 * 5 (egs/vocoder/vocos/emilia_singnet.json:15).  Mixed-radix Stockham autosort FFT in LDS, one workgroup per
// frame, complex n_fft-point transform: pass with radix r over Ns (the product of the radices done),
//     j in [0, N / r):  k = j mod Ns;  v_q = in[j + q N / r] * W_N^{q k N / (Ns r)};  u_p = sum_q v_q W_r^{p q};  out[(j - k) r + k + p Ns] = u_p
// with the whole unit circle W_N^m in LDS (one sincospif per entry and frame), radices = the prime factors of n_fft, twos last (compile-time
// butterflies for 2 .. 13, generic_pass below for larger primes).
// Window, reflect padding, |X|, mel projection, log: mel_kernel's.  A fallback, not a fast path: the frame-rate front end is 0.15 % of a
// vocoder step, and every config the reference trains with is 1 024 (mel1024_kernel).
// ------------------------------------------------------------------------------------------------
struct MelRadices {
    int n;          // passes
    int r[16];      // radix of each pass
};

template <int R>
__device__ __forceinline__ void mixed_pass(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw, int N, int Ns, int tid) {
    const int M = N / R;                 // butterflies
    const int tstep = N / (Ns * R);      // twiddle index step of this pass
    const int rstep = N / R;             // W_R = W_N^{N / R}
    for (int j = tid; j < M; j += 256) {
        const int k = j % Ns;
        float2 v[R];
#pragma unroll
        for (int q = 0; q < R; ++q) {
            const float2 x = in[j + q * M];
            const float2 w = tw[(q * k * tstep) % N];
            v[q] = make_float2(x.x * w.x - x.y * w.y, x.x * w.y + x.y * w.x);
        }
        const int j0 = (j - k) * R + k;
#pragma unroll
        for (int pq = 0; pq < R; ++pq) {
            float2 u = v[0];
#pragma unroll
            for (int q = 1; q < R; ++q) {
                const float2 w = tw[((pq * q) % R) * rstep];
                u.x += v[q].x * w.x - v[q].y * w.y;
                u.y += v[q].x * w.y + v[q].y * w.x;
            }
            out[j0 + pq * Ns] = u;
        }
    }
}

// The same pass for a radix known only at run time (prime factors above 13; round 5): one OUTPUT per thread and step -- output p of butterfly j is
// sum_q in[j + q M] W_N^{q k tstep + (p q mod R) rstep} -- R complex MACs each, N R per pass (a prime n_fft is one pass of N^2: a direct DFT).  Slow and
// simple: it exists so that no n_fft in [64, 4096] is refused, as torch.stft refuses none.
__device__ __forceinline__ void generic_pass(const float2* __restrict__ in, float2* __restrict__ out, const float2* __restrict__ tw, int N, int Ns, int R, int tid) {
    const int M = N / R;
    const int tstep = N / (Ns * R);
    const int rstep = N / R;
    for (int o = tid; o < N; o += 256) {
        const int pq = o / M, j = o - pq * M;       // consecutive threads: consecutive butterflies
        const int k = j % Ns;
        const int kt = k * tstep;                    // q * kt < N for every q < R
        float2 u = make_float2(0.f, 0.f);
        int pm = 0;                                  // (pq * q) mod R, incrementally
        for (int q = 0; q < R; ++q) {
            int m = q * kt + pm * rstep;             // < 2 N
            m = m >= N ? m - N : m;
            const float2 x = in[j + q * M];
            const float2 w = tw[m];
            u.x += x.x * w.x - x.y * w.y;
            u.y += x.x * w.y + x.y * w.x;
            pm += pq; pm = pm >= R ? pm - R : pm;
        }
        out[(j - k) * R + k + pq * Ns] = u;
    }
}

__global__ __launch_bounds__(256) void mel_mixed_kernel(const float* __restrict__ wav, const int* __restrict__ lens, int L, int F,
                                                        int n_fft, const MelRadices rad, int hop, int pad, int n_mel, float mag_eps,
                                                        float log_clip, const float* __restrict__ window,
                                                        const float* __restrict__ melbasis, float* __restrict__ mel,
                                                        float* __restrict__ mag, float* __restrict__ re_out, float* __restrict__ im_out,
                                                        const MelRange rng) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    mel_range_begin(rng);
    float2* buf0 = reinterpret_cast<float2*>(smem);   // [n_fft]
    float2* buf1 = buf0 + n_fft;                      // [n_fft]
    float2* tw = buf1 + n_fft;                        // [n_fft]  exp(-2 pi i m / n_fft), the whole circle
    float* magl = reinterpret_cast<float*>(tw + n_fft);   // [n_fft / 2 + 1]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / F;
    const int f = blockIdx.x - b * F;
    const int bins = (n_fft >> 1) + 1;
    const float* wb = wav + (size_t)b * L;
    int Li = L;
    if (lens) { Li = lens[b] < L ? lens[b] : L; if (f > (Li + 2 * pad - n_fft) / hop || Li <= pad) return; }

    float rlo = 0.f, rhi = 0.f;
    for (int n = tid; n < n_fft; n += 256) {
        const int s = reflect_index(f * hop + n - pad, Li);
        const float xv = wb[s];
        rlo = fminf(rlo, xv); rhi = fmaxf(rhi, xv);
        buf0[n] = make_float2(xv * window[n], 0.f);
        float sn, cs;
        sincospif(-2.0f * (float)n / (float)n_fft, &sn, &cs);
        tw[n] = make_float2(cs, sn);
    }
    mel_range_end(rng, rlo, rhi);
    __syncthreads();

    float2* in = buf0;
    float2* out = buf1;
    int Ns = 1;
    for (int s = 0; s < rad.n; ++s) {
        const int r = rad.r[s];
        switch (r) {
            case 2: mixed_pass<2>(in, out, tw, n_fft, Ns, tid); break;
            case 3: mixed_pass<3>(in, out, tw, n_fft, Ns, tid); break;
            case 5: mixed_pass<5>(in, out, tw, n_fft, Ns, tid); break;
            case 7: mixed_pass<7>(in, out, tw, n_fft, Ns, tid); break;
            case 11: mixed_pass<11>(in, out, tw, n_fft, Ns, tid); break;
            case 13: mixed_pass<13>(in, out, tw, n_fft, Ns, tid); break;
            default: generic_pass(in, out, tw, n_fft, Ns, r, tid); break;
        }
        Ns *= r;
        __syncthreads();
        float2* t = in; in = out; out = t;
    }
    for (int k = tid; k < bins; k += 256) {
        const float2 v = in[k];
        const float m = sqrtf(v.x * v.x + v.y * v.y + mag_eps);
        magl[k] = m;
        const size_t o = ((size_t)b * bins + k) * F + f;
        if (mag) mag[o] = m;
        if (re_out) re_out[o] = v.x;
        if (im_out) im_out[o] = v.y;
    }
    if (!mel) return;
    __syncthreads();
    const int lane = tid & 63, wave = tid >> 6;
    for (int m = wave; m < n_mel; m += 4) {
        const float* row = melbasis + (size_t)m * bins;
        float acc = 0.f;
        for (int k = lane; k < bins; k += 64) acc = fmaf(row[k], magl[k], acc);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
        if (lane == 0) {
            float v = acc;
            if (log_clip > 0.f) v = logf(fmaxf(v, log_clip));
            mel[((size_t)b * n_mel + m) * F + f] = v;
        }
    }
}

// prime factors of n: those above 13 first (run-time radix passes), then 13 .. 3, the twos last
static bool mel_radices(int n, MelRadices* out) {
    out->n = 0;
    if (n < 1) return false;
    int small = n;                              // strip the small primes to find what is left
    for (int p : {2, 3, 5, 7, 11, 13}) while (small % p == 0) small /= p;
    for (int p = 17; small > 1; p += 2) {
        if ((long long)p * p > small) p = small;             // what is left is prime
        while (small % p == 0) {
            if (out->n >= 16) return false;
            out->r[out->n++] = p;
            small /= p;
            n /= p;
        }
    }
    const int primes[6] = {13, 11, 7, 5, 3, 2};
    for (int p : primes)
        while (n % p == 0) {
            if (out->n >= 16) return false;
            out->r[out->n++] = p;
            n /= p;
        }
    return n == 1;
}
// THE gate of the four entry points (one check, one message).  Every length in the range runs: smooth ones on compile-time butterflies,
// a prime factor p > 13 as a run-time radix pass of N p complex MACs per frame -- a PRIME n_fft is one pass of N^2 (n_fft = 4093: 16.7 M MACs per frame in one
// 256-thread workgroup, three orders of magnitude slower per sample than 1024; documented in amphion_hip.h).
bool mel_nfft_supported(int n_fft) {
    MelRadices r;
    return n_fft >= 64 && n_fft <= 4096 && mel_radices(n_fft, &r);
}
static bool mel_nfft_check(const char* who, int n_fft) {
    if (mel_nfft_supported(n_fft)) return true;
    set_error("%s: n_fft=%d must lie in [64, 4096] (torch.stft semantics for every length in that range)", who, n_fft);
    return false;
}

// ------------------------------------------------------------------------------------------------
// n_fft = 1024 (every 22.05 / 24 kHz config of the reference: config/fs2.json:25-31, config/vocoder.json:34-40,
// config/vits.json): one WAVE per frame, radix-8, real-input FFT, nothing recomputed per frame.
//
//   z[m] = x[2m] + i x[2m+1]  (m < 512)  ->  512-point complex FFT as 8 x 8 x 8 Stockham passes: lane j owns the
//   radix-8 butterfly j of each pass (8 complex values in registers); between passes the wave exchanges through its
//   own 4 KB of LDS (no workgroup barrier: LDS operations of one wave execute in order)  ->  real-FFT split
//   X[k] = E - i W^k O with the partner Z[512 - k] fetched by a lane permute  ->  |X| -> LDS  ->  mel filterbank
//   over each filter's non-zero band  ->  log  ->  LDS tile [mel][32 frames]  ->  128-B row stores.
//
// The twiddles of passes 1 / 2 and of the split depend only on (lane, q): read from a 4.5-KB device table (g_mel_tw, filled
// from the host with correctly rounded values at the first launch on each device -- 15 sincospif calls per lane were a fifth
// of the kernel's VALU work), per frame, in the same batch of loads as the frame's samples.  A workgroup is
// 8 waves x 4 frames = 32 consecutive frames of one utterance: the samples are read straight from global memory
// (each lane 2 consecutive samples per slot = one 512-B wave access; the 4x frame overlap is served by L1 / L2).
// mag / re / im outputs (extract_linear_features, amplitude_phase_spectrum) are written per frame at stride F.
// ------------------------------------------------------------------------------------------------
struct __attribute__((packed, aligned(4))) Float2U { float2 v; };   // a sample pair: 4-byte aligned (odd utterance offsets)

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-place 8-point DFT (forward, e^{-2 pi i / 8}), natural order in, natural order out
__device__ __forceinline__ void dft8(float2 (&v)[8]) {
    const float h = 0.70710678118654752440f;
    // stage 1: pairs (q, q + 4)
    float2 a0 = make_float2(v[0].x + v[4].x, v[0].y + v[4].y), a4 = make_float2(v[0].x - v[4].x, v[0].y - v[4].y);
    float2 a1 = make_float2(v[1].x + v[5].x, v[1].y + v[5].y), a5 = make_float2(v[1].x - v[5].x, v[1].y - v[5].y);
    float2 a2 = make_float2(v[2].x + v[6].x, v[2].y + v[6].y), a6 = make_float2(v[2].x - v[6].x, v[2].y - v[6].y);
    float2 a3 = make_float2(v[3].x + v[7].x, v[3].y + v[7].y), a7 = make_float2(v[3].x - v[7].x, v[3].y - v[7].y);
    // twiddles of the odd half: a5 *= w8, a6 *= -i, a7 *= w8^3
    a5 = make_float2((a5.x + a5.y) * h, (a5.y - a5.x) * h);
    a6 = make_float2(a6.y, -a6.x);
    a7 = make_float2((a7.y - a7.x) * h, -(a7.x + a7.y) * h);
    // stage 2: 4-point DFTs of (a0, a1, a2, a3) -> even outputs, (a4, a5, a6, a7) -> odd outputs
    float2 b0 = make_float2(a0.x + a2.x, a0.y + a2.y), b2 = make_float2(a0.x - a2.x, a0.y - a2.y);
    float2 b1 = make_float2(a1.x + a3.x, a1.y + a3.y), b3 = make_float2(a1.x - a3.x, a1.y - a3.y);
    b3 = make_float2(b3.y, -b3.x);
    float2 c0 = make_float2(a4.x + a6.x, a4.y + a6.y), c2 = make_float2(a4.x - a6.x, a4.y - a6.y);
    float2 c1 = make_float2(a5.x + a7.x, a5.y + a7.y), c3 = make_float2(a5.x - a7.x, a5.y - a7.y);
    c3 = make_float2(c3.y, -c3.x);
    v[0] = make_float2(b0.x + b1.x, b0.y + b1.y);
    v[4] = make_float2(b0.x - b1.x, b0.y - b1.y);
    v[2] = make_float2(b2.x + b3.x, b2.y + b3.y);
    v[6] = make_float2(b2.x - b3.x, b2.y - b3.y);
    v[1] = make_float2(c0.x + c1.x, c0.y + c1.y);
    v[5] = make_float2(c0.x - c1.x, c0.y - c1.y);
    v[3] = make_float2(c2.x + c3.x, c2.y + c3.y);
    v[7] = make_float2(c2.x - c3.x, c2.y - c3.y);
}

// cos / sin of pi q / 8, q = 0..7: exp(-2 pi i q / 16) of the real-FFT split
__device__ constexpr float kC16[8] = {1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f,
                                      0.f, -0.38268343236508977f, -0.70710678118654752f, -0.92387953251128674f};
__device__ constexpr float kS16[8] = {0.f, 0.38268343236508977f, 0.70710678118654752f, 0.92387953251128674f,
                                      1.f, 0.92387953251128674f, 0.70710678118654752f, 0.38268343236508977f};

// twiddle table of mel1024_kernel: tw1[q - 1][k] = exp(-2 pi i q k / 64), tw2[q - 1][j] = exp(-2 pi i q j / 512),
// wsp[j] = exp(-2 pi i j / 1024)
struct MelTwiddles {
    float2 tw1[7][8];
    float2 tw2[7][64];
    float2 wsp[64];
};
__device__ MelTwiddles g_mel_tw;

// exp(-2 pi i n / N) rounded to fp32 from long double; the multiples of an eighth turn come out exact
static float2 unit_root(int n, int N) {
    n %= N;
    if ((8 * n) % N == 0) {
        const float h = 0.70710678118654752440f;
        const float c[8] = {1.f, h, 0.f, -h, -1.f, -h, 0.f, h};
        const float sn[8] = {0.f, -h, -1.f, -h, 0.f, h, 1.f, h};
        return make_float2(c[8 * n / N], sn[8 * n / N]);
    }
    const long double a = -2.0L * 3.14159265358979323846264338327950288L * (long double)n / (long double)N;
    return make_float2((float)cosl(a), (float)sinl(a));
}

static hipError_t upload_mel_twiddles() {
    MelTwiddles t;
    for (int q = 1; q < 8; ++q) {
        for (int k = 0; k < 8; ++k) t.tw1[q - 1][k] = unit_root(q * k, 64);
        for (int j = 0; j < 64; ++j) t.tw2[q - 1][j] = unit_root(q * j, 512);
    }
    for (int j = 0; j < 64; ++j) t.wsp[j] = unit_root(j, 1024);
    return hipMemcpyToSymbol(HIP_SYMBOL(g_mel_tw), &t, sizeof(t), 0, hipMemcpyHostToDevice);
}

// One-time set-up of the n_fft = 1024 kernel on the CURRENT device (its LDS attribute and the 4.5-KB twiddle table), under a lock:
// amp_mel_init() does it explicitly; the first launch does it lazily -- but never inside a stream capture, where the blocking copy
// would invalidate the capture (hipErrorStreamCaptureUnsupported is returned instead and the caller is told to call amp_mel_init).
static std::mutex g_mel_init_mutex;
static std::atomic<unsigned long long> g_mel_init_done{0};   // bit per device
static hipError_t mel1024_device_init(hipStream_t stream_or_null, bool have_stream);

constexpr int MEL_WAVES = 8;       // waves per workgroup
constexpr int MEL_FPW = 4;         // frames per wave
constexpr int MEL_FPB = MEL_WAVES * MEL_FPW;   // 32 frames per workgroup: one 128-B row of every mel channel
constexpr int MEL_XROW = 10;       // exchange-buffer row: 8 complex + 2 pad (80 B: the 16-B stores of pass 0 stay aligned)
constexpr int MEL_MAGROW = 520;    // 513 bins, padded
constexpr int MEL_MAXMEL = 256;
constexpr int MEL_WCAP = 1664;     // packed filter weights kept in LDS, every band padded to a multiple of 8 (triangular filters
                                   // on 513 bins hold <= ~1030 non-zeros; + <= 7 per filter: 1600 for 128 filters)

// LDS floats of mel1024_kernel for n_mel filters: packed weights, magnitude rows, exchange buffers, mel tile, band tables
// ([n_mel + 1] + [n_mel] 16-bit entries + the 32-bit total)
constexpr size_t mel1024_lds_floats(int n_mel) {
    return (size_t)(MEL_WCAP + 4) + (size_t)(MEL_WAVES * MEL_MAGROW + 2 * MEL_WAVES * 64 * MEL_XROW)
           + (size_t)n_mel * (MEL_FPB + 1) + (size_t)(1 + (2 * n_mel + 2) / 2);
}

__global__ __launch_bounds__(64 * MEL_WAVES, 4) void mel1024_kernel(const float* __restrict__ wav, const int* __restrict__ lens,
                                                                  int L, int F, int hop, int pad, int n_mel, float mag_eps,
                                                                  float log_clip, const float* __restrict__ window,
                                                                  const float* __restrict__ melbasis,
                                                                  const int* __restrict__ bands, float* __restrict__ mel,
                                                                  float* __restrict__ mag, float* __restrict__ re_out,
                                                                  float* __restrict__ im_out, const MelRange rng) {
    constexpr int N = 1024, M = 512, BINS = 513;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    mel_range_begin(rng);
    float* const wl = smem;                                                   // [MEL_WCAP + 4] packed filter weights
    float* const magl = wl + MEL_WCAP + 4;                                    // [MEL_WAVES][MEL_MAGROW]
    float2* const xch = reinterpret_cast<float2*>(magl + MEL_WAVES * MEL_MAGROW);   // [MEL_WAVES][64 * MEL_XROW]
    float* const melt = magl + MEL_WAVES * MEL_MAGROW + 2 * MEL_WAVES * 64 * MEL_XROW;   // [n_mel][MEL_FPB + 1]
    const int tid = threadIdx.x;
    const int j = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fblocks = (F + MEL_FPB - 1) / MEL_FPB;
    const int b = blockIdx.x / fblocks;
    const int f0 = (blockIdx.x - b * fblocks) * MEL_FPB;
    const float* wb = wav + (size_t)b * L;
    // ragged batch: item b holds lens[b] samples; its reflection padding mirrors at ITS end and it has
    // (lens[b] + 2*pad - n_fft) / hop + 1 frames -- later frames of the row are left untouched
    int Li = L, Fi = F;
    if (lens) {
        Li = lens[b] < L ? lens[b] : L;
        Fi = Li <= pad ? 0 : (Li + 2 * pad - N) / hop + 1;
        Fi = Fi < F ? Fi : F;
    }
    if (f0 >= Fi) return;   // workgroup-uniform

    // mel projection tables: the non-zero band of every filter is copied ONCE per workgroup into LDS, filter after filter, each
    // padded with zeros to a multiple of eight (wl[offs[m] + i] = basis[m][lo_m + i]), so the per-frame projection reads LDS
    // instead of one cache line per lane and step from L2.  Same products in the same order (k ascending inside the band) as
    // the loop over the global rows it replaces; the padding adds 0 * (a finite magnitude) = nothing.
    int* const total_p = reinterpret_cast<int*>(melt + n_mel * (MEL_FPB + 1));  // sum of the padded widths
    unsigned short* const offs = reinterpret_cast<unsigned short*>(total_p + 1);  // [n_mel + 1] exclusive prefix of padded widths
    unsigned short* const los = offs + n_mel + 1;                                 // [n_mel] first bin of each band
    bool packed = false;                                                          // workgroup-uniform
    if (mel && bands) {
        if (w == 0) {                                // one wave: exclusive scan over the widths, four filters per lane
            int wd[4], sum = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = 4 * j + t;
                wd[t] = 0;
                if (m < n_mel) {
                    const int lo = bands[2 * m], hi = bands[2 * m + 1];
                    wd[t] = hi > lo ? (hi - lo + 7) & ~7 : 0;
                    los[m] = (unsigned short)lo;
                }
                sum += wd[t];
            }
            int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if (j >= d) incl += o;
            }
            int base = incl - sum;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = 4 * j + t;
                if (m <= n_mel) offs[m] = (unsigned short)base;       // (wraps only when the total exceeds MEL_WCAP: unused then)
                base += wd[t];
            }
            if (j == 63) { *total_p = incl; if (n_mel == MEL_MAXMEL) offs[n_mel] = (unsigned short)incl; }
        }
        if (j < MEL_MAGROW - BINS) magl[w * MEL_MAGROW + BINS + j] = 0.f;      // the row's pad: read under a zero weight
        __syncthreads();
        const int total = *total_p;
        packed = total <= MEL_WCAP;
        if (packed) {
            for (int e = tid; e < total; e += 64 * MEL_WAVES) {
                int ml = 0, mh = n_mel;              // offs[ml] <= e < offs[mh]
                while (mh - ml > 1) {
                    const int mid = (ml + mh) >> 1;
                    if ((int)offs[mid] <= e) ml = mid; else mh = mid;
                }
                const int k = (int)los[ml] + (e - (int)offs[ml]);
                wl[e] = k < bands[2 * ml + 1] ? melbasis[(size_t)ml * BINS + k] : 0.f;
            }
        }
        __syncthreads();
    }

    // lane constants: the window slots and the twiddles are re-read every frame with the frame's samples (L1 hits, one
    // batch of loads) -- held in registers across the frames they cost 44 VGPRs and the compiler spilled them
    const float2* win = reinterpret_cast<const float2*>(window) + j;
    const int k1 = j & 7;
    const float2 wsp = g_mel_tw.wsp[j];                           // split: exp(-2 pi i j / 1024)
    float2* xw = xch + w * (64 * MEL_XROW);
    float* mg = magl + w * MEL_MAGROW;
    float rlo = 0.f, rhi = 0.f;                    // extreme samples this lane read (utils/mel.py:21-24, mel_range_end)

    for (int fi = 0; fi < MEL_FPW; ++fi) {
        const int fl = w * MEL_FPW + fi;            // frame inside the workgroup tile
        const int f = f0 + fl;
        if (f >= Fi) break;                         // wave-uniform
        float2 v[8];
        const int s0 = f * hop - pad + 2 * j;
        // all 30 loads of the frame (window slots, twiddles, samples: L1 hits but for the samples) are issued before the
        // first use.  The asm statements are the fence: left alone, the compiler saves registers by waiting for each sample
        // pair in turn -- eight memory round trips per frame -- and one fence per branch keeps it from merging the two load
        // sequences into sixteen scalar loads.
        float2 wq[8], tw1[8], tw2[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) wq[q] = win[64 * q];
#pragma unroll
        for (int q = 1; q < 8; ++q) {
            tw1[q] = g_mel_tw.tw1[q - 1][k1];                     // pass 1: exp(-2 pi i q k / 64),  k = j mod 8
            tw2[q] = g_mel_tw.tw2[q - 1][j];                      // pass 2: exp(-2 pi i q j / 512)
        }
        if (s0 - 2 * j >= 0 && f * hop - pad + N <= Li) {          // interior frame (wave-uniform): no reflection
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = reinterpret_cast<const Float2U*>(wb + s0 + 128 * q)->v;   // one 8-byte load
            asm volatile("; interior frame: loads issued" ::: "memory");
        } else {
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int s = s0 + 128 * q;
                v[q] = make_float2(wb[reflect_index(s, Li)], wb[reflect_index(s + 1, Li)]);
            }
            asm volatile("; edge frame: loads issued" ::: "memory");
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            rlo = fminf(rlo, fminf(v[q].x, v[q].y)); rhi = fmaxf(rhi, fmaxf(v[q].x, v[q].y));
            v[q] = make_float2(__fmul_rn(v[q].x, wq[q].x), __fmul_rn(v[q].y, wq[q].y));   // (rounded: never fused into pass 0's adds)
        }
        // pass 0 (Ns = 1): no twiddles; out[8 j + q]
        dft8(v);
        {
            float4* row = reinterpret_cast<float4*>(xw + j * MEL_XROW);
#pragma unroll
            for (int q = 0; q < 4; ++q) row[q] = make_float4(v[2 * q].x, v[2 * q].y, v[2 * q + 1].x, v[2 * q + 1].y);
        }
        __builtin_amdgcn_wave_barrier();
        // pass 1 (Ns = 8): in[j + 64 q] -> twiddle exp(-2 pi i q k / 64) -> out[8 (j - k) + k + 8 q]
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = j + 64 * q;
            v[q] = xw[(i >> 3) * MEL_XROW + (i & 7)];
        }
#pragma unroll
        for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw1[q]);
        dft8(v);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = 8 * (j - k1) + k1 + 8 * q;
            xw[(i >> 3) * MEL_XROW + (i & 7)] = v[q];
        }
        __builtin_amdgcn_wave_barrier();
        // pass 2 (Ns = 64): in[j + 64 q] -> twiddle exp(-2 pi i q j / 512) -> Z[j + 64 q]
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = j + 64 * q;
            v[q] = xw[(i >> 3) * MEL_XROW + (i & 7)];
        }
#pragma unroll
        for (int q = 1; q < 8; ++q) v[q] = cmul(v[q], tw2[q]);
        dft8(v);
        __builtin_amdgcn_wave_barrier();
        // real-FFT split, k = j + 64 q: partner Z[512 - k] sits in lane (64 - j) & 63, register 7 - q (lane 0: (8 - q) & 7)
        const int src = (64 - j) & 63;
        const size_t ob = ((size_t)b * BINS) * F + f;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float2 zp = make_float2(__shfl(v[7 - q].x, src, 64), __shfl(v[7 - q].y, src, 64));
            if (j == 0) zp = v[(8 - q) & 7];
            const float2 z = v[q];
            const float2 e = make_float2(0.5f * (z.x + zp.x), 0.5f * (z.y - zp.y));       // (Z + conj Zp) / 2
            const float2 o = make_float2(0.5f * (z.x - zp.x), 0.5f * (z.y + zp.y));       // (Z - conj Zp) / 2
            // W^k = wsp * exp(-2 pi i q / 16)
            const float2 wk = cmul(wsp, make_float2(kC16[q], -kS16[q]));
            const float2 t = cmul(wk, o);                                                 // X = E - i W^k O
            const float2 x = make_float2(e.x + t.y, e.y - t.x);
            const int k = j + 64 * q;
            const float m = sqrtf(x.x * x.x + x.y * x.y + mag_eps);
            mg[k] = m;
            if (mag) mag[ob + (size_t)k * F] = m;
            if (re_out) re_out[ob + (size_t)k * F] = x.x;
            if (im_out) im_out[ob + (size_t)k * F] = x.y;
            if (q == 0 && j == 0) {                                                       // Nyquist bin from Z[0]
                const float xn = z.x - z.y;
                const float mn = sqrtf(xn * xn + mag_eps);
                mg[M] = mn;
                if (mag) mag[ob + (size_t)M * F] = mn;
                if (re_out) re_out[ob + (size_t)M * F] = xn;
                if (im_out) im_out[ob + (size_t)M * F] = 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (mel && packed) {
            // a lane runs filter m, then filter m + 64, as ONE stream of eight-step trips (the narrow low filters share a lane
            // with the wide high ones, and a lane that is through leaves the loop: no reads past a band)
            for (int m0 = j; m0 < n_mel; m0 += 128) {
                const int m1 = m0 + 64;
                const int oa = offs[m0], na = (int)offs[m0 + 1] - oa;
                int ob = 0, nb = 0, lb = 0;
                if (m1 < n_mel) { ob = offs[m1]; nb = (int)offs[m1 + 1] - ob; lb = los[m1]; }
                const int nt = na + nb;
                const float* pw = wl + oa;
                const float* pg = mg + (int)los[m0];
                float acc = 0.f, acca = 0.f;
                for (int i = 0; i < nt; i += 8) {
                    if (i == na) { acca = acc; acc = 0.f; pw = wl + ob; pg = mg + lb; }     // first filter done
                    const float4 w4 = *reinterpret_cast<const float4*>(pw), w8 = *reinterpret_cast<const float4*>(pw + 4);
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = pg[u];
                    acc = fmaf(w4.x, x[0], acc); acc = fmaf(w4.y, x[1], acc); acc = fmaf(w4.z, x[2], acc); acc = fmaf(w4.w, x[3], acc);
                    acc = fmaf(w8.x, x[4], acc); acc = fmaf(w8.y, x[5], acc); acc = fmaf(w8.z, x[6], acc); acc = fmaf(w8.w, x[7], acc);
                    pw += 8; pg += 8;
                }
                float accb = 0.f;
                if (nb > 0) accb = acc; else acca = acc;
                if (log_clip > 0.f) { acca = logf(fmaxf(acca, log_clip)); accb = logf(fmaxf(accb, log_clip)); }
                melt[m0 * (MEL_FPB + 1) + fl] = acca;
                if (m1 < n_mel) melt[m1 * (MEL_FPB + 1) + fl] = accb;
            }
        } else if (mel) {
            for (int m = j; m < n_mel; m += 64) {
                const int lo = bands ? bands[2 * m] : 0;
                const int hi = bands ? bands[2 * m + 1] : BINS;
                const float* row = melbasis + (size_t)m * BINS;
                float acc = 0.f;
                for (int k = lo; k < hi; ++k) acc = fmaf(row[k], mg[k], acc);
                if (log_clip > 0.f) acc = logf(fmaxf(acc, log_clip));
                melt[m * (MEL_FPB + 1) + fl] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    mel_range_end(rng, rlo, rhi);
    if (!mel) return;
    __syncthreads();
    // melt [n_mel][32 frames] -> mel[b][m][f0 .. f0 + 32): one 128-B segment per channel
    const int nfr = (Fi - f0) < MEL_FPB ? (Fi - f0) : MEL_FPB;
    for (int idx = tid; idx < n_mel * MEL_FPB; idx += 64 * MEL_WAVES) {
        const int m = idx >> 5, fl = idx & 31;
        if (fl < nfr) mel[((size_t)b * n_mel + m) * F + f0 + fl] = melt[m * (MEL_FPB + 1) + fl];
    }
}

static hipError_t mel1024_device_init(hipStream_t stream, bool have_stream) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if ((g_mel_init_done.load(std::memory_order_acquire) >> dev) & 1ull) return hipSuccess;
    if (have_stream) {
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            set_error("amp_mel_forward: the first n_fft = 1024 call on a device uploads its twiddle table with a blocking copy; "
                      "call amp_mel_init() (or one mel forward) before capturing the stream");
            return hipErrorStreamCaptureUnsupported;
        }
    }
    std::lock_guard<std::mutex> lock(g_mel_init_mutex);
    if ((g_mel_init_done.load(std::memory_order_acquire) >> dev) & 1ull) return hipSuccess;
    const size_t mx = mel1024_lds_floats(MEL_MAXMEL) * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&mel1024_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)mx);
    if (e != hipSuccess) return e;
    e = upload_mel_twiddles();          // pageable copy: returns when the table is on the device
    if (e != hipSuccess) return e;
    e = hipDeviceSynchronize();         // ... and is ordered before work on EVERY stream of the device, blocking or not
    if (e != hipSuccess) return e;
    g_mel_init_done.fetch_or(1ull << dev, std::memory_order_release);
    return hipSuccess;
}

// ------------------------------------------------------------------------------------------------
// Wave-per-frame front end for n_fft = 128 P with P = 16 | 15 | 4: n_fft = 2048, 512 and 1920 = 2^7 * 3 * 5 -- the n_fft of 24 of the reference's 38 JSON
// configs (egs/vocoder/vocos/emilia_singnet.json:15; utils/mel.py:145-169) -- in mel1024_kernel's scheme (round 6; mel_mixed_kernel, one WORKGROUP
// per frame with every pass through LDS, took 0.99 / 0.86 ms for 64 x 65 536 samples where 1024 takes 0.041):
//   one WAVE per frame; the real frame as M = 64 P complex points z[n] = x[2n] + i x[2n + 1]; lane j loads z[j + 64 q], q < P, and runs pass 0 -- a
//   radix-P butterfly (16 = 2 x 8; 15 = 3 x 5) -- in registers; two radix-8 passes (8 P butterflies: two per lane) through the wave's own LDS
//   exchange buffer (Stockham autosort: the result arrives in natural order); the real-FFT split X[k] = E[k] - i W^k O[k] from Z[k] and Z[M - k];
//   magnitudes into the wave's LDS row; the band-limited mel projection from packed filter rows in LDS; 32-frame workgroups write whole 128-B
//   segments of every mel channel.  Twiddles: ONE table exp(-2 pi i n / n_fft) per workgroup in LDS, built from sincospi in double precision.
// Same semantics as the other kernels (torch.stft, center = False after the reflection padding): golden vectors of the REAL reference at both
// lengths (tests/golden/golden_nfft.npz) and the oracle.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

// 3-point DFT (forward), natural order
__device__ __forceinline__ void dft3(float2& a0, float2& a1, float2& a2) {
    const float s = 0.86602540378443864676f;
    const float2 t = cadd(a1, a2), d = csub(a1, a2);
    const float2 m = make_float2(a0.x - 0.5f * t.x, a0.y - 0.5f * t.y);
    a0 = cadd(a0, t);
    a1 = make_float2(m.x + s * d.y, m.y - s * d.x);      // m - i s d
    a2 = make_float2(m.x - s * d.y, m.y + s * d.x);      // m + i s d
}

// 5-point DFT (forward), natural order
__device__ __forceinline__ void dft5(float2& a0, float2& a1, float2& a2, float2& a3, float2& a4) {
    const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;     // cos(2 pi / 5), cos(4 pi / 5)
    const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;      // sin(2 pi / 5), sin(4 pi / 5)
    const float2 t1 = cadd(a1, a4), t2 = cadd(a2, a3), t3 = csub(a1, a4), t4 = csub(a2, a3);
    const float2 p1 = make_float2(a0.x + c1 * t1.x + c2 * t2.x, a0.y + c1 * t1.y + c2 * t2.y);
    const float2 p2 = make_float2(a0.x + c2 * t1.x + c1 * t2.x, a0.y + c2 * t1.y + c1 * t2.y);
    const float2 q1 = make_float2(s1 * t3.x + s2 * t4.x, s1 * t3.y + s2 * t4.y);
    const float2 q2 = make_float2(s2 * t3.x - s1 * t4.x, s2 * t3.y - s1 * t4.y);
    a0 = cadd(a0, cadd(t1, t2));
    a1 = make_float2(p1.x + q1.y, p1.y - q1.x);          // p1 - i q1
    a4 = make_float2(p1.x - q1.y, p1.y + q1.x);          // p1 + i q1
    a2 = make_float2(p2.x + q2.y, p2.y - q2.x);
    a3 = make_float2(p2.x - q2.y, p2.y + q2.x);
}

// in-place P-point DFT of v[0 .. P), natural order in and out: P = 16 as two 8-point DFTs + one radix-2 stage, P = 15 as 3 x 5 (Cooley-Tukey)
template <int P>
__device__ __forceinline__ void dftP(float2 (&v)[P]);

template <>
__device__ __forceinline__ void dftP<4>(float2 (&v)[4]) {
    const float2 a0 = cadd(v[0], v[2]), a1 = csub(v[0], v[2]), a2 = cadd(v[1], v[3]), a3 = csub(v[1], v[3]);
    v[0] = cadd(a0, a2);
    v[2] = csub(a0, a2);
    v[1] = make_float2(a1.x + a3.y, a1.y - a3.x);       // a1 - i a3
    v[3] = make_float2(a1.x - a3.y, a1.y + a3.x);       // a1 + i a3
}

template <>
__device__ __forceinline__ void dftP<8>(float2 (&v)[8]) { dft8(v); }

template <>
__device__ __forceinline__ void dftP<16>(float2 (&v)[16]) {
    float2 e[8], o[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { e[q] = v[2 * q]; o[q] = v[2 * q + 1]; }
    dft8(e);
    dft8(o);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float2 t = cmul(o[k], make_float2(kC16[k], -kS16[k]));   // exp(-2 pi i k / 16)
        v[k] = cadd(e[k], t);
        v[k + 8] = csub(e[k], t);
    }
}

template <>
__device__ __forceinline__ void dftP<15>(float2 (&v)[15]) {
    // n = 5 n1 + n2, k = k1 + 3 k2:  X[k1 + 3 k2] = sum_n2 w5^(n2 k2) [ w15^(n2 k1) sum_n1 w3^(n1 k1) x[5 n1 + n2] ]
    float2 b[5][3];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
        b[n2][0] = v[n2]; b[n2][1] = v[5 + n2]; b[n2][2] = v[10 + n2];
        dft3(b[n2][0], b[n2][1], b[n2][2]);
    }
    // exp(-2 pi i m / 15), m = n2 * k1
    constexpr float c15[9] = {1.f, 0.91354545764260089550f, 0.66913060635885821383f, 0.30901699437494742410f, -0.10452846326765347140f,
                              -0.5f, -0.80901699437494742410f, -0.97814760073380563793f, -0.97814760073380563793f};
    constexpr float s15[9] = {0.f, 0.40673664307580020775f, 0.74314482547739423501f, 0.95105651629515357212f, 0.99452189536827333692f,
                              0.86602540378443864676f, 0.58778525229247312917f, 0.20791169081775933710f, -0.20791169081775933710f};
#pragma unroll
    for (int n2 = 1; n2 < 5; ++n2)
#pragma unroll
        for (int k1 = 1; k1 < 3; ++k1) {
            const int m = n2 * k1;
            b[n2][k1] = cmul(b[n2][k1], make_float2(c15[m], -s15[m]));
        }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        dft5(b[0][k1], b[1][k1], b[2][k1], b[3][k1], b[4][k1]);
#pragma unroll
        for (int k2 = 0; k2 < 5; ++k2) v[k1 + 3 * k2] = b[k2][k1];
    }
}

constexpr int MELW_WCAP = 4096;    // packed filter weights in LDS (every band padded to a multiple of 8)
template <int P>
struct MelW {
    static constexpr int M = 64 * P, N = 2 * M, BINS = M + 1;
    static constexpr int PSH = P == 16 ? 4 : P == 8 ? 3 : P == 4 ? 2 : 31;   // a power-of-two P pads one slot in P: pass 0 writes at stride P (bank conflicts)
    static constexpr int XW = M + (M >> PSH) + 8;             // exchange buffer of a wave, float2
    static __host__ __device__ constexpr int phys(int n) { return n + (n >> PSH); }
    static constexpr int MAGROW = ((BINS + 7) & ~7) + 8;      // magnitude row of a wave: the packed bands read up to 7 bins past a band's end, as zeros
    static __host__ __device__ constexpr size_t lds_floats(int n_mel) {
        return (size_t)2 * N + (size_t)(MELW_WCAP + 4) + (size_t)MEL_WAVES * MAGROW + (size_t)2 * MEL_WAVES * XW
               + (size_t)n_mel * (MEL_FPB + 1) + (size_t)(1 + (2 * n_mel + 2) / 2);
    }
};

template <int P>
__global__ __launch_bounds__(64 * MEL_WAVES, 2) void mel_wave_kernel(const float* __restrict__ wav, const int* __restrict__ lens,
                                                                   int L, int F, int hop, int pad, int n_mel, float mag_eps,
                                                                   float log_clip, const float* __restrict__ window,
                                                                   const float* __restrict__ melbasis,
                                                                   const int* __restrict__ bands, float* __restrict__ mel,
                                                                   float* __restrict__ mag, float* __restrict__ re_out,
                                                                   float* __restrict__ im_out, const MelRange rng) {
    constexpr int M = MelW<P>::M, N = MelW<P>::N, BINS = MelW<P>::BINS, XW = MelW<P>::XW, MAGROW = MelW<P>::MAGROW;
    constexpr int I8 = 8 * P;                       // radix-8 butterflies per pass
    constexpr int NU = (I8 + 63) / 64;              // ... per lane
    extern __shared__ __attribute__((aligned(16))) float smem[];
    mel_range_begin(rng);
    float2* const twl = reinterpret_cast<float2*>(smem);                      // [N] exp(-2 pi i n / N)
    float* const wl = smem + 2 * N;                                           // [MELW_WCAP + 4] packed filter weights
    float* const magl = wl + MELW_WCAP + 4;                                   // [MEL_WAVES][MAGROW]
    float2* const xch = reinterpret_cast<float2*>(magl + MEL_WAVES * MAGROW); // [MEL_WAVES][XW]
    float* const melt = magl + MEL_WAVES * MAGROW + 2 * MEL_WAVES * XW;       // [n_mel][MEL_FPB + 1]
    const int tid = threadIdx.x;
    const int j = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fblocks = (F + MEL_FPB - 1) / MEL_FPB;
    const int b = blockIdx.x / fblocks;
    const int f0 = (blockIdx.x - b * fblocks) * MEL_FPB;
    const float* wb = wav + (size_t)b * L;
    int Li = L, Fi = F;        // ragged batch, see mel1024_kernel
    if (lens) {
        Li = lens[b] < L ? lens[b] : L;
        Fi = Li <= pad ? 0 : (Li + 2 * pad - N) / hop + 1;
        Fi = Fi < F ? Fi : F;
    }
    if (f0 >= Fi) return;   // workgroup-uniform

    for (int n = tid; n < N; n += 64 * MEL_WAVES) {
        double sn, cs;
        sincospi(2.0 * (double)n / (double)N, &sn, &cs);
        twl[n] = make_float2((float)cs, (float)-sn);
    }
    // packed filter rows, as in mel1024_kernel (same products in the same order as the loop over the global rows)
    int* const total_p = reinterpret_cast<int*>(melt + n_mel * (MEL_FPB + 1));
    unsigned short* const offs = reinterpret_cast<unsigned short*>(total_p + 1);
    unsigned short* const los = offs + n_mel + 1;
    bool packed = false;
    if (mel && bands) {
        if (w == 0) {
            int wd[4], sum = 0;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = 4 * j + t;
                wd[t] = 0;
                if (m < n_mel) {
                    const int lo = bands[2 * m], hi = bands[2 * m + 1];
                    wd[t] = hi > lo ? (hi - lo + 7) & ~7 : 0;
                    los[m] = (unsigned short)lo;
                }
                sum += wd[t];
            }
            int incl = sum;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int o = __shfl_up(incl, d, 64);
                if (j >= d) incl += o;
            }
            int base = incl - sum;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int m = 4 * j + t;
                if (m <= n_mel) offs[m] = (unsigned short)base;
                base += wd[t];
            }
            if (j == 63) { *total_p = incl; if (n_mel == MEL_MAXMEL) offs[n_mel] = (unsigned short)incl; }
        }
        for (int e = BINS + j; e < MAGROW; e += 64) magl[w * MAGROW + e] = 0.f;      // the row's pad: read under a zero weight
        __syncthreads();
        const int total = *total_p;
        packed = total <= MELW_WCAP;
        if (packed) {
            for (int e = tid; e < total; e += 64 * MEL_WAVES) {
                int ml = 0, mh = n_mel;
                while (mh - ml > 1) {
                    const int mid = (ml + mh) >> 1;
                    if ((int)offs[mid] <= e) ml = mid; else mh = mid;
                }
                const int k = (int)los[ml] + (e - (int)offs[ml]);
                wl[e] = k < bands[2 * ml + 1] ? melbasis[(size_t)ml * BINS + k] : 0.f;
            }
        }
    }
    __syncthreads();

    auto phys = [](int n) { return MelW<P>::phys(n); };
    const float2* win = reinterpret_cast<const float2*>(window) + j;
    float2* xw = xch + w * XW;
    float* mg = magl + w * MAGROW;
    float rlo = 0.f, rhi = 0.f;

    for (int fi = 0; fi < MEL_FPW; ++fi) {
        const int fl = w * MEL_FPW + fi;
        const int f = f0 + fl;
        if (f >= Fi) break;                         // wave-uniform
        float2 v[P];
        const int s0 = f * hop - pad + 2 * j;
        {
            float2 wq[P];
#pragma unroll
            for (int q = 0; q < P; ++q) wq[q] = win[64 * q];
            if (s0 - 2 * j >= 0 && f * hop - pad + N <= Li) {          // interior frame (wave-uniform): no reflection
#pragma unroll
                for (int q = 0; q < P; ++q) v[q] = reinterpret_cast<const Float2U*>(wb + s0 + 128 * q)->v;
                asm volatile("; interior frame: loads issued" ::: "memory");
            } else {
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const int sidx = s0 + 128 * q;
                    v[q] = make_float2(wb[reflect_index(sidx, Li)], wb[reflect_index(sidx + 1, Li)]);
                }
                asm volatile("; edge frame: loads issued" ::: "memory");
            }
#pragma unroll
            for (int q = 0; q < P; ++q) {
                rlo = fminf(rlo, fminf(v[q].x, v[q].y)); rhi = fmaxf(rhi, fmaxf(v[q].x, v[q].y));
                v[q] = make_float2(__fmul_rn(v[q].x, wq[q].x), __fmul_rn(v[q].y, wq[q].y));
            }
        }
        // pass 0 (radix P, Ns = 1): no twiddles; out[P j + q]
        dftP<P>(v);
#pragma unroll
        for (int q = 0; q < P; ++q) xw[phys(P * j + q)] = v[q];
        __builtin_amdgcn_wave_barrier();
        // passes 1 and 2 (radix 8; Ns = P, then 8 P): butterfly i = j + 64 u reads in[i + 8 P q], twiddle exp(-2 pi i q k / (8 Ns)), k = i mod Ns,
        // writes out[8 (i - k) + k + Ns q]; the second pass leaves Z in natural order
#pragma unroll
        for (int pass = 1; pass <= 2; ++pass) {
            const int Ns = pass == 1 ? P : I8;
            const int tstep = N / (8 * Ns);          // table index of exp(-2 pi i / (8 Ns)): 16, then 2
            float2 a[NU][8];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = j + 64 * u;
                if (i < I8) {
                    const int k = i % Ns;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[u][q] = xw[phys(i + I8 * q)];
#pragma unroll
                    for (int q = 1; q < 8; ++q) a[u][q] = cmul(a[u][q], twl[tstep * q * k]);
                    dft8(a[u]);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = j + 64 * u;
                if (i < I8) {
                    const int k = i % Ns;
#pragma unroll
                    for (int q = 0; q < 8; ++q) xw[phys(8 * (i - k) + k + Ns * q)] = a[u][q];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // real-FFT split, bins k = j + 64 t: X[k] = E - i W^k O from Z[k] and Z[M - k]; bin M (Nyquist) from Z[0]
        const size_t ob = ((size_t)b * BINS) * F + f;
#pragma unroll
        for (int t = 0; t < P; ++t) {
            const int k = j + 64 * t;
            const float2 z = xw[phys(k)];
            const float2 zp = xw[phys(k == 0 ? 0 : M - k)];
            const float2 e = make_float2(0.5f * (z.x + zp.x), 0.5f * (z.y - zp.y));       // (Z + conj Zp) / 2
            const float2 o = make_float2(0.5f * (z.x - zp.x), 0.5f * (z.y + zp.y));       // (Z - conj Zp) / 2
            const float2 tt = cmul(twl[k], o);
            const float2 x = make_float2(e.x + tt.y, e.y - tt.x);
            const float m = sqrtf(x.x * x.x + x.y * x.y + mag_eps);
            mg[k] = m;
            if (mag) mag[ob + (size_t)k * F] = m;
            if (re_out) re_out[ob + (size_t)k * F] = x.x;
            if (im_out) im_out[ob + (size_t)k * F] = x.y;
            if (k == 0) {
                const float xn = z.x - z.y;
                const float mn = sqrtf(xn * xn + mag_eps);
                mg[M] = mn;
                if (mag) mag[ob + (size_t)M * F] = mn;
                if (re_out) re_out[ob + (size_t)M * F] = xn;
                if (im_out) im_out[ob + (size_t)M * F] = 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (mel && packed) {
            for (int m0 = j; m0 < n_mel; m0 += 128) {
                const int m1 = m0 + 64;
                const int oa = offs[m0], na = (int)offs[m0 + 1] - oa;
                int ob2 = 0, nb = 0, lb = 0;
                if (m1 < n_mel) { ob2 = offs[m1]; nb = (int)offs[m1 + 1] - ob2; lb = los[m1]; }
                const int nt = na + nb;
                const float* pw = wl + oa;
                const float* pg = mg + (int)los[m0];
                float acc = 0.f, acca = 0.f;
                for (int i = 0; i < nt; i += 8) {
                    if (i == na) { acca = acc; acc = 0.f; pw = wl + ob2; pg = mg + lb; }
                    const float4 w4 = *reinterpret_cast<const float4*>(pw), w8 = *reinterpret_cast<const float4*>(pw + 4);
                    float x[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) x[u] = pg[u];
                    acc = fmaf(w4.x, x[0], acc); acc = fmaf(w4.y, x[1], acc); acc = fmaf(w4.z, x[2], acc); acc = fmaf(w4.w, x[3], acc);
                    acc = fmaf(w8.x, x[4], acc); acc = fmaf(w8.y, x[5], acc); acc = fmaf(w8.z, x[6], acc); acc = fmaf(w8.w, x[7], acc);
                    pw += 8; pg += 8;
                }
                float accb = 0.f;
                if (nb > 0) accb = acc; else acca = acc;
                if (log_clip > 0.f) { acca = logf(fmaxf(acca, log_clip)); accb = logf(fmaxf(accb, log_clip)); }
                melt[m0 * (MEL_FPB + 1) + fl] = acca;
                if (m1 < n_mel) melt[m1 * (MEL_FPB + 1) + fl] = accb;
            }
        } else if (mel) {
            for (int m = j; m < n_mel; m += 64) {
                const int lo = bands ? bands[2 * m] : 0;
                const int hi = bands ? bands[2 * m + 1] : BINS;
                const float* row = melbasis + (size_t)m * BINS;
                float acc = 0.f;
                for (int k = lo; k < hi; ++k) acc = fmaf(row[k], mg[k], acc);
                if (log_clip > 0.f) acc = logf(fmaxf(acc, log_clip));
                melt[m * (MEL_FPB + 1) + fl] = acc;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    mel_range_end(rng, rlo, rhi);
    if (!mel) return;
    __syncthreads();
    const int nfr = (Fi - f0) < MEL_FPB ? (Fi - f0) : MEL_FPB;
    for (int idx = tid; idx < n_mel * MEL_FPB; idx += 64 * MEL_WAVES) {
        const int m = idx >> 5, fl = idx & 31;
        if (fl < nfr) mel[((size_t)b * n_mel + m) * F + f0 + fl] = melt[m * (MEL_FPB + 1) + fl];
    }
}

template <int P>
static hipError_t launch_mel_wave(const amp_mel_desc& d, const float* wav, const int* lens, int B, int L, int F, int pad, int n_mel,
                                  const float* window, const float* melbasis, float* mel, float* mag, float* re, float* im, const MelRange& rng,
                                  hipStream_t stream) {
    const size_t lds = MelW<P>::lds_floats(n_mel) * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds<&mel_wave_kernel<P>>(160 * 1024); e != hipSuccess) return e;
    const int fblocks = (F + MEL_FPB - 1) / MEL_FPB;
    hipLaunchKernelGGL(mel_wave_kernel<P>, dim3((unsigned)((size_t)B * fblocks)), dim3(64 * MEL_WAVES), lds, stream, wav, lens, L, F,
                       d.hop_size, pad, n_mel, d.mag_eps, d.log_clip, window, melbasis, static_cast<const int*>(d.mel_bands_dev), mel, mag, re, im, rng);
    return hipGetLastError();
}

hipError_t launch_mel(const amp_mel_desc& d, const float* wav, const int* lens, int B, int L, int F, const float* window,
                      const float* melbasis, float* mel, float* mag, float* re, float* im, hipStream_t stream) {
    const int pad = d.pad_mode == 0 ? (d.n_fft - d.hop_size) / 2 : d.n_fft / 2;
    const int n_mel = mel ? d.n_mel : 0;
    const MelRange rng{d.range_dev, d.range_reset_dev, d.range_seq};
    if (d.n_fft == 1024 && n_mel <= MEL_MAXMEL && (pad & 1) == 0 && (d.hop_size & 1) == 0) {
        // wave-per-frame radix-8 real FFT (every shipped config of the reference)
        const size_t lds = mel1024_lds_floats(n_mel) * sizeof(float);
        {
            const hipError_t e = mel1024_device_init(stream, true);
            if (e != hipSuccess) return e;
        }
        const int fblocks = (F + MEL_FPB - 1) / MEL_FPB;
        hipLaunchKernelGGL(mel1024_kernel, dim3((unsigned)((size_t)B * fblocks)), dim3(64 * MEL_WAVES), lds, stream, wav, lens, L, F,
                           d.hop_size, pad, n_mel, d.mag_eps, d.log_clip, window, melbasis,
                           static_cast<const int*>(d.mel_bands_dev), mel, mag, re, im, rng);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && d.range_dev && d.range_host)
            e = hipMemcpyAsync(d.range_host, d.range_dev, 3 * sizeof(int), hipMemcpyDeviceToHost, stream);
        return e;
    }
    if ((d.n_fft == 2048 || d.n_fft == 1920 || d.n_fft == 512) && (pad & 1) == 0 && (d.hop_size & 1) == 0 && pad >= 0 && n_mel <= MEL_MAXMEL &&
        (d.n_fft == 2048 ? MelW<16>::lds_floats(n_mel) : d.n_fft == 1920 ? MelW<15>::lds_floats(n_mel) : MelW<4>::lds_floats(n_mel)) * sizeof(float) <= 160 * 1024) {
        // wave-per-frame radix-16 / radix-15 / radix-4 + 8 x 8 real FFT (round 6)
        hipError_t e = d.n_fft == 2048 ? launch_mel_wave<16>(d, wav, lens, B, L, F, pad, n_mel, window, melbasis, mel, mag, re, im, rng, stream)
                       : d.n_fft == 1920 ? launch_mel_wave<15>(d, wav, lens, B, L, F, pad, n_mel, window, melbasis, mel, mag, re, im, rng, stream)
                                         : launch_mel_wave<4>(d, wav, lens, B, L, F, pad, n_mel, window, melbasis, mel, mag, re, im, rng, stream);
        if (e == hipSuccess && d.range_dev && d.range_host)
            e = hipMemcpyAsync(d.range_host, d.range_dev, 3 * sizeof(int), hipMemcpyDeviceToHost, stream);
        return e;
    }
    if ((d.n_fft & (d.n_fft - 1)) != 0) {
        // any other length: mixed-radix Stockham, one workgroup per frame
        MelRadices rad;
        if (!mel_radices(d.n_fft, &rad)) return hipErrorInvalidValue;
        const size_t lds = (size_t)(3 * d.n_fft) * sizeof(float2) + (size_t)(d.n_fft / 2 + 1) * sizeof(float);
        if (hipError_t e = ensure_dynamic_lds<&mel_mixed_kernel>(112 * 1024); e != hipSuccess) return e;   // per device, amp_internal.h
        hipLaunchKernelGGL(mel_mixed_kernel, dim3((unsigned)((size_t)B * F)), dim3(256), lds, stream, wav, lens, L, F, d.n_fft, rad, d.hop_size, pad,
                           n_mel, d.mag_eps, d.log_clip, window, melbasis, mel, mag, re, im, rng);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && d.range_dev && d.range_host)
            e = hipMemcpyAsync(d.range_host, d.range_dev, 3 * sizeof(int), hipMemcpyDeviceToHost, stream);
        return e;
    }
    // generic power-of-two n_fft: one workgroup per frame, radix-2
    int log2n = 0;
    while ((1 << log2n) < d.n_fft) ++log2n;
    const size_t lds = (size_t)(2 * d.n_fft + d.n_fft / 2) * sizeof(float2) + (size_t)(d.n_fft / 2 + 1) * sizeof(float);
    if (lds > 64 * 1024) {                 // n_fft = 4096: 90 KB (round 5: the launch used to fail for want of the attribute)
        if (hipError_t e = ensure_dynamic_lds<&mel_kernel>(96 * 1024); e != hipSuccess) return e;
    }
    dim3 grid((unsigned)((size_t)B * F));
    hipLaunchKernelGGL(mel_kernel, grid, dim3(256), lds, stream, wav, lens, L, F, d.n_fft, log2n, d.hop_size, pad,
                       n_mel, d.mag_eps, d.log_clip, window, melbasis, mel, mag, re, im, rng);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && d.range_dev && d.range_host)
        e = hipMemcpyAsync(d.range_host, d.range_dev, 3 * sizeof(int), hipMemcpyDeviceToHost, stream);
    return e;
}

// ------------------------------------------------------------------------------------------------
// Inverse STFT (utils/stft.py:183-222, STFT.inverse): the reference runs conv_transpose1d with the
// windowed pseudo-inverse Fourier basis [2*(n_fft/2+1), 1, n_fft], divides by the window-sum-square
// envelope and crops n_fft/2 per side.  pinv of the stacked [Re; Im] real-DFT matrix is the inverse real
// FFT (Im X_0 and Im X_{N/2} have all-zero basis rows and drop out), so:
//   frame_f[n] = irfft(mag_f * e^{i phase_f})[n] * window[n] * hop/n_fft        (kernel 1, FFT in LDS)
//   out[m]     = (n_fft/hop) * sum_f frame_f[m - f*hop] / wss[m]   where wss[m] > tiny      (kernel 2)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void istft_frames_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                           int polar, int F, int n_fft, int log2n, float inv_scale,
                                                           const float* __restrict__ window, float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* buf0 = reinterpret_cast<float2*>(smem);   // [n_fft]
    float2* buf1 = buf0 + n_fft;                      // [n_fft]
    float2* tw = buf1 + n_fft;                        // [n_fft/2]  exp(-2*pi*i*m/n_fft)
    const int tid = threadIdx.x;
    const int b = blockIdx.x / F;
    const int f = blockIdx.x - b * F;
    const int half = n_fft >> 1;
    const int bins = half + 1;
    // ifft(X) = conj(fft(conj(X))) / N with the Hermitian extension X[N-k] = conj(X[k])
    for (int k = tid; k < bins; k += 256) {
        const size_t o = ((size_t)b * bins + k) * F + f;
        float re, im;
        if (polar) {          // (magnitude, phase)
            const float m = mag[o];
            float sn, cs;
            sincosf(phase[o], &sn, &cs);
            re = m * cs;
            im = m * sn;
        } else {              // (real, imaginary) -- APNet's ISTFT head (apnet.py:385-393)
            re = mag[o];
            im = phase[o];
        }
        if (k == 0 || k == half) im = 0.f;   // irfft ignores them (zero rows of the pseudo-inverse basis)
        buf0[k] = make_float2(re, -im);
        if (k > 0 && k < half) buf0[n_fft - k] = make_float2(re, im);
    }
    for (int m = tid; m < half; m += 256) {
        float sn, cs;
        sincospif(-2.0f * (float)m / (float)n_fft, &sn, &cs);
        tw[m] = make_float2(cs, sn);
    }
    __syncthreads();
    float2* in = buf0;
    float2* out = buf1;
    for (int s = 0; s < log2n; ++s) {
        const int Ns = 1 << s;
        const int tstride = half >> s;
        for (int j = tid; j < half; j += 256) {
            const int k = j & (Ns - 1);
            const float2 w = tw[k * tstride];
            const float2 v0 = in[j];
            const float2 x1 = in[j + half];
            const float2 v1 = make_float2(x1.x * w.x - x1.y * w.y, x1.x * w.y + x1.y * w.x);
            const int j0 = ((j - k) << 1) + k;
            out[j0] = make_float2(v0.x + v1.x, v0.y + v1.y);
            out[j0 + Ns] = make_float2(v0.x - v1.x, v0.y - v1.y);
        }
        __syncthreads();
        float2* t = in; in = out; out = t;
    }
    const float sc = inv_scale / (float)n_fft;
    float* fr = frames + ((size_t)b * F + f) * n_fft;
    for (int n = tid; n < n_fft; n += 256) fr[n] = in[n].x * sc * window[n];
}

// The same for any other n_fft (round 5: the inverse of mel_mixed_kernel's transform): mixed-radix Stockham passes over
// the Hermitian extension; an odd n_fft has no Nyquist bin (bins = (n_fft - 1) / 2 + 1, every k >= 1 has a partner n_fft - k).
__global__ __launch_bounds__(256) void istft_frames_mixed_kernel(const float* __restrict__ mag, const float* __restrict__ phase,
                                                                 int polar, int F, int n_fft, const MelRadices rad, float inv_scale,
                                                                 const float* __restrict__ window, float* __restrict__ frames) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* buf0 = reinterpret_cast<float2*>(smem);   // [n_fft]
    float2* buf1 = buf0 + n_fft;                      // [n_fft]
    float2* tw = buf1 + n_fft;                        // [n_fft]
    const int tid = threadIdx.x;
    const int b = blockIdx.x / F;
    const int f = blockIdx.x - b * F;
    const int half = n_fft >> 1;
    const int bins = half + 1;
    const int nyq = (n_fft & 1) ? -1 : half;
    for (int k = tid; k < bins; k += 256) {
        const size_t o = ((size_t)b * bins + k) * F + f;
        float re, im;
        if (polar) {
            const float m = mag[o];
            float sn, cs;
            sincosf(phase[o], &sn, &cs);
            re = m * cs;
            im = m * sn;
        } else {
            re = mag[o];
            im = phase[o];
        }
        if (k == 0 || k == nyq) im = 0.f;
        buf0[k] = make_float2(re, -im);
        if (k > 0 && k != nyq) buf0[n_fft - k] = make_float2(re, im);
    }
    for (int n = tid; n < n_fft; n += 256) {
        float sn, cs;
        sincospif(-2.0f * (float)n / (float)n_fft, &sn, &cs);
        tw[n] = make_float2(cs, sn);
    }
    __syncthreads();
    float2* in = buf0;
    float2* out = buf1;
    int Ns = 1;
    for (int s = 0; s < rad.n; ++s) {
        const int r = rad.r[s];
        switch (r) {
            case 2: mixed_pass<2>(in, out, tw, n_fft, Ns, tid); break;
            case 3: mixed_pass<3>(in, out, tw, n_fft, Ns, tid); break;
            case 5: mixed_pass<5>(in, out, tw, n_fft, Ns, tid); break;
            case 7: mixed_pass<7>(in, out, tw, n_fft, Ns, tid); break;
            case 11: mixed_pass<11>(in, out, tw, n_fft, Ns, tid); break;
            case 13: mixed_pass<13>(in, out, tw, n_fft, Ns, tid); break;
            default: generic_pass(in, out, tw, n_fft, Ns, r, tid); break;
        }
        Ns *= r;
        __syncthreads();
        float2* t = in; in = out; out = t;
    }
    const float sc = inv_scale / (float)n_fft;
    float* fr = frames + ((size_t)b * F + f) * n_fft;
    for (int n = tid; n < n_fft; n += 256) fr[n] = in[n].x * sc * window[n];
}

// Wave-per-frame INVERSE real FFT for n_fft = 128 P (built for P = 15: n_fft = 1920; round 6: the frames of STFT.inverse, utils/stft.py:183-222, and of
// the mel-loss gradient ran a full complex mixed-radix transform of n_fft points per WORKGROUP; P = 4 | 8 | 16 work and are level with the radix-2 kernel): the
// Hermitian spectrum X[0 .. M] -> Z[k] = E[k] + i W^-k D[k] with E = (X[k] + conj X[M - k]) / 2, D = (X[k] - conj X[M - k]) / 2 -- the inverse of
// mel_wave_kernel's split --, z = IDFT_M(Z) = conj(DFT_M(conj Z)) / M through the same three passes, x[2n] = Re z[n], x[2n + 1] = Im z[n],
// times the window and inv_scale.  (irfft ignores Im X[0] and Im X[M]: zero rows of the pseudo-inverse basis.)
template <int P>
__global__ __launch_bounds__(64 * MEL_WAVES, 2) void istft_wave_kernel(const float* __restrict__ mag, const float* __restrict__ phase, int polar,
                                                                     int F, long long nframes, float inv_scale,
                                                                     const float* __restrict__ window, float* __restrict__ frames) {
    constexpr int M = MelW<P>::M, N = MelW<P>::N, BINS = MelW<P>::BINS, XW = MelW<P>::XW;
    constexpr int I8 = 8 * P, NU = (I8 + 63) / 64;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float2* const twl = reinterpret_cast<float2*>(smem);                      // [N] exp(-2 pi i n / N)
    float2* const xch = twl + N;                                              // [MEL_WAVES][XW]
    const int tid = threadIdx.x;
    const int j = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int n = tid; n < N; n += 64 * MEL_WAVES) {
        double sn, cs;
        sincospi(2.0 * (double)n / (double)N, &sn, &cs);
        twl[n] = make_float2((float)cs, (float)-sn);
    }
    __syncthreads();
    auto phys = [](int n) { return MelW<P>::phys(n); };
    float2* xw = xch + w * XW;
    for (int fi = 0; fi < MEL_FPW; ++fi) {
        const long long gf = (long long)blockIdx.x * MEL_FPB + w * MEL_FPW + fi;    // flat frame index b * F + f
        if (gf >= nframes) break;                                                   // wave-uniform
        const long long b = gf / F;
        const int f = (int)(gf - b * F);
        // X[k], k = j + 64 q (and k = M by lane 0), into the exchange buffer as complex numbers
        auto load_bin = [&](int k) {
            const size_t o = ((size_t)b * BINS + k) * F + f;
            float re, im;
            if (polar) {
                const float m = mag[o];
                float sn, cs;
                sincosf(phase[o], &sn, &cs);
                re = m * cs; im = m * sn;
            } else {
                re = mag[o]; im = phase[o];
            }
            if (k == 0 || k == M) im = 0.f;
            return make_float2(re, im);
        };
#pragma unroll 4
        for (int q = 0; q < P; ++q) xw[phys(j + 64 * q)] = load_bin(j + 64 * q);     // (four at a time: sixteen sincosf side by side spill)
        if (j == 0) xw[phys(M)] = load_bin(M);     // (slot M: inside the buffer's 8 spare entries)
        __builtin_amdgcn_wave_barrier();
        float2 v[P];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            const int k = j + 64 * q;
            const float2 xq = xw[phys(k)];
            const float2 xp = xw[phys(M - k)];                                         // X[M - k] (k = 0: X[M])
            const float2 e = make_float2(0.5f * (xq.x + xp.x), 0.5f * (xq.y - xp.y));  // (X[k] + conj X[M - k]) / 2
            const float2 dd = make_float2(0.5f * (xq.x - xp.x), 0.5f * (xq.y + xp.y)); // (X[k] - conj X[M - k]) / 2
            const float2 tw = twl[k];                                                  // W^k; W^-k = conj
            const float2 wd = make_float2(tw.x * dd.x + tw.y * dd.y, tw.x * dd.y - tw.y * dd.x);   // W^-k D
            const float2 z = make_float2(e.x - wd.y, e.y + wd.x);                      // E + i W^-k D
            v[q] = make_float2(z.x, -z.y);                                             // conj Z
        }
        __builtin_amdgcn_wave_barrier();
        dftP<P>(v);
#pragma unroll
        for (int q = 0; q < P; ++q) xw[phys(P * j + q)] = v[q];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 1; pass <= 2; ++pass) {
            const int Ns = pass == 1 ? P : I8;
            const int tstep = N / (8 * Ns);
            float2 a[NU][8];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = j + 64 * u;
                if (i < I8) {
                    const int k = i % Ns;
#pragma unroll
                    for (int q = 0; q < 8; ++q) a[u][q] = xw[phys(i + I8 * q)];
#pragma unroll
                    for (int q = 1; q < 8; ++q) a[u][q] = cmul(a[u][q], twl[tstep * q * k]);
                    dft8(a[u]);
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const int i = j + 64 * u;
                if (i < I8) {
                    const int k = i % Ns;
#pragma unroll
                    for (int q = 0; q < 8; ++q) xw[phys(8 * (i - k) + k + Ns * q)] = a[u][q];
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // z[n] = conj(Y[n]) / M: x[2n] = Y.x / M, x[2n + 1] = -Y.y / M; times window and scale; one 8-byte store per lane: whole 512-B rows
        const float sc = inv_scale / (float)M;
        float2* fr = reinterpret_cast<float2*>(frames + (size_t)gf * N);
        const float2* win = reinterpret_cast<const float2*>(window);
#pragma unroll
        for (int t = 0; t < P; ++t) {
            const int n = j + 64 * t;
            const float2 y = xw[phys(n)];
            const float2 wn = win[n];
            fr[n] = make_float2(y.x * sc * wn.x, -y.y * sc * wn.y);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int P>
static hipError_t launch_istft_wave(const float* a, const float* b, int polar, int B, int F, float inv_scale, const float* window, float* frames,
                                    hipStream_t stream) {
    const size_t lds = ((size_t)MelW<P>::N + (size_t)MEL_WAVES * MelW<P>::XW) * sizeof(float2);
    if (hipError_t e = ensure_dynamic_lds<&istft_wave_kernel<P>>(lds); e != hipSuccess) return e;
    const long long nframes = (long long)B * F;
    hipLaunchKernelGGL(istft_wave_kernel<P>, dim3((unsigned)((nframes + MEL_FPB - 1) / MEL_FPB)), dim3(64 * MEL_WAVES), lds, stream, a, b, polar, F,
                       nframes, inv_scale, window, frames);
    return hipGetLastError();
}

// irfft(spectrum) * window * inv_scale / n_fft per frame: the radix-2 kernel for powers of two, the mixed-radix one otherwise
static hipError_t launch_istft_frames(int n_fft, const float* a, const float* b, int polar, int B, int F, float inv_scale, const float* window,
                                      float* frames, hipStream_t stream) {
    // n_fft = 1920: the wave-per-frame kernel (needs 8-byte aligned frames and window: hipMalloc'ed tensors always are).  Measured, same box, alternating
    // (profiles/r6_b_mel_wave_kernel.txt): STFT.inverse of 64 x 137 frames 0.272 -> 0.177 ms.  For powers of two it is level with the radix-2 kernel
    // (512 / 1024 / 2048: 0.129 / 0.134 / 0.175 against 0.131 / 0.129 / 0.160 ms): the inverse is bound by its strided spectrum reads, the frame round trip
    // through HBM and the overlap-add pass, not by the transform -- those lengths stay where they were.
    if (n_fft == 1920 && (reinterpret_cast<uintptr_t>(window) & 7) == 0 && (reinterpret_cast<uintptr_t>(frames) & 7) == 0)
        return launch_istft_wave<15>(a, b, polar, B, F, inv_scale, window, frames, stream);
    if ((n_fft & (n_fft - 1)) == 0) {
        int log2n = 0;
        while ((1 << log2n) < n_fft) ++log2n;
        const size_t lds = (size_t)(2 * n_fft + n_fft / 2) * sizeof(float2);
        if (lds > 64 * 1024) {
            if (hipError_t e = ensure_dynamic_lds<&istft_frames_kernel>(96 * 1024); e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)((size_t)B * F)), dim3(256), lds, stream, a, b, polar, F, n_fft, log2n, inv_scale, window, frames);
        return hipGetLastError();
    }
    MelRadices rad;
    if (!mel_radices(n_fft, &rad)) return hipErrorInvalidValue;
    const size_t lds = (size_t)(3 * n_fft) * sizeof(float2);
    if (hipError_t e = ensure_dynamic_lds<&istft_frames_mixed_kernel>(112 * 1024); e != hipSuccess) return e;
    hipLaunchKernelGGL(istft_frames_mixed_kernel, dim3((unsigned)((size_t)B * F)), dim3(256), lds, stream, a, b, polar, F, n_fft, rad, inv_scale, window, frames);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void istft_ola_kernel(const float* __restrict__ frames, const float* __restrict__ wss,
                                                        int F, int n_fft, int hop, int crop, int Lout, float scale,
                                                        float tiny, float* __restrict__ wav) {
    const int b = blockIdx.y;
    const int mo = blockIdx.x * 256 + threadIdx.x;
    if (mo >= Lout) return;
    const int m = mo + crop;
    int f0 = (m - n_fft + hop) / hop;          // ceil((m - n_fft + 1) / hop) for m - n_fft + 1 > 0
    if (m - n_fft + 1 <= 0) f0 = 0;
    int f1 = m / hop;
    if (f1 > F - 1) f1 = F - 1;
    const float* fb = frames + (size_t)b * F * n_fft;
    float acc = 0.f;
    for (int f = f0; f <= f1; ++f) acc += fb[(size_t)f * n_fft + (m - f * hop)];
    const float w = wss[m];
    if (w > tiny) acc /= w;
    wav[(size_t)b * Lout + mo] = acc * scale;
}

// mode 0: STFT.inverse (stft.py:183-222): polar input, crop n_fft/2, out = scale * sum / wss where wss > tiny
// mode 1: APNet ISTFT "same" (apnet.py:46-101): re/im input, crop (win - hop)/2, out = sum / envelope, L = F * hop
hipError_t launch_istft(const amp_mel_desc& d, int mode, const float* a, const float* b, int B, int F, const float* window,
                        const float* wss, float* frames, float* wav, hipStream_t stream) {
    const float scale = mode == 0 ? (float)d.n_fft / (float)d.hop_size : 1.0f;
    hipError_t e = launch_istft_frames(d.n_fft, a, b, mode == 0 ? 1 : 0, B, F, 1.0f / scale, window, frames, stream);
    if (e != hipSuccess) return e;
    const int crop = mode == 0 ? d.n_fft / 2 : (d.win_size - d.hop_size) / 2;
    const int Lout = mode == 0 ? d.hop_size * (F - 1) + (d.n_fft & 1) : d.hop_size * (F - 1) + d.win_size - 2 * crop;   // (odd n_fft: crop = floor(n_fft / 2) twice)
    if (Lout <= 0) return hipSuccess;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)((Lout + 255) / 256), (unsigned)B), dim3(256), 0, stream, frames, wss, F,
                       d.n_fft, d.hop_size, crop, Lout, scale, mode == 0 ? 1.17549435e-38f : -1.0f, wav);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------------
// Backward of the log-mel front end (the training-time mel loss, gan_vocoder_trainer.py:387-392:
// loss = 45 * L1(extract_mel_features(y_gt), extract_mel_features(y_pred)) needs d loss / d y_pred).
//
//   lm = log(max(mel, clip)),  mel = B |X|,  |X| = sqrt(re^2 + im^2 + eps),  X = rDFT(window * frame of reflect-pad(x))
//
//   g_mel = g_lm / mel  where mel >= clip (torch.clamp passes the gradient there), else 0
//   g_|X| = B^T g_mel ;  g_re = g_|X| re / |X| ;  g_im = g_|X| im / |X|
//   g_frame[n] = window[n] * Re sum_{k=0}^{N/2} (g_re + i g_im)[k] e^{+2 pi i k n / N}
//              = window[n] * N * irfft(H)[n],   H = G at k = 0, N/2 and G / 2 in between (the one-sided sum counts
//                interior bins once, the Hermitian extension twice)                    -> istft_frames_kernel
//   g_xpad = overlap-add of the frames ;  g_x[s] = g_xpad[s + pad] + the mirrored positions of the reflection padding
// ------------------------------------------------------------------------------------------------
// kernel 1: (g_lm, mel, |X|, re, im) -> H_re, H_im  [B, bins, F];  block = 32 frames x 8 bin lanes
__global__ __launch_bounds__(256) void mel_grad_spec_kernel(const float* __restrict__ g_lm, const float* __restrict__ mel,
                                                            const float* __restrict__ mag, const float* __restrict__ re,
                                                            const float* __restrict__ im, const float* __restrict__ melbasis,
                                                            int F, int bins, int nyq, int n_mel, float log_clip,
                                                            float* __restrict__ h_re, float* __restrict__ h_im) {
    extern __shared__ float gm[];                 // [n_mel][32]  g_mel of this tile  (nyq = n_fft / 2 for an even n_fft, -1 for an odd one)
    const int fblocks = (F + 31) / 32;
    const int b = blockIdx.x / fblocks;
    const int f0 = (blockIdx.x - b * fblocks) * 32;
    const int fl = threadIdx.x & 31, kl = threadIdx.x >> 5;
    const int f = f0 + fl;
    for (int m = kl; m < n_mel; m += 8) {
        float g = 0.f;
        if (f < F) {
            const size_t o = ((size_t)b * n_mel + m) * F + f;
            const float v = mel[o];
            if (log_clip > 0.f) g = (v >= log_clip) ? g_lm[o] / v : 0.f;   // mel holds the LINEAR mel energy here
            else g = g_lm[o];
        }
        gm[m * 32 + fl] = g;
    }
    __syncthreads();
    if (f >= F) return;
    for (int k = kl; k < bins; k += 8) {
        float acc = 0.f;
        for (int m = 0; m < n_mel; ++m) acc = fmaf(melbasis[(size_t)m * bins + k], gm[m * 32 + fl], acc);
        const size_t o = ((size_t)b * bins + k) * F + f;
        const float mg = mag[o];
        const float wk = (k == 0 || k == nyq) ? 1.f : 0.5f;
        const float s = mg > 0.f ? wk * acc / mg : 0.f;
        h_re[o] = s * re[o];
        h_im[o] = s * im[o];
    }
}

// kernel 3: overlap-add of the frame gradients + fold the reflection padding back onto the samples
__global__ __launch_bounds__(256) void mel_grad_ola_kernel(const float* __restrict__ frames, const int* __restrict__ lens, int L,
                                                           int F, int n_fft, int hop, int pad, float* __restrict__ g_wav) {
    const int b = blockIdx.y;
    const int s = blockIdx.x * 256 + threadIdx.x;
    if (s >= L) return;
    int Li = L, Fi = F;
    if (lens) {
        Li = lens[b] < L ? lens[b] : L;
        Fi = Li <= pad ? 0 : (Li + 2 * pad - n_fft) / hop + 1;
        Fi = Fi < F ? Fi : F;
    }
    float acc = 0.f;
    if (s < Li) {
        const float* fb = frames + (size_t)b * F * n_fft;
        // padded positions that read sample s: t = s + pad, and through the reflection t = pad - s (s >= 1) and
        // t = pad + 2 (Li - 1) - s (s <= Li - 2); the padded signal is Li + 2 pad long
        int tpos[3];
        tpos[0] = s + pad;
        tpos[1] = (s >= 1 && s <= pad) ? pad - s : -1;
        tpos[2] = (s <= Li - 2 && 2 * (Li - 1) - s < Li + pad) ? pad + 2 * (Li - 1) - s : -1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t = tpos[i];
            if (t < 0) continue;
            int fa = (t - n_fft + hop) / hop;              // first frame covering t
            if (t - n_fft + 1 <= 0) fa = 0;
            int fz = t / hop;
            if (fz > Fi - 1) fz = Fi - 1;
            for (int f = fa; f <= fz; ++f) acc += fb[(size_t)f * n_fft + (t - f * hop)];
        }
    }
    g_wav[(size_t)b * L + s] = acc;
}

hipError_t launch_mel_backward(const amp_mel_desc& d, const int* lens, int B, int L, int F, const float* window,
                               const float* melbasis, const float* mel_lin, const float* mag, const float* re, const float* im,
                               const float* g_lm, float* h_re, float* h_im, float* frames, float* g_wav, hipStream_t stream) {
    const int bins = d.n_fft / 2 + 1;
    const int pad = d.pad_mode == 0 ? (d.n_fft - d.hop_size) / 2 : d.n_fft / 2;
    const int fblocks = (F + 31) / 32;
    hipLaunchKernelGGL(mel_grad_spec_kernel, dim3((unsigned)((size_t)B * fblocks)), dim3(256), (size_t)d.n_mel * 32 * sizeof(float), stream,
                       g_lm, mel_lin, mag, re, im, melbasis, F, bins, (d.n_fft & 1) ? -1 : d.n_fft / 2, d.n_mel, d.log_clip, h_re, h_im);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // frames = irfft(H) * window * inv_scale / n_fft * n_fft ... the frames kernel multiplies by inv_scale / n_fft: we want x n_fft
    e = launch_istft_frames(d.n_fft, h_re, h_im, 0, B, F, (float)d.n_fft, window, frames, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(mel_grad_ola_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)B), dim3(256), 0, stream, frames, lens, L, F,
                       d.n_fft, d.hop_size, pad, g_wav);
    return hipGetLastError();
}

}  // namespace amp

using namespace amp;

extern "C" {

// The caller's descriptor, as far as the caller's struct_size says it exists (fields added by later versions of the header
// read as 0 / NULL for a consumer compiled against an older one).
static bool mel_desc_in(const amp_mel_desc* in, amp_mel_desc* out, const char* who) {
    if (!in) { set_error("%s: null descriptor", who); return false; }
    const size_t sz = in->struct_size;
    if (sz < offsetof(amp_mel_desc, range_dev) || sz > 4096) {
        set_error("%s: amp_mel_desc.struct_size = %zu: set it to sizeof(amp_mel_desc) (the library reads nothing beyond it)", who, sz);
        return false;
    }
    memset(out, 0, sizeof(*out));
    memcpy(out, in, sz < sizeof(*out) ? sz : sizeof(*out));
    out->struct_size = (uint32_t)sizeof(*out);
    return true;
}

int amp_mel_init(void) {
    if (amp_device_count() <= 0) { set_error("amp_mel_init: no HIP device visible"); return AMP_ERR_HIP; }
    const hipError_t e = mel1024_device_init(nullptr, false);
    if (e != hipSuccess) { set_error("amp_mel_init: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_mel_num_frames(const amp_mel_desc* d, int L) {
    if (!d || d->hop_size <= 0 || d->n_fft <= 0) return 0;
    const int pad = d->pad_mode == 0 ? (d->n_fft - d->hop_size) / 2 : d->n_fft / 2;
    const int Lp = L + 2 * pad;
    if (Lp < d->n_fft) return 0;
    return (Lp - d->n_fft) / d->hop_size + 1;
}

int amp_mel_forward(const amp_mel_desc* d, const float* wav_dev, int B, int L, const float* window_dev,
                    const float* melbasis_dev, float* mel_dev, float* mag_dev, float* re_dev, float* im_dev,
                    void* stream) {
    return amp_mel_forward_ragged(d, wav_dev, nullptr, B, L, window_dev, melbasis_dev, mel_dev, mag_dev, re_dev, im_dev, stream);
}

int amp_mel_forward_ragged(const amp_mel_desc* d_in, const float* wav_dev, const int32_t* lens_dev, int B, int L,
                           const float* window_dev, const float* melbasis_dev, float* mel_dev, float* mag_dev,
                           float* re_dev, float* im_dev, void* stream) {
    amp_mel_desc dn;
    if (!mel_desc_in(d_in, &dn, "amp_mel_forward_ragged")) return AMP_ERR_INVALID;
    const amp_mel_desc* d = &dn;
    if (!d || !wav_dev || !window_dev) { set_error("amp_mel_forward: null argument"); return AMP_ERR_INVALID; }
    if (!mel_nfft_check("amp_mel_forward", d->n_fft)) {
        return AMP_ERR_UNSUPPORTED;
    }
    if (d->hop_size <= 0 || B <= 0 || L <= 0) { set_error("amp_mel_forward: hop=%d B=%d L=%d", d->hop_size, B, L); return AMP_ERR_INVALID; }
    const int pad = d->pad_mode == 0 ? (d->n_fft - d->hop_size) / 2 : d->n_fft / 2;
    if (pad >= L) { set_error("amp_mel_forward: reflect padding %d needs more than %d samples", pad, L); return AMP_ERR_INVALID; }
    if (mel_dev && (d->n_mel <= 0 || !melbasis_dev)) { set_error("amp_mel_forward: mel output needs n_mel > 0 and a mel basis"); return AMP_ERR_INVALID; }
    const int F = amp_mel_num_frames(d, L);
    if (F <= 0) { set_error("amp_mel_forward: no frames for L=%d", L); return AMP_ERR_INVALID; }
    hipError_t e = launch_mel(*d, wav_dev, lens_dev, B, L, F, window_dev, melbasis_dev, mel_dev, mag_dev, re_dev, im_dev, (hipStream_t)stream);
    if (e == hipErrorStreamCaptureUnsupported) return AMP_ERR_STATE;   // message set by mel1024_device_init
    if (e != hipSuccess) { set_error("amp_mel_forward: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_istft_forward(const amp_mel_desc* d_in, const float* mag_dev, const float* phase_dev, int B, int F,
                      const float* window_dev, const float* wss_dev, float* frames_ws_dev, float* wav_dev, void* stream) {
    amp_mel_desc dn;
    if (!mel_desc_in(d_in, &dn, "amp_istft_forward")) return AMP_ERR_INVALID;
    const amp_mel_desc* d = &dn;
    if (!d || !mag_dev || !phase_dev || !window_dev || !wss_dev || !frames_ws_dev || !wav_dev) { set_error("amp_istft_forward: null argument"); return AMP_ERR_INVALID; }
    if (!mel_nfft_check("amp_istft_forward", d->n_fft)) {
        return AMP_ERR_UNSUPPORTED;
    }
    if (d->hop_size <= 0 || d->hop_size > d->n_fft || B <= 0 || F <= 1) { set_error("amp_istft_forward: hop=%d B=%d F=%d", d->hop_size, B, F); return AMP_ERR_INVALID; }
    hipError_t e = launch_istft(*d, 0, mag_dev, phase_dev, B, F, window_dev, wss_dev, frames_ws_dev, wav_dev, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("amp_istft_forward: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_istft_same(const amp_mel_desc* d_in, const float* re_dev, const float* im_dev, int B, int F, const float* window_dev,
                   const float* envelope_dev, float* frames_ws_dev, float* wav_dev, void* stream) {
    amp_mel_desc dn;
    if (!mel_desc_in(d_in, &dn, "amp_istft_same")) return AMP_ERR_INVALID;
    const amp_mel_desc* d = &dn;
    if (!d || !re_dev || !im_dev || !window_dev || !envelope_dev || !frames_ws_dev || !wav_dev) { set_error("amp_istft_same: null argument"); return AMP_ERR_INVALID; }
    if (!mel_nfft_check("amp_istft_same", d->n_fft) || d->win_size != d->n_fft) {
        if (d->win_size != d->n_fft && mel_nfft_supported(d->n_fft)) set_error("amp_istft_same: n_fft=%d must equal win_size=%d", d->n_fft, d->win_size);
        return AMP_ERR_UNSUPPORTED;
    }
    if (d->hop_size <= 0 || d->hop_size > d->n_fft || ((d->win_size - d->hop_size) & 1) || B <= 0 || F <= 0) { set_error("amp_istft_same: hop=%d B=%d F=%d", d->hop_size, B, F); return AMP_ERR_INVALID; }
    hipError_t e = launch_istft(*d, 1, re_dev, im_dev, B, F, window_dev, envelope_dev, frames_ws_dev, wav_dev, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("amp_istft_same: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_mel_backward(const amp_mel_desc* d_in, const int32_t* lens_dev, int B, int L, const float* window_dev,
                     const float* melbasis_dev, const float* mel_linear_dev, const float* mag_dev, const float* re_dev,
                     const float* im_dev, const float* grad_logmel_dev, float* spec_ws_dev, float* frames_ws_dev,
                     float* grad_wav_dev, void* stream) {
    amp_mel_desc dn;
    if (!mel_desc_in(d_in, &dn, "amp_mel_backward")) return AMP_ERR_INVALID;
    const amp_mel_desc* d = &dn;
    if (!d || !window_dev || !melbasis_dev || !mel_linear_dev || !mag_dev || !re_dev || !im_dev || !grad_logmel_dev || !spec_ws_dev ||
        !frames_ws_dev || !grad_wav_dev) { set_error("amp_mel_backward: null argument"); return AMP_ERR_INVALID; }
    if (!mel_nfft_check("amp_mel_backward", d->n_fft)) {
        return AMP_ERR_UNSUPPORTED;
    }
    if (d->hop_size <= 0 || d->n_mel <= 0 || B <= 0 || L <= 0) { set_error("amp_mel_backward: hop=%d n_mel=%d B=%d L=%d", d->hop_size, d->n_mel, B, L); return AMP_ERR_INVALID; }
    const int F = amp_mel_num_frames(d, L);
    if (F <= 0) { set_error("amp_mel_backward: no frames for L=%d", L); return AMP_ERR_INVALID; }
    const size_t nspec = (size_t)B * (d->n_fft / 2 + 1) * F;
    hipError_t e = launch_mel_backward(*d, lens_dev, B, L, F, window_dev, melbasis_dev, mel_linear_dev, mag_dev, re_dev, im_dev,
                                       grad_logmel_dev, spec_ws_dev, spec_ws_dev + nspec, frames_ws_dev, grad_wav_dev, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("amp_mel_backward: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

}  // extern "C"
