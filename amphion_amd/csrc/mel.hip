// Mel / STFT front end on gfx950: framing (reflect padding) * window -> radix-2 Stockham FFT in LDS ->
// |X| -> mel filterbank -> log, one workgroup per frame, nothing but the final features leaves the CU.
//
// Replaces utils/mel.py:20-170 (torch.stft + sqrt + matmul + log as 5 separate tensor ops) and the
// conv1d-with-Fourier-basis STFT of utils/stft.py:152-181,259-278 (TacotronSTFT).
#include "amp_internal.h"

namespace amp {

__device__ __forceinline__ int reflect_index(int s, int L) {
    // torch reflect padding (no edge repeat); pad < L is checked on the host
    if (s < 0) s = -s;
    if (s >= L) s = 2 * (L - 1) - s;
    return s;
}

__global__ __launch_bounds__(256) void mel_kernel(const float* __restrict__ wav, const int* __restrict__ lens, int L, int F,
                                                  int n_fft, int log2n, int hop, int pad, int n_mel, float mag_eps,
                                                  float log_clip,
                                                  const float* __restrict__ window, const float* __restrict__ melbasis,
                                                  float* __restrict__ mel, float* __restrict__ mag,
                                                  float* __restrict__ re_out, float* __restrict__ im_out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
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

    for (int n = tid; n < n_fft; n += 256) {
        const int s = reflect_index(f * hop + n - pad, Li);
        buf0[n] = make_float2(wb[s] * window[n], 0.f);
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

hipError_t launch_mel(const amp_mel_desc& d, const float* wav, const int* lens, int B, int L, int F, const float* window,
                      const float* melbasis, float* mel, float* mag, float* re, float* im, hipStream_t stream) {
    int log2n = 0;
    while ((1 << log2n) < d.n_fft) ++log2n;
    const int pad = d.pad_mode == 0 ? (d.n_fft - d.hop_size) / 2 : d.n_fft / 2;
    const size_t lds = (size_t)(2 * d.n_fft + d.n_fft / 2) * sizeof(float2) + (size_t)(d.n_fft / 2 + 1) * sizeof(float);
    dim3 grid((unsigned)((size_t)B * F));
    hipLaunchKernelGGL(mel_kernel, grid, dim3(256), lds, stream, wav, lens, L, F, d.n_fft, log2n, d.hop_size, pad,
                       mel ? d.n_mel : 0, d.mag_eps, d.log_clip, window, melbasis, mel, mag, re, im);
    return hipGetLastError();
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
    int log2n = 0;
    while ((1 << log2n) < d.n_fft) ++log2n;
    const size_t lds = (size_t)(2 * d.n_fft + d.n_fft / 2) * sizeof(float2);
    const float scale = mode == 0 ? (float)d.n_fft / (float)d.hop_size : 1.0f;
    hipLaunchKernelGGL(istft_frames_kernel, dim3((unsigned)((size_t)B * F)), dim3(256), lds, stream, a, b, mode == 0 ? 1 : 0, F,
                       d.n_fft, log2n, 1.0f / scale, window, frames);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const int crop = mode == 0 ? d.n_fft / 2 : (d.win_size - d.hop_size) / 2;
    const int Lout = mode == 0 ? d.hop_size * (F - 1) : d.hop_size * (F - 1) + d.win_size - 2 * crop;
    if (Lout <= 0) return hipSuccess;
    hipLaunchKernelGGL(istft_ola_kernel, dim3((unsigned)((Lout + 255) / 256), (unsigned)B), dim3(256), 0, stream, frames, wss, F,
                       d.n_fft, d.hop_size, crop, Lout, scale, mode == 0 ? 1.17549435e-38f : -1.0f, wav);
    return hipGetLastError();
}

}  // namespace amp

using namespace amp;

extern "C" {

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

int amp_mel_forward_ragged(const amp_mel_desc* d, const float* wav_dev, const int32_t* lens_dev, int B, int L,
                           const float* window_dev, const float* melbasis_dev, float* mel_dev, float* mag_dev,
                           float* re_dev, float* im_dev, void* stream) {
    if (!d || !wav_dev || !window_dev) { set_error("amp_mel_forward: null argument"); return AMP_ERR_INVALID; }
    if (d->n_fft < 64 || d->n_fft > 4096 || (d->n_fft & (d->n_fft - 1)) != 0) {
        set_error("amp_mel_forward: n_fft=%d must be a power of two in [64, 4096]", d->n_fft);
        return AMP_ERR_UNSUPPORTED;
    }
    if (d->hop_size <= 0 || B <= 0 || L <= 0) { set_error("amp_mel_forward: hop=%d B=%d L=%d", d->hop_size, B, L); return AMP_ERR_INVALID; }
    const int pad = d->pad_mode == 0 ? (d->n_fft - d->hop_size) / 2 : d->n_fft / 2;
    if (pad >= L) { set_error("amp_mel_forward: reflect padding %d needs more than %d samples", pad, L); return AMP_ERR_INVALID; }
    if (mel_dev && (d->n_mel <= 0 || !melbasis_dev)) { set_error("amp_mel_forward: mel output needs n_mel > 0 and a mel basis"); return AMP_ERR_INVALID; }
    const int F = amp_mel_num_frames(d, L);
    if (F <= 0) { set_error("amp_mel_forward: no frames for L=%d", L); return AMP_ERR_INVALID; }
    hipError_t e = launch_mel(*d, wav_dev, lens_dev, B, L, F, window_dev, melbasis_dev, mel_dev, mag_dev, re_dev, im_dev, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("amp_mel_forward: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_istft_forward(const amp_mel_desc* d, const float* mag_dev, const float* phase_dev, int B, int F,
                      const float* window_dev, const float* wss_dev, float* frames_ws_dev, float* wav_dev, void* stream) {
    if (!d || !mag_dev || !phase_dev || !window_dev || !wss_dev || !frames_ws_dev || !wav_dev) { set_error("amp_istft_forward: null argument"); return AMP_ERR_INVALID; }
    if (d->n_fft < 64 || d->n_fft > 4096 || (d->n_fft & (d->n_fft - 1)) != 0) {
        set_error("amp_istft_forward: n_fft=%d must be a power of two in [64, 4096]", d->n_fft);
        return AMP_ERR_UNSUPPORTED;
    }
    if (d->hop_size <= 0 || d->hop_size > d->n_fft || B <= 0 || F <= 1) { set_error("amp_istft_forward: hop=%d B=%d F=%d", d->hop_size, B, F); return AMP_ERR_INVALID; }
    hipError_t e = launch_istft(*d, 0, mag_dev, phase_dev, B, F, window_dev, wss_dev, frames_ws_dev, wav_dev, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("amp_istft_forward: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

int amp_istft_same(const amp_mel_desc* d, const float* re_dev, const float* im_dev, int B, int F, const float* window_dev,
                   const float* envelope_dev, float* frames_ws_dev, float* wav_dev, void* stream) {
    if (!d || !re_dev || !im_dev || !window_dev || !envelope_dev || !frames_ws_dev || !wav_dev) { set_error("amp_istft_same: null argument"); return AMP_ERR_INVALID; }
    if (d->n_fft < 64 || d->n_fft > 4096 || (d->n_fft & (d->n_fft - 1)) != 0 || d->win_size != d->n_fft) {
        set_error("amp_istft_same: n_fft=%d must be a power of two in [64, 4096] and equal win_size=%d", d->n_fft, d->win_size);
        return AMP_ERR_UNSUPPORTED;
    }
    if (d->hop_size <= 0 || d->hop_size > d->n_fft || ((d->win_size - d->hop_size) & 1) || B <= 0 || F <= 0) { set_error("amp_istft_same: hop=%d B=%d F=%d", d->hop_size, B, F); return AMP_ERR_INVALID; }
    hipError_t e = launch_istft(*d, 1, re_dev, im_dev, B, F, window_dev, envelope_dev, frames_ws_dev, wav_dev, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("amp_istft_same: %s", hipGetErrorString(e)); return AMP_ERR_HIP; }
    return AMP_OK;
}

}  // extern "C"
