/*
 * vocoder_ref.c -- plain-C restatement of the arithmetic on the vocoder-inference hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/vocoder_oracle.py): an independent, loop-level statement
 * of the formulas in SURVEY.md Appendix C, accumulated in double, used to cross-check the
 * torch-functional oracle (tests/test_oracle_c.py) and as a slow but dependency-free checker.
 * The reference has no native code on this path; each function cites the reference call site
 * whose torch op it restates (paths relative to the reference tree).
 *
 * Layout everywhere: [B, C, T] contiguous, time fastest, fp32 in/out.
 */
#include <math.h>
#include <stddef.h>

/* nn.Conv1d(cin, cout, k, 1, dilation=d, padding=pad)      hifigan.py:24-52,93-100,204,216
 * y[b,o,t] = bias[o] + sum_i sum_j w[o,i,j] * x[b,i,t - pad + j*d],  x == 0 outside [0,T) */
void ref_conv1d(const float* x, const float* w, const float* bias, float* y, int B, int cin, int cout, int T,
                int k, int d, int pad) {
    const int Tout = T + 2 * pad - d * (k - 1);
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < cout; ++o)
            for (int t = 0; t < Tout; ++t) {
                double acc = bias ? (double)bias[o] : 0.0;
                for (int i = 0; i < cin; ++i)
                    for (int j = 0; j < k; ++j) {
                        const int s = t - pad + j * d;
                        if (s >= 0 && s < T)
                            acc += (double)w[((size_t)o * cin + i) * k + j] * (double)x[((size_t)b * cin + i) * T + s];
                    }
                y[((size_t)b * cout + o) * Tout + t] = (float)acc;
            }
}

/* nn.ConvTranspose1d(cin, cout, k, stride=u, padding=p), weight [cin, cout, k]   hifigan.py:176-186,207
 * y[b,o,n] = bias[o] + sum_i sum_{m,j : m*u - p + j = n} w[i,o,j] * x[b,i,m] */
void ref_conv_transpose1d(const float* x, const float* w, const float* bias, float* y, int B, int cin, int cout,
                          int T, int k, int u, int p) {
    const int Tout = (T - 1) * u - 2 * p + k;
    for (int b = 0; b < B; ++b)
        for (int o = 0; o < cout; ++o)
            for (int n = 0; n < Tout; ++n) {
                double acc = bias ? (double)bias[o] : 0.0;
                for (int m = 0; m < T; ++m) {
                    const int j = n + p - m * u;
                    if (j < 0 || j >= k) continue;
                    for (int i = 0; i < cin; ++i)
                        acc += (double)w[((size_t)i * cout + o) * k + j] * (double)x[((size_t)b * cin + i) * T + m];
                }
                y[((size_t)b * cout + o) * Tout + n] = (float)acc;
            }
}

/* F.leaky_relu(x, slope)                                     hifigan.py:95,97,206,215 */
void ref_leaky_relu(const float* x, float* y, size_t n, float slope) {
    for (size_t i = 0; i < n; ++i) y[i] = x[i] > 0.f ? x[i] : x[i] * slope;
}

/* torch.nn.utils.weight_norm (dim=0): w[r,:] = g[r] * v[r,:] / ||v[r,:]||_2     hifigan.py:23,157,176,199 */
void ref_fold_weight_norm(const float* g, const float* v, float* w, int d0, int inner) {
    for (int r = 0; r < d0; ++r) {
        double ss = 0.0;
        for (int i = 0; i < inner; ++i) ss += (double)v[(size_t)r * inner + i] * (double)v[(size_t)r * inner + i];
        const double sc = (double)g[r] / sqrt(ss);
        for (int i = 0; i < inner; ++i) w[(size_t)r * inner + i] = (float)((double)v[(size_t)r * inner + i] * sc);
    }
}

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* Activation1d(Snake[Beta]) = UpSample1d(2,12) -> snake -> DownSample1d(2,12)
 * act.py:31-36, resample.py:36-45,62-65, filter.py:92-99, snake.py:51-61,110-122.
 * a[c] = alpha (already exp'ed when log-scale), b[c] = beta (likewise; pass a for plain Snake). */
void ref_activation1d(const float* x, float* y, int B, int C, int T, const float* a, const float* b,
                      const float* fu, const float* fd) {
    for (int bc = 0; bc < B * C; ++bc) {
        const int c = bc % C;
        const float* xr = x + (size_t)bc * T;
        float* yr = y + (size_t)bc * T;
        for (int t = 0; t < T; ++t) {
            double acc = 0.0;
            for (int j = 0; j < 12; ++j) {
                /* sp = replicate_pad(s, 5, 6); y[t] = sum_j fd[j] * sp[2t + j] */
                const int n = clampi(2 * t + j - 5, 0, 2 * T - 1);
                /* u[n] = 2 * sum_m xp[m] * fu[n + 15 - 2m], xp = replicate_pad(x, 5, 5) */
                double u = 0.0;
                for (int m = 0; m < T + 10; ++m) {
                    const int q = n + 15 - 2 * m;
                    if (q < 0 || q >= 12) continue;
                    u += (double)xr[clampi(m - 5, 0, T - 1)] * (double)fu[q];
                }
                u *= 2.0;
                const double sn = sin(u * (double)a[c]);
                const double s = u + (1.0 / ((double)b[c] + 0.000000001)) * sn * sn;
                acc += (double)fd[j] * s;
            }
            yr[t] = (float)acc;
        }
    }
}

/* torch.stft(center=False, onesided) after reflect padding, as a direct DFT    utils/mel.py:145-166
 * wav [B, L]; pad = (n_fft - hop)/2 (pad_mode 0) or n_fft/2 (pad_mode 1, utils/stft.py:152-165);
 * re/im [B, n_fft/2+1, F].  window has n_fft entries. */
void ref_stft(const float* wav, int B, int L, int n_fft, int hop, int pad, const float* window, float* re, float* im) {
    const int bins = n_fft / 2 + 1;
    const int F = (L + 2 * pad - n_fft) / hop + 1;
    const double PI2 = 6.283185307179586476925286766559;
    for (int b = 0; b < B; ++b)
        for (int f = 0; f < F; ++f)
            for (int k = 0; k < bins; ++k) {
                double sr = 0.0, si = 0.0;
                for (int n = 0; n < n_fft; ++n) {
                    int s = f * hop + n - pad;
                    if (s < 0) s = -s;
                    if (s >= L) s = 2 * (L - 1) - s;
                    const double v = (double)wav[(size_t)b * L + s] * (double)window[n];
                    const double ang = PI2 * (double)((long long)k * n % n_fft) / (double)n_fft;
                    sr += v * cos(ang);
                    si -= v * sin(ang);
                }
                re[((size_t)b * bins + k) * F + f] = (float)sr;
                im[((size_t)b * bins + k) * F + f] = (float)si;
            }
}

/* log(clamp(melbasis @ sqrt(re^2 + im^2 + eps), clip))                        utils/mel.py:165-169,10-12 */
void ref_logmel(const float* re, const float* im, int B, int bins, int F, const float* basis, int n_mel, float eps,
                float clip, float* mel) {
    for (int b = 0; b < B; ++b)
        for (int m = 0; m < n_mel; ++m)
            for (int f = 0; f < F; ++f) {
                double acc = 0.0;
                for (int k = 0; k < bins; ++k) {
                    const size_t o = ((size_t)b * bins + k) * F + f;
                    const float mag = sqrtf(re[o] * re[o] + im[o] * im[o] + eps);
                    acc += (double)basis[(size_t)m * bins + k] * (double)mag;
                }
                const float v = (float)acc;
                mel[((size_t)b * n_mel + m) * F + f] = logf(v > clip ? v : clip);
            }
}

/* fp32 -> signed 16-bit PCM as the reference writes wavs: save_audio utils/io.py:68-76 ->
 * torchaudio.save(encoding="PCM_S", bits_per_sample=16).  torchaudio 2.0.2 sox_io (effects_chain.cpp,
 * tensor_input_drain: double(x) * 2^31, clamp to int32, truncate) then libsox 14.4.2 SOX_SAMPLE_TO_SIGNED_16BIT
 * (sox.h: saturate above INT32_MAX - 2^15, else add 2^15 and drop 16 bits).  Integer restatement in plain C;
 * parity unpinned (neither library is installed), see oracle/pcm16.py. */
#include <stdint.h>
void ref_pcm16(const float* x, int16_t* y, size_t n) {
    for (size_t i = 0; i < n; ++i) {
        const double v = (double)x[i] * 2147483648.0;
        int32_t d;
        if (v != v) d = INT32_MIN;                 /* NaN: what x86's cvttsd2si yields for torch's undefined cast */
        else if (v >= 2147483647.0) d = INT32_MAX;
        else if (v <= -2147483648.0) d = INT32_MIN;
        else d = (int32_t)v;                       /* C conversion truncates toward zero */
        if (d > INT32_MAX - (1 << 15)) y[i] = 32767;
        else y[i] = (int16_t)((((uint32_t)d ^ 0x80000000u) + (1u << 15)) >> 16 ^ 0x8000u);
    }
}
