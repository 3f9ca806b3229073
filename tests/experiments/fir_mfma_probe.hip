// Probe for profiles/negative_kernels/act1d_mfma.h (round 5; result: profiles/r5_b_fir_mfma.txt): Activation1d with both FIRs as split-f16 Toeplitz products on the matrix pipe.
//   1. f16 MFMA operands below 2^-14 (subnormal halves of a split) are NOT flushed
//   2. act_run_mfma on P-layout runs == fp64 Activation1d of the same signal (interior columns; halos given)
//   3. time per run and wave at two waves per SIMD, against the fp32 VALU chains of round 4's act_run
// hipcc --offload-arch=gfx950 -O3 -std=c++17 -I profiles/negative_kernels -I amphion_amd/csrc -I include tests/experiments/fir_mfma_probe.hip -o tests/experiments/fir_mfma_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "act1d_mfma.h"

using namespace amp;
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
#define AMP_PT(c) ((c) >> 4)
#define AMP_PR(c) ((c) & 15)

// ---- round 4's in-register activation (ampb_f16x3.hip act_run), kept here as the timing baseline ----
__device__ __forceinline__ void act_run_valu(f32x16 (&v)[4], const float (&hl)[5], const float (&hr)[5], const float a, const float invb,
                                             const float (&fu2)[12], const float (&fd)[12]) {
    f32x2 P[72];
#pragma unroll
    for (int g = 0; g < 18; ++g) {
        f32x2 uv[4], xa[4], sv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) uv[q] = pk_splat(0.f);
#pragma unroll
        for (int k = 0; k < 6; ++k) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = 4 * g + q - k;
                const float xv = c < 0 ? hl[c + 5 < 0 ? 0 : c + 5] : (c > 63 ? hr[c - 64 > 4 ? 4 : c - 64] : v[AMP_PT(c & 63)][AMP_PR(c & 63)]);
                uv[q] = pk_fma(pk_splat(xv), (f32x2){fu2[2 * k], fu2[2 * k + 1]}, uv[q]);
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) xa[q] = uv[q] * a;
        snake_sin2_pk4(xa, sv);
#pragma unroll
        for (int q = 0; q < 4; ++q) P[4 * g + q] = pk_fma(pk_splat(invb), sv[q], uv[q]);
        {
            f32x2 oa[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) oa[e] = pk_splat(0.f);
#pragma unroll
            for (int m = 0; m < 6; ++m) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int t = 4 * g - 5 + e;
                    if (t >= 0 && t < 64) oa[e] = pk_fma((f32x2){fd[2 * m], fd[2 * m + 1]}, P[(t + m) < 72 ? (t + m) : 71], oa[e]);
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = 4 * g - 5 + e;
                if (t >= 0 && t < 64) v[AMP_PT(t & 63)][AMP_PR(t & 63)] = oa[e].x + oa[e].y;
            }
        }
        if (g >= 1 && g < 17) {
            float t0 = v[AMP_PT((4 * g + 4) & 63)][AMP_PR((4 * g + 4) & 63)], t1 = v[AMP_PT((4 * g + 5) & 63)][AMP_PR((4 * g + 5) & 63)];
            float t2 = v[AMP_PT((4 * g + 6) & 63)][AMP_PR((4 * g + 6) & 63)], t3 = v[AMP_PT((4 * g + 7) & 63)][AMP_PR((4 * g + 7) & 63)];
            float o0 = v[AMP_PT((4 * g - 5) & 63)][AMP_PR((4 * g - 5) & 63)], o1 = v[AMP_PT((4 * g - 4) & 63)][AMP_PR((4 * g - 4) & 63)];
            float o2 = v[AMP_PT((4 * g - 3) & 63)][AMP_PR((4 * g - 3) & 63)], o3 = v[AMP_PT((4 * g - 2) & 63)][AMP_PR((4 * g - 2) & 63)];
            asm volatile("" : "+v"(t0), "+v"(t1), "+v"(t2), "+v"(t3), "+v"(o0), "+v"(o1), "+v"(o2), "+v"(o3));
            v[AMP_PT((4 * g + 4) & 63)][AMP_PR((4 * g + 4) & 63)] = t0; v[AMP_PT((4 * g + 5) & 63)][AMP_PR((4 * g + 5) & 63)] = t1;
            v[AMP_PT((4 * g + 6) & 63)][AMP_PR((4 * g + 6) & 63)] = t2; v[AMP_PT((4 * g + 7) & 63)][AMP_PR((4 * g + 7) & 63)] = t3;
            v[AMP_PT((4 * g - 5) & 63)][AMP_PR((4 * g - 5) & 63)] = o0; v[AMP_PT((4 * g - 4) & 63)][AMP_PR((4 * g - 4) & 63)] = o1;
            v[AMP_PT((4 * g - 3) & 63)][AMP_PR((4 * g - 3) & 63)] = o2; v[AMP_PT((4 * g - 2) & 63)][AMP_PR((4 * g - 2) & 63)] = o3;
        }
    }
}

// 1. subnormal operands: D = A * B with A = 2^-20 (an f16 subnormal) in every entry, B = 1.0: flushed -> 0, kept -> 16 * 2^-20
__global__ void denorm_kernel(float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f; b[i] = (_Float16)1.0f; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
    c = (f32x16){0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, c, 0, 0, 0);
    out[64 + threadIdx.x] = c[0];
}

// X: [nwaves][32 channels][138 columns] (column 0 = signal column -5); Y: [nwaves][32][128]
template <int MODE>   // 0: MFMA form, 1: VALU form
__global__ __launch_bounds__(256, 2) void act_kernel(const float* X, float* Y, const float* fu2p, const float* fdp, const uint4* tabg, float a,
                                                     float invb, int iters, float* rmax) {
    __shared__ uint4 tab[kActTabFrags * 64];
    for (int i = threadIdx.x; i < kActTabFrags * 64; i += blockDim.x) tab[i] = tabg[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int m = lane & 31, h = lane >> 5;
    const float* row = X + ((size_t)wave * 32 + m) * 138 + 64 * h;
    f32x16 v[4];
    float hl[5], hr[5];
    for (int i = 0; i < 5; ++i) { hl[i] = row[i] * 16.f; hr[i] = row[69 + i] * 16.f; }
    for (int c = 0; c < 64; ++c) v[AMP_PT(c)][AMP_PR(c)] = row[5 + c] * 16.f;
    float fu2[12], fd[12];
    for (int k = 0; k < 12; ++k) { fu2[k] = fu2p[k]; fd[k] = fdp[k]; }
    float range_max = 0.f;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) act_run_mfma(v, hl, hr, a * 0.0625f, invb * 16.f, tab + lane, range_max);
        else act_run_valu(v, hl, hr, a * 0.0625f, invb * 16.f, fu2, fd);
        if (it + 1 < iters) {       // keep the values bounded for the timing loop
            for (int i = 0; i < 5; ++i) { hl[i] = v[0][i]; hr[i] = v[3][11 + i]; }
        }
    }
    float* out = Y + ((size_t)wave * 32 + m) * 128 + 64 * h;
    for (int c = 0; c < 64; ++c) out[c] = v[AMP_PT(c)][AMP_PR(c)] * 0.0625f;
    if (rmax && lane == 0) rmax[wave] = range_max;
}

static void ref_act(const double* x /* 138: col -5 .. 132 */, double* y /* 128 */, const float* fu2, const float* fd, double a, double invb) {
    // columns 0 .. 127 of the signal; u[v], v = 2 Q + p, Q = 0 .. 127 + 5 + ..., from x[Q - k] (x index = column + 5)
    std::vector<double> s(2 * 140 + 16, 0.0);
    for (int Q = 0; Q < 133; ++Q)
        for (int p = 0; p < 2; ++p) {
            double u = 0;
            for (int k = 0; k < 6; ++k) u += x[Q - k + 5] * (double)fu2[2 * k + p];    // X[Q - k], stored at +5
            s[2 * Q + p] = u + invb * sin(a * u) * sin(a * u);
        }
    for (int t = 0; t < 128; ++t) {
        double acc = 0;
        for (int j = 0; j < 12; ++j) acc += (double)fd[j] * s[2 * t + j];
        y[t] = acc;
    }
}

int main() {
    // Kaiser-sinc taps of BigVGAN's Activation1d (filter.py:30-61, cutoff 0.25, half width 0.3, 12 taps), computed as the reference does
    double w[12], sum = 0;
    const double A = 2.285 * 5 * M_PI * (4 * 0.3) + 7.95, beta = 0.1102 * (A - 8.7);
    auto bessel0 = [](double x) { double s = 1, t = 1; for (int k = 1; k < 40; ++k) { t *= (x / (2 * k)) * (x / (2 * k)); s += t; } return s; };
    for (int n = 0; n < 12; ++n) {
        const double r = 2.0 * n / 11 - 1, win = bessel0(beta * sqrt(1 - r * r)) / bessel0(beta), tt = n - 6 + 0.5, z = 2 * 0.25 * tt;
        w[n] = 2 * 0.25 * win * (z == 0 ? 1 : sin(M_PI * z) / (M_PI * z));
        sum += w[n];
    }
    float fu2[12], fd[12];
    for (int n = 0; n < 12; ++n) { fd[n] = (float)(w[n] / sum); fu2[n] = 2.f * fd[n]; }
    std::vector<_Float16> tab_h((size_t)kActTabFrags * 64 * 8);
    act_mfma_table(fu2, fd, tab_h.data());

    const int NWG = 512, NW = NWG * 4;
    std::vector<float> X((size_t)NW * 32 * 138), Y((size_t)NW * 32 * 128);
    srand(5);
    for (auto& v : X) v = 3.f * ((float)rand() / RAND_MAX * 2.f - 1.f) * ((float)rand() / RAND_MAX);
    float *dX, *dY, *dfu, *dfd, *dout, *drm;
    uint4* dtab;
    hipMalloc(&dX, X.size() * 4); hipMalloc(&dY, Y.size() * 4); hipMalloc(&dfu, 48); hipMalloc(&dfd, 48); hipMalloc(&dout, 128 * 4);
    hipMalloc(&dtab, kActTabBytes); hipMalloc(&drm, NW * 4);
    hipMemcpy(dX, X.data(), X.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dfu, fu2, 48, hipMemcpyHostToDevice); hipMemcpy(dfd, fd, 48, hipMemcpyHostToDevice);
    hipMemcpy(dtab, tab_h.data(), kActTabBytes, hipMemcpyHostToDevice);

    float o[128];
    denorm_kernel<<<1, 64>>>(dout);
    hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
    printf("1 subnormal f16 MFMA operands: A-side %g, B-side %g (kept: %g)  %s\n", o[0], o[64], 16 * 9.5367431640625e-07,
           (o[0] != 0.f && o[64] != 0.f) ? "kept" : "FLUSHED");

    const float a = 1.3f, invb = 0.8f;
    int bad = 0;
#ifdef AMP_ACT_DEBUG
    {
        float* ddbg; hipMalloc(&ddbg, 64 * 64 * 4); hipMemset(ddbg, 0, 64 * 64 * 4);
        hipMemcpyToSymbol(HIP_SYMBOL(amp_act_dbg), &ddbg, sizeof(ddbg));
        act_kernel<0><<<1, 64>>>(dX, dY, dfu, dfd, dtab, a, invb, 1, drm);
        hipDeviceSynchronize();
        std::vector<float> dbg(64 * 64);
        hipMemcpy(dbg.data(), ddbg, 64 * 64 * 4, hipMemcpyDeviceToHost);
        for (int lane : {0, 1, 32}) {
            const float* o = &dbg[lane * 64];
            const int m = lane & 31, h = lane >> 5;
            const float* row = &X[((size_t)0 * 32 + m) * 138 + 64 * h];
            printf("lane %d\n  Z   :", lane); for (int r = 0; r < 8; ++r) printf(" %.6f", o[48 + r]);
            printf("\n  16x :"); for (int r = 0; r < 8; ++r) printf(" %.6f", row[r] * 16.f);
            printf("\n  zh0 :"); for (int r = 0; r < 8; ++r) printf(" %.6f", o[16 + r]);
            printf("\n  zl0 :"); for (int r = 0; r < 8; ++r) printf(" %.6f", o[24 + r]);
            printf("\n  h+l :"); for (int r = 0; r < 8; ++r) printf(" %.6f", o[16 + r] + o[24 + r]);
            printf("\n  u   :"); for (int r = 0; r < 16; ++r) printf(" %.5f", o[r]);
            printf("\n  uref:");
            for (int v = 0; v < 16; ++v) { double u = 0; for (int k = 0; k < 6; ++k) u += 16.0 * row[(v >> 1) + 5 - k] * fu2[2 * k + (v & 1)]; printf(" %.5f", u); }
            printf("\n");
        }
        float* nul = nullptr; hipMemcpyToSymbol(HIP_SYMBOL(amp_act_dbg), &nul, sizeof(nul));
    }
#endif
    for (int mode = 0; mode < 2; ++mode) {
        if (mode == 0) act_kernel<0><<<NWG, 256>>>(dX, dY, dfu, dfd, dtab, a, invb, 1, drm);
        else act_kernel<1><<<NWG, 256>>>(dX, dY, dfu, dfd, dtab, a, invb, 1, nullptr);
        if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
        hipMemcpy(Y.data(), dY, Y.size() * 4, hipMemcpyDeviceToHost);
        double emax = 0, ymax = 0;
        double epos[128] = {0};
        for (int wv = 0; wv < 64; ++wv)
            for (int m = 0; m < 32; ++m) {
                double x[138], y[128];
                for (int i = 0; i < 138; ++i) x[i] = X[((size_t)wv * 32 + m) * 138 + i];
                // each half is an independent run with its own halos: run h covers columns 64 h .. 64 h + 63 of the same signal, so
                // one reference over the 128 columns serves both (x[-5 ..] / x[.. 132] are the outer halos, the seam is shared)
                ref_act(x, y, fu2, fd, a, invb);
                for (int t = 0; t < 128; ++t) {
                    const double e = fabs(Y[((size_t)wv * 32 + m) * 128 + t] - y[t]);
                    emax = e > emax ? e : emax;
                    epos[t] = e > epos[t] ? e : epos[t];
                    ymax = fabs(y[t]) > ymax ? fabs(y[t]) : ymax;
                }
            }
        printf("2 %s form vs fp64: max |err| %.3g (|y| max %.3g)\n", mode == 0 ? "MFMA" : "VALU", emax, ymax);
        if (emax > (mode == 0 ? 2e-6 : 1e-6)) {
            bad = 1;
            int nb0 = 0, nb64 = 0, first = -1;
            for (int wv = 0; wv < 64; ++wv)
                for (int m = 0; m < 32; ++m) {
                    double x[138], y[128];
                    for (int i = 0; i < 138; ++i) x[i] = X[((size_t)wv * 32 + m) * 138 + i];
                    ref_act(x, y, fu2, fd, a, invb);
                    if (fabs(Y[((size_t)wv * 32 + m) * 128] - y[0]) > 1e-5) { ++nb0; if (first < 0) first = wv * 32 + m; }
                    if (fabs(Y[((size_t)wv * 32 + m) * 128 + 64] - y[64]) > 1e-5) ++nb64;
                }
            printf("  rows (of 2048) off by > 1e-5 at column 0: %d, at column 64: %d; first bad row %d\n", nb0, nb64, first);
            printf("  max |err| by column:");
            for (int t = 0; t < 128; ++t) printf(" %.1e", epos[t]);
            printf("\n");
        }
    }
    {
        std::vector<float> rm(NW);
        hipMemcpy(rm.data(), drm, NW * 4, hipMemcpyDeviceToHost);
        printf("  range_max (16 x |operand|) of wave 0: %g\n", rm[0]);
    }
    for (int mode = 0; mode < 2; ++mode) {
        const int iters = 200;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) act_kernel<0><<<NWG, 256>>>(dX, dY, dfu, dfd, dtab, a, invb, iters, nullptr);
            else act_kernel<1><<<NWG, 256>>>(dX, dY, dfu, dfd, dtab, a, invb, iters, nullptr);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        // 512 workgroups of 4 waves on 256 CUs: two waves per SIMD, one round
        printf("3 %s form: %.3f ms for %d runs per wave at 2 waves / SIMD = %.2f us per run\n", mode == 0 ? "MFMA" : "VALU", best, iters, best * 1e3 / iters);
    }
    printf(bad ? "PROBE FAIL\n" : "PROBE PASS\n");
    return bad;
}
