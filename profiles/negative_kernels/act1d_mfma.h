// NEGATIVE RESULT (round 5, profiles/r5_b_fir_mfma.txt): correct, 45 % slower than the fp32 VALU chains it was to replace.  Not built into the library.
// Activation1d with both 12-tap FIRs on the matrix pipe -- an in-register form for ampb_f16x3.hip.
//
// Activation1d = 2x up-sampling FIR -> Snake -> 2x down-sampling FIR (modules/anti_aliasing/act.py:31-36, resample.py:36-65,
// filter.py:92-99) and its filters are the SAME for every channel.  In the kernel's P layout a lane owns 64 consecutive columns of
// one channel, so eight consecutive columns of a lane ARE a B fragment of v_mfma_f32_32x32x16_f16 (lane (m, h) supplies
// k = 8 h .. 8 h + 7 of column m), the banded Toeplitz matrix of the filter is a CONSTANT A fragment, and the C layout hands 16
// consecutive outputs back to the same lane -- register to register.  The two half-waves of a lane pair hold different runs, so
// the 32 x 16 Toeplitz tile is block diagonal (rows of half h use k = 8 h .. 8 h + 7 only); half of every MFMA multiplies zeros,
// on a matrix pipe that idles 70-85 % of this kernel's time (profiles/r4_c3_sq_counters_in_forward.txt).
//
// Arithmetic: the split-f16 form of the convs.  x = hi + lo f16 after the exact x16 every value inside the block
// carries, taps = hi + lo f16, products hi*hi + lo*hi + hi*lo into the fp32 accumulator (the dropped lo*lo is 2^-22 relative).
// NOT bit-identical to act1d_kernel's fp32 chains: |difference| ~ 1e-7 per activation (tests/experiments/ampb_fir_mfma_model.py:
// BigVGAN-base end to end 7e-7 against fp64, the reference's own fp32 run 6e-7).
//
// Index algebra, per lane and run (all values 16 x true):
//   Z[i]  = X[i - 5],  i = 0 .. 79        X[-5 .. -1] = hl, X[0 .. 63] = the run, X[64 .. 68] = hr, zero beyond
//   u[v]  = sum_k Z[Q + 5 - k] * fu2[2 k + p],   v = 2 Q + p   (v = n + 5 of resample.py's up-sampled index n; fu2 = 2 * taps)
//   s[v]  = u[v] + (16 / b) * sin^2(u[v] * a / 16)
//   y[t]  = sum_j fd[j] * s[2 t + j],  j = 0 .. 11,  t = 0 .. 63
// K blocks of 8: Z block j = Z[8 j ..], s block i = s[8 i ..].  Up block b (16 values v = 16 b + r) = Toeplitz(d = 0) * Z_b +
// Toeplitz(d = 1) * Z_{b+1}; output block B (t = 16 B + r) = sum_{e = 0 .. 5} Toeplitz_dn(e) * s_{4 B + e}.
// 9 up blocks x 2 x 3 + 4 output blocks x 6 x 3 = 126 MFMAs per run instead of ~880 packed FMAs (1 700 issue slots).
#pragma once
#include "act1d_math.h"
#include "amp_internal.h"

namespace amp {

constexpr int kActTabFrags = 16;                       // uint4 [16][64]: up (d = 0 hi, lo; d = 1 hi, lo), down (e = 0 .. 5: hi, lo)
constexpr int kActTabBytes = kActTabFrags * 64 * 16;

// Host: the constant A fragments of one Activation1d.  Lane l = (row = l % 32, hk = l / 32) holds A[row][8 hk .. 8 hk + 7]; row
// 8 a + 4 h + j is output register r = 4 a + j of the lanes of half h (the MFMA C layout), so the fragment is zero unless hk == h.
inline void act_mfma_table(const float* fu2, const float* fd, _Float16* out /* [16][64][8] */) {
    for (int f = 0; f < kActTabFrags; ++f)
        for (int l = 0; l < 64; ++l) {
            const int row = l & 31, hk = l >> 5;
            const int r = 4 * (row >> 3) + (row & 3), hrow = (row >> 2) & 1;
            const bool lo = (f & 1) != 0;
            for (int kk = 0; kk < 8; ++kk) {
                float tap = 0.f;
                if (hk == hrow) {
                    if (f < 4) {
                        const int d = f >> 1;
                        const int k = (r >> 1) + 5 - 8 * d - kk;
                        if (k >= 0 && k <= 5) tap = fu2[2 * k + (r & 1)];
                    } else {
                        const int e = (f - 4) >> 1;
                        const int j = 8 * e + kk - 2 * r;
                        if (j >= 0 && j <= 11) tap = fd[j];
                    }
                }
                const _Float16 hi = (_Float16)tap;
                out[((size_t)f * 64 + l) * 8 + kk] = lo ? (_Float16)(tap - (float)hi) : hi;
            }
        }
}

#if defined(__HIP_DEVICE_COMPILE__) || defined(__HIPCC__)

typedef float act_f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 act_f16x8 __attribute__((ext_vector_type(8)));
union ActFrag {
    uint4 u;
    act_f16x8 h;
};

// split4_f16 (amp_internal.h) for operands that go STRAIGHT into an MFMA.  The four v_fma_mix* are one asm statement, so their order is
// fixed (both low halves, then both high halves: a partial write of a register is never followed at once by the other half's), and it
// ends in s_nop 1: on gfx950 an MFMA that reads a VGPR needs two wait states after the VALU instruction that wrote it.  hipcc inserts
// them for instructions it knows, it cannot see into an asm: the first version of this header used split4_f16 and 3 % of the waves
// read a stale half of a fragment register (tests/experiments/fir_mfma_probe.hip, profiles/r5_b_fir_mfma_probe.txt: the MFMA sat two
// instructions behind the last v_fma_mixhi_f16; in the conv kernels the halves go to LDS stores, which interlock).
__device__ __forceinline__ void act_split4(amp_f32x2 v01, amp_f32x2 v23, uint2& h, uint2& l) {
    asm("" : "+v"(v01));
    asm("" : "+v"(v23));
    const amp_f16x2 h01 = __builtin_convertvector(v01, amp_f16x2), h23 = __builtin_convertvector(v23, amp_f16x2);
    h.x = __builtin_bit_cast(unsigned, h01); h.y = __builtin_bit_cast(unsigned, h23);
    asm("v_fma_mixlo_f16 %0, %2, 1.0, -%6 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixlo_f16 %1, %4, 1.0, -%7 op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %3, 1.0, -%6 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %1, %5, 1.0, -%7 op_sel:[0,0,1] op_sel_hi:[0,0,1]\n\t"
        "s_nop 1"
        : "=&v"(l.x), "=&v"(l.y)
        : "v"(v01.x), "v"(v01.y), "v"(v23.x), "v"(v23.y), "v"(h.x), "v"(h.y));
}

// eight fp32 values -> one K block: hi and lo fragments (v_cvt_pk_f16_f32 + v_fma_mix per value, 1.5 instructions each)
__device__ __forceinline__ void act_split8(const float (&z)[8], ActFrag& hi, ActFrag& lo) {
    uint2 h0, l0, h1, l1;
    act_split4((amp_f32x2){z[0], z[1]}, (amp_f32x2){z[2], z[3]}, h0, l0);
    act_split4((amp_f32x2){z[4], z[5]}, (amp_f32x2){z[6], z[7]}, h1, l1);
    hi.u = make_uint4(h0.x, h0.y, h1.x, h1.y);
    lo.u = make_uint4(l0.x, l0.y, l1.x, l1.y);
}

#ifdef AMP_ACT_DEBUG
__device__ float* amp_act_dbg = nullptr;
#endif
#define AMP_ACT_MFMA(A, B, C) __builtin_amdgcn_mfma_f32_32x32x16_f16((A).h, (B).h, (C), 0, 0, 0)

// v: the run in the P layout (v[c >> 4][c & 15] = 16 * x[c]), replaced by 16 * Activation1d(x); hl / hr: the five columns either side
// (x16 as well); a16 = alpha / 16, invb16 = 16 / (beta + 1e-9); tab: this lane's slice of the fragment table (tab[f * 64], LDS or
// global); range_max: running maximum of |f16 operand| (the range guard: beyond 65504 the split left the f16 range).
__device__ __forceinline__ void act_run_mfma(act_f32x16 (&v)[4], const float (&hl)[5], const float (&hr)[5], const float a16,
                                             const float invb16, const uint4* tab, float& range_max) {
    auto Z = [&](int i) __attribute__((always_inline)) -> float {
        return i < 5 ? hl[i < 0 ? 0 : i] : (i < 69 ? v[((i - 5) & 63) >> 4][(i - 5) & 15] : (i < 74 ? hr[i - 69] : 0.f));
    };
    auto zblock = [&](int j, ActFrag& hi, ActFrag& lo) __attribute__((always_inline)) {
        float z[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            z[i] = Z(8 * j + i);
            if (8 * j + i < 74) range_max = __builtin_fmaxf(range_max, __builtin_fabsf(z[i]));
        }
        act_split8(z, hi, lo);
    };
    ActFrag au[4];
#pragma unroll
    for (int f = 0; f < 4; ++f) au[f].u = tab[f * 64];
    const act_f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    act_f32x16 y[4];
    ActFrag zh0, zl0;
    zblock(0, zh0, zl0);
#pragma unroll
    for (int b = 0; b < 9; ++b) {
        ActFrag zh1, zl1;
        zblock(b + 1, zh1, zl1);
        // ---- up-sampling FIR: 16 values of u ----
        act_f32x16 d = AMP_ACT_MFMA(au[0], zh0, zero);
        d = AMP_ACT_MFMA(au[0], zl0, d);
        d = AMP_ACT_MFMA(au[1], zh0, d);
        d = AMP_ACT_MFMA(au[2], zh1, d);
        d = AMP_ACT_MFMA(au[2], zl1, d);
        d = AMP_ACT_MFMA(au[3], zh1, d);
#ifdef AMP_ACT_DEBUG
        if (b == 0 && amp_act_dbg) {
            float* o = amp_act_dbg + (threadIdx.x & 63) * 64;
            for (int r = 0; r < 16; ++r) o[r] = d[r];
            for (int r = 0; r < 8; ++r) { o[16 + r] = (float)zh0.h[r]; o[24 + r] = (float)zl0.h[r]; o[32 + r] = (float)zh1.h[r]; o[40 + r] = (float)zl1.h[r]; }
            for (int r = 0; r < 8; ++r) o[48 + r] = Z(r);
        }
#endif
        zh0 = zh1;
        zl0 = zl1;
        // ---- Snake ----
        float s[16];
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            f32x2 uv[4], xa[4], sv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uv[q] = (f32x2){d[8 * g + 2 * q], d[8 * g + 2 * q + 1]};
                xa[q] = uv[q] * a16;
            }
            snake_sin2_pk4(xa, sv);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x2 sq = pk_fma(pk_splat(invb16), sv[q], uv[q]);
                s[8 * g + 2 * q] = sq.x;
                s[8 * g + 2 * q + 1] = sq.y;
                range_max = __builtin_fmaxf(range_max, __builtin_fmaxf(__builtin_fabsf(sq.x), __builtin_fabsf(sq.y)));
            }
        }
        // ---- down-sampling FIR: s blocks 2 b and 2 b + 1 into the output blocks they reach ----
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            ActFrag sh, sl;
            {
                float z[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) z[i] = s[8 * q + i];
                act_split8(z, sh, sl);
            }
            const int i = 2 * b + q, B = i >> 2, e = i & 3;
            if (B <= 3) {
                ActFrag ah, al;
                ah.u = tab[(4 + 2 * e) * 64];
                al.u = tab[(5 + 2 * e) * 64];
                act_f32x16 acc = e == 0 ? AMP_ACT_MFMA(ah, sh, zero) : AMP_ACT_MFMA(ah, sh, y[B]);
                acc = AMP_ACT_MFMA(ah, sl, acc);
                y[B] = AMP_ACT_MFMA(al, sh, acc);
            }
            if (e < 2 && B >= 1) {
                ActFrag ah, al;
                ah.u = tab[(4 + 2 * (e + 4)) * 64];
                al.u = tab[(5 + 2 * (e + 4)) * 64];
                act_f32x16 acc = AMP_ACT_MFMA(ah, sh, y[B - 1]);
                acc = AMP_ACT_MFMA(ah, sl, acc);
                y[B - 1] = AMP_ACT_MFMA(al, sh, acc);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) v[t] = y[t];
}

#endif  // device

}  // namespace amp
