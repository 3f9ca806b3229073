// A whole WN layer (modules/flow/modules.py:126-151) in ONE launch (round 4):
//
//     acts = tanh((in_i(x) + g_i)[:H]) * sigmoid((in_i(x) + g_i)[H:])          in_layers[i]: Conv1d(H, 2H, k, dilation), gated
//     rs   = res_skip_i(acts)                                                   Conv1d(H, 2H | H, 1)
//     not last:  x' = (x + rs[:H]) * mask,  output += rs[H:];      last:  output += rs
//
// Rounds 2-3 ran this as two conv_small_kernel launches (EPI_GATE, then EPI_WNACC) with acts [B, H, T] through HBM in between: 18 + 11 us
// at B = 16, T = 165 (VITS' flow), 48 such layers in the VITS decode path (config 5) and 16 in text -> wave, every one latency-bound.
// Here a workgroup of NW waves owns one 32-column tile of one item for the WHOLE layer:
//   1. the [H, 32 + halo] tile of x is staged once (leaky-free: x16, hi / lo split) exactly as conv_small_f16x3.hip stages it;
//   2. gate passes: wave w of pass p owns the 32 packed rows of block p * NW + w (16 channels: their tanh and sigmoid pre-activations),
//      runs the K loop of the gated conv out of LDS, forms acts in registers and writes them -- x16, hi / lo split -- into a second LDS
//      tile in the B-operand layout, chunk = its row block (the layout the second conv would have staged them into from HBM);
//   3. one barrier; res_skip passes: the 1 x 1 conv's K loop out of the acts tile, epilogue = the residual / skip update.
// x is NOT updated in place (other workgroups still read this tile's columns as their halo): the caller ping-pongs two x buffers.
// Per output element every product, sum and rounding is that of the two launches it replaces -- same bits
// (tests/test_gpu_vits.py::test_wn_layer_kernel_bitwise).  Covered: k in {1, 3, 5} with (k - 1) * dilation <= 32, H a multiple of 32, <= 256.
#include "conv_small_f16x3_body.h"

namespace amp {

template <int KT, int NW>
__global__ __launch_bounds__(64 * NW, 1) void wn_layer_kernel(const WnLayerArgs p) {
    constexpr int S = 64;                      // staged x columns: 32 + the 32-column halo buffer
    constexpr int BUF = 4 * S;                 // uint4 per x chunk buffer: [plane hi|lo][octet h][S]
    constexpr int S2 = 32;                     // acts columns (1 x 1 conv: no halo)
    constexpr int BUF2 = 4 * S2;
    constexpr int AR = ARing<KT>::n, AR2 = ARing<1>::n;
    constexpr int G = NW / 4;                  // thread groups of 256 staging items: group g stages chunks g, g + G, ...
    constexpr int MAXCT = (kSmallMaxChunks + G - 1) / G;
    static_assert(NW == 4 || NW == 8 || NW == 12, "4, 8 or 12 waves");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];  // [nch_pad][BUF] x tile, then [nch2_pad][BUF2] acts tile
    const ConvArgs& a = p.g;                   // the gated conv
    const ConvArgs& r = p.r;                   // the res_skip conv (its wn_* fields; wn_x = the x this layer READS)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int bx = (int)blockIdx.x;
    const int item = bx / a.tiles_per_item;
    const int tile = bx - item * a.tiles_per_item;
    const int q0 = tile * 32;
    int Tv = a.Tin;
    int len_item = a.Tout;
    if (a.lens) {
        const int l0 = __builtin_amdgcn_readfirstlane(a.lens[item]);
        if (q0 >= l0) return;                  // block-uniform: a tile wholly beyond the utterance's end (conv_small_body: the same rule)
        len_item = l0;
        Tv = l0 < Tv ? l0 : Tv;
    }
    const int nchunks = a.nchunks;
    const int nch_pad = ((nchunks + AR - 1) / AR) * AR;
    const int nch2 = r.nchunks;                // H / 16
    const int nch2_pad = ((nch2 + AR2 - 1) / AR2) * AR2;
    uint4* const acts4 = smem4 + nch_pad * BUF;

    // ---- 1. stage the x tile: every global load first, then convert + store, one barrier ----
    const float* xb = a.x + (size_t)item * (size_t)a.xbs;
    const int tbase = q0 - a.halo_left;
    const int sit = tid & 255, sgrp = tid >> 8;            // staging item (channel quad x column) and chunk group of this thread
    const int sqd = sit >> 6, scol = sit & 63;             // wave-uniform quad
    float range_max = 0.f;
    {
        int t = tbase + scol;
        const bool tok = (scol < a.wd) && (t >= 0) && (t < Tv);
        t = t < 0 ? 0 : t;
        t = t > a.Tin - 1 ? a.Tin - 1 : t;
        float xs[MAXCT][4];
#pragma unroll
        for (int k = 0; k < MAXCT; ++k) {
            const int c = sgrp + G * k;
            if (c < nchunks) {
                const int ch0 = c * KC16 + 4 * sqd;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int ch = ch0 + e;
                    ch = ch > a.Cin - 1 ? a.Cin - 1 : ch;
                    xs[k][e] = xb[(size_t)ch * (size_t)a.Tin + (unsigned)t];
                }
            }
        }
        const int o2 = (((sqd >> 1) * S + scol) << 1) + (sqd & 1);
#pragma unroll
        for (int k = 0; k < MAXCT; ++k) {
            const int c = sgrp + G * k;
            if (c < nchunks) {
                const int ch0 = c * KC16 + 4 * sqd;
                uint2* dst = reinterpret_cast<uint2*>(smem4 + c * BUF);
                struct { uint2 u; } fh, fl;
                stage4_f16((tok && (ch0 + 0) < a.Cin) ? xs[k][0] : 0.f, (tok && (ch0 + 1) < a.Cin) ? xs[k][1] : 0.f,
                           (tok && (ch0 + 2) < a.Cin) ? xs[k][2] : 0.f, (tok && (ch0 + 3) < a.Cin) ? xs[k][3] : 0.f, 16.f, 16.f * a.slope_in,
                           range_max, fh.u, fl.u);
                dst[o2] = fh.u;
                dst[4 * S + o2] = fl.u;
            }
        }
        for (int i = nchunks * BUF + tid; i < nch_pad * BUF; i += 64 * NW) smem4[i] = make_uint4(0u, 0u, 0u, 0u);
        for (int i = nch2 * BUF2 + tid; i < nch2_pad * BUF2; i += 64 * NW) acts4[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    __syncthreads();
    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);

    const int q = q0 + l31;                                // this lane's output column
    const bool qok = q < a.Tq;
    const int Tqc = a.Tq - 1;
    const unsigned qcl = (unsigned)(q < Tqc ? q : Tqc);
    const unsigned hi4T = (unsigned)(4 * hi) * (unsigned)a.Tout;

    // ---- 2. gate passes ----
    {
        const uint4* lbase = smem4 + (hi * S + l31 + a.halo_left + a.off0);
        const int dstep = a.dstep;
        const int Mc = a.M - 1;
        const bool has_c = a.gate_cond != nullptr;
        const float* gc = has_c ? a.gate_cond + (size_t)item * a.gate_cond_bs : a.bias;
        for (int mb = wave; mb * 32 < a.M; mb += NW) {                        // wave-uniform trip count per wave (M = 2H: a multiple of 32)
            const uint4* wa0 = static_cast<const uint4*>(a.wp) + (size_t)mb * nchunks * (KT * 128) + lane;
            FragS a_h[AR][KT], a_l[AR][KT];
#pragma unroll
            for (int j = 0; j < AR; ++j) {
                const int cj = j < nchunks ? j : nchunks;
                const uint4* wj = wa0 + (size_t)cj * (KT * 128);
#pragma unroll
                for (int g = 0; g < KT; ++g) {
                    a_h[j][g].u = wj[g * 128];
                    a_l[j][g].u = wj[g * 128 + 64];
                }
            }
            f32x16 acc;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                const int i = rr & 3, jj = rr >> 2, s = jj & 1, u = jj >> 1;
                int prow = mb * 32 + i + 8 * jj + 4 * hi;
                int orow = s * a.wn_H + 16 * mb + i + 4 * hi + 8 * u;
                prow = prow < Mc ? prow : Mc;
                orow = orow < Mc ? orow : Mc;
                const float cv = gc[orow];
                acc[rr] = a.bias[prow] + (has_c ? cv : 0.f);
            }
            acc *= a.acc_scale;
            FragS bh[2], bl[2];
            bh[0].u = lbase[0];
            bl[0].u = lbase[2 * S];
            for (int c0 = 0; c0 < nch_pad; c0 += AR) {
#pragma unroll
                for (int j = 0; j < AR; ++j) {
                    const int c = c0 + j;
                    const uint4* base = lbase + c * BUF;
                    const uint4* base_next = lbase + (c + 1 < nch_pad ? c + 1 : c) * BUF;
                    int cn = c + AR;
                    cn = cn < nchunks ? cn : nchunks;
                    const uint4* wn_ = wa0 + (size_t)cn * (KT * 128);
#pragma unroll
                    for (int g = 0; g < KT; ++g) {
                        const int cur = (j * KT + g) & 1, nxt = cur ^ 1;
                        const uint4* bn = (g + 1 < KT) ? base + (g + 1) * dstep : base_next;
                        bh[nxt].u = bn[0];
                        bl[nxt].u = bn[2 * S];
                        __builtin_amdgcn_sched_barrier(0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[j][g].h, bh[cur].h, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[j][g].h, bl[cur].h, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[j][g].h, bh[cur].h, acc, 0, 0, 0);
                        a_h[j][g].u = wn_[g * 128];
                        a_l[j][g].u = wn_[g * 128 + 64];
                        AMP_PIN_VMEM_S();
                    }
                }
            }
            // gate -> acts (the fp32 values the two-launch form stores), then what the second conv's staging makes of them: zero beyond
            // the valid length / the tile, x16, hi / lo split; chunk = this row block, quads 2u + hi
            const bool keep = qok && q < Tv;
            uint2* dst = reinterpret_cast<uint2*>(acts4 + mb * BUF2);
            float dummy = 0.f;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float v[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float at = acc[4 * (2 * u) + i] * a.inv_scale;
                    const float as = acc[4 * (2 * u + 1) + i] * a.inv_scale;
                    const float act = fast_tanh(at) * fast_sigmoid(as);
                    v[i] = keep ? act : 0.f;
                }
                struct { uint2 u; } fh, fl;
                stage4_f16(v[0], v[1], v[2], v[3], 16.f, 16.f, dummy, fh.u, fl.u);
                const int o2 = ((u * S2 + l31) << 1) + hi;
                dst[o2] = fh.u;
                dst[4 * S2 + o2] = fl.u;
            }
        }
    }
    __syncthreads();

    // ---- 3. res_skip passes: 1 x 1 conv out of the acts tile, residual / skip update ----
    {
        const uint4* lbase = acts4 + (hi * S2 + l31);
        const int Mc = r.M - 1;
        for (int mb = wave; mb * 32 < r.M; mb += NW) {
            const uint4* wa0 = static_cast<const uint4*>(r.wp) + (size_t)mb * nch2 * 128 + lane;
            FragS a_h[AR2], a_l[AR2];
#pragma unroll
            for (int j = 0; j < AR2; ++j) {
                const int cj = j < nch2 ? j : nch2;
                a_h[j].u = wa0[(size_t)cj * 128];
                a_l[j].u = wa0[(size_t)cj * 128 + 64];
            }
            const bool res_part = !r.wn_last && (mb * 32 < r.wn_H);
            int orow0 = mb * 32 - ((r.wn_last || res_part) ? 0 : r.wn_H);
            orow0 = orow0 + 32 <= r.wn_H ? orow0 : r.wn_H - 32;
            const bool use_src = (res_part || !r.wn_first) && mb * 32 < r.M;
            const float* src = (res_part ? r.wn_x : r.wn_out) + ((size_t)item * r.wn_H + orow0) * r.Tout;
            float ldv[16];
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) ldv[rr] = src[(size_t)((rr & 3) + 8 * (rr >> 2)) * r.Tout + hi4T + qcl];
            f32x16 acc;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) {
                int prow = mb * 32 + (rr & 3) + 8 * (rr >> 2) + 4 * hi;
                prow = prow < Mc ? prow : Mc;
                float ld = ldv[rr];
                AMP_OPAQUE(ld);
                acc[rr] = r.bias[prow] + ((use_src && qok) ? ld : 0.f);
            }
            acc *= r.acc_scale;
            FragS bh[2], bl[2];
            bh[0].u = lbase[0];
            bl[0].u = lbase[2 * S2];
            for (int c0 = 0; c0 < nch2_pad; c0 += AR2) {
#pragma unroll
                for (int j = 0; j < AR2; ++j) {
                    const int c = c0 + j;
                    const uint4* base_next = lbase + (c + 1 < nch2_pad ? c + 1 : c) * BUF2;
                    int cn = c + AR2;
                    cn = cn < nch2 ? cn : nch2;
                    const int cur = j & 1, nxt = cur ^ 1;
                    bh[nxt].u = base_next[0];
                    bl[nxt].u = base_next[2 * S2];
                    __builtin_amdgcn_sched_barrier(0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[j].h, bh[cur].h, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[j].h, bl[cur].h, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[j].h, bh[cur].h, acc, 0, 0, 0);
                    a_h[j].u = wa0[(size_t)cn * 128];
                    a_l[j].u = wa0[(size_t)cn * 128 + 64];
                    AMP_PIN_VMEM_S();
                }
            }
            if (mb * 32 < r.M) {
                const int orow_w = mb * 32 - ((r.wn_last || res_part) ? 0 : r.wn_H);
                float* dstw = (res_part ? p.x_out : r.wn_out) + ((size_t)item * r.wn_H + orow_w) * r.Tout;
#pragma unroll
                for (int rr = 0; rr < 16; ++rr) {
                    float* dr = dstw + (size_t)((rr & 3) + 8 * (rr >> 2)) * r.Tout;
                    if (qok) {
                        float v = acc[rr] * r.inv_scale;
                        if (res_part) v = q < len_item ? v : 0.f;
                        dr[hi4T + (unsigned)q] = v;
                    }
                }
            }
        }
    }
}

template <int KT, int NW>
static hipError_t launch_wn_one(const WnLayerArgs& p, hipStream_t stream) {
    constexpr int AR = ARing<KT>::n, AR2 = ARing<1>::n;
    const int nch_pad = (p.g.nchunks + AR - 1) / AR * AR, nch2_pad = (p.r.nchunks + AR2 - 1) / AR2 * AR2;
    const size_t lds = ((size_t)nch_pad * 4 * 64 + (size_t)nch2_pad * 4 * 32) * sizeof(uint4);
    if (p.g.nchunks > kSmallMaxChunks || p.r.nchunks > kSmallMaxChunks || lds > kSmallMaxLds || p.g.wd > 64) return hipErrorInvalidValue;
    static unsigned long long attr_set = 0;   // per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!((attr_set >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&wn_layer_kernel<KT, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)kSmallMaxLds);
        if (e != hipSuccess) return e;
        attr_set |= 1ull << dev;
    }
    note_kernel("wn_layer_kernel", KT, NW);
    note_work((unsigned long long)p.g.B * p.g.tiles_per_item, (2.0 * p.g.M * p.g.Cin * KT + 2.0 * p.r.M * p.r.Cin) * (double)p.g.Tq * p.g.B / 1e9,
              4.0 * p.g.B * (double)p.g.Tin * (3.0 * p.g.Cin) / 1e6, "WN layer H=%d k=%d T=%d B=%d", p.g.Cin, KT, p.g.Tin, p.g.B);
    hipLaunchKernelGGL((wn_layer_kernel<KT, NW>), dim3((unsigned)(p.g.B * p.g.tiles_per_item)), dim3(64 * NW), lds, stream, p);
    return hipGetLastError();
}

// kt: taps of the gated conv (1, 3, 5); nw: waves per workgroup (4, 8 or 12)
hipError_t launch_wn_layer(int kt, int nw, const WnLayerArgs& p, hipStream_t stream) {
    if (nw == 12) {
        if (kt == 1) return launch_wn_one<1, 12>(p, stream);
        if (kt == 3) return launch_wn_one<3, 12>(p, stream);
        if (kt == 5) return launch_wn_one<5, 12>(p, stream);
    } else if (nw == 8) {
        if (kt == 1) return launch_wn_one<1, 8>(p, stream);
        if (kt == 3) return launch_wn_one<3, 8>(p, stream);
        if (kt == 5) return launch_wn_one<5, 8>(p, stream);
    } else if (nw == 4) {
        if (kt == 1) return launch_wn_one<1, 4>(p, stream);
        if (kt == 3) return launch_wn_one<3, 4>(p, stream);
        if (kt == 5) return launch_wn_one<5, 4>(p, stream);
    }
    return hipErrorInvalidValue;
}

}  // namespace amp
