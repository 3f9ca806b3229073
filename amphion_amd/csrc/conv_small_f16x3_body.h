// The body of the whole-K frame-rate conv kernel (conv_small_f16x3.hip: see there for the algorithm), as a device function of (arguments,
// column-tile index, row-group index): conv_small_kernel runs it over a 2-D grid of its own; conv_small3_kernel (conv_small3_f16x3.hip)
// runs the same conv of a generator stage's three resblocks side by side in ONE grid.
#pragma once
#include "amp_internal.h"

namespace amp {


typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union FragS {
    uint4 u;
    f16x8 h;
};

#define AMP_PIN_VMEM_S() __builtin_amdgcn_sched_barrier(0x386)
// hides a loaded value behind an empty asm: hipcc otherwise turns `cond ? loaded : 0` into a branch around the load
// (load sunk into the taken side) and, loads returning in order, waits with vmcnt(0) for the whole tile at every one
#define AMP_OPAQUE(v) asm("" : "+v"(v))

constexpr int kSmallMaxChunks = kSmallConvMaxChunks;   // Cin <= 256
constexpr size_t kSmallMaxLds = 128 * 1024;

// A-fragment prefetch distance in chunk-sets (8 * KT VGPRs each)
template <int KT> struct ARing { static constexpr int n = KT <= 3 ? 4 : 2; };

// tanh / sigmoid of the gate on the hardware exp / rcp (v_exp_f32, v_rcp_f32: ~1e-7 absolute on outputs in [-1, 1];
// the accurate libm forms cost ~45 instructions each and the launch is issue-bound)
__device__ __forceinline__ float fast_sigmoid(float v) { return __frcp_rn(1.0f + __expf(-v)); }
__device__ __forceinline__ float fast_tanh(float v) { return 2.0f * fast_sigmoid(2.0f * v) - 1.0f; }

template <int KT, int NI, int HALO, int EPI>
__device__ __forceinline__ void conv_small_body(const ConvArgs a, const int bx, const int by) {   // (bx, by): this workgroup's column tile / 128-row group
    constexpr int WM = 4;
    constexpr int NT = 32 * NI;                // output columns per workgroup
    constexpr int S = NT + HALO;               // staged columns
    constexpr int NST = (4 * S) / 256;         // staging items per thread and chunk
    constexpr int BUF = 4 * S;                 // uint4 per chunk buffer: [plane hi|lo][octet h][S]
    constexpr int AR = ARing<KT>::n;
    constexpr int MAXC = kSmallMaxChunks;
    static_assert(S % 64 == 0, "the channel quad of a staging item must be wave-uniform");
    static_assert(AR % 2 == 0, "the B double buffer alternates with (chunk * KT + tap)");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];  // [nch_pad][BUF]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int item = bx / a.tiles_per_item;
    const int tile = bx - item * a.tiles_per_item;
    const int q0 = tile * NT;
    // ragged batch: a tile that lies entirely beyond this utterance's valid length produces only samples the contract
    // leaves unspecified (nothing downstream reads them: every layer takes its input as zero / replicated beyond the valid
    // length) -- skip it.  A batch of 60..400-frame utterances is 40 % such tiles.
    if (a.lens && a.Tout == a.Tin && (long long)q0 >= (long long)a.lens[item] * a.len_mul) return;   // block-uniform
    const int mb = by * WM + wave;             // 32-row block of W'
    const bool mb_ok = mb * 32 < a.Mpad;       // a wave past the packed rows only helps staging
    const int nchunks = a.nchunks;

    // ---- 1. every global load of the tile, and the first AR chunk-sets of A, in flight together ----
    const float* xb = a.x + (size_t)item * (size_t)a.xbs;
    const int tbase = q0 - a.halo_left;
    int Tv = a.Tin;
    int len_item = a.Tout;                      // EPI_WNACC: valid frames of this item (the mask of the x update)
    if (a.lens) {
        const int l0 = __builtin_amdgcn_readfirstlane(a.lens[item]);   // fetched ONCE, here: a load in the epilogue would
        len_item = l0;                                                 // make every store wait for the one before it
        const int l = l0 * a.len_mul;
        Tv = l < Tv ? l : Tv;
    }
    (void)len_item;
    const float kpos = 16.f, kneg = 16.f * a.slope_in;

    // staging item `it` of a thread: channel quad (wave-uniform) and column; the clamped column is the 32-bit lane
    // offset of every load of that item, the row pointer is uniform
    unsigned tcl[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int ibase = wave * 64 + 256 * it;
        const int qd = ibase / S;
        int t = tbase + (ibase - qd * S + lane);
        t = t < 0 ? 0 : t;
        t = t > a.Tin - 1 ? a.Tin - 1 : t;
        tcl[it] = (unsigned)t;
    }
    float xs[MAXC][NST][4];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nchunks) {
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int qd = (wave * 64 + 256 * it) / S;
                const int ch0 = c * KC16 + 4 * qd;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    int ch = ch0 + e;
                    ch = ch > a.Cin - 1 ? a.Cin - 1 : ch;                   // scalar
                    const float* rowp = xb + (size_t)ch * (size_t)a.Tin;     // uniform
                    xs[c][it][e] = rowp[tcl[it]];
                }
            }
        }
    }

    const uint4* wa0 = static_cast<const uint4*>(a.wp) + (size_t)(mb_ok ? mb : 0) * nchunks * (KT * 128) + lane;
    FragS a_h[AR][KT], a_l[AR][KT];
#pragma unroll
    for (int j = 0; j < AR; ++j) {
        const int cj = j < nchunks ? j : nchunks;               // chunk `nchunks` = next block / allocation pad
        const uint4* wj = wa0 + (size_t)cj * (KT * 128);
#pragma unroll
        for (int g = 0; g < KT; ++g) {
            a_h[j][g].u = wj[g * 128];
            a_l[j][g].u = wj[g * 128 + 64];
        }
    }

    // ---- accumulators: bias (+ condition / residual / running sums), scaled ----
    // (every load below is unconditional from a clamped address and masked by a select afterwards: a branch around a
    // load would make the compiler wait for it -- and, loads returning in order, for the whole tile -- on the spot)
    const int qw = q0 + l31;
    const float asc = a.acc_scale;
    const int Mc = a.M - 1, Tqc = a.Tq - 1;
    f32x16 acc[NI];
    bool qok[NI];
    unsigned qcl[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int q = qw + 32 * t;
        qok[t] = q < a.Tq;
        qcl[t] = (unsigned)(q < Tqc ? q : Tqc);
    }
    // row r of a lane: (r & 3) + 8 * (r >> 2) + 4 * hi inside the 32-row block; hi4T = the lane part in elements
    const unsigned hi4T = (unsigned)(4 * hi) * (unsigned)a.Tout;
    if constexpr (EPI == 0) {
        // (bias), (+ residual), (+ running MRF sum): three straight-line passes in the order of conv_f16x3.hip, each
        // behind ONE uniform branch.  Rows past M (last row block) clamp to row M - 1 and are masked.
        const bool full = mb * 32 + 32 <= a.M;                       // wave-uniform
        bool mokr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) mokr[r] = full || (mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi < a.M);
        if (a.bias) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const float bl = a.bias[m < Mc ? m : Mc];
#pragma unroll
                for (int t = 0; t < NI; ++t) acc[t][r] = mokr[r] ? bl : 0.f;
            }
        } else {
#pragma unroll
            for (int t = 0; t < NI; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
        auto add_rows = [&](const float* p) {     // acc += p[item, row, q] on the valid part of the tile: all loads first
            float ldv[16][NI];
            if (full) {
                const float* pw = p + ((size_t)item * a.Cout + (size_t)mb * 32) * a.Tout;   // uniform
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* rp = pw + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;      // uniform
#pragma unroll
                    for (int t = 0; t < NI; ++t) ldv[r][t] = rp[hi4T + qcl[t]];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    const float* rp = p + ((size_t)item * a.Cout + (m < Mc ? m : Mc)) * a.Tout;
#pragma unroll
                    for (int t = 0; t < NI; ++t) ldv[r][t] = rp[qcl[t]];
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    float ld = ldv[r][t];
                    AMP_OPAQUE(ld);
                    acc[t][r] += (mokr[r] && qok[t]) ? ld : 0.f;
                }
            }
        };
        if (a.res) add_rows(a.res);
        if (a.mode != 0) add_rows(a.y);
    } else if constexpr (EPI == 1) {
        // packed row rho = i + 4*hi + 8*(2u + s)  <->  original row s*H + 16*mb + (i + 4*hi + 8*u); bias is packed
        // in the same order (amp_conv_create_gated), the condition is indexed by the original row.  M = 2H is a
        // multiple of 32 and the bias is always present (host).
        const bool has_c = a.gate_cond != nullptr;
        const float* gc = has_c ? a.gate_cond + (size_t)item * a.gate_cond_bs : a.bias;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = r & 3, jj = r >> 2, s = jj & 1, u = jj >> 1;
            int prow = mb * 32 + i + 8 * jj + 4 * hi;
            int orow = s * a.wn_H + 16 * mb + i + 4 * hi + 8 * u;
            prow = prow < Mc ? prow : Mc;
            orow = orow < Mc ? orow : Mc;
            const float cv = gc[orow];
            const float bv = a.bias[prow] + (has_c ? cv : 0.f);
#pragma unroll
            for (int t = 0; t < NI; ++t) acc[t][r] = bv;
        }
    } else {
        // EPI_WNACC: rows < H (not last) start from x, the others from the running output (zero on the first layer).
        // M (2H, or H on the last layer) is a multiple of 32 and the bias is always present (host).
        const bool res_part = !a.wn_last && (mb * 32 < a.wn_H);
        int orow0 = mb * 32 - ((a.wn_last || res_part) ? 0 : a.wn_H);
        orow0 = orow0 + 32 <= a.wn_H ? orow0 : a.wn_H - 32;            // waves past M: any valid block (masked)
        const bool use_src = (res_part || !a.wn_first) && mb * 32 < a.M;
        const float* src = (res_part ? a.wn_x : a.wn_out) + ((size_t)item * a.wn_H + orow0) * a.Tout;   // uniform
        float ldv[16][NI];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* sp = src + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
            for (int t = 0; t < NI; ++t) ldv[r][t] = sp[hi4T + qcl[t]];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            int prow = mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            prow = prow < Mc ? prow : Mc;
            const float bv = a.bias[prow];
#pragma unroll
            for (int t = 0; t < NI; ++t) {
                float ld = ldv[r][t];
                AMP_OPAQUE(ld);
                acc[t][r] = bv + ((use_src && qok[t]) ? ld : 0.f);
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) acc[t] *= asc;

    // ---- 2. convert + store every chunk, one barrier ----
    float range_max = 0.f;
    bool tokv[NST];
    int o2v[NST];
#pragma unroll
    for (int it = 0; it < NST; ++it) {
        const int ibase = wave * 64 + 256 * it;
        const int qd = ibase / S;
        const int col = ibase - qd * S + lane;
        const int t = tbase + col;
        tokv[it] = (col < a.wd) && (t >= 0) && (t < Tv);
        o2v[it] = (((qd >> 1) * S + col) << 1) + (qd & 1);   // uint2 index inside a plane
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        if (c < nchunks) {
            uint2* dst = reinterpret_cast<uint2*>(smem4 + c * BUF);
#pragma unroll
            for (int it = 0; it < NST; ++it) {
                const int qd = (wave * 64 + 256 * it) / S;
                const int ch0 = c * KC16 + 4 * qd;
                struct { uint2 u; } fh, fl;
                stage4_f16((tokv[it] && (ch0 + 0) < a.Cin) ? xs[c][it][0] : 0.f, (tokv[it] && (ch0 + 1) < a.Cin) ? xs[c][it][1] : 0.f, (tokv[it] && (ch0 + 2) < a.Cin) ? xs[c][it][2] : 0.f, (tokv[it] && (ch0 + 3) < a.Cin) ? xs[c][it][3] : 0.f,
                           kpos, kneg, range_max, fh.u, fl.u);
                dst[o2v[it]] = fh.u;
                dst[4 * S + o2v[it]] = fl.u;
            }
        }
    }
    // the K loop below runs in whole groups of AR chunks without a branch inside: the buffers of the chunks that pad the
    // last group hold zeros (their A fragments are the next row block's / the allocation pad: finite)
    const int nch_pad = ((nchunks + AR - 1) / AR) * AR;
    for (int i = nchunks * BUF + tid; i < nch_pad * BUF; i += 256) smem4[i] = make_uint4(0u, 0u, 0u, 0u);
    __syncthreads();
    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);

    // ---- 3. K loop out of LDS, A fragments AR chunk-sets ahead ----
    // Nothing but this wave hides its LDS latency, so the B fragments of tap g + 1 (or of the next chunk's first tap) are
    // read into a second register set before the MFMAs of tap g are issued.  (c * KT + g) & 1 is a compile-time value
    // inside the unrolled body because AR is even.
    const uint4* lbase = smem4 + (hi * S + l31 + a.halo_left + a.off0);
    const int dstep = a.dstep;
    if (mb_ok) {
        FragS bh[2][NI], bl[2][NI];
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            bh[0][t].u = lbase[32 * t];
            bl[0][t].u = lbase[2 * S + 32 * t];
        }
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
#pragma unroll
                    for (int t = 0; t < NI; ++t) {
                        bh[nxt][t].u = bn[32 * t];
                        bl[nxt][t].u = bn[2 * S + 32 * t];
                    }
                    __builtin_amdgcn_sched_barrier(0);   // the reads above are issued BEFORE this tap's MFMAs (hipcc sinks them to their use)
#pragma unroll
                    for (int t = 0; t < NI; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[j][g].h, bh[cur][t].h, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NI; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[j][g].h, bl[cur][t].h, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < NI; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[j][g].h, bh[cur][t].h, acc[t], 0, 0, 0);
                    a_h[j][g].u = wn_[g * 128];
                    a_l[j][g].u = wn_[g * 128 + 64];
                    AMP_PIN_VMEM_S();
                }
            }
        }
    }

    // ---- epilogue (uniform row pointers + the lane offset hi4T + q) ----
    if constexpr (EPI == 0) {
        const float slope_out = a.slope_out;
        const bool exact_div = a.mode == 2;
        float* yw = a.y + ((size_t)item * a.Cout + (size_t)mb * 32) * a.Tout;
        auto store_rows = [&](auto fn) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = (r & 3) + 8 * (r >> 2);
                if (mb * 32 + rr + 4 * hi < a.M) {
                    float* yr = yw + (size_t)rr * a.Tout;
#pragma unroll
                    for (int t = 0; t < NI; ++t) {
                        if (qok[t]) {
                            float v = acc[t][r] * a.inv_scale;
                            if (exact_div) v = v / a.div;
                            v = v > 0.f ? v : v * slope_out;
                            yr[hi4T + (unsigned)(qw + 32 * t)] = fn(v);
                        }
                    }
                }
            }
        };
        if (a.tanh_out) store_rows([](float v) { return tanhf(v); });
        else store_rows([](float v) { return v; });
    } else if constexpr (EPI == 1) {
        float* yw = a.y + ((size_t)item * a.wn_H + (size_t)mb * 16) * a.Tout;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float* yr = yw + (size_t)(i + 8 * u) * a.Tout;     // channel 16*mb + i + 4*hi + 8*u
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    if (qok[t] && mb * 32 < a.M) {     // (M = 2H; the packed rows are padded to a multiple of 128: a block beyond M holds no channel --
                                                       //  round 4: it was stored, over the next item's first rows, when H was no multiple of 64)
                        const float at = acc[t][4 * (2 * u) + i] * a.inv_scale;
                        const float as = acc[t][4 * (2 * u + 1) + i] * a.inv_scale;
                        yr[hi4T + (unsigned)(qw + 32 * t)] = fast_tanh(at) * fast_sigmoid(as);
                    }
                }
            }
        }
    } else {
        const bool res_part = !a.wn_last && (mb * 32 < a.wn_H);
        const int orow0 = mb * 32 - ((a.wn_last || res_part) ? 0 : a.wn_H);
        if (mb * 32 < a.M) {
            float* dstw = (res_part ? a.wn_x : a.wn_out) + ((size_t)item * a.wn_H + orow0) * a.Tout;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float* dr = dstw + (size_t)((r & 3) + 8 * (r >> 2)) * a.Tout;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    if (qok[t]) {
                        float v = acc[t][r] * a.inv_scale;
                        if (res_part) v = (qw + 32 * t) < len_item ? v : 0.f;
                        dr[hi4T + (unsigned)(qw + 32 * t)] = v;
                    }
                }
            }
        }
    }
}


}  // namespace amp
