// Persistent, software-pipelined form of the f16x3 implicit-GEMM Conv1d (round 6) for the square convs of the MRF / AMP blocks
// that run UNFUSED (bigvgan.py:137-146: an Activation1d sits between the two convs of an AMPBlock pair; hifigan.py:93-100 at
// C = 256): Cin = Cout = 64 | 128 | 256, k = 3 | 5 | 7 | 11, any dilation whose receptive field fits 64 staged columns.
//
// conv_f16x3.hip / conv_blk_f16x3.hip give every workgroup ONE column tile and keep two workgroups per CU so that one's prologue
// (bias + residual into the accumulators, first staging round) and epilogue (128 stores per lane) may hide under the other's MFMAs.
// The clock stamps of round 3 / 5 show they mostly do not: the arbiter serves the older wave of a SIMD first, a tile's fixed
// phases are 25-45 % of its life, and with 256 registers there is no room to read the B fragments ahead of the MFMAs that use
// them -- those kernels sit at 0.29-0.44 of the f16x3 peak where pair_strip_f16x3.hip (one wave per SIMD, 512 registers, B
// fragments a half-tap ahead, A-fragment ring) reaches 0.49.  Here the strip kernel's wave shape is carried over to a single conv
// and the fixed phases are taken OUT of the critical path instead of being hidden by occupancy:
//
//   * one workgroup per CU, 4 waves, a wave owns 64 rows x 128 columns (2 x 4 MFMA tiles): C = 256 -> 4 x 1 waves (128-column
//     steps), C = 128 -> 2 x 2 (256), C = 64 -> 1 x 4 (512);
//   * a workgroup walks `strip_steps` consecutive column tiles of one utterance;
//   * a SIDE register set Y next to the accumulators X (both in AGPRs: 2 x 128 registers): while step s accumulates into X, Y --
//     step s - 1's raw results -- is un-scaled, activated and stored in pieces between the half-taps of step s's FIRST staging round,
//     then receives step s + 1's residual in pieces during the second-to-last round and is turned into step s + 1's starting
//     accumulators ((bias + residual) * scale) during the last one.  At the step boundary X and Y change places through the matrix
//     pipe (D = 0 * 0 + C: 24 MFMAs, ~800 cycles, no VALU; two register sets that merely swap ROLES every step made hipcc shuffle
//     a whole set through scratch between the two copies of the loop body).  No instruction of a step boundary waits for memory:
//     the next step's first x chunk is requested at the head of the last round and staged at its end.  (A -0.0 that passes the
//     exchange becomes +0.0, as it would in the next MFMA that accumulates onto it.)
//   * loads and stores go through buffer descriptors: a lane outside the utterance (zero padding, the ragged tail, the tile beyond
//     the end) carries an out-of-range offset, so that the hardware returns 0 / drops the store -- no selects while staging and no
//     exec-mask branches between the MFMAs (a branch would make hipcc's counted s_waitcnt vmcnt conservative: every A-fragment
//     wait would also wait for the HBM loads issued after it);
//   * CM 16-channel chunks per staging round / barrier (two; k = 3: four -- 200+ MFMAs per wave between barriers), staged one chunk at a
//     time between the MFMAs.
//
// Per output element the order of operations -- accumulator start (bias + residual [+ running sum]) * scale, chunks, taps, the
// three MFMAs of a term, un-scale, leaky ReLU -- is that of conv_f16x3.hip: bit-identical results (tests/test_gpu_f16x3_kernels.py).
//
// Compiled once per tap count:  -DAMP_KT=<3|5|7|11>.
#include "amp_internal.h"

#include <type_traits>

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

union FragC {
    uint4 u;
    f16x8 h;
};

// VMEM and MFMA may not cross (VALU, SALU, DS may)
#define AMP_PIN_VMEM() __builtin_amdgcn_sched_barrier(0x386)
// LDS reads stay above, MFMAs below (VALU / SALU / VMEM / LDS writes may cross)
#define AMP_PIN_DSREAD() __builtin_amdgcn_sched_barrier(0x276)
// end of a half-tap: VALU may not cross either (the scheduler would lift all 128 un-scale / activation results of a store round to its top
// and hold them in registers until their stores' positions); SALU and DS may
#define AMP_PIN_HALFTAP() __builtin_amdgcn_sched_barrier(0x384)

constexpr int kStripOOB = (int)0x80000000u;   // a byte offset beyond every descriptor (num_records < 2^31, checked on the host)

enum { RK_ST = 1, RK_LD = 2, RK_CV = 4, RK_LAST = 8 };
#ifndef AMP_STRIP_ST_TAPS
#define AMP_STRIP_ST_TAPS 99
#endif
#ifndef AMP_STRIP_LD_TAPS
#define AMP_STRIP_LD_TAPS 99
#endif

// staged halo columns for a step of `ntw` columns: at least 64, and a staged width that is a multiple of 64
constexpr int strip_halo(int ntw) { return (ntw + 64) % 64 == 0 ? 64 : 64 + (64 - (ntw + 64) % 64); }

template <int KT, int WM, int WN, int NI, int CM, int RING, bool RES, bool SUM>
__global__ __launch_bounds__(256, 1) void conv_strip_kernel(const ConvArgs a) {
    constexpr int MI = 2, BH = (NI % 2 == 0) ? 2 : 1, NB = NI / BH;   // the B fragments of a tap in BH parts, each read one part ahead
    constexpr int NT = 32 * NI;                // output columns per wave
    constexpr int NTW = NT * WN;               // ... per step
    constexpr int HALO = strip_halo(NTW);
    constexpr int S = NTW + HALO;              // staged columns
    constexpr int NST = (4 * S) / 256;         // staging items (column x channel quad) per thread and chunk
    constexpr int BUF = 4 * S;                 // uint4 per chunk: [plane hi|lo][octet][S]
    constexpr int C = 64 * WM;                 // Cin = Cout
    constexpr int NCH = C / KC16;              // 16-channel chunks
    constexpr int NR = NCH / CM;               // staging rounds per step
    constexpr int VT = CM * KT;                // taps per round
    constexpr int NH = VT * BH;                // half-taps per round
    constexpr int G = (32 + NH - 1) / NH;      // conversions per half-tap: row groups (one (row block, register) = NI column tiles)
    // The memory operations of the boundary are spread over ALL half-taps of their round (AMP_STRIP_ST_TAPS / _LD_TAPS = 99): a vector-memory
    // instruction costs the issuing wave ~30 cycles when they come five to an MFMA (four waves share the CU's address unit, which the A
    // fragments already keep a quarter busy) -- bursts of 32 stores / 96 loads in the first taps of a round cost 1 000-3 300 cycles per tap
    // (profiles/r6_strip_conv.txt), one or two behind each MFMA 100-200
    constexpr int ST_TAPS = AMP_STRIP_ST_TAPS < NH ? AMP_STRIP_ST_TAPS : NH;   // half-taps over which the stores of a step are spread
    constexpr int LD_TAPS = AMP_STRIP_LD_TAPS < NH ? AMP_STRIP_LD_TAPS : NH;   // ... the residual loads of the next step
    constexpr int GST = (32 + ST_TAPS - 1) / ST_TAPS, GLD = (32 + LD_TAPS - 1) / LD_TAPS;
    constexpr int SLO = KT >= 7 ? 3 : KT - 1;        // a slot's chunk (of the next round) is split and written to LDS over the slot's taps SLO .. KT - 1
    constexpr int LDR = NR >= 4 ? NR - 2 : NR - 1;   // the round that requests the next step's residual
    constexpr bool CV_IN_ROUND = NR >= 4;            // ... which the LAST round turns into accumulators (else: at the boundary)
    static_assert(WM * WN == 4, "four waves");
    static_assert(NCH % CM == 0 && NR >= 2 && NR % 2 == 0, "rounds alternate between two staging buffers, also across steps");
    static_assert(RING <= VT && RING >= 2, "A-fragment ring");
    static_assert(S % 64 == 0, "the channel quad of a staging item must be wave-uniform");
    extern __shared__ __attribute__((aligned(16))) uint4 smem4[];  // [2][CM][BUF] + bias[C]
    float* const bias_s = reinterpret_cast<float*>(smem4 + 2 * CM * BUF);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, l31 = lane & 31;
    const int nbx = gridDim.x;   // XCD-contiguous strip runs, see conv_f16x3.hip
    int bx = ((nbx & 7) == 0 && !a.lens) ? (int)(blockIdx.x & 7) * (nbx >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (a.rev) bx = nbx - 1 - bx;
    const int item = bx / a.strips_per_item;
    const int strip = bx - item * a.strips_per_item;
    const int tile0 = strip * a.strip_steps;
    int Tv = a.Tin;                            // valid input columns of this item (ragged batch)
    if (a.lens) { const int l = a.lens[item] * a.len_mul; Tv = l < Tv ? l : Tv; }
    // tiles that lie entirely beyond the utterance's valid length produce only samples the contract leaves unspecified (conv_f16x3.hip)
    const int Tl = Tv < a.Tq ? Tv : a.Tq;
    const int ntiles = (Tl + NTW - 1) / NTW;
    int nst = ntiles - tile0;
    nst = nst < a.strip_steps ? nst : a.strip_steps;
    if (nst <= 0) return;                      // workgroup-uniform

    const int wm = wave / WN, wnc = (wave % WN) * NT;
    const int mb0 = wm * MI;                   // first 32-row block of this wave
    const int colw = wnc + l31;

    // buffer descriptors of this item (wave-uniform: kernel arguments and blockIdx only)
    const int Tin4 = a.Tin * 4, Tout4 = a.Tout * 4;
    const float* xb = a.x + (size_t)item * (size_t)a.xbs;
    float* yb = a.y + (size_t)item * C * a.Tout;
    const float* rb = RES ? a.res + (size_t)item * C * a.Tout : yb;
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xb), 0, C * Tin4, 0x00020000);
    const __amdgpu_buffer_rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(yb, 0, C * Tout4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rb), 0, C * Tout4, 0x00020000);
    auto ldf = [](const __amdgpu_buffer_rsrc_t r, int voff, int soff) __attribute__((always_inline)) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };
    auto stf = [](float v, const __amdgpu_buffer_rsrc_t r, int voff, int soff) __attribute__((always_inline)) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
    };

    const float asc = a.acc_scale, isc = a.inv_scale;
    const float slope_out = a.slope_out;
    const float cvm = SUM ? 1.f : asc;         // the last round's conversion leaves (bias + residual) un-scaled when a running sum follows
    const float kpos = 16.f, kneg = 16.f * a.slope_in;
    float range_max = 0.f;

    // byte offsets of this lane's four output columns of step tile `tl` (row 4 * hi of a register quad's rows included), out of range
    // beyond the end / for a step that does not exist
    auto set_voy = [&](int tl, bool exists, int (&vo)[NI]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < NI; ++t) {
            const int col = tl * NTW + colw + 32 * t;
            vo[t] = (exists && col < a.Tq) ? (4 * hi * a.Tout + col) * 4 : kStripOOB;
        }
    };
    // ... of this thread's staging items: columns outside [0, Tv) read as zero (the conv's padding, the ragged tail)
    int vox[NST];
    auto set_vox = [&](int tl, bool exists) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NST; ++it) {
            const int ibase = wave * 64 + 256 * it;          // wave-uniform
            const int qd = ibase / S;
            const int col = ibase - qd * S + lane;
            const int t = tl * NTW - a.halo_left + col;
            vox[it] = (exists && t >= 0 && t < Tv) ? t * 4 : kStripOOB;
        }
    };
    // One 16-channel chunk at a time (a SLOT of KT taps): its loads go out behind the slot's first tap and the split values enter LDS behind
    // tap SPOS of the same slot, between MFMAs -- nothing of the staging is left for the end of a round but the barrier, and a round of CM
    // chunks needs the registers of one.
    float xs[NST][4];
    auto stage_load_one = [&](int chunk, int it, int e) __attribute__((always_inline)) {
        const int qd = (wave * 64 + 256 * it) / S;       // wave-uniform
        int t4 = Tin4;
        asm volatile("" : "+s"(t4));                     // opaque: hipcc would hoist all NCH * 16 row offsets out of the step loop (SGPR spills)
        xs[it][e] = ldf(rx, vox[it], (chunk * KC16 + 4 * qd + e) * t4);
    };
    auto stage_store_one = [&](int par, int cc, int it) __attribute__((always_inline)) {
        uint2* dst = reinterpret_cast<uint2*>(smem4 + (par * CM + cc) * BUF);
        const int ibase = wave * 64 + 256 * it;
        const int qd = ibase / S;
        const int col = ibase - qd * S + lane;
        struct { uint2 u; } fh, fl;
        stage4_f16(xs[it][0], xs[it][1], xs[it][2], xs[it][3], kpos, kneg, range_max, fh.u, fl.u);
        const int o2 = (((qd >> 1) * S + col) << 1) + (qd & 1);   // uint2 index inside a plane
        dst[o2] = fh.u;
        dst[4 * S + o2] = fl.u;
    };
    auto stage_load = [&](int chunk) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NST; ++it)
#pragma unroll
            for (int e = 0; e < 4; ++e) stage_load_one(chunk, it, e);
    };
    auto stage_store = [&](int par, int cc) __attribute__((always_inline)) {
#pragma unroll
        for (int it = 0; it < NST; ++it) stage_store_one(par, cc, it);
    };

    // A fragments [mb][chunk][tap][plane][lane] x uint4 as a ring of RING taps per row block (conv_blk_f16x3.hip): the tap in slot
    // v % RING is replaced right after its use by tap v + RING of this round or tap v % RING of the next round (the next STEP's
    // first round after the last one: the same weights again)
    // Through a buffer descriptor as well: lane offset in a VGPR, everything else (row block, round, tap, plane) a scalar byte offset -- with
    // global pointers hipcc forms every 64-bit address with VALU adds, hoists them out of the loops and spills them (first build: 370 spills).
    constexpr int MBS = NCH * (KT * 128) * 16;              // bytes per 32-row block of packed A fragments
    constexpr int RNDB = VT * 128 * 16;                     // ... per staging round
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.wp), 0, 0x7fffffff, 0x00020000);
    const int wlane = lane * 16;
    const int wa0 = mb0 * MBS;                              // scalar byte offset of this wave's first row block
    auto ldw = [&](int soff) __attribute__((always_inline)) {
        return __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rw, wlane, soff, 0));
    };
    FragC a_h[MI][RING], a_l[MI][RING];
    auto reload = [&](int mi, int v, int wr, int wn) __attribute__((always_inline)) {
        const int src = (v + RING < VT) ? wr + (v + RING) * 2048 : wn + (v % RING) * 2048;
        a_h[mi][v % RING].u = ldw(src + mi * MBS);
        a_l[mi][v % RING].u = ldw(src + mi * MBS + 1024);
    };
    const int rd0 = hi * S + wnc + l31 + a.halo_left + a.off0;
    const int dstep = a.dstep;

    // this lane's rows of register r of row block mi: 32 * (mb0 + mi) + (r & 3) + 8 * (r >> 2) [+ 4 * hi, part of the lane offset]
    auto row_of = [&](int mi, int r) __attribute__((always_inline)) { return 32 * (mb0 + mi) + (r & 3) + 8 * (r >> 2); };
    // through an opaque index: read where it is used -- a plain read is hoisted out of the step loop (32 values per lane held across it,
    // then spilled, and every reload sits behind an s_waitcnt vmcnt(0) in the middle of the MFMAs: the first build's last round took twice
    // a plain one; a volatile read loses the LDS address space and becomes a flat load)
    auto bias_at = [&](int mi, int r) __attribute__((always_inline)) {
        int h4 = 4 * hi;                   // (only the lane part goes through the asm: its INPUT is loop-invariant and would be hoisted and spilled per row)
        asm volatile("" : "+v"(h4));
        return bias_s[row_of(mi, r) + h4];
    };
    auto fin = [&](float v) __attribute__((always_inline)) {
        v *= isc;
        return __builtin_fmaxf(v, v * slope_out);    // leaky ReLU for slopes <= 1 (host), 1.0 = identity
    };

#ifdef AMP_STRIP_STAMPS
    // experiment build (tools/strip_stamps.py): shader-clock stamps of every wave -- entry, prologue, the rounds of step 1 (start, end of the
    // MFMAs, barrier passed), every half-tap of its first two and last two rounds, the boundary, the flush
    unsigned long long* const stp = a.stamps ? a.stamps + ((size_t)blockIdx.x * 4 + wave) * 256 : nullptr;
#define AMP_STAMP(cond, i) do { if (stp && (cond) && lane == 0) stp[(i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define AMP_STAMP(cond, i) do { } while (0)
#endif
    AMP_STAMP(true, 0);
    // ---------------- prologue: bias table, first staging round, first A taps, step 0's accumulators (the one exposed start) ----------------
    for (int i = tid; i < C; i += 256) bias_s[i] = a.bias ? a.bias[i] : 0.f;
    set_vox(tile0, true);
    stage_load(0);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int v = 0; v < RING; ++v) {
            a_h[mi][v].u = ldw(wa0 + mi * MBS + v * 2048);
            a_l[mi][v].u = ldw(wa0 + mi * MBS + v * 2048 + 1024);
        }
    AMP_PIN_VMEM();
    __syncthreads();
    f32x16 X[MI][NI], Y[MI][NI];
    {
        int vo[NI];
        set_voy(tile0, true, vo);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float bv = bias_at(mi, r);
                const int soff = row_of(mi, r) * Tout4;
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    float v = bv;
                    if (RES) v += ldf(rr, vo[t], soff);
                    X[mi][t][r] = v;
                    Y[mi][t][r] = 0.f;
                }
                if ((r & 3) == 3) asm volatile("" ::: "memory");   // 16 values' loads in flight at a time: left alone, all of them are hoisted (registers)
            }
        if (SUM) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int soff = row_of(mi, r) * Tout4;
#pragma unroll
                    for (int t = 0; t < NI; ++t) X[mi][t][r] += ldf(ry, vo[t], soff);
                    if ((r & 3) == 3) asm volatile("" ::: "memory");
                }
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int t = 0; t < NI; ++t) X[mi][t] *= asc;
    }
    stage_store(0, 0);
#pragma unroll
    for (int cc = 1; cc < CM; ++cc) {   // (the first round of a strip: its other chunks one after the other, exposed)
        stage_load(cc);
        stage_store(0, cc);
    }
    __syncthreads();

    AMP_STAMP(true, 1);
    int vst[NI], vld[NI];
    // where the boundary's rotation left the results of tile (mi, t)
    auto yres = [&](int mi, int t) __attribute__((always_inline)) -> f32x16& {
        const int k = (mi * NI + t + MI * NI - 1) % (MI * NI);
        return Y[k / NI][k % NI];
    };

    // ---------------- one staging round of a step: NH half-taps of 12 MFMAs, with the step boundary's work between them ----------------
    auto round = [&](auto kind_c, const int c, const int s) __attribute__((always_inline)) {
        constexpr int kind = decltype(kind_c)::value;
        constexpr bool last = (kind & RK_LAST) != 0;
        int wr = wa0 + c * RNDB;
        int wn = last ? wa0 : wr + RNDB;
        asm volatile("" : "+s"(wr), "+s"(wn));              // opaque: tap offsets are formed where they are used, not hoisted (SGPR spills)
        const uint4* base = smem4 + ((c & 1) * CM) * BUF + rd0;
        AMP_STAMP(s == 1, 2 + 3 * c);
        if (last) set_vox(tile0 + s + 1, s + 1 < nst);     // every staging load of this step has been issued; a step that does not exist reads zeros
        FragC bh[2][NB], bl[2][NB];
#pragma unroll
        for (int t = 0; t < NB; ++t) {
            bh[0][t].u = base[32 * t];
            bl[0][t].u = base[2 * S + 32 * t];
        }
#pragma unroll
        for (int h = 0; h < NH; ++h) {
            const int v = h / BH, th = h % BH, cur = h & 1;
            if (h + 1 < NH) {   // the B fragments of half-tap h + 1 are read while the MFMAs of half-tap h run (pair_strip_f16x3.hip)
                const int vn = (h + 1) / BH, tn = (h + 1) % BH;
                const uint4* bn = base + (vn / KT) * BUF + (vn % KT) * dstep;
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    bh[cur ^ 1][t].u = bn[32 * (tn * NB + t)];
                    bl[cur ^ 1][t].u = bn[2 * S + 32 * (tn * NB + t)];
                }
            }
            AMP_PIN_DSREAD();
            // Everything that is not an MFMA rides BETWEEN the MFMAs of this half-tap, a unit behind each (a bunch of 20 buffer loads behind
            // the last MFMA of a tap cost that tap 500 cycles, 120 staging VALU instructions in one tap 300-2 000: r6_c / r6_d stamps):
            //   the loads of the slot's chunk (of the next round) behind the MFMAs of the slot's first tap,
            //   its staging items (split + LDS write) one or two per tap over the slot's later taps,
            //   the boundary's pieces (stores / residual loads / conversions) a row group at a time.
            const int slot = v / KT, g = v % KT;
            const int next_chunk = (last ? 0 : (c + 1) * CM) + slot;
            constexpr int NM = MI * NB * 3;              // MFMAs of a half-tap
            constexpr int NL = 4 * NST;                  // loads of a chunk
            auto piece_st = [&](int g4) __attribute__((always_inline)) {        // the previous step's results leave
                if (g4 >= 32) return;
                const int mi = g4 / 16, r = g4 % 16;
                int t4 = Tout4;
                asm volatile("" : "+s"(t4));             // opaque, as in stage_load
                const int soff = row_of(mi, r) * t4;
#pragma unroll
                for (int t = 0; t < NI; ++t) stf(fin(yres(mi, t)[r]), ry, vst[t], soff);
            };
            auto piece_ld = [&](int g4) __attribute__((always_inline)) {        // the next step's residual arrives
                if (g4 >= 32) return;
                const int mi = g4 / 16, r = g4 % 16;
                int t4 = Tout4;
                asm volatile("" : "+s"(t4));
                const int soff = row_of(mi, r) * t4;
#pragma unroll
                for (int t = 0; t < NI; ++t) Y[mi][t][r] = ldf(rr, vld[t], soff);
            };
            auto piece_cv = [&](int g4) __attribute__((always_inline)) {        // ... and becomes its starting accumulators
                if (g4 >= 32) return;
                const int mi = g4 / 16, r = g4 % 16;
                const float bv = bias_at(mi, r);
#pragma unroll
                for (int t = 0; t < NI; ++t) {
                    float v = RES ? (bv + Y[mi][t][r]) * cvm : bv * cvm;
                    // anchored HERE, in an accumulator register: a pure computation has no position of its own -- left alone all of
                    // them (and their 32 bias values) sink to the boundary, where their only consumer (the exchange) is
                    asm volatile("" : "+a"(v));
                    Y[mi][t][r] = v;
                }
            };
            auto side = [&](const int mf) __attribute__((always_inline)) {   // behind MFMA mf of this half-tap
                if (th == 0 && g == 0) {
#pragma unroll
                    for (int q = (mf * NL) / NM; q < ((mf + 1) * NL) / NM; ++q) stage_load_one(next_chunk, q / 4, q % 4);
                }
                if ((kind & RK_ST) && h < ST_TAPS) {
#pragma unroll
                    for (int gi = 0; gi < GST; ++gi)
                        if ((gi * NM) / GST == mf) piece_st(h * GST + gi);
                }
                if ((kind & RK_LD) && RES && h < LD_TAPS) {
#pragma unroll
                    for (int gi = 0; gi < GLD; ++gi)
                        if ((gi * NM) / GLD == mf) piece_ld(h * GLD + gi);
                }
                if (kind & RK_CV) {
#pragma unroll
                    for (int gi = 0; gi < G; ++gi)
                        if ((gi * NM) / G == mf) piece_cv(h * G + gi);
                }
                AMP_PIN_VMEM();
            };
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    X[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[mi][v % RING].h, bh[cur][t].h, X[mi][th * NB + t], 0, 0, 0);
                    side((mi * 3 + 0) * NB + t);
                }
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    X[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_h[mi][v % RING].h, bl[cur][t].h, X[mi][th * NB + t], 0, 0, 0);
                    side((mi * 3 + 1) * NB + t);
                }
#pragma unroll
                for (int t = 0; t < NB; ++t) {
                    X[mi][th * NB + t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_l[mi][v % RING].h, bh[cur][t].h, X[mi][th * NB + t], 0, 0, 0);
                    side((mi * 3 + 2) * NB + t);
                }
                if (th == BH - 1) reload(mi, v, wr, wn);
            }
            if (th == BH - 1) {   // this tap's share of the slot's staging items
#pragma unroll
                for (int it = 0; it < NST; ++it)
                    if (g == SLO + (it * (KT - SLO)) / NST) stage_store_one((c + 1) & 1, slot, it);
            }
            AMP_PIN_HALFTAP();
            AMP_STAMP(s == 1 && (c <= 1 || c >= NR - 2), 64 + 32 * (c <= 1 ? c : c - (NR - 4)) + h);
        }
        AMP_STAMP(s == 1, 3 + 3 * c);
        __syncthreads();
        AMP_STAMP(s == 1, 4 + 3 * c);
    };

    FragC zf;
    unsigned zero = 0u;
#pragma clang loop unroll(disable)
    for (int s = 0; s < nst; ++s) {
        set_voy(tile0 + s - 1, s > 0, vst);
        set_voy(tile0 + s + 1, s + 1 < nst, vld);
        round(std::integral_constant<int, RK_ST>{}, 0, s);
#pragma unroll 1
        for (int c = 1; c < LDR; ++c) round(std::integral_constant<int, 0>{}, c, s);
        if constexpr (CV_IN_ROUND) {
            round(std::integral_constant<int, RK_LD>{}, LDR, s);
            round(std::integral_constant<int, RK_CV | RK_LAST>{}, NR - 1, s);
        } else {
            round(std::integral_constant<int, RK_LD | RK_LAST>{}, LDR, s);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float bv = bias_at(mi, r);
#pragma unroll
                    for (int t = 0; t < NI; ++t) Y[mi][t][r] = RES ? (bv + Y[mi][t][r]) * cvm : bv * cvm;
                }
        }
        if (SUM) {   // running MRF sum (one conv in six, its own instantiation): the next tile's sum is read here, exposed
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int soff = row_of(mi, r) * Tout4;
#pragma unroll
                    for (int t = 0; t < NI; ++t) Y[mi][t][r] = (Y[mi][t][r] + ldf(ry, vld[t], soff)) * asc;
                    if ((r & 3) == 3) asm volatile("" ::: "memory");   // 16 loads in flight at a time (registers)
                }
        }
        // X <-> Y through the matrix pipe (D = 0 * 0 + C), as a ROTATION over the eight 32 x 32 tiles k = 4 * mi + t: X_k <- Y_k (the next step's
        // start), Y_k <- the old X_(k + 1): 15 MFMAs and one spare tile.  Afterwards tile k's results sit in Y_((k + 7) % 8) -- yres() below.
        // The scheduling barriers keep the order: left alone the scheduler batches the independent copies and needs a spare tile for each
        // (128 registers: the A ring was spilled across the boundary).
        AMP_STAMP(s == 1, 56);
        asm volatile("" : "+v"(zero));       // opaque: a zero the compiler cannot fold the MFMAs around
        zf.u = make_uint4(zero, zero, zero, zero);
        {
            // (the spare tile lives in VGPRs, moved by hand: every MFMA of this kernel has its C / D in AGPRs, which X and Y fill completely)
            float spare[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                spare[r] = X[0][0][r];
                asm volatile("" : "+v"(spare[r]));
            }
#pragma unroll
            for (int k = 0; k < MI * NI; ++k) {
                X[k / NI][k % NI] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf.h, zf.h, Y[k / NI][k % NI], 0, 0, 0);
                if (k + 1 < MI * NI) {
                    Y[k / NI][k % NI] = __builtin_amdgcn_mfma_f32_32x32x16_f16(zf.h, zf.h, X[(k + 1) / NI][(k + 1) % NI], 0, 0, 0);
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[k / NI][k % NI][r] = spare[r];
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        AMP_STAMP(s == 1, 57);
    }
    AMP_STAMP(true, 58);
    {   // the last step's results
        int vo[NI];
        set_voy(tile0 + nst - 1, true, vo);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int soff = row_of(mi, r) * Tout4;
#pragma unroll
                for (int t = 0; t < NI; ++t) stf(fin(yres(mi, t)[r]), ry, vo[t], soff);
            }
    }
#ifdef AMP_STRIP_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    AMP_STAMP(true, 59);
    if (a.range_flag && __any(range_max > 65504.f) && lane == 0) atomicOr(a.range_flag, 1u);
}

template <int KT, int WM, int WN, int NI, int CM, int RING, bool RES, bool SUM>
static hipError_t launch_strip_conv_one(const ConvArgs& a, hipStream_t stream) {
    constexpr int S = 32 * NI * WN + strip_halo(32 * NI * WN);
    constexpr int C = 64 * WM;
    const size_t lds = (size_t)2 * CM * 4 * S * sizeof(uint4) + C * sizeof(float);
    if (hipError_t e = ensure_dynamic_lds<&conv_strip_kernel<KT, WM, WN, NI, CM, RING, RES, SUM>>(lds); e != hipSuccess) return e;
    dim3 grid((unsigned)(a.B * a.strips_per_item));
    note_kernel("conv_strip_kernel", KT, WM, WN, NI, CM, RING, (int)RES, (int)SUM);
    note_conv_work(a, KT, grid);
    hipLaunchKernelGGL((conv_strip_kernel<KT, WM, WN, NI, CM, RING, RES, SUM>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

// Columns per step for a C-channel conv with this tap count, 0 = not covered
constexpr int kStripNI = 3;   // 64 rows x 96 columns per wave: X and Y are 2 x 96 of the 256 AGPRs.  (With 128-column wave tiles X and Y fill the
                              // accumulator file exactly, and hipcc spills residual values as they arrive -- behind an s_waitcnt vmcnt(0) each.)
int AMP_CAT(conv_strip_nt_kt, AMP_KT)(int C, int halo_total) {
    if (halo_total > 64) return 0;
    return C == 256 ? 32 * kStripNI : C == 128 ? 64 * kStripNI : C == 64 ? 128 * kStripNI : 0;
}

// the caller guarantees: Conv1d, Cin == Cout == M == C, Tout == Tin, k == AMP_KT, mode in {0, 1} (1 only with a residual), slope_out <= 1, no reflection padding,
// no tanh, C * T * 4 < 2^31, tiles of conv_strip_nt_kt*() columns, strip_steps / strips_per_item set
hipError_t AMP_CAT(launch_conv_strip_kt, AMP_KT)(const ConvArgs& a, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    constexpr int CMX = KT == 3 ? 4 : 2;      // chunks per staging round / barrier: 200+ MFMAs per wave between barriers (C = 64 has four chunks: two)
    constexpr int RING = 4;
    const bool res = a.res != nullptr, sum = a.mode != 0;
    if (a.mode != 0 && a.mode != 1) return hipErrorInvalidValue;
    if (sum && !res) return hipErrorInvalidValue;
#define AMP_STRIP_SHAPE(WM_, WN_)                                                                              \
    return sum ? launch_strip_conv_one<KT, WM_, WN_, kStripNI, (WM_ == 1 && CMX > 2 ? 2 : CMX), RING, true, true>(a, stream)                 \
               : res ? launch_strip_conv_one<KT, WM_, WN_, kStripNI, (WM_ == 1 && CMX > 2 ? 2 : CMX), RING, true, false>(a, stream)          \
                     : launch_strip_conv_one<KT, WM_, WN_, kStripNI, (WM_ == 1 && CMX > 2 ? 2 : CMX), RING, false, false>(a, stream);
    switch (a.Cout) {
        case 256: AMP_STRIP_SHAPE(4, 1)
        case 128: AMP_STRIP_SHAPE(2, 2)
        case 64: AMP_STRIP_SHAPE(1, 4)
    }
#undef AMP_STRIP_SHAPE
    return hipErrorInvalidValue;
}

}  // namespace amp
