// Fused ResBlock pair on the gfx950 f16 matrix cores (split-f16 operands, see conv_f16x3.hip):
//
//     y = x + c2( lrelu( c1( lrelu(x) ) ) )         [+ the running MRF sum, / num_kernels]
//
// i.e. ONE iteration of ResBlock1.forward (hifigan.py:93-100: xt = lrelu(x); xt = c1(xt); xt = lrelu(xt);
// xt = c2(xt); x = xt + x) in one kernel.  Unfused, a pair moves 5 tensors through HBM (read x, write
// xt, read xt, read x again as the residual, write y); here xt never leaves the CU and the residual
// re-read is an L2 hit, so a pair is one read and one write of the [B, C, T] tensor.
//
//   workgroup = all C channels x NT output columns of one batch item.
//   phase 1   conv1 (kernel KT, dilation d) as in conv_f16x3.hip -- K loop over 16-channel chunks of x
//             staged through a double-buffered LDS tile -- on N1 = NT + (KT - 1) columns: exactly the
//             columns conv2 needs, so there is no halo recompute beyond the tile seam itself
//             (N1 is a multiple of 32; NT is not: rows are stored at 4-B granularity).
//   seam      bias + leaky_relu + conv2's zero padding + x16 + hi/lo split in registers, written to the
//             LDS xt tile in B-fragment layout [chunk][plane hi|lo][octet][column][8 x f16].
//   phase 2   conv2 (kernel KT, dilation 1): K loop over the xt chunks in LDS -- no staging, no
//             barriers; A fragments streamed from L2 with the one-chunk-ahead rotation.
//   epilogue  + bias, + residual x (global, L2-hot), MRF accumulate, store.
//
// Compiled once per tap count:  -DAMP_KT=<3|5|7|11>.
#include "pair_f16x3_body.h"

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

template <int KT, int WM, int WN, int NI, int SX>
__global__ __launch_bounds__(256, 2) void pair_f16x3_kernel(const PairArgs a) {
    pair_f16x3_body<KT, WM, WN, NI, SX>(a, (int)blockIdx.x, (int)gridDim.x);
}

template <int KT, int WM, int WN, int NI, int SX>
static hipError_t launch_pair_one(const PairArgs& a, hipStream_t stream) {
    constexpr int N1 = 32 * NI * WN;
    constexpr int XT = N1 + 12;
    const size_t lds = ((size_t)2 * 4 * SX + (size_t)2 * WM * 4 * XT) * sizeof(uint4);
    if (hipError_t e = ensure_dynamic_lds<&pair_f16x3_kernel<KT, WM, WN, NI, SX>>(lds); e != hipSuccess) return e;
    dim3 grid((unsigned)(a.B * a.tiles_per_item));
    note_kernel("pair_f16x3_kernel", KT, WM, WN, NI, SX);
    note_work(grid.x, 2 * 2.0 * a.C * a.C * KT * (double)a.T * a.B / 1e9, 2 * 4.0 * a.B * (double)a.C * a.T * (1.0 + (a.mode ? 0.5 : 0.0)) / 1e6,
              "fused pair C=%d k=%d d=%d T=%d B=%d%s", a.C, KT, a.dil, a.T, a.B, a.mode ? " +sum" : "");
    hipLaunchKernelGGL((pair_f16x3_kernel<KT, WM, WN, NI, SX>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

// Output columns per workgroup for C channels, or 0 when (C, KT, dilation) is not covered.
int AMP_CAT(pair_tile_kt, AMP_KT)(int C, int dil) {
    constexpr int KT = AMP_KT;
    const int span = (KT - 1) * dil;   // 2 * h1
    if (C == 128) return (96 + span <= 192) ? 96 - (KT - 1) : 0;
    if (C == 64) return (128 + span <= 192) ? 128 - (KT - 1) : 0;
    if (C == 32) return (256 + span <= 320) ? 256 - (KT - 1) : 0;
    return 0;
}

hipError_t AMP_CAT(launch_pair_kt, AMP_KT)(const PairArgs& a, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    if (a.C == 128) return launch_pair_one<KT, 4, 1, 3, 192>(a, stream);
    if (a.C == 64) return launch_pair_one<KT, 2, 2, 2, 192>(a, stream);
    if (a.C == 32) return launch_pair_one<KT, 1, 4, 2, 320>(a, stream);
    return hipErrorInvalidValue;
}

}  // namespace amp
