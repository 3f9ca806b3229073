// Frame-rate convolutions (short time axis, short contraction) on the gfx950 f16 matrix cores: the f16x3 arithmetic of
// conv_f16x3.hip with the WHOLE K extent of the input tile staged in LDS at once.
//
// Why a second kernel.  conv_f16x3.hip is built for the vocoder stacks: K = C*k up to 2816, grids of thousands of
// workgroups, hundreds of MFMAs per 16-channel chunk, two workgroups per CU covering each other's latencies.  The
// frame-rate convs around the VITS decoder (modules/flow/modules.py:106-124 WN in_layers / res_skip_layers,
// models/tts/vits/vits.py:135,143 pre / proj, modules/transformer/attentions.py FFN / projections) have K = 192 .. 960
// and grids of <= 400 workgroups: there a launch is bound by the INSTRUCTIONS each wave issues around its few MFMAs
// (rocprofv3: 15 us for the 1x1 192 -> 384 conv of a WN layer at B = 16, T = 256 with 1 us of MFMA work per wave, 23 us
// for the k5 conv with 5 us) and by the element-wise launches between them.  Here a workgroup
//   1. issues ALL global loads of its input tile (every chunk) and the first A fragments at once: one memory latency,
//   2. converts (leaky_relu-on-load, zero padding, x16, hi/lo split: identical to conv_f16x3.hip) into
//      nchunks LDS buffers, ONE barrier,
//   3. runs the K loop barrier-free out of LDS with the A fragments prefetched AR chunk-sets ahead and the B fragments
//      one tap ahead,
// with straight-line, branch-free address arithmetic (uniform row pointers + 32-bit lane offsets).
// Per output element the chunk / tap / (hh, hl, lh) order is that of conv_f16x3.hip, so the standard epilogue gives the
// same bits as that kernel (tests/test_gpu_conv.py::test_small_conv_bitwise).
//
// Two more epilogues fuse a WN layer (modules/flow/modules.py:126-151) into two launches instead of four:
//   EPI_GATE  in_layers[i] + fused_add_tanh_sigmoid_multiply (utils/util.py:602-609): the 2H GEMM rows are packed
//             on the host so that every 32-row block holds the tanh AND the sigmoid half of 16 channels (a lane
//             then owns both pre-activations of its 8 channels): acts = tanh(a + g_t) * sigmoid(b + g_s) is formed in
//             registers and [B, H, T] is written instead of [B, 2H, T].
//   EPI_WNACC res_skip_layers[i] + the residual / skip update (:144-151): rows < H go to x = (x + v) * mask in place,
//             rows >= H to output (+)= v; the last layer's H rows all go to output.
//
// Tile variants: 128 rows x 32 columns with a 32-column halo (S = 64 staged columns, two workgroups per CU) or
// 128 x 64 with a 64-column halo (S = 128, one workgroup per CU).
//
// Compiled once per tap count: -DAMP_KT=<1|3|5|7|11> (7 and 11: the C = 256 stage of a single utterance, whose 64-column
// tiles of the pipelined kernel leave three quarters of the chip idle).
#include "conv_small_f16x3_body.h"

#ifndef AMP_KT
#error "compile with -DAMP_KT=<taps>"
#endif

namespace amp {

template <int KT, int NI, int HALO, int EPI>
__global__ __launch_bounds__(256, ((NI == 1 && KT <= 5) ? 2 : 1)) void conv_small_kernel(const ConvArgs a) {   // KT >= 7: the A ring alone is 112 / 176 VGPRs
    conv_small_body<KT, NI, HALO, EPI>(a, (int)blockIdx.x, (int)blockIdx.y);
}

template <int KT, int NI, int HALO, int EPI>
static hipError_t launch_small_one(const ConvArgs& a, hipStream_t stream) {
    constexpr int S = 32 * NI + HALO;
    constexpr int AR = ARing<KT>::n;
    const size_t lds = (size_t)((a.nchunks + AR - 1) / AR * AR) * 4 * S * sizeof(uint4);
    if (a.nchunks > kSmallMaxChunks || lds > kSmallMaxLds || a.wd > S) return hipErrorInvalidValue;
    if (hipError_t e = ensure_dynamic_lds<&conv_small_kernel<KT, NI, HALO, EPI>>(kSmallMaxLds); e != hipSuccess) return e;
    dim3 grid((unsigned)(a.B * a.tiles_per_item), (unsigned)((a.M + 127) / 128));
    note_kernel("conv_small_kernel", KT, NI, HALO, EPI);
    note_conv_work(a, KT, grid);
    hipLaunchKernelGGL((conv_small_kernel<KT, NI, HALO, EPI>), grid, dim3(256), lds, stream, a);
    return hipGetLastError();
}

#define AMP_CAT2(a, b) a##b
#define AMP_CAT(a, b) AMP_CAT2(a, b)

// epi: 0 standard, 1 gate, 2 WN accumulate; ni: 1 (128 x 32 tiles, halo <= 32) or 2 (128 x 64 tiles, halo <= 64).
// The caller sets a.tiles_per_item / a.wd for the tile width it asks for.
hipError_t AMP_CAT(launch_conv_small_kt, AMP_KT)(int ni, int epi, const ConvArgs& a, hipStream_t stream) {
    constexpr int KT = AMP_KT;
    if (ni == 1) {
        if (epi == 0) return launch_small_one<KT, 1, 32, 0>(a, stream);
        if (epi == 1) return launch_small_one<KT, 1, 32, 1>(a, stream);
        return launch_small_one<KT, 1, 32, 2>(a, stream);
    }
    if (epi == 0) return launch_small_one<KT, 2, 64, 0>(a, stream);
    if (epi == 1) return launch_small_one<KT, 2, 64, 1>(a, stream);
    return launch_small_one<KT, 2, 64, 2>(a, stream);
}

}  // namespace amp
