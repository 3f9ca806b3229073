// Horizontal fusion of a generator stage's resblocks where they run as UNFUSED whole-K convs (round 4; pair3_f16x3.hip is the same for
// the fused pairs): ONE grid runs the same conv -- c1 of a dilation, or c2 -- of the stage's three resblocks, kernel sizes 11 / 7 / 3,
// each on its own tensors.  This is the C = 256 stage of a single utterance (conv_small_f16x3.hip: 64-workgroup launches of 13-38 us,
// 18 of them in a row).  Per element the arithmetic is conv_small_body's own: same bits as three conv_small_kernel launches.
#include "conv_small_f16x3_body.h"

namespace amp {

// slot 0: k = 11 with NI0 x 32-column tiles (64-column tiles when its dilated receptive field needs the 64-column halo); slots 1 / 2:
// k = 7 / 3, 32-column tiles
template <int NI0, int HALO0>
__global__ __launch_bounds__(256, 1) void conv_small3_kernel(const ConvSmall3Args p) {
    const int b = (int)blockIdx.x;                         // workgroup-uniform dispatch; within a slot: column tile fastest
    const int n0 = p.nx[0] * p.ny[0], n1 = p.nx[1] * p.ny[1];
    if (b < n0) conv_small_body<11, NI0, HALO0, 0>(p.a[0], b % p.nx[0], b / p.nx[0]);
    else if (b < n0 + n1) conv_small_body<7, 1, 32, 0>(p.a[1], (b - n0) % p.nx[1], (b - n0) / p.nx[1]);
    else conv_small_body<3, 1, 32, 0>(p.a[2], (b - n0 - n1) % p.nx[2], (b - n0 - n1) / p.nx[2]);
}

template <int NI0, int HALO0>
static hipError_t launch_small3_one(const ConvSmall3Args& p, hipStream_t stream) {
    size_t lds = 0;
    for (int j = 0; j < 3; ++j) {
        const int kt = j == 0 ? 11 : j == 1 ? 7 : 3;
        const int S = j == 0 ? 32 * NI0 + HALO0 : 64;
        const int AR = kt <= 3 ? 4 : 2;                    // ARing<KT>::n
        const size_t l = (size_t)((p.a[j].nchunks + AR - 1) / AR * AR) * 4 * S * sizeof(uint4);
        if (p.a[j].nchunks > kSmallMaxChunks || l > kSmallMaxLds || p.a[j].wd > S) return hipErrorInvalidValue;
        lds = l > lds ? l : lds;
    }
    if (hipError_t e = ensure_dynamic_lds<&conv_small3_kernel<NI0, HALO0>>(kSmallMaxLds); e != hipSuccess) return e;
    dim3 grid((unsigned)(p.nx[0] * p.ny[0] + p.nx[1] * p.ny[1] + p.nx[2] * p.ny[2]));
    note_kernel("conv_small3_kernel", NI0, HALO0);
    note_work(grid.x, 0.0, 0.0, "three frame-rate convs side by side (work not itemised)");
    hipLaunchKernelGGL((conv_small3_kernel<NI0, HALO0>), grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

// ni[j]: tile width of slot j in 32-column units as conv_run chose it; covered: (1, 1, 1) and (2, 1, 1)
hipError_t launch_conv_small3(const ConvSmall3Args& p, const int ni[3], hipStream_t stream) {
    if (ni[1] != 1 || ni[2] != 1) return hipErrorInvalidValue;
    if (ni[0] == 1) return launch_small3_one<1, 32>(p, stream);
    if (ni[0] == 2) return launch_small3_one<2, 64>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace amp
