// Horizontal fusion of a generator stage's resblocks (round 4): ONE grid runs the fused ResBlock pair (pair_f16x3_body.h) of up to three
// resblocks side by side -- kernel sizes 11 / 7 / 3 of HiFi-GAN's MRF (hifigan.py:208-214) -- each on its own tensors.
//
// Why.  The resblocks of a stage are independent until the MRF mean, but a stream runs them one after the other, and for a single
// utterance none of their launches fills the chip (50-300 workgroups on 256 CUs): a 3-s utterance is 52 dependent launches = 1.04 ms.
// Running them on three streams (generator.hip, concurrent mode) overlaps them but pays ~12 us for every cross-queue dependency
// (fork, join): 0.81 ms.  Here the three bodies share a launch: no events, one stream, a third of the launches, and the widest kernel's
// workgroups are dispatched first.  The per-element arithmetic is the body's own: same bits as three pair_f16x3_kernel launches
// (tests/test_gpu_resblock.py).
#include "pair_f16x3_body.h"

namespace amp {

template <int K0, int K1, int K2, int WM, int WN, int NI, int SX>
__global__ __launch_bounds__(256, 2) void pair3_kernel(const Pair3Args p) {
    const int b = (int)blockIdx.x;                         // workgroup-uniform dispatch
    if (b < p.n[0]) pair_f16x3_body<K0, WM, WN, NI, SX>(p.a[0], b, p.n[0]);
    else if (b < p.n[0] + p.n[1]) pair_f16x3_body<K1, WM, WN, NI, SX>(p.a[1], b - p.n[0], p.n[1]);
    else pair_f16x3_body<K2, WM, WN, NI, SX>(p.a[2], b - p.n[0] - p.n[1], p.n[2]);
}

template <int WM, int WN, int NI, int SX>
static hipError_t launch_pair3_one(const Pair3Args& p, hipStream_t stream) {
    constexpr int N1 = 32 * NI * WN;
    constexpr int XT = N1 + 12;
    const size_t lds = ((size_t)2 * 4 * SX + (size_t)2 * WM * 4 * XT) * sizeof(uint4);     // independent of the tap count
    if (hipError_t e = ensure_dynamic_lds<&pair3_kernel<11, 7, 3, WM, WN, NI, SX>>(lds); e != hipSuccess) return e;
    dim3 grid((unsigned)(p.n[0] + p.n[1] + p.n[2]));
    note_kernel("pair3_kernel", 11, 7, 3, WM, WN);
    note_work(grid.x, 0.0, 0.0, "three fused pairs side by side (work not itemised)");
    hipLaunchKernelGGL((pair3_kernel<11, 7, 3, WM, WN, NI, SX>), grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

// p.a[0] / [1] / [2]: the pairs with kernel sizes 11 / 7 / 3, all on C = a[0].C channels (the tile forms of pair_f16x3.hip)
hipError_t launch_pair3(const Pair3Args& p, hipStream_t stream) {
    const int C = p.a[0].C;
    if (p.a[1].C != C || p.a[2].C != C) return hipErrorInvalidValue;
    if (C == 128) return launch_pair3_one<4, 1, 3, 192>(p, stream);
    if (C == 64) return launch_pair3_one<2, 2, 2, 192>(p, stream);
    if (C == 32) return launch_pair3_one<1, 4, 2, 320>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace amp
