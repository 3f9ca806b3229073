// What does a workgroup cost before it computes anything?  (round 5: ampb_f16x3 with every phase knocked out still takes 0.3-0.4 ms for 2 500-5 000
// workgroups, profiles/r5_c_ampb_knockouts.txt.)  An almost empty kernel -- one float4 load and one store per lane, NSYNC barriers -- launched with the grid,
// block, dynamic LDS and register footprint of the real forms.
// hipcc --offload-arch=gfx950 -O3 tests/experiments/wg_launch_cost.hip -o tests/experiments/wg_launch_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int NTHR, int NREG>
__global__ __launch_bounds__(NTHR, 2) void k(const float4* in, float4* out, int nsync, int spin) {
    extern __shared__ float4 lds[];
    float r[NREG];
    const float4 v = in[blockIdx.x * NTHR + threadIdx.x];
#pragma unroll
    for (int i = 0; i < NREG; ++i) r[i] = v.x * (float)(i + 1);
    lds[threadIdx.x] = v;
    for (int s = 0; s < nsync; ++s) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NREG; ++i) r[i] = __builtin_fmaf(r[i], 1.0001f, lds[(threadIdx.x + s) % NTHR].y);
        for (int j = 0; j < spin; ++j)
#pragma unroll
            for (int i = 0; i < NREG; ++i) r[i] = __builtin_fmaf(r[i], 1.0001f, 0.5f);
    }
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < NREG; ++i) acc += r[i];
    out[blockIdx.x * NTHR + threadIdx.x] = make_float4(acc, v.y, v.z, v.w);
}

template <int NTHR, int NREG>
static float run(int wgs, size_t lds, int nsync, int spin, const float4* in, float4* out) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k<NTHR, NREG>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(e0);
        k<NTHR, NREG><<<wgs, NTHR, lds>>>(in, out, nsync, spin);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best * 1e3f;
}

int main() {
    float4 *in, *out;
    hipMalloc(&in, 8192 * 512 * 16); hipMalloc(&out, 8192 * 512 * 16);
    hipMemset(in, 0, 8192 * 512 * 16);
    printf("us per launch                                 LDS 1 KB   40 KB   80 KB  150 KB\n");
    for (int nsync : {0, 12}) {
        printf("4 960 WGs x 256 thr, 200 regs, %2d barriers     %7.1f %7.1f %7.1f %7.1f\n", nsync, run<256, 200>(4960, 1024, nsync, 0, in, out),
               run<256, 200>(4960, 40 << 10, nsync, 0, in, out), run<256, 200>(4960, 80 << 10, nsync, 0, in, out), run<256, 200>(4960, 150 << 10, nsync, 0, in, out));
        printf("4 960 WGs x 256 thr,  32 regs, %2d barriers     %7.1f %7.1f %7.1f %7.1f\n", nsync, run<256, 32>(4960, 1024, nsync, 0, in, out),
               run<256, 32>(4960, 40 << 10, nsync, 0, in, out), run<256, 32>(4960, 80 << 10, nsync, 0, in, out), run<256, 32>(4960, 150 << 10, nsync, 0, in, out));
        printf("2 528 WGs x 512 thr, 200 regs, %2d barriers     %7.1f %7.1f %7.1f %7.1f\n", nsync, run<512, 200>(2528, 1024, nsync, 0, in, out),
               run<512, 200>(2528, 40 << 10, nsync, 0, in, out), run<512, 200>(2528, 80 << 10, nsync, 0, in, out), run<512, 200>(2528, 150 << 10, nsync, 0, in, out));
    }
    // the same with ~20 us of arithmetic per workgroup (does the fixed cost hide behind a neighbour's work?)
    printf("4 960 WGs x 256 thr, 200 regs, 12 barriers, spin 40:   80 KB %7.1f   1 KB %7.1f\n", run<256, 200>(4960, 80 << 10, 12, 40, in, out), run<256, 200>(4960, 1024, 12, 40, in, out));
    return 0;
}
