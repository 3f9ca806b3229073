// Accuracy of the hardware sine (v_sin_f32, input in revolutions) for Snake's sin(a u)^2, against double precision:
// would it replace the 13-FMA Cody-Waite + Taylor path of act1d (small_kernels.hip: snake_sin2, 1.1e-7)?
//   hipcc --offload-arch=gfx950 -O3 vsin_accuracy.hip -o vsin_accuracy
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <vector>

__global__ void k(const float* x, float* hw, float* hw_red, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    // (a) straight: revolutions = x / (2 pi)
    const float s = __builtin_amdgcn_sinf(v * 0.15915494309189535f);
    hw[i] = s * s;
    // (b) with an explicit fract first (sin^2 has period pi -> half revolutions)
    float r = v * 0.31830988618379067f;     // x / pi
    r = r - rintf(r);                        // [-0.5, 0.5] half-revolutions
    const float s2 = __builtin_amdgcn_sinf(0.5f * r);
    hw_red[i] = s2 * s2;
}

int main() {
    const int n = 1 << 22;
    std::vector<float> h(n);
    srand(1);
    for (int i = 0; i < n; ++i) {
        const float u = (float)rand() / (float)RAND_MAX * 2.f - 1.f;
        const int cls = i & 3;
        h[i] = cls == 0 ? u * 3.2f : cls == 1 ? u * 30.f : cls == 2 ? u * 1000.f : u * 1e5f;
    }
    float *dx, *d1, *d2;
    hipMalloc(&dx, n * 4); hipMalloc(&d1, n * 4); hipMalloc(&d2, n * 4);
    hipMemcpy(dx, h.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d1, d2, n);
    std::vector<float> a(n), b(n);
    hipMemcpy(a.data(), d1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d2, n * 4, hipMemcpyDeviceToHost);
    double e1[4] = {0, 0, 0, 0}, e2[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const double s = sin((double)h[i]), ref = s * s;
        const int cls = i & 3;
        e1[cls] = fmax(e1[cls], fabs(a[i] - ref));
        e2[cls] = fmax(e2[cls], fabs(b[i] - ref));
    }
    const char* names[4] = {"|x|<=3.2", "|x|<=30", "|x|<=1e3", "|x|<=1e5"};
    for (int c = 0; c < 4; ++c) printf("sin^2 max abs err %-9s: v_sin_f32(x/2pi) %.3e   fract + v_sin_f32 %.3e\n", names[c], e1[c], e2[c]);
    return 0;
}
