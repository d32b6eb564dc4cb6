// Probe: absolute accuracy of the hardware v_cos_f32 (input in revolutions) for the SnakeBeta term
//   sin^2(theta) = (1 - cos(2 theta)) / 2,  cos(2 theta) = v_cos(theta / pi)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* th, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        float t = th[i] * 0.31830988618379067f;      // theta / pi
        out[i] = 0.5f * (1.0f - __builtin_amdgcn_cosf(t));
    }
}
int main() {
    const int n = 1 << 22;
    std::vector<float> h(n), o(n);
    unsigned s = 12345;
    float scales[4] = {1.f, 8.f, 64.f, 400.f};
    for (int sc = 0; sc < 4; ++sc) {
        for (int i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) / 8388608.0f - 1.0f) * scales[sc]; }
        float *d, *e; hipMalloc(&d, n * 4); hipMalloc(&e, n * 4);
        hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        k<<<n / 256, 256>>>(d, e, n);
        hipMemcpy(o.data(), e, n * 4, hipMemcpyDeviceToHost);
        double mx = 0, ref_mx = 0;
        for (int i = 0; i < n; ++i) {
            double tr = sin((double)h[i]); tr *= tr;
            double er = fabs((double)o[i] - tr); if (er > mx) mx = er;
            float sf = sinf(h[i]); double e2 = fabs((double)(sf * sf) - tr); if (e2 > ref_mx) ref_mx = e2;
        }
        printf("|theta| <= %g: max abs err of (1-v_cos)/2 = %.3e   (libm sinf^2: %.3e)\n", scales[sc], mx, ref_mx);
        hipFree(d); hipFree(e);
    }
    return 0;
}
