// Probe: does v_mfma_f32_16x16x32_f16 honour SUBNORMAL fp16 inputs (or flush them to zero)?  A = 1.0 everywhere, B = 2^-20 (a subnormal half:
// 16 * 2^-24) everywhere: D[i][j] = 32 * 2^-20 = 3.0518e-05 if subnormals are honoured, 0 if flushed.  Also B = 2^-14 (smallest normal) as a control.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void probe(float* out, unsigned short bbits) {
    f16x8 a, b;
    const _Float16 one = (_Float16)1.0f;
    const _Float16 bv = __builtin_bit_cast(_Float16, bbits);
    for (int i = 0; i < 8; ++i) { a[i] = one; b[i] = bv; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 64 * sizeof(float));
    float h[64];
    const unsigned short cases[3] = {0x0010 /* 2^-20: subnormal */, 0x0001 /* 2^-24: smallest subnormal */, 0x0400 /* 2^-14: smallest normal */};
    const char* names[3] = {"B = 2^-20 (subnormal)", "B = 2^-24 (smallest subnormal)", "B = 2^-14 (smallest normal)"};
    const double want[3] = {32.0 / 1048576.0, 32.0 / 16777216.0, 32.0 / 16384.0};
    for (int k = 0; k < 3; ++k) {
        probe<<<1, 64>>>(d, cases[k]);
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("%s: D = %.6e  (honoured: %.6e)  -> %s\n", names[k], h[0], want[k], h[0] == (float)want[k] ? "HONOURED" : (h[0] == 0.f ? "FLUSHED" : "OTHER"));
    }
    return 0;
}
