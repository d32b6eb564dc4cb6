// Probe: what does the fp32 matrix pipe SUSTAIN on this chip?  Nothing but v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32 from registers:
// 8 independent accumulators per wave, 8 waves per CU, no memory traffic, ~45 ms per run.  Peak at 2.4 GHz is 256 CUs x 4 SIMDs x 64 FLOP /
// cycle = 157.3 TFLOP/s; what comes out below it is the clock the chip holds under this load (DVFS).  TOGGLE = 0: the same operand pair every
// time (multiplier inputs do not toggle: the low-power case); TOGGLE = 1: consecutive MFMAs multiply different pseudo-random operand pairs
// (eight pairs per lane in registers, static indices: no extra instruction), the data-dependent power of a real GEMM.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// operand bits of a float in [-0.5, 0.5) from a 32-bit state: mantissa = state bits (finite, no denormals)
__device__ __forceinline__ float as_operand(unsigned u) { return __builtin_bit_cast(float, 0x3f800000u | (u & 0x007fffffu)) - 1.5f; }
template <int SHAPE, int TOGGLE>
__global__ __launch_bounds__(512) void burn(float* out, int iters, float seed) {
    // TOGGLE: eight pseudo-random operand pairs per lane held in registers; consecutive MFMAs take DIFFERENT pairs (static register
    // indices in the unrolled loop: no extra instruction), so every multiplier input bit toggles between instructions
    float a[8], b[8];
    unsigned u = 0x9e3779b9u * (threadIdx.x + 1) + (unsigned)(seed * 1000.f);
    for (int k = 0; k < 8; ++k) {
        u ^= u << 13; u ^= u >> 17; u ^= u << 5; a[k] = TOGGLE ? as_operand(u) : 0.37f;
        u ^= u << 13; u ^= u >> 17; u ^= u << 5; b[k] = TOGGLE ? as_operand(u) : -0.21f;
    }
    float s = 0.f;
    if (SHAPE == 32) {
        f32x16 c[8];
        for (int k = 0; k < 8; ++k) for (int r = 0; r < 16; ++r) c[k][r] = 0.f;
        for (int i = 0; i < iters; i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(k + j) & 7], b[(3 * k + j) & 7], c[k], 0, 0, 0);
        for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][15];
    } else {
        f32x4 c[8];
        for (int k = 0; k < 8; ++k) for (int r = 0; r < 4; ++r) c[k][r] = 0.f;
        for (int i = 0; i < iters; i += 8)
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + j) & 7], b[(3 * k + j) & 7], c[k], 0, 0, 0);
        for (int k = 0; k < 8; ++k) s += c[k][0] + c[k][3];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int SHAPE, int TOGGLE>
static void run(const char* name, int blocks, int iters) {
    float* d; hipMalloc(&d, (size_t)blocks * 512 * sizeof(float));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    burn<SHAPE, TOGGLE><<<blocks, 512>>>(d, iters / 10, 0.5f);           // warm up (clocks ramp)
    hipDeviceSynchronize();
    for (int rep = 0; rep < (iters > 1000000 ? 1 : 3); ++rep) {
        hipEventRecord(e0);
        burn<SHAPE, TOGGLE><<<blocks, 512>>>(d, iters, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 8 /* waves */ * iters * 8 /* accumulators */ * (SHAPE == 32 ? 4096.0 : 2048.0);
        printf("%s: %d blocks x 8 waves, %.1f ms -> %.1f TFLOP/s = %.3f of 157.3 (clock-equivalent %.2f GHz)\n", name, blocks, ms, flop / ms / 1e9,
               flop / ms / 1e9 / 157.3, 2.4 * flop / ms / 1e9 / 157.3);
    }
    hipFree(d);
}
int main() {
    run<32, 0>("v_mfma_f32_32x32x2_f32, constant operands", 256, 100000);
    run<32, 1>("v_mfma_f32_32x32x2_f32, toggling random operands", 256, 100000);
    run<16, 0>("v_mfma_f32_16x16x4_f32, constant operands", 256, 200000);
    run<16, 1>("v_mfma_f32_16x16x4_f32, toggling random operands", 256, 200000);
    run<32, 1>("v_mfma_f32_32x32x2_f32, toggling random operands, 3 s of load", 256, 100000 * 30);
    return 0;
}
