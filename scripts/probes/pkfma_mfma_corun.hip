// Standalone reproduction attempt (round 6): does v_pk_fma_f32 lose a product in lanes 48..63 while another wave on the SIMD runs
// v_mfma_f32_16x16x32_f16?  Victim: chains of in-place packed FMAs on small integers (exact in fp32), one kernel per operand-select
// form; culprit: a loop of matrix-core instructions of one shape.  Both are launched on two streams at once; the victim counts, per
// lane, results that differ from the closed form.
//   hipcc --offload-arch=gfx950 -O2 scripts/probes/pkfma_mfma_corun.hip -o /tmp/pkfma && /tmp/pkfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int FORM>
__global__ __launch_bounds__(256) void victim(unsigned* bad_per_lane, int iters) {
    const int lane = threadIdx.x & 63;
    f32x2 x = {(float)(lane % 7 + 1), (float)(lane % 5 + 2)};
    f32x2 g = {1.f, 1.f};
    f32x2 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x2{0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(acc[k]) : "v"(g), "v"(x));                 // lo += g.lo * x.HI, hi += g.hi * x.hi
            else if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(g), "v"(x));         // lo += g.lo * x.lo, hi += g.hi * x.LO
            else if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[k]) : "v"(g), "v"(x));                           // plain
            else if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel:[1,0,0]" : "+v"(acc[k]) : "v"(g), "v"(x));            // lo += x.HI * g.lo (the swap in src0)
            else if (FORM == 4) { f32x2 t; asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\tv_pk_add_f32 %3, %3, %0" : "=&v"(t), "+v"(g) , "+v"(x), "+v"(acc[k])); }      // pk_mul with the swap, then a plain pk_add
            else if (FORM == 5) { f32x2 t; asm volatile("v_pk_mul_f32 %0, %1, %2\n\tv_pk_add_f32 %3, %3, %0 op_sel:[0,1]" : "=&v"(t), "+v"(g) , "+v"(x), "+v"(acc[k])); }      // plain pk_mul, pk_add with the swap: lo += t.HI
            else if (FORM == 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "+v"(acc[k]) : "v"(g), "v"(x));       // both halves swapped: lo += x.HI, hi += x.LO
            else if (FORM == 7) asm volatile("v_pk_fma_f32 %0, %2, %1, %0 op_sel_hi:[0,1,1]" : "+v"(acc[k]) : "v"(g), "v"(x));        // src0 low splat (the library's most common form): lo += x.lo, hi += x.LO
            else if (FORM == 8) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,1,0]" : "+v"(acc[k]) : "v"(g), "v"(x));        // src2: hi = x.hi + acc.LO
            else if (FORM == 9) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,1]" : "+v"(acc[k]) : "v"(g), "v"(x));           // src2: lo = x.lo + acc.HI
            else if (FORM == 10) { f32x2 t; asm volatile("v_pk_mul_f32 %0, %2, %1 op_sel_hi:[0,1]\n\tv_pk_add_f32 %3, %3, %0" : "=&v"(t), "+v"(g) , "+v"(x), "+v"(acc[k])); }   // pk_mul src0 low splat
            else { unsigned d; asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(d) : "v"(x[0]), "v"(x[1])); acc[k][0] += (float)(d & 0xffffu); acc[k][1] += (float)(d >> 16); }   // the 16-bit kernels' conversion (small integers: exact)
        }
    }
    const float n = (float)iters;
    float want_lo, want_hi;
    const float xl = x[0], xh = x[1];
    if (FORM == 0 || FORM == 3 || FORM == 4 || FORM == 5) { want_lo = n * xh; want_hi = n * xh; }
    else if (FORM == 1 || FORM == 7 || FORM == 10) { want_lo = n * xl; want_hi = n * xl; }
    else if (FORM == 6) { want_lo = n * xh; want_hi = n * xl; }
    else if (FORM == 8) { want_lo = n * xl; want_hi = (n - 1.f) * xl + xh; }
    else if (FORM == 9) { want_hi = n * xh; want_lo = (n - 1.f) * xh + xl; }
    else if (FORM == 11) { want_lo = n * (float)(__builtin_bit_cast(unsigned, xl) >> 16); want_hi = n * (float)(__builtin_bit_cast(unsigned, xh) >> 16); }
    else { want_lo = n * xl; want_hi = n * xh; }
    unsigned bad = 0;
    for (int k = 0; k < 8; ++k) bad += (acc[k][0] != want_lo) + (acc[k][1] != want_hi);
    if (bad) atomicAdd(&bad_per_lane[lane], bad);
}

template <int SHAPE>
__global__ __launch_bounds__(256) void culprit(float* sink, int iters) {
    f32x4 acc[8];
    for (int k = 0; k < 8; ++k) acc[k] = f32x4{0.f, 0.f, 0.f, 0.f};
    f16x8 a, b;
    f32x16 acc16[2];
    for (int k = 0; k < 2; ++k) for (int e = 0; e < 16; ++e) acc16[k][e] = 0.f;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(threadIdx.x % 3); b[e] = (_Float16)(threadIdx.x % 2); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (SHAPE == 0) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[k], 0, 0, 0);
            else if (SHAPE == 1) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc[k], 0, 0, 0);
            else if (SHAPE == 2) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[0], (float)b[0], acc[k], 0, 0, 0);
            else if (SHAPE == 3) acc[k] = __builtin_amdgcn_mfma_f32_16x16x16f16(__builtin_shufflevector(a, a, 0, 1, 2, 3), __builtin_shufflevector(b, b, 0, 1, 2, 3), acc[k], 0, 0, 0);
            else if (SHAPE == 4) { acc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc16[k & 1], 0, 0, 0); }
            else if (SHAPE == 5) { acc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x8f16(__builtin_shufflevector(a, a, 0, 1, 2, 3), __builtin_shufflevector(b, b, 0, 1, 2, 3), acc16[k & 1], 0, 0, 0); }
            else { acc16[k & 1] = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[0], (float)b[0], acc16[k & 1], 0, 0, 0); }
        }
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
    s += acc16[0][0] + acc16[1][5];
    if (s == 12345.678f) sink[0] = s;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int FORM, int SHAPE>
int trial(const char* fname, const char* sname, bool with_culprit) {
    unsigned* bad; float* sink;
    CK(hipMalloc(&bad, 64 * 4)); CK(hipMalloc(&sink, 4)); CK(hipMemset(bad, 0, 64 * 4));
    hipStream_t s0, s1; CK(hipStreamCreate(&s0)); CK(hipStreamCreate(&s1));
    for (int rep = 0; rep < 6; ++rep) {
        if (with_culprit) hipLaunchKernelGGL((culprit<SHAPE>), dim3(2048), dim3(256), 0, s0, sink, 4000);
        for (int q = 0; q < 4; ++q) hipLaunchKernelGGL((victim<FORM>), dim3(4096), dim3(256), 0, s1, bad, 400);
        CK(hipDeviceSynchronize());
    }
    std::vector<unsigned> h(64); CK(hipMemcpy(h.data(), bad, 64 * 4, hipMemcpyDeviceToHost));
    unsigned q[4] = {0, 0, 0, 0}; for (int l = 0; l < 64; ++l) q[l / 16] += h[l];
    printf("victim %-34s beside %-28s: wrong results by lane quarter [0-15 | 16-31 | 32-47 | 48-63] = %u %u %u %u\n", fname, with_culprit ? sname : "nothing", q[0], q[1], q[2], q[3]);
    hipFree(bad); hipFree(sink); hipStreamDestroy(s0); hipStreamDestroy(s1);
    return 0;
}

int main() {
    const char* F[12] = {"pk_fma op_sel:[0,1,0]", "pk_fma op_sel_hi:[1,0,1]", "pk_fma (plain)", "pk_fma op_sel:[1,0,0]", "pk_mul op_sel:[0,1] + pk_add", "pk_mul + pk_add op_sel:[0,1]", "pk_fma op_sel:[0,1,0] op_sel_hi:[1,0,1]",
                         "pk_fma op_sel_hi:[0,1,1]", "pk_fma op_sel_hi:[1,1,0]", "pk_fma op_sel:[0,0,1]", "pk_mul op_sel_hi:[0,1] + pk_add", "v_cvt_pk_bf16_f32"};
    const char* S[7] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x32_bf16", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x8_f16", "v_mfma_f32_32x32x2_f32"};
    trial<0, 0>(F[0], S[0], false);
#define ROW(f) trial<f, 0>(F[f], S[0], true); trial<f, 1>(F[f], S[1], true); trial<f, 2>(F[f], S[2], true); trial<f, 3>(F[f], S[3], true); trial<f, 4>(F[f], S[4], true); trial<f, 5>(F[f], S[5], true); trial<f, 6>(F[f], S[6], true);
#define ROW3(f) trial<f, 0>(F[f], S[0], true); trial<f, 1>(F[f], S[1], true); trial<f, 4>(F[f], S[4], true);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW3(7) ROW3(8) ROW3(9) ROW3(10) ROW3(11)
    return 0;
}
