// svcmi_rt.h -- the one include every kernel file uses.
//
// Product build (hipcc --offload-arch=gfx950): pulls in the HIP runtime and defines the few
// wrappers below on top of gfx950 builtins.  There is no other backend in the product.
//
// Test build (-DSVCMI_EMU, g++): tests/emu/hip_emu.h supplies a fiber-based SIMT emulator so the
// SAME kernel source can be executed on the CPU by `pytest -m "not gpu"` to check indexing/tiling
// logic where no GPU exists.  It is test infrastructure (see tests/emu/README.md); the Python
// package never loads it.
#pragma once

#ifdef SVCMI_EMU
#include "hip_emu.h"
#else
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float svcmi_f32x16 __attribute__((ext_vector_type(16)));
typedef float svcmi_f32x4 __attribute__((ext_vector_type(4)));

// D = A(32x2) * B(2x32) + C, exact fp32 (v_mfma_f32_32x32x2_f32).  Lane l supplies
// A[i=l&31][k=l>>5] and B[k=l>>5][j=l&31]; C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
__device__ __forceinline__ svcmi_f32x16 svcmi_mfma_32x32x2(float a, float b, svcmi_f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// D = A(16x4) * B(4x16) + C, exact fp32 (v_mfma_f32_16x16x4_f32; 32-cycle issue, 40-cycle dependent latency, so
// keep >= 2 independent accumulators in flight).  Lane l supplies A[i=l&15][k=l>>4] and B[k=l>>4][j=l&15];
// C/D: col = l&15, row = 4*(l>>4) + r.
__device__ __forceinline__ svcmi_f32x4 svcmi_mfma_16x16x4(float a, float b, svcmi_f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// Asynchronous global -> LDS copy (LDS-DMA, global_load_lds_dword[x4]).  `lds_wave_base` is wave-uniform; lane
// l's 16 (4) bytes land at lds_wave_base + 16*l (4*l) bytes.  Issued through inline asm ON PURPOSE: with the
// builtin hipcc models the DMA as a pending LDS write and puts s_waitcnt vmcnt(0) in front of the next
// ds_read whenever it cannot prove the buffers disjoint (runtime double-buffer index), which serialises copy
// and compute.  The asm form is invisible to that bookkeeping, so the caller owns completion:
// svcmi_dma_wait() then a barrier before any wave reads the data (cdna_hip_programming.md section 5.7).
__device__ __forceinline__ unsigned svcmi_lds_addr(const float* lds_ptr) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds_ptr);
}
__device__ __forceinline__ void svcmi_glds16(const float* g, float* lds_wave_base) {
    const unsigned dst = svcmi_lds_addr(lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ __forceinline__ void svcmi_glds4(const float* g, float* lds_wave_base) {
    const unsigned dst = svcmi_lds_addr(lds_wave_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ __forceinline__ void svcmi_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Tell hipcc a threadIdx-derived value is wave-uniform (unlocks scalar loads / SGPR operands).
#define SVCMI_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

#define SVCMI_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__)
#define SVCMI_LAST_ERROR() ((int)hipGetLastError())
#endif

#define SVCMI_WAVE 64
