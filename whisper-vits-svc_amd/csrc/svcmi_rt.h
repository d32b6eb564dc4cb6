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

// Tell hipcc a threadIdx-derived value is wave-uniform (unlocks scalar loads / SGPR operands).
#define SVCMI_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

#define SVCMI_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__)
#define SVCMI_LAST_ERROR() ((int)hipGetLastError())
#endif

#define SVCMI_WAVE 64
