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
typedef float svcmi_f32x2 __attribute__((ext_vector_type(2)));      // packed fp32 (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32)
__device__ __forceinline__ svcmi_f32x2 svcmi_fma2(svcmi_f32x2 a, svcmi_f32x2 b, svcmi_f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// lo + hi of a pair as ONE scalar v_add_f32.  Written plainly (`p[0] + p[1]`) the compiler emits v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]
// (both halves = the sum): a packed instruction whose LOW lane takes the HIGH half of src1 -- the form that computes lanes 48..63 wrongly on
// MI355X while another wave of the SIMD executes v_mfma_f32_16x16x32_{f16,bf16} (round 6: scripts/probes/pkfma_mfma_corun.hip,
// profiles/r06y_pkfma_mfma_corun.log; DESIGN.md "the packed-fp32 operand-select erratum").  The empty asm takes the high half out of the
// pair for the optimiser; no instruction, no hazard.
__device__ __forceinline__ float svcmi_hsum2(svcmi_f32x2 p) {
    float hi = p[1];
#ifndef SVCMI_EMU
    asm("" : "+v"(hi));
#endif
    return p[0] + hi;
}
__device__ __forceinline__ svcmi_f32x2 svcmi_mul2(svcmi_f32x2 a, svcmi_f32x2 b) { return a * b; }
__device__ __forceinline__ svcmi_f32x2 svcmi_splat2(float v) { return svcmi_f32x2{v, v}; }
// (lo, lo) / (hi, hi) of a register pair: folds into the op_sel modifiers of the packed instruction that consumes it
__device__ __forceinline__ svcmi_f32x2 svcmi_splat_lo(svcmi_f32x2 p) { return __builtin_shufflevector(p, p, 0, 0); }
__device__ __forceinline__ svcmi_f32x2 svcmi_splat_hi(svcmi_f32x2 p) { return __builtin_shufflevector(p, p, 1, 1); }
// A constant the compiler must keep in a scalar register: packed instructions cannot encode literals, and given a literal the
// instruction selector prefers two scalar v_fmaak_f32 over one v_pk_fma_f32 -- an SGPR operand keeps the polynomial packed.
__device__ __forceinline__ float svcmi_sgpr_const(float v) {
    int i = __builtin_bit_cast(int, v);
    asm("" : "+s"(i));
    return __builtin_bit_cast(float, i);
}

// 16 bytes from READ-ONLY memory at a wave-uniform address as a scalar load (s_load_dwordx4 -> SGPR operands of the FMAs that
// follow).  A plain `*reinterpret_cast<const float4*>(p)` gets the scalar path only while the compiler can prove that nothing in the
// kernel clobbers it (MemorySSA walk with a visit limit): in a kernel with a phase loop and hundreds of LDS stores the proof
// times out and every weight fetch becomes global_load + s_waitcnt vmcnt(0) (measured: the fused AMP block 2x slower).  The
// constant address space states the invariant instead.  Only for memory no kernel on the device writes while this one runs (weights).
__device__ __forceinline__ svcmi_f32x4 svcmi_load_uniform4(const float* p) {
    return *(const __attribute__((address_space(4))) svcmi_f32x4*)(p);
}
// a wave-uniform pointer the optimiser knows nothing about any more (stays in an SGPR pair)
__device__ __forceinline__ const float* svcmi_opaque_uniform(const float* p) {
    asm volatile("" : "+s"(p));
    return p;
}
// One float at (wave-uniform base) + (per-lane 32-bit BYTE offset): `global_load_dword v, v_off, s[base]`.  The generic form spends a
// 64-bit multiply-add and a 64-bit add per load on address arithmetic; here it is one v_add_u32 per further row.  `base` must come
// from svcmi_opaque_uniform (an SGPR pair the optimiser cannot fold lane terms into); byte offsets < 2^32.
__device__ __forceinline__ float svcmi_load_saddr(const float* base, unsigned byte_off) {
    return *(const __attribute__((address_space(1))) float*)((const __attribute__((address_space(1))) char*)base + byte_off);
}
__device__ __forceinline__ float svcmi_load_uniform1(const float* p) { return *(const __attribute__((address_space(4))) float*)(p); }

#define SVCMI_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)      // nothing is scheduled across this point

// 2^x as ONE transcendental instruction (v_exp_f32, 1 ulp; inputs below -126 give 0).  libm's expf is ~15 instructions; a softmax
// that keeps its scores in the log2 domain (scores * log2 e folded into the scale) needs nothing more.
__device__ __forceinline__ float svcmi_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

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

// 16-bit-operand matrix instructions (bf16 / fp16 in, fp32 accumulate; 16x the fp32 MFMA rate).  Operands travel as four
// packed dwords = 8 values; lane l supplies A[i = l&31][k = 8*(l>>5) .. +7] and B[k = 8*(l>>5) .. +7][j = l&31]
// (32x32x16) resp. A[l&15][8*(l>>4) .. +7], B[8*(l>>4) .. +7][l&15] (16x16x32); C/D layouts as the fp32 forms above.
typedef unsigned svcmi_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned svcmi_u32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 svcmi_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 svcmi_bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 svcmi_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 svcmi_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ svcmi_u32x4 svcmi_as_u32x4(svcmi_f32x4 v) { return __builtin_bit_cast(svcmi_u32x4, v); }
__device__ __forceinline__ float svcmi_bits_f32(unsigned u) { return __builtin_bit_cast(float, u); }
__device__ __forceinline__ float svcmi_f16_bits_f32(unsigned h) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(h & 0xffffu)); }   // the fp16 value in the low 16 bits
// (lo, hi) halves = round-to-nearest-even of (a, b): v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32
__device__ __forceinline__ unsigned svcmi_cvt_pk_bf16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(svcmi_f32x2{a, b}, svcmi_bf16x2));
}
__device__ __forceinline__ unsigned svcmi_cvt_pk_f16(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(svcmi_f32x2{a, b}, svcmi_f16x2));
}
template <bool F16>
__device__ __forceinline__ svcmi_f32x16 svcmi_mfma16_32x32x16(svcmi_u32x4 a, svcmi_u32x4 b, svcmi_f32x16 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(svcmi_f16x8, a), __builtin_bit_cast(svcmi_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(svcmi_bf16x8, a), __builtin_bit_cast(svcmi_bf16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ svcmi_f32x4 svcmi_mfma16_16x16x32(svcmi_u32x4 a, svcmi_u32x4 b, svcmi_f32x4 c) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(svcmi_f16x8, a), __builtin_bit_cast(svcmi_f16x8, b), c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(svcmi_bf16x8, a), __builtin_bit_cast(svcmi_bf16x8, b), c, 0, 0, 0);
}

// (lo16(a) | lo16(b) << 16) and (hi16(a) | hi16(b) << 16) as ONE v_perm_b32 each: the 16-bit transposes of the attention16 V staging
__device__ __forceinline__ unsigned svcmi_pack_lo16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x05040100u); }
__device__ __forceinline__ unsigned svcmi_pack_hi16(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// Asynchronous global -> LDS copy (LDS-DMA) through a buffer descriptor: `buffer_load_dword[x4] voff, rsrc, 0 offen lds`.
// Lane l fetches 16 (4) bytes at rsrc.base + voff[l]; they land at lds_wave_base + 16*l (4*l) bytes, where
// `lds_wave_base` is wave-uniform (it travels in M0).  The hardware range-checks voff against rsrc.num_records
// and an out-of-range lane writes ZEROS to its LDS slot (probe: scripts/probes/oob_lds_dma.hip) -- that is how
// padding taps, masked rows and ragged tile edges are zero-filled without a branch or a select on data.
// Issued through inline asm ON PURPOSE: with the builtin hipcc models the DMA as a pending LDS write and puts
// s_waitcnt vmcnt(0) in front of the next ds_read whenever it cannot prove the buffers disjoint (runtime
// double-buffer index), which serialises copy and compute.  The asm form is invisible to that bookkeeping, so
// the caller owns completion: svcmi_dma_wait() then a barrier before any wave reads the data
// (cdna_hip_programming.md section 5.7).
typedef int svcmi_rsrc __attribute__((ext_vector_type(4)));
__device__ __forceinline__ svcmi_rsrc svcmi_make_rsrc(const void* base, unsigned bytes) {   // wave-uniform inputs only
    const size_t a = (size_t)base;
    svcmi_rsrc r;
    r.x = (int)(unsigned)a;
    r.y = (int)(unsigned)(a >> 32);      // stride 0: raw buffer
    r.z = (int)bytes;                    // num_records, in bytes
    r.w = 0x00020000;
    return r;
}
// Timing probes of the GEMM K loop (scripts/build_variant.sh ... -DSVCMI_PROBE_NODMA=1 / _NOMFMA / _NOLDSREAD; results are garbage, the
// time is the point): which of {global -> LDS fill, LDS fragment reads, matrix instructions} bounds a K-step.  Never set in the product build.
// Experiment (round 6): LDS-DMA issue WITHOUT saving / restoring M0 around it (2 scalar moves less per piece).  hipcc reserves M0 and rejects an
// "m0" clobber, so this is only sound in a translation unit whose ISA touches M0 nowhere else (checked on the assembly, not by the compiler).
#ifndef SVCMI_DMA_M0_RAW
#define SVCMI_DMA_M0_RAW 0
#endif
#ifndef SVCMI_PROBE_NODMA
#define SVCMI_PROBE_NODMA 0
#endif
#ifndef SVCMI_PROBE_NOLDSREAD
#define SVCMI_PROBE_NOLDSREAD 0
#endif
#ifndef SVCMI_PROBE_NOMFMA
#define SVCMI_PROBE_NOMFMA 0
#endif
typedef unsigned svcmi_ldsaddr;          // wave-uniform LDS byte address (what M0 takes)
__device__ __forceinline__ svcmi_ldsaddr svcmi_lds_addr(const float* lds_ptr) {
    return __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds_ptr);
}
__device__ __forceinline__ svcmi_ldsaddr svcmi_lds_advance(svcmi_ldsaddr a, int floats) { return a + 4u * (unsigned)floats; }
__device__ __forceinline__ void svcmi_bdma16(unsigned voff, svcmi_ldsaddr lds_wave_base, svcmi_rsrc rsrc) {
    if (SVCMI_PROBE_NODMA) return;
#if SVCMI_DMA_M0_RAW
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds" : : "v"(voff), "s"(lds_wave_base), "s"(rsrc) : "memory");
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_wave_base), "s"(rsrc) : "memory");
#endif
}
// The same DMA pinned inside an MFMA stream by register ties (svcmi_lds_read16: the A fragments the surrounding MFMAs consume)
__device__ __forceinline__ void svcmi_bdma16_at(unsigned voff, svcmi_ldsaddr lds_wave_base, svcmi_rsrc rsrc, svcmi_f32x4& tie) {
    if (SVCMI_PROBE_NODMA) return;
#if SVCMI_DMA_M0_RAW
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds" : "+v"(tie) : "v"(voff), "s"(lds_wave_base), "s"(rsrc) : "memory");
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "+v"(tie) : "v"(voff), "s"(lds_wave_base), "s"(rsrc) : "memory");
#endif
}
__device__ __forceinline__ void svcmi_bdma16_at(unsigned voff, svcmi_ldsaddr lds_wave_base, svcmi_rsrc rsrc, svcmi_f32x4& tie, svcmi_f32x4& tie2) {
    if (SVCMI_PROBE_NODMA) return;
#if SVCMI_DMA_M0_RAW
    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %4, 0 offen lds" : "+v"(tie), "+v"(tie2) : "v"(voff), "s"(lds_wave_base), "s"(rsrc) : "memory");
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\tbuffer_load_dwordx4 %3, %5, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep), "+v"(tie), "+v"(tie2) : "v"(voff), "s"(lds_wave_base), "s"(rsrc) : "memory");
#endif
}
__device__ __forceinline__ void svcmi_bdma16_at(unsigned voff, svcmi_ldsaddr lds_wave_base, svcmi_rsrc rsrc) { svcmi_bdma16(voff, lds_wave_base, rsrc); }
__device__ __forceinline__ void svcmi_bdma4(unsigned voff, svcmi_ldsaddr lds_wave_base, svcmi_rsrc rsrc) {
#if SVCMI_DMA_M0_RAW
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dword %0, %2, 0 offen lds" : : "v"(voff), "s"(lds_wave_base), "s"(rsrc) : "memory");
#else
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dword %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_wave_base), "s"(rsrc) : "memory");
#endif
}
__device__ __forceinline__ void svcmi_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// Wait until at most N of this wave's VMEM operations (LDS-DMAs included, counted in issue order) are outstanding.
template <int N>
__device__ __forceinline__ void svcmi_dma_wait_n() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit counter");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Register-prefetched LDS fragment reads.  hipcc sinks a plain `ds_read` to just above its first use, i.e. BELOW the
// MFMAs of the running sub-step, so each sub-step opens with an exposed LDS round trip.  These two statements pin
// the software pipeline instead (cdna_hip_programming.md section 5.7, form ii):
//   svcmi_lds_read16(dst, p, tie)  issues ds_read_b128 dst <- p; `tie` (a register of the fragment the MFMAs
//                                  about to be issued consume) is marked read-write so those MFMAs stay below it;
//   svcmi_lds_arrive(dst)          s_waitcnt lgkmcnt(0) naming dst read-write, so no consumer floats above it.
// hipcc does not count asm LDS operations; extra ones in flight only make its own lgkmcnt waits stricter.
__device__ __forceinline__ void svcmi_lds_read16(svcmi_f32x4& dst, const float* p, svcmi_f32x4& tie) {
    if (SVCMI_PROBE_NOLDSREAD) { asm volatile("" : "=v"(dst), "+v"(tie)); return; }
    const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
    asm volatile("ds_read_b128 %0, %2" : "=v"(dst), "+v"(tie) : "v"(a) : "memory");
}
// Two ties (the 128-row tiles issue MFMAs on two A fragments): the statement sits below every earlier consumer of either register and
// above every later one -- with both A fragments of a sub-step named, an exact position in its MFMA stream (conv_gemm_body.h, "SPREAD").
__device__ __forceinline__ void svcmi_lds_read16(svcmi_f32x4& dst, const float* p, svcmi_f32x4& tie, svcmi_f32x4& tie2) {
    if (SVCMI_PROBE_NOLDSREAD) { asm volatile("" : "=v"(dst), "+v"(tie), "+v"(tie2)); return; }
    const unsigned a = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)p;
    asm volatile("ds_read_b128 %0, %3" : "=v"(dst), "+v"(tie), "+v"(tie2) : "v"(a) : "memory");
}
__device__ __forceinline__ void svcmi_lds_arrive(svcmi_f32x4& d) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d)::"memory"); }
// ... with the register(s) the MFMAs of the running sub-step consume named too: they stay ABOVE the wait (they are earlier readers of a
// register the statement rewrites), which is what the accumulator pins (svcmi_pin) do for the plain placement
__device__ __forceinline__ void svcmi_lds_arrive(svcmi_f32x4& d, svcmi_f32x4& tie) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d), "+v"(tie)::"memory"); }
__device__ __forceinline__ void svcmi_lds_arrive(svcmi_f32x4& d, svcmi_f32x4& tie, svcmi_f32x4& tie2) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d), "+v"(tie), "+v"(tie2)::"memory");
}
// The same ordering pin without the instruction: after ONE svcmi_lds_arrive every outstanding LDS read of the wave has landed, the
// other fragment registers only need to be named so that no consumer floats above that wait.
__device__ __forceinline__ void svcmi_lds_landed(svcmi_f32x4& d) { asm volatile("" : "+v"(d)::"memory"); }

// Order fence for a register-only value: nothing that produces `v` is scheduled below, nothing that consumes it above.
// (a translation unit built with -mllvm -amdgpu-mfma-vgpr-form=1 keeps its accumulators in architectural registers and says so with
// -DSVCMI_ACC_IN_VGPRS=1: an "a" constraint there would cost a copy into the accumulator file and back at every pin -- build.py FILE_FLAGS)
#ifdef SVCMI_ACC_IN_VGPRS
__device__ __forceinline__ void svcmi_pin(svcmi_f32x16& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void svcmi_pin(svcmi_f32x4& v) { asm volatile("" : "+v"(v)); }
#else
__device__ __forceinline__ void svcmi_pin(svcmi_f32x16& v) { asm volatile("" : "+a"(v)); }   // "a": stays in the accumulator file
__device__ __forceinline__ void svcmi_pin(svcmi_f32x4& v) { asm volatile("" : "+a"(v)); }
#endif

// Cross-workgroup hand-off inside one launch (cdna_hip_programming.md Guideline 16): agent-scope release by the
// producer, a relaxed agent-scope ticket, agent-scope acquire by the consumer.  Workgroup scope is NOT enough.
#define SVCMI_RELEASE_AGENT() __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent")
#define SVCMI_ACQUIRE_AGENT() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
__device__ __forceinline__ int svcmi_ticket(int* counter) {
    return __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 16-byte write-through (sc1) store through a buffer descriptor: the data is visible to other XCDs once the issuing
// wave's vmcnt drains -- no release fence (L2 write-back) needed for a hand-off (Guideline 16, R1).
__device__ __forceinline__ void svcmi_store16_sc1(svcmi_f32x4 v, svcmi_rsrc r, unsigned byte_off) {
    asm volatile("buffer_store_dwordx4 %0, %1, %2, 0 offen sc1\n\ts_nop 1" ::"v"(v), "v"(byte_off), "s"(r) : "memory");
}

// Plain 16-byte loads / stores through a buffer descriptor whose per-lane 32-bit byte offset is RANGE-CHECKED by the hardware: an offset
// >= num_records (0x80000000 for "this lane does nothing") loads zeros / drops the store.  No branch around the memory operation, so hipcc
// can count vmcnt exactly across a loop's back edge (with a conditional load in the loop it opens every iteration with s_waitcnt vmcnt(0),
// which on gfx9 -- loads and stores share the counter -- also waits for the previous iteration's STORE: conv_gemm_body.h, epilogue).
typedef __amdgpu_buffer_rsrc_t svcmi_brsrc;
__device__ __forceinline__ svcmi_brsrc svcmi_make_brsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ svcmi_f32x4 svcmi_buf_load16(svcmi_brsrc r, unsigned byte_off) {
    return __builtin_bit_cast(svcmi_f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0));
}
__device__ __forceinline__ void svcmi_buf_store16(svcmi_f32x4 v, svcmi_brsrc r, unsigned byte_off) {
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(svcmi_u32x4, v), r, (int)byte_off, 0, 0);
}

// Scheduling hint: the next `n` instructions of class `mask` (0x008 MFMA, 0x100 DS read, 0x020 VMEM read, 0x002 VALU)
// form a group, groups are emitted in source order (cdna_hip_programming.md T19).
#define SVCMI_SCHED_GROUP(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)

// Tell hipcc a threadIdx-derived value is wave-uniform (unlocks scalar loads / SGPR operands).
#define SVCMI_UNIFORM(x) __builtin_amdgcn_readfirstlane(x)

#define SVCMI_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    hipLaunchKernelGGL(kernel, grid, block, shmem, (hipStream_t)(stream), __VA_ARGS__)
#define SVCMI_LAST_ERROR() ((int)hipGetLastError())
#endif

#define SVCMI_WAVE 64

// GELU(v) = v * Phi(v), exact-erf form (torch.nn.functional.gelu default; whisper/model.py:122, hubert, crepe), branch-free (round 6).
// libm's erff is two polynomial branches plus an expf expansion, ~45 instructions per element when a wave takes both -- 4 us of the 63 us
// Whisper MLP-up launch were its epilogue's GELU.  Here: t = |v| / sqrt 2, erfc(t) = 2^(t * q(t)) with q a degree-7 polynomial fitted to
// log2(erfc(t)) / t on [0, 4.2] under the weight erfc(t) (max |erf error| 1.6e-8, scripts/fit_gelu.py), ONE v_exp_f32, and
//   v > 0:  v - (0.5 v) erfc(t)        v <= 0:  (0.5 v) erfc(t)
// which needs no 1 - erfc cancellation: max |error| 2.7e-7 over [-12, 12] against the fp64 function (torch's own fp32 GELU: 1.2e-6).
#ifndef SVCMI_EMU
__device__ __forceinline__
#else
static inline
#endif
float svcmi_gelu(float v) {
    const float a = fabsf(v) * 0.70710678118654752440f;
    const float t = a < 4.2f ? a : 4.2f;                 // erfc(4.2) = 3e-9: below half an ulp of 1
    float q = -4.535877815214917e-05f;
    q = fmaf(q, t, 0.0004455065354704857f);
    q = fmaf(q, t, -0.0014894307823851705f);
    q = fmaf(q, t, -0.0007746575865894556f);
    q = fmaf(q, t, 0.028253713622689247f);
    q = fmaf(q, t, -0.14848163723945618f);
    q = fmaf(q, t, -0.9184163808822632f);
    q = fmaf(q, t, -1.6279085874557495f);
    const float h = 0.5f * v * svcmi_exp2(q * t);
    return v > 0.f ? v - h : h;
}

// Four consecutive values as a 16-bit copy (8-byte store): fmt 0 bf16, 1 f16, 2 split bf16 -- hi = rne(x) at dst, lo = rne(x - hi) at
// dst + lo_off: the activation-row format of the _A16 GEMM kernels (conv_gemm_body.h).
#ifndef SVCMI_EMU
__device__ __forceinline__
#else
static inline
#endif
void svcmi_store4_16(unsigned short* dst, int lo_off, float a, float b, float c, float d, int fmt) {
    unsigned* h = reinterpret_cast<unsigned*>(dst);
    if (fmt == 1) {
        h[0] = svcmi_cvt_pk_f16(a, b);
        h[1] = svcmi_cvt_pk_f16(c, d);
        return;
    }
    const unsigned h0 = svcmi_cvt_pk_bf16(a, b), h1 = svcmi_cvt_pk_bf16(c, d);
    h[0] = h0;
    h[1] = h1;
    if (fmt == 2) {
        unsigned* l = reinterpret_cast<unsigned*>(dst + lo_off);
        l[0] = svcmi_cvt_pk_bf16(a - svcmi_bits_f32(h0 << 16), b - svcmi_bits_f32(h0 & 0xffff0000u));
        l[1] = svcmi_cvt_pk_bf16(c - svcmi_bits_f32(h1 << 16), d - svcmi_bits_f32(h1 & 0xffff0000u));
    }
}
// Host side: the kernels' format code of a 16-bit activation output from an `enum svcmi_precision` value (BF16 = 2 -> 0, F16 = 3 -> 1,
// BF16X3 = 1 -> 2 = split rows); -1 = not an activation format.  Row check: `ld` 16-bit values hold n outputs (split: two planes of ld/2).
static inline int svcmi_fmt16(int format) { return format == 2 ? 0 : format == 3 ? 1 : format == 1 ? 2 : -1; }
static inline bool svcmi_fmt16_row_ok(int format, int ld, int n) { return format == 1 ? (ld % 8 == 0 && ld / 2 >= n) : (ld % 4 == 0 && ld >= n); }

// Lane movement for wave scans on the DPP row operations of gfx9 (no LDS crossbar: ~8 cycles where a ds_bpermute shuffle takes ~100).
// Every function returns the moved value where a source lane exists and the lane's OWN value elsewhere, so `x = op(x, moved(x))` is a
// no-op there.  row_shr<N>: lane l - N inside the lane's row of 16;  row_bcast15: rows 1 and 3 receive lane 15 of the row before;
// row_bcast31: rows 2 and 3 receive lane 31;  wave_shr1: lane l - 1 (lane 0 keeps its own).  The classic inclusive scan is
// row_shr 1, 2, 4, 8, then row_bcast15, then row_bcast31.
#ifdef SVCMI_EMU
template <int N> static inline int svcmi_dpp_row_shr(int x) { const int l = threadIdx.x & 63; return __shfl(x, (l & 15) >= N ? l - N : l); }
static inline int svcmi_dpp_row_bcast15(int x) { const int l = threadIdx.x & 63, r = l >> 4; return __shfl(x, (r == 1 || r == 3) ? 16 * r - 1 : l); }
static inline int svcmi_dpp_row_bcast31(int x) { const int l = threadIdx.x & 63; return __shfl(x, l >= 32 ? 31 : l); }
static inline int svcmi_dpp_wave_shr1(int x) { const int l = threadIdx.x & 63; return __shfl(x, l > 0 ? l - 1 : l); }
#define SVCMI_DEV static inline
#else
template <int N> __device__ __forceinline__ int svcmi_dpp_row_shr(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x110 + N, 0xf, 0xf, false); }
__device__ __forceinline__ int svcmi_dpp_row_bcast15(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x142, 0xa, 0xf, false); }
__device__ __forceinline__ int svcmi_dpp_row_bcast31(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x143, 0xc, 0xf, false); }
__device__ __forceinline__ int svcmi_dpp_wave_shr1(int x) { return __builtin_amdgcn_update_dpp(x, x, 0x138, 0xf, 0xf, false); }
#define SVCMI_DEV __device__ __forceinline__
#endif
// the same moves on a double (two 32-bit halves)
template <class F>
SVCMI_DEV double svcmi_dpp_f64(double v, F move) {
    unsigned long long u;
    memcpy(&u, &v, 8);
    const unsigned lo = (unsigned)move((int)(unsigned)u), hi = (unsigned)move((int)(unsigned)(u >> 32));
    u = ((unsigned long long)hi << 32) | lo;
    memcpy(&v, &u, 8);
    return v;
}

