// hip_emu.h -- TEST-ONLY fiber-based SIMT emulator for the subset of HIP that svcmi kernels use.
//
// Purpose: execute the product's kernel source (whisper-vits-svc_amd/csrc/*.hip, compiled with
// g++ -DSVCMI_EMU) on the CPU so that `pytest -m "not gpu"` can check tiling / indexing / masking
// logic in a container without a GPU.  Each workgroup runs as blockDim cooperative fibers
// (ucontext) on one OS thread; __syncthreads and wave-level collectives (shuffles, the fp32 MFMA)
// are rendezvous points.  Blocks run sequentially by default, so `__shared__` is plain static storage (with red zones in the
// sanitizer build); the -DSVCMI_EMU_TLS build runs K blocks at once on K lock-stepped OS threads, each with its own copy.
// Nothing here is a fallback for the product: svcmi (Python) only ever loads the hipcc-built
// library and refuses to run without it.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#ifdef SVCMI_EMU_TLS
#define __shared__ static thread_local      // one copy per resident block (= per executing OS thread): the SVCMI_EMU_BLOCKS=K mode of hip_emu.cpp
#else
#define __shared__ static                   // exactly the declared bytes; the sanitizer build puts red zones around it
#endif

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(16) int4 { int x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef float svcmi_f32x16 __attribute__((vector_size(64)));
typedef float svcmi_f32x4 __attribute__((vector_size(16)));
typedef float svcmi_f32x2 __attribute__((vector_size(8)));
static inline svcmi_f32x2 svcmi_fma2(svcmi_f32x2 a, svcmi_f32x2 b, svcmi_f32x2 c) { return svcmi_f32x2{fmaf(a[0], b[0], c[0]), fmaf(a[1], b[1], c[1])}; }
static inline svcmi_f32x2 svcmi_splat2(float v) { return svcmi_f32x2{v, v}; }
static inline svcmi_f32x2 svcmi_mul2(svcmi_f32x2 a, svcmi_f32x2 b) { return svcmi_f32x2{a[0] * b[0], a[1] * b[1]}; }
static inline float svcmi_hsum2(svcmi_f32x2 p) { return p[0] + p[1]; }
static inline svcmi_f32x2 svcmi_splat_lo(svcmi_f32x2 p) { return svcmi_f32x2{p[0], p[0]}; }
static inline svcmi_f32x2 svcmi_splat_hi(svcmi_f32x2 p) { return svcmi_f32x2{p[1], p[1]}; }
static inline float svcmi_sgpr_const(float v) { return v; }
static inline float svcmi_exp2(float x) { return exp2f(x); }
static inline svcmi_f32x4 svcmi_load_uniform4(const float* p) { svcmi_f32x4 v; memcpy(&v, p, 16); return v; }
static inline float svcmi_load_uniform1(const float* p) { return *p; }
static inline const float* svcmi_opaque_uniform(const float* p) { return p; }
static inline float svcmi_load_saddr(const float* base, unsigned byte_off) { return *(const float*)((const char*)base + byte_off); }

typedef void* hipStream_t;

namespace emu {
struct Fiber;
extern thread_local dim3 g_blockIdx;      // (one executing OS thread per resident block in the SVCMI_EMU_BLOCKS mode)
extern dim3 g_blockDim, g_gridDim;
const dim3& cur_tid();
int cur_lane();
void syncthreads();
void wave_exchange(const void* mine, size_t bytes, void* all64);  // all-gather within the wave
void launch(dim3 grid, dim3 block, const std::function<void()>& body);
extern int g_last_error;
}  // namespace emu

#define threadIdx (emu::cur_tid())
#define blockIdx (emu::g_blockIdx)
#define blockDim (emu::g_blockDim)
#define gridDim (emu::g_gridDim)

static inline void __syncthreads() { emu::syncthreads(); }

template <class T>
static inline T __shfl(T v, int src, int width = 64) {
    T all[64];
    emu::wave_exchange(&v, sizeof(T), all);
    int lane = emu::cur_lane();
    int base = lane & ~(width - 1);
    return all[base + (src & (width - 1))];
}
template <class T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    T all[64];
    emu::wave_exchange(&v, sizeof(T), all);
    int lane = emu::cur_lane();
    int t = lane ^ mask;
    if ((t & ~(width - 1)) != (lane & ~(width - 1))) t = lane;
    return all[t];
}
template <class T>
static inline T __shfl_down(T v, int delta, int width = 64) {
    T all[64];
    emu::wave_exchange(&v, sizeof(T), all);
    int lane = emu::cur_lane();
    int t = lane + delta;
    if ((t & ~(width - 1)) != (lane & ~(width - 1))) t = lane;
    return all[t];
}
template <class T>
static inline T __shfl_up(T v, int delta, int width = 64) {
    T all[64];
    emu::wave_exchange(&v, sizeof(T), all);
    int lane = emu::cur_lane();
    int t = lane - delta;
    if (t < 0 || (t & ~(width - 1)) != (lane & ~(width - 1))) t = lane;
    return all[t];
}

// v_mfma_f32_32x32x2_f32 with the gfx950 operand layout (cdna_hip_programming.md section 3):
// lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]; D: col=l&31, row=(r&3)+8*(r>>2)+4*(l>>5);
// numerics = k-ordered fmaf chain.
static inline svcmi_f32x16 svcmi_mfma_32x32x2(float a, float b, svcmi_f32x16 c) {
    float ab[2] = {a, b};
    float all[64][2];
    emu::wave_exchange(ab, sizeof(ab), all);
    int l = emu::cur_lane();
    int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        acc = fmaf(all[i][0], all[j][1], acc);
        acc = fmaf(all[i + 32][0], all[j + 32][1], acc);
        c[r] = acc;
    }
    return c;
}

// v_mfma_f32_16x16x4_f32: lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D: col=l&15, row=4*(l>>4)+r.
static inline svcmi_f32x4 svcmi_mfma_16x16x4(float a, float b, svcmi_f32x4 c) {
    float ab[2] = {a, b};
    float all[64][2];
    emu::wave_exchange(ab, sizeof(ab), all);
    int l = emu::cur_lane();
    int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(all[i + 16 * k][0], all[j + 16 * k][1], acc);
        c[r] = acc;
    }
    return c;
}

// 16-bit-operand MFMAs (bf16 / fp16 in, fp32 accumulate): operands as 4 packed dwords = 8 values per lane.
typedef unsigned svcmi_u32x4 __attribute__((vector_size(16)));
typedef unsigned svcmi_u32x2 __attribute__((vector_size(8)));
static inline svcmi_u32x4 svcmi_as_u32x4(svcmi_f32x4 v) { svcmi_u32x4 r; memcpy(&r, &v, 16); return r; }
static inline float svcmi_bits_f32(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned emu_bf16_rne(float f) {
    unsigned u; memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;      // NaN stays NaN
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline float emu_bf16_f32(unsigned h) { return svcmi_bits_f32((h & 0xffffu) << 16); }
static inline unsigned emu_f16_rne(float f) {          // IEEE binary16, round to nearest even, overflow -> inf
    unsigned u; memcpy(&u, &f, 4);
    const unsigned sign = (u >> 16) & 0x8000u;
    const unsigned a = u & 0x7fffffffu;
    if (a > 0x7f800000u) return sign | 0x7e00u;                         // NaN
    if (a >= 0x477ff000u) return sign | 0x7c00u;                        // >= 65520 rounds to inf
    if (a < 0x33000001u) return sign;                                   // <= 2^-25 rounds to zero
    int e = (int)(a >> 23) - 127;
    unsigned m = (a & 0x7fffffu) | 0x800000u;                           // 24-bit significand
    int shift = e < -14 ? 13 + (-14 - e) : 13;                          // subnormal results lose more bits
    unsigned r = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (r & 1))) ++r;
    if (e < -14) return sign | r;                                       // subnormal (a carry into 0x400 is the smallest normal)
    return sign | (((unsigned)(e + 15) << 10) + (r - 0x400u));         // a mantissa carry bumps the exponent
}
static inline float emu_f16_f32(unsigned h) {
    const unsigned sign = (h & 0x8000u) << 16, e = (h >> 10) & 31u, m = h & 0x3ffu;
    if (e == 31) return svcmi_bits_f32(sign | 0x7f800000u | (m << 13));
    if (e == 0) return (sign ? -1.f : 1.f) * ldexpf((float)m, -24);
    return svcmi_bits_f32(sign | ((e + 112) << 23) | (m << 13));
}
static inline float svcmi_f16_bits_f32(unsigned h) { return emu_f16_f32(h & 0xffffu); }
static inline unsigned svcmi_cvt_pk_bf16(float a, float b) { return emu_bf16_rne(a) | (emu_bf16_rne(b) << 16); }
static inline unsigned svcmi_cvt_pk_f16(float a, float b) { return emu_f16_rne(a) | (emu_f16_rne(b) << 16); }
template <bool F16>
static inline float emu_h16(const svcmi_u32x4& v, int e) {
    const unsigned h = (v[e >> 1] >> (16 * (e & 1))) & 0xffffu;
    return F16 ? emu_f16_f32(h) : emu_bf16_f32(h);
}
// v_mfma_f32_32x32x16_{bf16,f16}: lane l holds A[i=l&31][k=8*(l>>5)..+7], B[k=8*(l>>5)..+7][j=l&31]; D as 32x32x2.
template <bool F16>
static inline svcmi_f32x16 svcmi_mfma16_32x32x16(svcmi_u32x4 a, svcmi_u32x4 b, svcmi_f32x16 c) {
    svcmi_u32x4 ab[2] = {a, b};
    svcmi_u32x4 all[64][2];
    emu::wave_exchange(ab, sizeof(ab), all);
    int l = emu::cur_lane();
    int j = l & 31;
    for (int r = 0; r < 16; ++r) {
        int i = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int h = 0; h < 2; ++h)
            for (int e = 0; e < 8; ++e) acc += emu_h16<F16>(all[i + 32 * h][0], e) * emu_h16<F16>(all[j + 32 * h][1], e);
        c[r] = acc;
    }
    return c;
}
// v_mfma_f32_16x16x32_{bf16,f16}: lane l holds A[l&15][8*(l>>4)..+7], B[8*(l>>4)..+7][l&15]; D as 16x16x4.
template <bool F16>
static inline svcmi_f32x4 svcmi_mfma16_16x16x32(svcmi_u32x4 a, svcmi_u32x4 b, svcmi_f32x4 c) {
    svcmi_u32x4 ab[2] = {a, b};
    svcmi_u32x4 all[64][2];
    emu::wave_exchange(ab, sizeof(ab), all);
    int l = emu::cur_lane();
    int j = l & 15;
    for (int r = 0; r < 4; ++r) {
        int i = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int h = 0; h < 4; ++h)
            for (int e = 0; e < 8; ++e) acc += emu_h16<F16>(all[i + 16 * h][0], e) * emu_h16<F16>(all[j + 16 * h][1], e);
        c[r] = acc;
    }
    return c;
}

static inline unsigned svcmi_pack_lo16(unsigned a, unsigned b) { return (a & 0xffffu) | (b << 16); }
static inline unsigned svcmi_pack_hi16(unsigned a, unsigned b) { return (a >> 16) | (b & 0xffff0000u); }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

#define SVCMI_LAUNCH(kernel, grid, block, shmem, stream, ...) \
    emu::launch(grid, block, [=]() { kernel(__VA_ARGS__); })
#define SVCMI_LAST_ERROR() (emu::g_last_error)
#define SVCMI_UNIFORM(x) (x)
#define SVCMI_SCHED_GROUP(mask, n) ((void)0)
#define SVCMI_SCHED_BARRIER() ((void)0)
#define SVCMI_RELEASE_AGENT() ((void)0)
#define SVCMI_ACQUIRE_AGENT() ((void)0)
static inline int svcmi_ticket(int* counter) { return (*counter)++; }   // blocks run one after another in the emulator
static inline void svcmi_lds_read16(svcmi_f32x4& dst, const float* p, svcmi_f32x4&) { memcpy(&dst, p, 16); }
static inline void svcmi_lds_read16(svcmi_f32x4& dst, const float* p, svcmi_f32x4&, svcmi_f32x4&) { memcpy(&dst, p, 16); }
static inline void svcmi_lds_arrive(svcmi_f32x4&) {}
static inline void svcmi_lds_arrive(svcmi_f32x4&, svcmi_f32x4&) {}
static inline void svcmi_lds_arrive(svcmi_f32x4&, svcmi_f32x4&, svcmi_f32x4&) {}
static inline void svcmi_lds_landed(svcmi_f32x4&) {}
static inline void svcmi_pin(svcmi_f32x16&) {}
static inline void svcmi_pin(svcmi_f32x4&) {}
// LDS-DMA emulation (buffer form): lane l copies its 16 (4) bytes from rsrc.base + voff to lds_wave_base + 16*l
// (4*l); an out-of-range lane writes zeros, like the hardware.  Synchronous here.  `lds_wave_base` is the LDS
// "address" svcmi_lds_addr() returned -- in the emulator simply the pointer.
struct svcmi_rsrc { const char* base; unsigned bytes; };
static inline svcmi_rsrc svcmi_make_rsrc(const void* base, unsigned bytes) { return svcmi_rsrc{(const char*)base, bytes}; }
typedef float* svcmi_ldsaddr;
static inline svcmi_ldsaddr svcmi_lds_addr(float* lds_ptr) { return lds_ptr; }
static inline svcmi_ldsaddr svcmi_lds_advance(svcmi_ldsaddr a, int floats) { return a + floats; }
static inline void svcmi_bdma16(unsigned voff, float* lds_wave_base, svcmi_rsrc r) {
    float* dst = lds_wave_base + 4 * emu::cur_lane();
    if ((unsigned long long)voff + 16 <= r.bytes) memcpy(dst, r.base + voff, 16); else memset(dst, 0, 16);
}
static inline void svcmi_bdma16_at(unsigned voff, float* lds_wave_base, svcmi_rsrc r) { svcmi_bdma16(voff, lds_wave_base, r); }
static inline void svcmi_bdma16_at(unsigned voff, float* lds_wave_base, svcmi_rsrc r, svcmi_f32x4&) { svcmi_bdma16(voff, lds_wave_base, r); }
static inline void svcmi_bdma16_at(unsigned voff, float* lds_wave_base, svcmi_rsrc r, svcmi_f32x4&, svcmi_f32x4&) { svcmi_bdma16(voff, lds_wave_base, r); }
static inline void svcmi_bdma4(unsigned voff, float* lds_wave_base, svcmi_rsrc r) {
    float* dst = lds_wave_base + emu::cur_lane();
    if ((unsigned long long)voff + 4 <= r.bytes) memcpy(dst, r.base + voff, 4); else memset(dst, 0, 4);
}
static inline void svcmi_dma_wait() {}
template <int N>
static inline void svcmi_dma_wait_n() {}
// range-checked plain buffer accesses (svcmi_rt.h): out-of-range lanes load zeros / store nothing
typedef svcmi_rsrc svcmi_brsrc;
static inline svcmi_brsrc svcmi_make_brsrc(const void* base, unsigned bytes) { return svcmi_rsrc{(const char*)base, bytes}; }
static inline svcmi_f32x4 svcmi_buf_load16(svcmi_brsrc r, unsigned byte_off) {
    svcmi_f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if ((unsigned long long)byte_off + 16 <= r.bytes) memcpy(&v, r.base + byte_off, 16);
    return v;
}
static inline void svcmi_buf_store16(svcmi_f32x4 v, svcmi_brsrc r, unsigned byte_off) {
    if ((unsigned long long)byte_off + 16 <= r.bytes) memcpy(const_cast<char*>(r.base) + byte_off, &v, 16);
}
static inline void svcmi_store16_sc1(svcmi_f32x4 v, svcmi_rsrc r, unsigned byte_off) {
    if ((unsigned long long)byte_off + 16 <= r.bytes) memcpy(const_cast<char*>(r.base) + byte_off, &v, 16);
}
