// snake_math.h -- the arithmetic of the anti-aliased SnakeBeta activation (vits_decoder/alias/act.py:124-129, resample.py,
// filter.py), shared by snake_alias_kernel (generator.hip) and the fused kernels (amp_fused.hip).  Include
// inside the including file's anonymous namespace.
//
// One work item = one channel x RT consecutive outputs, from a register window of RT + 10 inputs:
//   up-sampler (polyphase, 6 taps per phase) -> SnakeBeta on the 2x grid -> 12-tap decimating low-pass.
// Everything runs on float PAIRS -- (odd phase, even phase) of an up-sampled position, i.e. (s[2m], s[2m+1]) -- as packed
// fp32 instructions (v_pk_fma_f32 / v_pk_mul_f32: two lanes' worth of FMA per issue slot, the rate the 157 TFLOP/s vector
// peak is quoted at): the filter taps pair up the same way for the up-sampler, (f[2j], f[2j+1]) x x[.], and for the
// low-pass, sum_i (f[2i], f[2i+1]) . (s[2r+2i], s[2r+2i+1]).  PMC had the scalar version VALU-instruction-bound
// (SQ_INSTS_VALU x 4 cycles = 76 % of the fused kernel's SIMD time); the packed form issues ~45 % fewer VALU instructions.
#pragma once

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cold path of sin_sq as a real call: inlined, the unrolled copies of libm's large-argument reduction made the kernel
// 5600 instructions (45 KB) for ~1200 hot ones
static __device__ __attribute__((noinline)) float sin_sq_huge(float x) {
    const float sl = sinf(x);
    return sl * sl;
}

// sin^2(x) without the libm call: sin^2 has period pi, so x is reduced to r = x - n*pi in [-pi/2, pi/2] with a
// two-constant Cody-Waite step (the FMAs keep n*PI_HI exact), sin(r) is a degree-11 odd minimax polynomial and
// the result is squared.  Max abs error 2.6e-7 for |x| <= 1e5 (fp32 libm sin, squared: 1.3e-7).
__device__ __forceinline__ float sin_sq(float x) {
    if (__builtin_expect(fabsf(x) > 1.0e5f, 0)) return sin_sq_huge(x);   // never taken for audio-scale activations
    const float n = rintf(x * 0.31830987f);
    float r = fmaf(-n, 3.1415927f, x);
    r = fmaf(-n, -8.742278e-08f, r);
    const float u = r * r;
    float p = -2.3840804885821854e-08f;
    p = fmaf(p, u, 2.7522235086507862e-06f);
    p = fmaf(p, u, -0.00019840795721393079f);
    p = fmaf(p, u, 0.008333330042660236f);
    p = fmaf(p, u, -0.1666666716337204f);
    const float sn = fmaf(r * u, p, r);
    return sn * sn;
}
__device__ __forceinline__ float snake_fn(float y, float a, float inv_b) { return fmaf(inv_b, sin_sq(y * a), y); }

// the same on a pair; identical operation sequence per component, so both forms agree bit for bit.  The constants come in
// scalar registers (SnakeConsts, made once per work item) so that every step is one packed instruction.
struct SnakeConsts {
    float inv_pi, pi_hi, pi_lo, c0, c1, c2, c3, c4, two;
};
__device__ __forceinline__ SnakeConsts snake_consts() {
    return SnakeConsts{svcmi_sgpr_const(0.31830987f), svcmi_sgpr_const(3.1415927f), svcmi_sgpr_const(-8.742278e-08f),
                       svcmi_sgpr_const(-2.3840804885821854e-08f), svcmi_sgpr_const(2.7522235086507862e-06f),
                       svcmi_sgpr_const(-0.00019840795721393079f), svcmi_sgpr_const(0.008333330042660236f),
                       svcmi_sgpr_const(-0.1666666716337204f), svcmi_sgpr_const(2.0f)};
}
__device__ __forceinline__ svcmi_f32x2 sin_sq2(svcmi_f32x2 x, const SnakeConsts& k) {
    if (__builtin_expect(fmaxf(fabsf(x[0]), fabsf(x[1])) > 1.0e5f, 0)) return svcmi_f32x2{sin_sq_huge(x[0]), sin_sq_huge(x[1])};
    const svcmi_f32x2 q = svcmi_mul2(x, svcmi_splat2(k.inv_pi));
    const svcmi_f32x2 n = {rintf(q[0]), rintf(q[1])};
    svcmi_f32x2 r = svcmi_fma2(-n, svcmi_splat2(k.pi_hi), x);
    r = svcmi_fma2(-n, svcmi_splat2(k.pi_lo), r);
    const svcmi_f32x2 u = svcmi_mul2(r, r);
    svcmi_f32x2 p = svcmi_splat2(k.c0);
    p = svcmi_fma2(p, u, svcmi_splat2(k.c1));
    p = svcmi_fma2(p, u, svcmi_splat2(k.c2));
    p = svcmi_fma2(p, u, svcmi_splat2(k.c3));
    p = svcmi_fma2(p, u, svcmi_splat2(k.c4));
    const svcmi_f32x2 sn = svcmi_fma2(svcmi_mul2(r, u), p, r);
    return svcmi_mul2(sn, sn);
}
__device__ __forceinline__ svcmi_f32x2 snake_fn2(svcmi_f32x2 y, float a, float inv_b, const SnakeConsts& k) {
    return svcmi_fma2(svcmi_splat2(inv_b), sin_sq2(svcmi_mul2(y, svcmi_splat2(a)), k), y);
}
// sin_sq2 without the large-argument test: for the callers that test a whole work item at once (snake_fn2_all)
__device__ __forceinline__ svcmi_f32x2 sin_sq2_nocheck(svcmi_f32x2 x, const SnakeConsts& k) {
    const svcmi_f32x2 q = svcmi_mul2(x, svcmi_splat2(k.inv_pi));
    const svcmi_f32x2 n = {rintf(q[0]), rintf(q[1])};
    svcmi_f32x2 r = svcmi_fma2(-n, svcmi_splat2(k.pi_hi), x);
    r = svcmi_fma2(-n, svcmi_splat2(k.pi_lo), r);
    const svcmi_f32x2 u = svcmi_mul2(r, r);
    svcmi_f32x2 p = svcmi_splat2(k.c0);
    p = svcmi_fma2(p, u, svcmi_splat2(k.c1));
    p = svcmi_fma2(p, u, svcmi_splat2(k.c2));
    p = svcmi_fma2(p, u, svcmi_splat2(k.c3));
    p = svcmi_fma2(p, u, svcmi_splat2(k.c4));
    const svcmi_f32x2 sn = svcmi_fma2(svcmi_mul2(r, u), p, r);
    return svcmi_mul2(sn, sn);
}
// s[m] = snake_fn2(y[m]) for the N pairs of a work item, with ONE large-argument test for all of them (round 4).  With the test inside
// every pair (snake_fn2) each pair was its own basic block: a dependent chain of ~20 packed operations behind an exec-mask save / branch
// / restore, 13 of them per work item, which the scheduler cannot interleave.  Here the N chains are straight-line code (independent:
// the scheduler overlaps them) and the rare fix-up pass replaces only the components that needed libm.  Same value per component as
// snake_fn2: (|x| <= 1e5) the polynomial path, else inv_b * sin^2_libm(x) + y.
template <int N>
__device__ __forceinline__ void snake_fn2_all(const svcmi_f32x2 (&y)[N], float a, float inv_b, const SnakeConsts& k, svcmi_f32x2 (&s)[N]) {
    float mx = 0.f;
#pragma unroll
    for (int m = 0; m < N; ++m) {
        const svcmi_f32x2 x = svcmi_mul2(y[m], svcmi_splat2(a));
        mx = fmaxf(mx, fmaxf(fabsf(x[0]), fabsf(x[1])));
        s[m] = svcmi_fma2(svcmi_splat2(inv_b), sin_sq2_nocheck(x, k), y[m]);
    }
    if (__builtin_expect(mx > 1.0e5f, 0)) {               // never taken for audio-scale activations
#pragma unroll
        for (int m = 0; m < N; ++m) {
            const svcmi_f32x2 x = svcmi_mul2(y[m], svcmi_splat2(a));
            if (fmaxf(fabsf(x[0]), fabsf(x[1])) > 1.0e5f)  // (the pair as a whole, as sin_sq2 decides it)
                s[m] = svcmi_fma2(svcmi_splat2(inv_b), svcmi_f32x2{sin_sq_huge(x[0]), sin_sq_huge(x[1])}, y[m]);
        }
    }
}

// The x window of a work item as ALIGNED register pairs: element i lives in half (i & 1) of pair (i >> 1).  The up-sampler multiplies a
// tap pair (f[2j], f[2j+1]) by the same x value in both halves; with the value in a half of an aligned pair the broadcast is an
// operand-select modifier of v_pk_fma_f32 (op_sel), with the value in a lone register the compiler spends a v_mov per tap to build the
// pair (ISA of round 3: 5 of the ~29 VALU instructions per up-sampled pair).
// ORDER OF THE FACTORS (round 6): the window value is the FIRST operand of every packed FMA -- svcmi_fma2(xw.splat(i), taps, acc) -- so that the
// half-select of an odd element lands on src0 (v_pk_fma_f32 ... op_sel:[1,0,0]).  As the second operand it became op_sel:[0,1,0], the low lane
// taking the HIGH half of src1: on MI355X that form loses its product in lanes 48..63 whenever another wave of the SIMD is executing
// v_mfma_f32_16x16x32_f16 / _bf16 (the fp16 fused half-step of another clip in flight).  Reproduced outside the library by
// scripts/probes/pkfma_mfma_corun.hip (profiles/r06y_pkfma_mfma_corun.log: src1 select x {16x16x32 f16, bf16} only; src0 select, op_sel_hi, the
// 32x32x16 / 16x16x16 / fp32 shapes are clean); tests/test_isa_packed_operand_select.py keeps the form out of the shipped binary.
template <int N>
struct SnakeWindow {
    svcmi_f32x2 p[(N + 1) / 2];
    __device__ __forceinline__ void set(int i, float v) { p[i >> 1][i & 1] = v; }
    __device__ __forceinline__ svcmi_f32x2 splat(int i) const { return (i & 1) ? svcmi_splat_hi(p[i >> 1]) : svcmi_splat_lo(p[i >> 1]); }
};
// taps of the 2x up-sampler with its gain folded in: (2 f[2j], 2 f[2j+1]).  Scaling by two commutes with every rounding of the fma
// chain (no overflow / underflow at audio scale), so sum_j (2 f) x == 2 * sum_j f x bit for bit -- one packed multiply less per pair.
__device__ __forceinline__ void snake_up_taps(const float (&f)[12], svcmi_f32x2 (&g2)[6]) {
#pragma unroll
    for (int j = 0; j < 6; ++j) g2[j] = svcmi_f32x2{2.f * f[2 * j], 2.f * f[2 * j + 1]};
}

// The two 6-tap packed FMA chains of a work item: y2[m] = sum_j g2[j] * x[5 - j + m] (up-sampler) and out[r] = sum_i f2[i] . P[r + i]
// (decimating low-pass).  Written value by value the compiler emits each chain as six DEPENDENT v_pk_fma_f32 on one accumulator with an
// s_nop between them (ISA of round 4: 78 s_nop per U-fill work item).  Walking the taps in the outer loop and the values in the inner one
// (consecutive instructions from different chains; bit-identical) was measured on MI355X in round 5 and changes nothing (92.8 vs 91.3 us
// at 20 channels, 64.4 vs 64.6 at 10: profiles/r05a_variants.log) -- the kernels are not bound by these chains; the variant was removed.
template <int NP, class Window>
__device__ __forceinline__ void snake_upsample(const Window& xw, const svcmi_f32x2 (&g2)[6], svcmi_f32x2 (&y2)[NP]) {
#pragma unroll
    for (int m = 0; m < NP; ++m) y2[m] = svcmi_splat2(0.f);
#pragma unroll
    for (int j = 0; j < 6; ++j)
#pragma unroll
        for (int m = 0; m < NP; ++m) y2[m] = svcmi_fma2(xw.splat(5 - j + m), g2[j], y2[m]);
}
template <int NR>
__device__ __forceinline__ void snake_fir_taps(const svcmi_f32x2 (&P)[NR + 5], const svcmi_f32x2 (&f2)[6], float (&out)[NR]) {
    svcmi_f32x2 z[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) z[r] = svcmi_splat2(0.f);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < NR; ++r) z[r] = svcmi_fma2(f2[i], P[r + i], z[r]);
#pragma unroll
    for (int r = 0; r < NR; ++r) out[r] = svcmi_hsum2(z[r]);
}

// s_up[u] for one up-sampled index 0 <= u < 2n straight from global memory; only the runs that touch a sequence end evaluate
// it (once each), for the replicate padding of the low-pass input (filter.py:86-95).
__device__ __forceinline__ float snake_s_at(const float* xc, int ld, int n, int u, const float* f, float a, float inv_b) {
    const int tq = u >> 1, odd = u & 1;
    float y = 0.f;
#pragma unroll
    for (int j = 0; j < 6; ++j) y = fmaf(f[2 * j + 1 - odd], xc[(long long)clampi(tq + 2 + odd - j, 0, n - 1) * ld], y);
    return snake_fn(2.f * y, a, inv_b);
}

// out[r] = SnakeAlias(x)[t0 + r], r < RT, for one channel.  xw[i] = x[clamp(t0 - 5 + i, 0, n-1)] (replicate padding of the
// up-sampler, resample.py:25-27); f = the 12 filter taps; xc / ld address the channel's column for the two end-of-sequence
// values.  Outputs for t outside [0, n) are meaningless (callers mask them).
template <int RT>
__device__ __forceinline__ void snake_run(const SnakeWindow<RT + 10>& xw, const float (&f)[12], float a, float inv_b,
                                          const float* xc, int ld, int n, int t0, float (&out)[RT]) {
    const SnakeConsts k = snake_consts();
    svcmi_f32x2 f2[6], g2[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) f2[j] = svcmi_f32x2{f[2 * j], f[2 * j + 1]};
    snake_up_taps(f, g2);
    // s2[m] = (s_up[2*t0 - 5 + 2m], s_up[2*t0 - 5 + 2m + 1]): polyphase up-sampler + SnakeBeta, each value computed once
    svcmi_f32x2 s2[RT + 5], y2[RT + 5];
#pragma unroll
    for (int m = 0; m < RT + 5; ++m) {
        svcmi_f32x2 y = svcmi_splat2(0.f);          // (odd phase: even taps, even phase: odd taps)
#pragma unroll
        for (int j = 0; j < 6; ++j) y = svcmi_fma2(xw.splat(5 - j + m), g2[j], y);
        y2[m] = y;
    }
    snake_fn2_all<RT + 5>(y2, a, inv_b, k, s2);
    const int u0 = 2 * t0 - 5;
    if (u0 < 0 || u0 + 2 * RT + 9 > 2 * n - 1) {      // replicate padding of the low-pass input
        const float s_first = snake_s_at(xc, ld, n, 0, f, a, inv_b);
        const float s_last = snake_s_at(xc, ld, n, 2 * n - 1, f, a, inv_b);
#pragma unroll
        for (int m = 0; m < RT + 5; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int u = u0 + 2 * m + h;
                s2[m][h] = u < 0 ? s_first : (u > 2 * n - 1 ? s_last : s2[m][h]);
            }
    }
#pragma unroll
    for (int r = 0; r < RT; ++r) {
        svcmi_f32x2 z = svcmi_splat2(0.f);
#pragma unroll
        for (int i = 0; i < 6; ++i) z = svcmi_fma2(f2[i], s2[r + i], z);
        out[r] = svcmi_hsum2(z);
    }
}

// The two halves of snake_run as separate steps, for kernels that keep the up-sampled SnakeBeta values of a whole tile in LDS so that
// each is computed ONCE (snake_run recomputes the 5 pairs two neighbouring runs share: 13 pairs per 8 outputs).  Same operation
// sequence per value as snake_run, so the results are bit-identical to it.
//   snake_pairs:  s2[mm] = (s_up[2*tq0 - 5 + 2mm], s_up[2*tq0 - 5 + 2mm + 1]), mm < NP, from xw[i] = x[clamp(tq0 - 5 + i, 0, n-1)], i < NP + 5
//   snake_fir:    out[r] = sum_i f2[i] . P[r + i], r < NR, from NR + 5 consecutive pairs P
template <int NP>
__device__ __forceinline__ void snake_pairs(const SnakeWindow<NP + 5>& xw, const float (&f)[12], float a, float inv_b, const float* xc, int ld,
                                            int n, int tq0, svcmi_f32x2 (&s2)[NP]) {
    const SnakeConsts k = snake_consts();
    svcmi_f32x2 g2[6];
    snake_up_taps(f, g2);
    svcmi_f32x2 y2[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) {
        svcmi_f32x2 y = svcmi_splat2(0.f);
#pragma unroll
        for (int j = 0; j < 6; ++j) y = svcmi_fma2(xw.splat(5 - j + m), g2[j], y);
        y2[m] = y;
    }
    snake_fn2_all<NP>(y2, a, inv_b, k, s2);
    const int u0 = 2 * tq0 - 5;
    if (u0 < 0 || u0 + 2 * NP - 1 > 2 * n - 1) {      // replicate padding of the low-pass input (filter.py:86-95)
        const float s_first = snake_s_at(xc, ld, n, 0, f, a, inv_b);
        const float s_last = snake_s_at(xc, ld, n, 2 * n - 1, f, a, inv_b);
#pragma unroll
        for (int m = 0; m < NP; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int u = u0 + 2 * m + h;
                s2[m][h] = u < 0 ? s_first : (u > 2 * n - 1 ? s_last : s2[m][h]);
            }
    }
}

template <int NR>
__device__ __forceinline__ void snake_fir(const svcmi_f32x2 (&P)[NR + 5], const float (&f)[12], float (&out)[NR]) {
    svcmi_f32x2 f2[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) f2[j] = svcmi_f32x2{f[2 * j], f[2 * j + 1]};
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        svcmi_f32x2 z = svcmi_splat2(0.f);
#pragma unroll
        for (int i = 0; i < 6; ++i) z = svcmi_fma2(f2[i], P[r + i], z);
        out[r] = svcmi_hsum2(z);
    }
}
