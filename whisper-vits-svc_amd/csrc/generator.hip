// generator.hip -- NSF-BigVGAN pieces that are not convolutions: the anti-aliased SnakeBeta
// activation (the dominant cost of the reference CPU path), the harmonic source
// (SineGen + merge), and the int16 side output.  All are HBM-bound streaming kernels on
// time-major [batch][len][ld] tensors; a thread owns one channel and a run of RT consecutive samples
// so the 12-tap polyphase windows are reused from registers (3.25 sin evaluations per output
// instead of 12) while neighbouring lanes walk neighbouring channels (contiguous bytes).
#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

constexpr int TPB = 256;
constexpr int SNAKE_RT_BIG = 8;   // run length of the SnakeAlias stream kernel for arithmetic-bound (batched) launches: measured choice, see svcmi_snake_alias_group_f32
int g_snake_rt = 0;

#include "snake_math.h"

// Up to 3 activations of one shape per launch (blockIdx.z): the AMP blocks of a generator stage have their own alpha / beta
// and, after the first step, their own input.
constexpr int SNAKE_GROUP = 3;
struct SnakeArgs {
    const float* x[SNAKE_GROUP]; float* y[SNAKE_GROUP]; const float* alpha_log[SNAKE_GROUP]; const float* beta_log[SNAKE_GROUP];
    const float* filt;
    int n, c, ld;
    unsigned short* y16[SNAKE_GROUP];      // when set: the output goes out as bf16 / fp16 rows INSTEAD of fp32 (a following _A16 GEMM's A operand)
    int f16;                               // 0 bf16, 1 f16, 2 split bf16 rows [hi: ld | lo: ld] (svcmi_store4_16's codes)
};

// RTK = outputs per thread along time: a run of RTK outputs evaluates RTK + 5 up-sampled pairs (13 per 8 = 1.63x, 17 per 12 = 1.42x, 21 per
// 16 = 1.31x) at the price of registers; same operation sequence per value, so every run length gives the same bits.
template <int RTK>
__global__ __launch_bounds__(TPB) void snake_alias_kernel(SnakeArgs p) {
    constexpr int RT = RTK;
    const int b = blockIdx.y, gi = blockIdx.z;
    const int n = p.n, c = p.c, ld = p.ld;
    const float* filt = p.filt;
    const long long e = (long long)blockIdx.x * TPB + threadIdx.x;
    const int ch = (int)(e % c);
    const long long run = e / c;
    const long long t0l = run * RT;
    if (t0l >= n) return;                       // no barriers in this kernel
    const int t0 = (int)t0l;
    const float a = expf(p.alpha_log[gi][ch]);
    const float inv_b = 1.0f / (expf(p.beta_log[gi][ch]) + 1e-9f);
    const float* xc = p.x[gi] + (long long)b * n * ld + ch;
    float* yc = p.y[gi] ? p.y[gi] + (long long)b * n * ld + ch : nullptr;
    const int ld16 = p.f16 == 2 ? 2 * ld : ld;      // split rows: [hi: ld | lo: ld]
    unsigned short* yh = p.y16[gi] ? p.y16[gi] + (long long)b * n * ld16 + ch : nullptr;

    float f[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) f[k] = filt[k];
    // x window with replicate padding (resample.py:25-27): xw[i] = x[clamp(t0 - 5 + i)]
    SnakeWindow<RT + 10> xw;
    if (t0 >= 5 && t0 + RT + 4 < n) {
        // interior run (all but the first and last of a sequence): no clamps, and one 32-bit offset per lane from the wave-uniform batch
        // base, advanced by ld per row -- the general form costs a clamp + a 64-bit multiply-add + a 64-bit add per load (ISA: 90 of the
        // ~520 vector instructions of a work item were address arithmetic)
        const float* xu = svcmi_opaque_uniform(p.x[gi] + (long long)b * n * ld);
        const unsigned o = 4u * (unsigned)((t0 - 5) * ld + ch), step = 4u * (unsigned)ld;       // BYTE offsets inside one batch item: the entry point rejects len * ld * 4 >= 2^31
#pragma unroll
        for (int i = 0; i < RT + 10; ++i) xw.set(i, svcmi_load_saddr(xu, o + (unsigned)i * step));
    } else {
#pragma unroll
        for (int i = 0; i < RT + 10; ++i) xw.set(i, xc[(long long)clampi(t0 - 5 + i, 0, n - 1) * ld]);
    }
    float out[RT];
    snake_run<RT>(xw, f, a, inv_b, xc, ld, n, t0, out);
#pragma unroll
    for (int r = 0; r < RT; ++r)
        if (t0 + r < n) {
            if (yh) {
                unsigned short* q = yh + (long long)(t0 + r) * ld16;
                if (p.f16 == 1) {
                    q[0] = (unsigned short)(svcmi_cvt_pk_f16(out[r], 0.f) & 0xffffu);
                } else {
                    const unsigned h = svcmi_cvt_pk_bf16(out[r], 0.f) & 0xffffu;
                    q[0] = (unsigned short)h;
                    if (p.f16 == 2) q[ld] = (unsigned short)(svcmi_cvt_pk_bf16(out[r] - svcmi_bits_f32(h << 16), 0.f) & 0xffffu);
                }
            }
            else yc[(long long)(t0 + r) * ld] = out[r];
        }
}

// y = ((x0 + x1) + x2) / count -- `xs = xs + resblock(x)` ... `x = xs / num_kernels` (vits_decoder/generator.py:188-194), float4 stream
__global__ __launch_bounds__(TPB) void block_mean_kernel(const float* x0, const float* x1, const float* x2, float* y, long long n4,
                                                         float count) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n4; i += (long long)gridDim.x * TPB) {
        float4 a = reinterpret_cast<const float4*>(x0)[i];
        if (x1) { const float4 b = reinterpret_cast<const float4*>(x1)[i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        if (x2) { const float4 b = reinterpret_cast<const float4*>(x2)[i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        a.x /= count; a.y /= count; a.z /= count; a.w /= count;
        reinterpret_cast<float4*>(y)[i] = a;
    }
}

// ------------------------------------------------------------------------------------ harmonic source
constexpr int NH = 11;

__device__ __forceinline__ float rad_of(float f0, int k, float sr) {
    // (f0 * (k+1) / sr) % 1 in fp32, nsf.py:228,293-297
    return fmodf((f0 * (float)(k + 1)) / sr, 1.0f);
}

// Phase prefix of every frame, fp64, as a three-level scan per (batch item, harmonic): each of the 23 lanes a
// harmonic owns sums a contiguous chunk of frames, lane 0 of the harmonic scans the 23 chunk totals, then every
// lane replays its chunk from its offset.  All sums are taken mod 1 (only the fractional phase matters), so the
// re-association moves a prefix by ~1e-16.  One block per batch item; 11 harmonics x 23 lanes = 253 threads busy.
constexpr int PL = 23;      // lanes per harmonic: 11 * 23 <= 256

__global__ __launch_bounds__(TPB) void pitch_prefix_kernel(const float* f0, const float* rand_ini, double* prefix,
                                                           int t, int hop, float sr) {
    __shared__ double tot[NH * PL];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int k = tid / PL, l = tid - k * PL;
    const bool on = k < NH;
    const int per = (t + PL - 1) / PL;
    const int f_beg = l * per, f_end = (f_beg + per) < t ? (f_beg + per) : t;
    const float* fb = f0 + (long long)b * t;
    double sum = 0.0;
    if (on) {
        for (int f = f_beg; f < f_end; ++f) {
            sum += (double)hop * (double)rad_of(fb[f], k, sr);
            sum -= floor(sum);
        }
        tot[tid] = sum;
    }
    __syncthreads();
    if (on && l == 0) {       // exclusive scan of the chunk totals, seeded with the initial phase (column 0 forced to 0, :235)
        double acc = k > 0 ? (double)rand_ini[b * NH + k] : 0.0;
        for (int i = 0; i < PL; ++i) {
            const double v = tot[k * PL + i];
            tot[k * PL + i] = acc;
            acc += v;
            acc -= floor(acc);
        }
    }
    __syncthreads();
    if (on) {
        double acc = tot[tid];
        for (int f = f_beg; f < f_end; ++f) {
            prefix[((long long)b * t + f) * NH + k] = acc;
            acc += (double)hop * (double)rad_of(fb[f], k, sr);
            acc -= floor(acc);
        }
    }
}

__global__ __launch_bounds__(TPB) void pitch_source_kernel(const float* f0, const double* prefix, const float* noise,
                                                           const float* merge_w, float merge_b, float* out,
                                                           int t, int hop, float sr) {
    const int b = blockIdx.y;
    const long long L = (long long)t * hop;
    const long long s = (long long)blockIdx.x * TPB + threadIdx.x;
    if (s >= L) return;
    const int fr = (int)(s / hop), j = (int)(s - (long long)fr * hop);
    const float f0v = f0[(long long)b * t + fr];
    const float uv = f0v > 0.f ? 1.f : 0.f;
    const float namp = uv * 0.003f + (1.f - uv) * 0.1f / 3.f;      // nsf.py:310
    const double* pf = prefix + ((long long)b * t + fr) * NH;
    const float* nz = noise + ((long long)b * L + s) * NH;
    float acc = 0.f;
#pragma unroll
    for (int k = 0; k < NH; ++k) {
        const float rad = rad_of(f0v, k, sr);
        double ph = pf[k] + (double)(j + 1) * (double)rad;
        ph -= floor(ph);
        const float sine = sinf(((float)ph * 2.0f) * 3.14159265358979323846f) * 0.1f;
        const float hk = sine * uv + namp * nz[k];
        acc = fmaf(hk, merge_w[k], acc);
    }
    out[(long long)b * L + s] = tanhf(acc + merge_b);
}

__global__ __launch_bounds__(TPB) void source2wav_kernel(const float* x, int16_t* y, long long n) {
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < n; i += (long long)gridDim.x * TPB) {
        float v = 32768.0f * x[i];
        v = v < -32768.0f ? -32768.0f : (v > 32767.0f ? 32767.0f : v);
        y[i] = (int16_t)v;     // torch .short(): truncation toward zero
    }
}

}  // namespace

extern "C" int svcmi_snake_alias_group_f32(const float* const* x, float* const* y, const float* const* alpha_log,
                                           const float* const* beta_log, const float* filt, int32_t count, int32_t batch,
                                           int32_t len, int32_t c, int32_t ld, void* const* y16, int32_t y16_format, void* stream) {
    if (!x || (!y && !y16) || !alpha_log || !beta_log || !filt || count < 1 || count > SNAKE_GROUP) return SVCMI_EINVAL;
    if (y16 && svcmi_fmt16(y16_format) < 0) return SVCMI_EINVAL;
    if (batch <= 0 || len <= 0 || c <= 0 || ld < c) return SVCMI_EINVAL;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    if ((long long)len * ld * 4 >= (1LL << 31)) return SVCMI_EUNSUPPORTED;      // the interior runs form 32-bit byte offsets inside ONE batch item (int arithmetic)
    SnakeArgs a;
    for (int i = 0; i < SNAKE_GROUP; ++i) {
        const int j = i < count ? i : 0;
        float* yj = y ? y[j] : nullptr;
        unsigned short* hj = y16 ? static_cast<unsigned short*>(y16[j]) : nullptr;
        if (!x[j] || (!yj && !hj) || !alpha_log[j] || !beta_log[j]) return SVCMI_EINVAL;
        if (x[j] == yj) return SVCMI_EINVAL;              // halo reads: not an in-place op
        a.x[i] = x[j]; a.y[i] = yj; a.y16[i] = hj; a.alpha_log[i] = alpha_log[j]; a.beta_log[i] = beta_log[j];
    }
    a.filt = filt; a.n = len; a.c = c; a.ld = ld; a.f16 = y16 ? svcmi_fmt16(y16_format) : 0;
    // run length per thread: 8 for one clip (more, smaller work items: the launch is latency-bound), longer runs once the launch is
    // arithmetic-bound (tuning knob "snake_rt": 0 = this rule, 8 | 12 | 16 forced)
    const long long elems = (long long)batch * len * c * count;
    const int rt = g_snake_rt ? g_snake_rt : (elems >= (1LL << 25) ? SNAKE_RT_BIG : 8);
    const long long runs = ((long long)len + rt - 1) / rt;
    const long long threads = runs * c;
    const dim3 grid((unsigned)((threads + TPB - 1) / TPB), batch, count);
    if (rt == 16) SVCMI_LAUNCH(snake_alias_kernel<16>, grid, dim3(TPB), 0, stream, a);
    else if (rt == 12) SVCMI_LAUNCH(snake_alias_kernel<12>, grid, dim3(TPB), 0, stream, a);
    else SVCMI_LAUNCH(snake_alias_kernel<8>, grid, dim3(TPB), 0, stream, a);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_snake_tune_set(const char* name, int32_t value) {
    const char* k = "snake_rt";
    int i = 0;
    while (k[i] && name[i] == k[i]) ++i;
    if (k[i] == 0 && name[i] == 0 && (value == 0 || value == 8 || value == 12 || value == 16)) { g_snake_rt = value; return 0; }
    return SVCMI_EINVAL;
}

extern "C" int svcmi_snake_alias_f32(const float* x, float* y, const float* alpha_log, const float* beta_log,
                                     const float* filt, int32_t batch, int32_t len, int32_t c, int32_t ld, void* stream) {
    return svcmi_snake_alias_group_f32(&x, &y, &alpha_log, &beta_log, filt, 1, batch, len, c, ld, nullptr, 0, stream);
}

extern "C" int svcmi_block_mean_f32(const float* const* xs, int32_t count, float* y, int64_t n, void* stream) {
    if (!xs || !y || count < 1 || count > 3 || n <= 0) return SVCMI_EINVAL;
    for (int i = 0; i < count; ++i)
        if (!xs[i]) return SVCMI_EINVAL;
    if (n % 4) return SVCMI_EINVAL;
    if (((uintptr_t)y & 15) || ((uintptr_t)xs[0] & 15) || (count > 1 && ((uintptr_t)xs[1] & 15)) || (count > 2 && ((uintptr_t)xs[2] & 15))) return SVCMI_EALIGN;
    long long nb = (n / 4 + TPB - 1) / TPB;
    if (nb > 4096) nb = 4096;
    SVCMI_LAUNCH(block_mean_kernel, dim3((unsigned)nb), dim3(TPB), 0, stream, xs[0], count > 1 ? xs[1] : nullptr,
                 count > 2 ? xs[2] : nullptr, y, (long long)(n / 4), (float)count);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_pitch_prefix_f64(const float* f0, const float* rand_ini, double* prefix, int32_t batch, int32_t t,
                                      int32_t hop, float sr, void* stream) {
    if (!f0 || !rand_ini || !prefix || batch <= 0 || t <= 0 || hop <= 0 || !(sr > 0.f)) return SVCMI_EINVAL;
    SVCMI_LAUNCH(pitch_prefix_kernel, dim3(batch), dim3(TPB), 0, stream, f0, rand_ini, prefix, t, hop, sr);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_pitch_source_f32(const float* f0, const double* prefix, const float* noise, const float* merge_w,
                                      float merge_b, float* out, int32_t batch, int32_t t, int32_t hop, float sr, void* stream) {
    if (!f0 || !prefix || !noise || !merge_w || !out || batch <= 0 || t <= 0 || hop <= 0 || !(sr > 0.f)) return SVCMI_EINVAL;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    const long long L = (long long)t * hop;
    SVCMI_LAUNCH(pitch_source_kernel, dim3((unsigned)((L + TPB - 1) / TPB), batch), dim3(TPB), 0, stream, f0, prefix, noise,
                 merge_w, merge_b, out, t, hop, sr);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_source2wav_i16(const float* x, int16_t* y, int64_t n, void* stream) {
    if (!x || !y || n <= 0) return SVCMI_EINVAL;
    long long nb = (n + TPB - 1) / TPB;
    if (nb > 2048) nb = 2048;
    SVCMI_LAUNCH(source2wav_kernel, dim3((unsigned)nb), dim3(TPB), 0, stream, x, y, (long long)n);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_abi_version(void) { return SVCMI_ABI_VERSION; }

extern "C" const char* svcmi_build_info(void) {
#ifdef SVCMI_EMU
    return "emu";
#else
    return "hip:gfx950";
#endif
}
