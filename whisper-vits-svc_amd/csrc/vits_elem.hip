// vits_elem.hip -- the small HBM-bound glue of the prior encoder / flow: WaveNet gate and
// residual bookkeeping, speaker-normalised coupling, pitch embedding, prior sampling, and the
// NCL <-> time-major bridges at the API edge.  All are one-pass streaming kernels: float4 where the
// layout allows, grid-stride over rows, channel index fastest so wavefronts touch contiguous bytes.
#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

constexpr int TPB = 256;

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// a = bias + sum of `splits` slabs laid out [batch][splits][t][2h] (the raw split-K partials of the in_layer convolution,
// SVCMI_CONV_PARTIALS) -- or the finished [batch][t][lda] activations when splits == 1 and bias == NULL.
__global__ __launch_bounds__(TPB) void wn_gate_kernel(const float* a, const float* bias, float* out, int batch, int t, int h,
                                                      int lda, int ldo, int splits, long long slab_stride, long long a_bs) {
    const int h4 = h >> 2;
    const long long total = (long long)batch * t * h4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long r = i / h4;
        const int c = (int)(i - r * h4) * 4;
        const int b = (int)(r / t), tt = (int)(r - (long long)b * t);
        const float* ap = a + (long long)b * a_bs + (long long)tt * lda + c;
        float4 ta = *reinterpret_cast<const float4*>(ap);
        float4 sa = *reinterpret_cast<const float4*>(ap + h);
        for (int s0 = 1; s0 < splits; s0 += 3) {       // fixed slab order: deterministic; up to 3 slabs requested before the first add
            float4 t2[3], s2[3];
#pragma unroll
            for (int e = 0; e < 3; ++e) {
                const bool on = s0 + e < splits;
                t2[e] = on ? *reinterpret_cast<const float4*>(ap + (s0 + e) * slab_stride) : make_float4(0.f, 0.f, 0.f, 0.f);
                s2[e] = on ? *reinterpret_cast<const float4*>(ap + (s0 + e) * slab_stride + h) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int e = 0; e < 3; ++e)
                if (s0 + e < splits) {
                    ta.x += t2[e].x; ta.y += t2[e].y; ta.z += t2[e].z; ta.w += t2[e].w;
                    sa.x += s2[e].x; sa.y += s2[e].y; sa.z += s2[e].z; sa.w += s2[e].w;
                }
        }
        if (bias) {
            const float4 tb = *reinterpret_cast<const float4*>(bias + c);
            const float4 sb = *reinterpret_cast<const float4*>(bias + h + c);
            ta.x += tb.x; ta.y += tb.y; ta.z += tb.z; ta.w += tb.w;
            sa.x += sb.x; sa.y += sb.y; sa.z += sb.z; sa.w += sb.w;
        }
        float4 o;
        o.x = tanhf(ta.x) * sigmoidf_(sa.x);
        o.y = tanhf(ta.y) * sigmoidf_(sa.y);
        o.z = tanhf(ta.z) * sigmoidf_(sa.z);
        o.w = tanhf(ta.w) * sigmoidf_(sa.w);
        *reinterpret_cast<float4*>(out + r * ldo + c) = o;
    }
}

__global__ __launch_bounds__(TPB) void wn_update_kernel(const float* rs, float* x, float* skip, const int32_t* lengths,
                                                        int batch, int t, int h, int ldrs, int first, int last) {
    const int h4 = h >> 2;
    const long long total = (long long)batch * t * h4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long r = i / h4;              // b*t + tt
        const int c = (int)(i - r * h4) * 4;
        const int b = (int)(r / t), tt = (int)(r - (long long)b * t);
        const float m = (!lengths || tt < lengths[b]) ? 1.f : 0.f;
        const float* rp = rs + r * ldrs;
        float4 sk = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *reinterpret_cast<const float4*>(skip + r * h + c);
        if (!last) {
            const float4 res = *reinterpret_cast<const float4*>(rp + c);
            const float4 s2 = *reinterpret_cast<const float4*>(rp + h + c);
            float4 xv = *reinterpret_cast<const float4*>(x + r * h + c);
            xv.x = (xv.x + res.x) * m; xv.y = (xv.y + res.y) * m; xv.z = (xv.z + res.z) * m; xv.w = (xv.w + res.w) * m;
            *reinterpret_cast<float4*>(x + r * h + c) = xv;
            sk.x += s2.x; sk.y += s2.y; sk.z += s2.z; sk.w += s2.w;
        } else {
            const float4 s2 = *reinterpret_cast<const float4*>(rp + c);
            sk.x = (sk.x + s2.x) * m; sk.y = (sk.y + s2.y) * m; sk.z = (sk.z + s2.z) * m; sk.w = (sk.w + s2.w) * m;
        }
        *reinterpret_cast<float4*>(skip + r * h + c) = sk;
    }
}

__global__ __launch_bounds__(TPB) void coupling_pre_kernel(const float* x, int ldx, int x0_off, const float* ms_vs,
                                                           float* out, int ldo, const int32_t* lengths,
                                                           int batch, int t, int half) {
    const long long total = (long long)batch * t * half;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long r = i / half;
        const int c = (int)(i - r * half);
        const int b = (int)(r / t), tt = (int)(r - (long long)b * t);
        const float m = (!lengths || tt < lengths[b]) ? 1.f : 0.f;
        const float ms = ms_vs[(long long)b * 2 * half + c], vs = ms_vs[(long long)b * 2 * half + half + c];
        out[r * ldo + c] = (x[r * ldx + x0_off + c] - ms) * expf(-vs) * m;
    }
}

__global__ __launch_bounds__(TPB) void coupling_post_kernel(float* x, int ldx, int x1_off, const float* mm, int ldm,
                                                            const float* ms_vs, const int32_t* lengths,
                                                            int batch, int t, int half) {
    const long long total = (long long)batch * t * half;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long r = i / half;
        const int c = (int)(i - r * half);
        const int b = (int)(r / t), tt = (int)(r - (long long)b * t);
        const float m = (!lengths || tt < lengths[b]) ? 1.f : 0.f;
        const float ms = ms_vs[(long long)b * 2 * half + c], vs = ms_vs[(long long)b * 2 * half + half + c];
        float x1 = x[r * ldx + x1_off + c];
        x1 = (x1 - mm[r * ldm + c]) * m;          // logs == 0 (mean_only): exp(-logs) = 1
        x1 = (ms + x1 * expf(vs)) * m;
        x[r * ldx + x1_off + c] = x1;
    }
}

// vits/utils.py:20-33 in fp32, same operation order as the torch path.
__device__ __forceinline__ int f0_to_coarse_dev(float f0) {
    // f0_mel_min and (f0_mel_max - f0_mel_min) are float64 Python scalars in the reference
    // (utils.py:16-17), rounded to fp32 when they meet the fp32 tensor: 77.75496616579426, 986.6532669978451.
    const float mel_min = 77.75496616579426f;
    const float mel_span = 986.6532669978451f;
    float mel = 1127.0f * logf(1.0f + f0 / 700.0f);
    if (mel > 0.f) mel = (mel - mel_min) * 254.0f / mel_span + 1.0f;
    if (mel <= 1.f) mel = 1.f;
    if (mel > 255.f) mel = 255.f;
    return (int)(mel + 0.5f);
}

__global__ __launch_bounds__(TPB) void embed_pitch_kernel(float* x, int ldx, const float* pit, const float* emb,
                                                          const int32_t* lengths, int batch, int t, int c) {
    const int c4 = c >> 2;
    const long long total = (long long)batch * t * c4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long r = i / c4;
        const int cc = (int)(i - r * c4) * 4;
        const int b = (int)(r / t), tt = (int)(r - (long long)b * t);
        const float m = (!lengths || tt < lengths[b]) ? 1.f : 0.f;
        const int bin = f0_to_coarse_dev(pit[r]);
        const float4 e = *reinterpret_cast<const float4*>(emb + (long long)bin * c + cc);
        float4 v = *reinterpret_cast<const float4*>(x + r * ldx + cc);
        v.x = (v.x + e.x) * m; v.y = (v.y + e.y) * m; v.z = (v.z + e.z) * m; v.w = (v.w + e.w) * m;
        *reinterpret_cast<float4*>(x + r * ldx + cc) = v;
    }
}

// 32x32 LDS-tiled transposes: reads coalesced along the source's fast dim, writes along the destination's.
__global__ __launch_bounds__(TPB) void sample_prior_kernel(const float* stats, int lds_, const float* noise,
                                                           const int32_t* lengths, float* z, int ldz, int t, int ic) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    const float* nb = noise + (long long)b * ic * t;
    for (int k = ty; k < 32; k += 8) {                          // noise[b][c0+k][t0+tx]
        int c = c0 + k, tt = t0 + tx;
        tile[k][tx] = (c < ic && tt < t) ? nb[(long long)c * t + tt] : 0.f;
    }
    __syncthreads();
    const int len = lengths ? lengths[b] : t;
    for (int k = ty; k < 32; k += 8) {                          // z[b][t0+k][c0+tx]
        int tt = t0 + k, c = c0 + tx;
        if (tt < t && c < ic) {
            const float* sp = stats + ((long long)b * t + tt) * lds_;
            float v = sp[c] + tile[tx][k] * expf(sp[ic + c]);
            z[((long long)b * t + tt) * ldz + c] = tt < len ? v : 0.f;
        }
    }
}

__global__ __launch_bounds__(TPB) void ncl_to_nlc_kernel(const float* x, const float* add, float add_scale, float* y,
                                                         int c, int t, int ldy) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const long long base = (long long)b * c * t;
    for (int k = ty; k < 32; k += 8) {
        int cc = c0 + k, tt = t0 + tx;
        float v = 0.f;
        if (cc < c && tt < t) {
            v = x[base + (long long)cc * t + tt];
            if (add) v += add_scale * add[base + (long long)cc * t + tt];
        }
        tile[k][tx] = v;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int tt = t0 + k, cc = c0 + tx;
        if (tt < t && cc < c) y[((long long)b * t + tt) * ldy + cc] = tile[tx][k];
    }
}

__global__ __launch_bounds__(TPB) void nlc_to_ncl_kernel(const float* x, int ldx, float* y, int c, int t) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8) {
        int tt = t0 + k, cc = c0 + tx;
        tile[k][tx] = (tt < t && cc < c) ? x[((long long)b * t + tt) * ldx + cc] : 0.f;
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        int cc = c0 + k, tt = t0 + tx;
        if (cc < c && tt < t) y[((long long)b * c + cc) * t + tt] = tile[tx][k];
    }
}

inline unsigned blocks_for(long long total) {
    long long nb = (total + TPB - 1) / TPB;
    const long long cap = 256 * 8;   // CUs x blocks/CU, grid-stride beyond that
    return (unsigned)(nb < 1 ? 1 : (nb > cap ? cap : nb));
}

inline bool mis16(const void* p) { return ((uintptr_t)p & 15) != 0; }

// y[r][0:cols] = x[r][0:cols]: the strided device copies of the stage host (csrc/host_stages.hip) as a stream-ordered kernel
__global__ __launch_bounds__(TPB) void copy2d_kernel(const float* x, long long ldx, float* y, long long ldy, long long rows, long long cols, int vec) {
    if (vec) {
        const long long c4 = cols >> 2, total = rows * c4;
        for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
            const long long r = i / c4, c = (i - r * c4) * 4;
            *reinterpret_cast<float4*>(y + r * ldy + c) = *reinterpret_cast<const float4*>(x + r * ldx + c);
        }
    } else {
        const long long total = rows * cols;
        for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
            const long long r = i / cols, c = i - r * cols;
            y[r * ldy + c] = x[r * ldx + c];
        }
    }
}

}  // namespace

extern "C" int svcmi_wn_gate_f32(const float* a, const float* bias, float* out, int32_t batch, int32_t t, int32_t h, int32_t lda,
                                 int32_t ldo, int32_t splits, void* stream) {
    if (!a || !out || batch <= 0 || t <= 0 || h <= 0 || splits < 1 || lda < 2 * h || ldo < h) return SVCMI_EINVAL;
    if (splits > 1 && lda != 2 * h) return SVCMI_EINVAL;       // slabs are dense [t][2h]
    if (h % 4 || lda % 4 || ldo % 4 || mis16(a) || mis16(out) || mis16(bias)) return SVCMI_EALIGN;
    const long long slab = (long long)t * lda;
    SVCMI_LAUNCH(wn_gate_kernel, dim3(blocks_for((long long)batch * t * (h / 4))), dim3(TPB), 0, stream, a, bias, out, batch, t, h,
                 lda, ldo, splits, slab, slab * splits);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_wn_update_f32(const float* rs, float* x, float* skip, const int32_t* lengths, int32_t batch, int32_t t,
                                   int32_t h, int32_t ldrs, int32_t first, int32_t last, void* stream) {
    if (!rs || !skip || (!last && !x) || batch <= 0 || t <= 0 || h <= 0) return SVCMI_EINVAL;
    if (h % 4 || ldrs % 4 || mis16(rs) || mis16(x) || mis16(skip)) return SVCMI_EALIGN;
    SVCMI_LAUNCH(wn_update_kernel, dim3(blocks_for((long long)batch * t * (h / 4))), dim3(TPB), 0, stream, rs, x, skip,
                 lengths, batch, t, h, ldrs, first, last);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_coupling_pre_f32(const float* x, int32_t ldx, int32_t x0_off, const float* ms_vs, float* out,
                                      int32_t ldo, const int32_t* lengths, int32_t batch, int32_t t, int32_t half, void* stream) {
    if (!x || !ms_vs || !out || batch <= 0 || t <= 0 || half <= 0 || x0_off < 0) return SVCMI_EINVAL;
    SVCMI_LAUNCH(coupling_pre_kernel, dim3(blocks_for((long long)batch * t * half)), dim3(TPB), 0, stream, x, ldx, x0_off,
                 ms_vs, out, ldo, lengths, batch, t, half);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_coupling_post_f32(float* x, int32_t ldx, int32_t x1_off, const float* m, int32_t ldm, const float* ms_vs,
                                       const int32_t* lengths, int32_t batch, int32_t t, int32_t half, void* stream) {
    if (!x || !m || !ms_vs || batch <= 0 || t <= 0 || half <= 0 || x1_off < 0) return SVCMI_EINVAL;
    SVCMI_LAUNCH(coupling_post_kernel, dim3(blocks_for((long long)batch * t * half)), dim3(TPB), 0, stream, x, ldx, x1_off,
                 m, ldm, ms_vs, lengths, batch, t, half);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_embed_pitch_f32(float* x, int32_t ldx, const float* pit, const float* emb, const int32_t* lengths,
                                     int32_t batch, int32_t t, int32_t c, void* stream) {
    if (!x || !pit || !emb || batch <= 0 || t <= 0 || c <= 0) return SVCMI_EINVAL;
    if (c % 4 || ldx % 4 || mis16(x) || mis16(emb)) return SVCMI_EALIGN;
    SVCMI_LAUNCH(embed_pitch_kernel, dim3(blocks_for((long long)batch * t * (c / 4))), dim3(TPB), 0, stream, x, ldx, pit,
                 emb, lengths, batch, t, c);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_sample_prior_f32(const float* stats, int32_t lds, const float* noise_ncl, const int32_t* lengths,
                                      float* z, int32_t ldz, int32_t batch, int32_t t, int32_t i, void* stream) {
    if (!stats || !noise_ncl || !z || batch <= 0 || t <= 0 || i <= 0 || lds < 2 * i || ldz < i) return SVCMI_EINVAL;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    SVCMI_LAUNCH(sample_prior_kernel, dim3((t + 31) / 32, (i + 31) / 32, batch), dim3(TPB), 0, stream, stats, lds, noise_ncl,
                 lengths, z, ldz, t, i);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_ncl_to_nlc_f32(const float* x, const float* add, float add_scale, float* y, int32_t batch, int32_t c,
                                    int32_t t, int32_t ldy, void* stream) {
    if (!x || !y || batch <= 0 || c <= 0 || t <= 0 || ldy < c) return SVCMI_EINVAL;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    SVCMI_LAUNCH(ncl_to_nlc_kernel, dim3((t + 31) / 32, (c + 31) / 32, batch), dim3(TPB), 0, stream, x, add, add_scale, y, c, t, ldy);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_nlc_to_ncl_f32(const float* x, int32_t ldx, float* y, int32_t batch, int32_t c, int32_t t, void* stream) {
    if (!x || !y || batch <= 0 || c <= 0 || t <= 0 || ldx < c) return SVCMI_EINVAL;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    SVCMI_LAUNCH(nlc_to_ncl_kernel, dim3((t + 31) / 32, (c + 31) / 32, batch), dim3(TPB), 0, stream, x, ldx, y, c, t);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_copy2d_f32(const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols, void* stream) {
    if (!x || !y || rows <= 0 || cols <= 0 || ldx < cols || ldy < cols) return SVCMI_EINVAL;
    const int vec = cols % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && !mis16(x) && !mis16(y);
    SVCMI_LAUNCH(copy2d_kernel, dim3(blocks_for(vec ? rows * (cols / 4) : rows * cols)), dim3(TPB), 0, stream, x, (long long)ldx, y,
                 (long long)ldy, (long long)rows, (long long)cols, vec);
    return SVCMI_LAST_ERROR();
}
