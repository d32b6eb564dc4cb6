// norm_attn.hip -- row LayerNorm and fp32 multi-head attention (online softmax, optional
// relative-position band), both on time-major rows.  Reductions use wavefront shuffles (64 lanes).
#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// ------------------------------------------------------------------------------------ LayerNorm
// One wave per row; a lane owns float4 #(lane + 64*i).  Two-pass (mean, then centred variance) like
// torch's CPU layer_norm so results track the oracle to rounding.
constexpr int LN_MAXV = 8;   // c <= 64*4*8 = 2048

__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* res, const float* gamma,
                                                        const float* beta, float* y, int rows, int rows_per_batch,
                                                        int c, int ldx, int ldr, int ldy, int gb_bs, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;   // whole waves exit together; no block-level barrier below
    const int nv = c >> 2;
    const float* xr = x + (long long)row * ldx;
    const float* rr = res ? res + (long long)row * ldr : nullptr;
    float4 v[LN_MAXV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int q = lane + 64 * i;
        if (q < nv) {
            float4 t = *reinterpret_cast<const float4*>(xr + 4 * q);
            if (rr) {
                float4 u = *reinterpret_cast<const float4*>(rr + 4 * q);
                t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
            }
            v[i] = t;
            s += (t.x + t.y) + (t.z + t.w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float mean = wave_sum(s) / (float)c;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int q = lane + 64 * i;
        if (q < nv) {
            float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(ss) / (float)c + eps);
    const long long gb = (long long)(row / rows_per_batch) * gb_bs;
    float* yr = y + (long long)row * ldy;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
        int q = lane + 64 * i;
        if (q < nv) {
            float4 g = gamma ? *reinterpret_cast<const float4*>(gamma + gb + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
            float4 bb = beta ? *reinterpret_cast<const float4*>(beta + gb + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + bb.x;
            o.y = (v[i].y - mean) * rstd * g.y + bb.y;
            o.z = (v[i].z - mean) * rstd * g.z + bb.z;
            o.w = (v[i].w - mean) * rstd * g.w + bb.w;
            *reinterpret_cast<float4*>(yr + 4 * q) = o;
        }
    }
}

// ------------------------------------------------------------------------------------ attention
// lane = query.  q and the output accumulator of one head (D floats each) live in registers; K and V
// rows are addressed wave-uniformly, so hipcc fetches them with scalar loads and the FMAs take them as
// SGPR operands -- no LDS staging, no barrier in the key loop, 2*D FMAs per (query, key) with no
// redundancy.  The NW waves of a block share 64 queries and split the key range; their partial
// (max, sum, acc) states are merged through LDS at the end (flash-decoding style).  Softmax is the
// online form, rescaled once per chunk of CH keys.  The relative-position band (|j-i| <= window) only
// touches <= 2*window+1 keys per query and is handled by a divergent side branch.
struct AttnArgs {
    const float* q; const float* k; const float* v; float* o;
    int ldq, ldk, ldv, ldo;
    long long q_bs, k_bs, v_bs, o_bs;
    int t, heads;
    float scale;
    const float* rel_k; const float* rel_v;
    int window;
    const int32_t* lengths;
};

constexpr int CH = 4;    // keys per softmax rescale chunk
constexpr int NW = 4;    // waves per block = key-range splits
constexpr int MAXW = 4;  // largest relative window

template <int D>
__global__ __launch_bounds__(64 * NW) void attention_kernel(AttnArgs p) {
    __shared__ float part[(NW - 1) * (D + 2) * 64];   // states of waves 1..NW-1, [w-1][slot][lane]

    const int lane = threadIdx.x & 63;
    const int w = SVCMI_UNIFORM((int)(threadIdx.x >> 6));
    const int qi = blockIdx.x * 64 + lane;
    const int h = blockIdx.y, b = blockIdx.z;
    const int T = p.t;
    const int len = p.lengths ? p.lengths[b] : T;
    const bool qvalid = qi < T;
    const bool has_rel = p.rel_k != nullptr;
    const int W = p.window;

    float qreg[D], oacc[D];
    {
        const float* qp = p.q + (long long)b * p.q_bs + (long long)(qvalid ? qi : 0) * p.ldq + h * D;
#pragma unroll
        for (int d = 0; d < D; d += 4) {
            float4 t4 = *reinterpret_cast<const float4*>(qp + d);
            qreg[d] = t4.x; qreg[d + 1] = t4.y; qreg[d + 2] = t4.z; qreg[d + 3] = t4.w;
        }
#pragma unroll
        for (int d = 0; d < D; ++d) oacc[d] = 0.f;
    }
    float mrun = -3.0e38f, lrun = 0.f;
    const float* kb = p.k + (long long)b * p.k_bs + h * D;
    const float* vb = p.v + (long long)b * p.v_bs + h * D;

    // this wave's key range, CH-aligned
    const int per = ((T + NW - 1) / NW + CH - 1) / CH * CH;
    const int jbeg = w * per;
    const int jend = (jbeg + per) < T ? (jbeg + per) : T;

    for (int j0 = jbeg; j0 < jend; j0 += CH) {
        float s[CH];
        float cmax = -3.0e38f;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = j0 + u;
            const int jc = j < T ? j : T - 1;              // uniform clamp keeps the address valid
            const float* kr = kb + (long long)jc * p.ldk;  // wave-uniform -> scalar loads
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) a = fmaf(qreg[d], kr[d], a);
            if (has_rel) {
                const int rel = j - qi + W;
                if (rel >= 0 && rel <= 2 * W) {            // divergent: <= 2W+1 keys per query
                    const float* e = p.rel_k + rel * D;
                    float ae = 0.f;
#pragma unroll
                    for (int d = 0; d < D; ++d) ae = fmaf(qreg[d], e[d], ae);
                    a += ae;
                }
            }
            a *= p.scale;
            if (qi >= len || j >= len) a = -1.0e4f;         // masked_fill(mask == 0, -1e4)
            if (j >= T) a = -3.0e38f;                       // beyond the sequence: weight 0
            s[u] = a;
            cmax = fmaxf(cmax, a);
        }
        const float mnew = fmaxf(mrun, cmax);
        const float corr = expf(mrun - mnew);
        lrun *= corr;
#pragma unroll
        for (int d = 0; d < D; ++d) oacc[d] *= corr;
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int j = j0 + u;
            const int jc = j < T ? j : T - 1;
            const float pj = (j < T) ? expf(s[u] - mnew) : 0.f;
            lrun += pj;
            const float* vr = vb + (long long)jc * p.ldv;   // wave-uniform
#pragma unroll
            for (int d = 0; d < D; ++d) oacc[d] = fmaf(pj, vr[d], oacc[d]);
            if (has_rel) {
                const int rel = j - qi + W;
                if (rel >= 0 && rel <= 2 * W && j < T) {
                    const float* e = p.rel_v + rel * D;
#pragma unroll
                    for (int d = 0; d < D; ++d) oacc[d] = fmaf(pj, e[d], oacc[d]);
                }
            }
        }
        mrun = mnew;
    }

    // merge the NW partial states (wave 0 owns the result)
    if (w > 0) {
        float* pw = part + (w - 1) * (D + 2) * 64 + lane;
        pw[0] = mrun;
        pw[64] = lrun;
#pragma unroll
        for (int d = 0; d < D; ++d) pw[(2 + d) * 64] = oacc[d];
    }
    __syncthreads();
    if (w == 0) {
        float mall = mrun;
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) mall = fmaxf(mall, part[(ww - 1) * (D + 2) * 64 + lane]);
        float c0 = expf(mrun - mall);
        float lall = lrun * c0;
#pragma unroll
        for (int d = 0; d < D; ++d) oacc[d] *= c0;
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) {
            const float* pw = part + (ww - 1) * (D + 2) * 64 + lane;
            const float cw = expf(pw[0] - mall);
            lall += pw[64] * cw;
#pragma unroll
            for (int d = 0; d < D; ++d) oacc[d] = fmaf(pw[(2 + d) * 64], cw, oacc[d]);
        }
        if (qvalid) {
            const float inv = 1.0f / lall;
            float* op = p.o + (long long)b * p.o_bs + (long long)qi * p.ldo + h * D;
#pragma unroll
            for (int d = 0; d < D; d += 4)
                *reinterpret_cast<float4*>(op + d) = make_float4(oacc[d] * inv, oacc[d + 1] * inv, oacc[d + 2] * inv, oacc[d + 3] * inv);
        }
    }
}

template <int D>
int launch_attn(const AttnArgs& a, int batch, void* stream) {
    dim3 grid((a.t + 63) / 64, a.heads, batch);
    SVCMI_LAUNCH((attention_kernel<D>), grid, dim3(64 * NW), 0, stream, a);
    return SVCMI_LAST_ERROR();
}

}  // namespace

extern "C" int svcmi_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                   int32_t batch, int32_t rows_per_batch, int32_t c, int32_t ldx, int32_t ldr,
                                   int32_t ldy, int32_t gb_bstride, float eps, void* stream) {
    if (!x || !y || batch <= 0 || rows_per_batch <= 0 || c <= 0) return SVCMI_EINVAL;
    if (c % 4 != 0 || c > 64 * 4 * LN_MAXV) return SVCMI_EUNSUPPORTED;
    if (ldx % 4 || ldy % 4 || (res && ldr % 4) || gb_bstride % 4) return SVCMI_EALIGN;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)res & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15))
        return SVCMI_EALIGN;
    const long long rows = (long long)batch * rows_per_batch;
    if (rows > 0x7fffffffLL) return SVCMI_EUNSUPPORTED;
    SVCMI_LAUNCH(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, res, gamma, beta, y,
                 (int)rows, rows_per_batch, c, ldx, ldr, ldy, gb_bstride, eps);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_attention_f32(const float* q, const float* k, const float* v, float* o,
                                   int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                   int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                                   int32_t batch, int32_t t, int32_t heads, int32_t head_dim, float scale,
                                   const float* rel_k, const float* rel_v, int32_t window,
                                   const int32_t* lengths, void* stream) {
    if (!q || !k || !v || !o || batch <= 0 || t <= 0 || heads <= 0) return SVCMI_EINVAL;
    if ((rel_k == nullptr) != (rel_v == nullptr)) return SVCMI_EINVAL;
    if (rel_k && (window < 0 || window > MAXW)) return SVCMI_EUNSUPPORTED;
    if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4 || q_bstride % 4 || k_bstride % 4 || v_bstride % 4 || o_bstride % 4)
        return SVCMI_EALIGN;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 15)) return SVCMI_EALIGN;
    if (batch > 65535 || heads > 65535) return SVCMI_EUNSUPPORTED;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.q_bs = q_bstride; a.k_bs = k_bstride; a.v_bs = v_bstride; a.o_bs = o_bstride;
    a.t = t; a.heads = heads; a.scale = scale; a.rel_k = rel_k; a.rel_v = rel_v; a.window = window; a.lengths = lengths;
    switch (head_dim) {
        case 16: return launch_attn<16>(a, batch, stream);
        case 32: return launch_attn<32>(a, batch, stream);
        case 64: return launch_attn<64>(a, batch, stream);
        case 96: return launch_attn<96>(a, batch, stream);
        default: return SVCMI_EUNSUPPORTED;
    }
}
