// conv_gemm.hip -- implicit-GEMM 1-D convolution / linear layer on the gfx950 fp32 matrix cores.
//
// One kernel serves every Conv1d / Linear / (polyphase) ConvTranspose1d of the path (see svcmi.h).
// Time-major activations make the im2col matrix free: row t of the A operand is the contiguous
// span x[t*stride - pad ... ][0:c_in] for dilation 1, and a gather of `ksize` row segments otherwise.
//
// Tiling (wave64, 256 threads = 2x2 waves):  block tile (64*WM) x (64*WN), wave tile (32*WM) x (32*WN)
// as WM x WN accumulators of v_mfma_f32_32x32x2_f32 (16 VGPR each).  K is walked in steps of 32:
// both operands are K-contiguous in HBM, fetched with 16-byte loads, staged in LDS as [row][32+4]
// (the +4 pad makes the 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots) and double
// buffered so the next tile's global loads fly under the current tile's MFMAs.
// K-order trick: a lane's ds_read_b128 returns 4 consecutive k; lanes 0-31 take k = 8s+0..3 and
// lanes 32-63 take k = 8s+4..7, so MFMA #j of sub-step s multiplies k-pairs (8s+j, 8s+4+j) -- a
// permutation of the K summation shared by A and B, i.e. the same dot product.
#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

constexpr int BK = 32;
constexpr int LDS_LD = BK + 4;

struct ConvArgs {
    const float* x; const float* w; const float* bias; const float* res; float* y; const int32_t* lengths;
    long long x_bs, y_bs, r_bs;
    int t_in, t_out, c_in, ldx, n_out, ldw, ldy, ldr;
    int ksize, stride, dil, pad, rshift, act, flags;
    int ktot;        // ksize * c_in
    int vec;         // 16-byte A loads legal
    int chunk_tap;   // c_in % BK == 0: a K-step never straddles taps
    float alpha;
};

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case SVCMI_ACT_RELU: return v > 0.f ? v : 0.f;
        case SVCMI_ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));
        case SVCMI_ACT_MISH: {   // x * tanh(softplus(x)), softplus with torch's threshold 20
            float sp = v > 20.f ? v : log1pf(expf(v));
            return v * tanhf(sp);
        }
        case SVCMI_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// One 4-wide slice of the implicit im2col matrix: row t, flat K index kk..kk+3.
__device__ __forceinline__ float4 load_a4(const ConvArgs& p, const float* xb, int t, int kk, int t_lim,
                                          int tap_u, int ci_u) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (t >= p.t_out) return r;
    if (p.vec) {
        if (kk >= p.ktot) return r;
        int k, ci;
        if (p.chunk_tap) { k = tap_u; ci = ci_u; }
        else { k = kk / p.c_in; ci = kk - k * p.c_in; }
        int tin = t * p.stride + k * p.dil - p.pad;
        if (tin < 0 || tin >= t_lim) return r;
        return *reinterpret_cast<const float4*>(xb + (long long)(tin >> p.rshift) * p.ldx + ci);
    }
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        int q = kk + e;
        float val = 0.f;
        if (q < p.ktot) {
            int k = q / p.c_in, ci = q - k * p.c_in;
            int tin = t * p.stride + k * p.dil - p.pad;
            if (tin >= 0 && tin < t_lim) val = xb[(long long)(tin >> p.rshift) * p.ldx + ci];
        }
        v[e] = val;
    }
    return make_float4(v[0], v[1], v[2], v[3]);
}

template <int WM, int WN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(ConvArgs p) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int A_PER = BM / 32, B_PER = BN / 32;   // float4 loads per thread per K-step
    __shared__ float As[2][BM * LDS_LD];
    __shared__ float Bs[2][BN * LDS_LD];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int b = blockIdx.z;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const float* xb = p.x + (long long)b * p.x_bs;
    const int len = p.lengths ? p.lengths[b] : 0x7fffffff;
    const int t_lim = (p.flags & SVCMI_CONV_MASK_IN) ? (len < p.t_in ? len : p.t_in) : p.t_in;

    const int lrow = tid >> 3, lkq = (tid & 7) * 4;   // this thread's (row, k-offset) in a 32-row slab
    const int nk = (p.ktot + BK - 1) / BK;

    svcmi_f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 ra[A_PER], rb[B_PER];
    auto gload = [&](int it) {
        const int k0 = it * BK;
        int tap_u = 0, ci_u = 0;
        if (p.chunk_tap) { tap_u = k0 / p.c_in; ci_u = k0 - tap_u * p.c_in + lkq; }
#pragma unroll
        for (int i = 0; i < A_PER; ++i) ra[i] = load_a4(p, xb, m0 + lrow + 32 * i, k0 + lkq, t_lim, tap_u, ci_u);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            int n = n0 + lrow + 32 * i, kk = k0 + lkq;
            rb[i] = (n < p.n_out && kk < p.ldw) ? *reinterpret_cast<const float4*>(p.w + (long long)n * p.ldw + kk)
                                                  : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto sstore = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) *reinterpret_cast<float4*>(&As[buf][(lrow + 32 * i) * LDS_LD + lkq]) = ra[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) *reinterpret_cast<float4*>(&Bs[buf][(lrow + 32 * i) * LDS_LD + lkq]) = rb[i];
    };

    gload(0);
    sstore(0);
    __syncthreads();
    const int arow = (wm * 32 * WM + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    const int brow = (wn * 32 * WN + (lane & 31)) * LDS_LD + (lane >> 5) * 4;
    for (int it = 0; it < nk; ++it) {
        const int cur = it & 1;
        if (it + 1 < nk) gload(it + 1);
#pragma unroll
        for (int s = 0; s < BK / 8; ++s) {
            float4 a4[WM], b4[WN];
#pragma unroll
            for (int i = 0; i < WM; ++i) a4[i] = *reinterpret_cast<const float4*>(&As[cur][arow + i * 32 * LDS_LD + s * 8]);
#pragma unroll
            for (int j = 0; j < WN; ++j) b4[j] = *reinterpret_cast<const float4*>(&Bs[cur][brow + j * 32 * LDS_LD + s * 8]);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) {
                    acc[i][j] = svcmi_mfma_32x32x2(a4[i].x, b4[j].x, acc[i][j]);
                    acc[i][j] = svcmi_mfma_32x32x2(a4[i].y, b4[j].y, acc[i][j]);
                    acc[i][j] = svcmi_mfma_32x32x2(a4[i].z, b4[j].z, acc[i][j]);
                    acc[i][j] = svcmi_mfma_32x32x2(a4[i].w, b4[j].w, acc[i][j]);
                }
        }
        if (it + 1 < nk) sstore(cur ^ 1);
        __syncthreads();
    }

    // epilogue: D layout col = lane&31 (n), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (t)
    float* yb = p.y + (long long)b * p.y_bs;
    const float* rbp = p.res ? p.res + (long long)b * p.r_bs : nullptr;
    const bool accum = (p.flags & SVCMI_CONV_ACCUMULATE) != 0;
    const bool mask_out = (p.flags & SVCMI_CONV_MASK_OUT) != 0;
#pragma unroll
    for (int j = 0; j < WN; ++j) {
        const int n = n0 + wn * 32 * WN + j * 32 + (lane & 31);
        if (n >= p.n_out) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int t = m0 + wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (t >= p.t_out) continue;
                float v = act_apply(acc[i][j][r] + bv, p.act);
                if (rbp) v += rbp[(long long)t * p.ldr + n];
                v *= p.alpha;
                float* dst = yb + (long long)t * p.ldy + n;
                if (accum) v += *dst;
                if (mask_out && t >= len) v = 0.f;
                *dst = v;
            }
        }
    }
}

template <int WM, int WN>
int launch(const ConvArgs& a, int batch, void* stream) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    dim3 grid((a.t_out + BM - 1) / BM, (a.n_out + BN - 1) / BN, batch);
    SVCMI_LAUNCH((conv_gemm_kernel<WM, WN>), grid, dim3(256), 0, stream, a);
    return SVCMI_LAST_ERROR();
}

}  // namespace

extern "C" int svcmi_conv_gemm_f32(const svcmi_conv_desc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->y) return SVCMI_EINVAL;
    if (d->batch <= 0 || d->t_in <= 0 || d->t_out <= 0 || d->c_in <= 0 || d->n_out <= 0 || d->ksize <= 0) return SVCMI_EINVAL;
    if (d->stride <= 0 || d->dilation <= 0 || d->x_row_shift < 0 || d->x_row_shift > 1) return SVCMI_EINVAL;
    if (d->ldw % 4 != 0 || d->ldw < d->ksize * d->c_in) return SVCMI_EINVAL;
    if (d->ldy < d->n_out || (d->res && d->ldr < d->n_out) || d->ldx < d->c_in) return SVCMI_EINVAL;
    if ((d->flags & (SVCMI_CONV_MASK_IN | SVCMI_CONV_MASK_OUT)) && !d->lengths) return SVCMI_EINVAL;
    if (d->act < SVCMI_ACT_NONE || d->act > SVCMI_ACT_TANH) return SVCMI_EINVAL;
    if (((uintptr_t)d->w & 15) != 0) return SVCMI_EALIGN;
    if (d->batch > 65535 || (d->n_out + 63) / 64 > 65535) return SVCMI_EUNSUPPORTED;

    ConvArgs a;
    a.x = d->x; a.w = d->w; a.bias = d->bias; a.res = d->res; a.y = d->y; a.lengths = d->lengths;
    a.x_bs = d->x_bstride; a.y_bs = d->y_bstride; a.r_bs = d->res_bstride;
    a.t_in = d->t_in; a.t_out = d->t_out; a.c_in = d->c_in; a.ldx = d->ldx; a.n_out = d->n_out;
    a.ldw = d->ldw; a.ldy = d->ldy; a.ldr = d->ldr;
    a.ksize = d->ksize; a.stride = d->stride; a.dil = d->dilation; a.pad = d->pad; a.rshift = d->x_row_shift;
    a.act = d->act; a.flags = d->flags; a.alpha = d->alpha;
    a.ktot = d->ksize * d->c_in;
    a.vec = (d->c_in % 4 == 0) && (d->ldx % 4 == 0) && (d->x_bstride % 4 == 0) && (((uintptr_t)d->x & 15) == 0);
    a.chunk_tap = a.vec && (d->c_in % BK == 0);

    // Tile choice: fill the 256 CUs first, then grow the tile for operand reuse (or honour the override).
    switch (d->flags & SVCMI_CONV_TILE_MASK) {
        case SVCMI_CONV_TILE_64x64: return launch<1, 1>(a, d->batch, stream);
        case SVCMI_CONV_TILE_128x64: return launch<2, 1>(a, d->batch, stream);
        case SVCMI_CONV_TILE_128x128: return launch<2, 2>(a, d->batch, stream);
        default: break;
    }
    const long long mt64 = (d->t_out + 63) / 64, nt64 = (d->n_out + 63) / 64;
    const long long blocks64 = mt64 * nt64 * d->batch;
    if (d->n_out > 64 && blocks64 >= 4 * 1024 && d->t_out >= 128) return launch<2, 2>(a, d->batch, stream);
    if (blocks64 >= 2 * 1024 && d->t_out >= 128) return launch<2, 1>(a, d->batch, stream);
    return launch<1, 1>(a, d->batch, stream);
}
