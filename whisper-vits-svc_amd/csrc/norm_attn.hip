// norm_attn.hip -- row LayerNorm (wavefront-shuffle reductions) and fp32 multi-head attention on the
// exact-fp32 matrix cores (online softmax, optional relative-position band), both on time-major rows.
#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

// 16-bit copy of four consecutive outputs (the A operand of a following SVCMI_PREC_*_A16 GEMM), 8-byte store

// ------------------------------------------------------------------------------------ LayerNorm
// One wave per row; a lane owns float4 #(lane + 64*i).  Two-pass (mean, then centred variance) like
// torch's CPU layer_norm so results track the oracle to rounding.
constexpr int LN_MAXV = 8;   // c <= 64*4*8 = 2048

__global__ __launch_bounds__(256) void layernorm_kernel(const float* x, const float* res, const float* gamma,
                                                        const float* beta, float* y, int rows, int rows_per_batch,
                                                        int c, int ldx, int ldr, int ldy, int gb_bs, float eps,
                                                        unsigned short* y16, int ldy16, int f16) {
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
            if (y16) svcmi_store4_16(y16 + (long long)row * ldy16 + 4 * q, ldy16 >> 1, o.x, o.y, o.z, o.w, f16);
        }
    }
}

// ------------------------------------------------------------------------------------ per-channel norm over time
// GroupNorm(C, C) + GELU of HuBERT's first feature-extractor layer (hubert/hubert_model.py:78,88): every channel is
// normalised over the whole time axis of its batch item.  On time-major rows that is a column reduction: (1) fp64
// partial sums per (batch item, time chunk, channel), lanes walking channels so loads stay coalesced; (2) a tiny
// finalise to mean / rstd; (3) one elementwise pass.  Fixed chunking and summation order: deterministic.
constexpr int CN_CHUNKS_MAX = 64;

__global__ __launch_bounds__(256) void colstats_kernel(const float* x, double* part, int t, int c, int ldx, int nch) {
    __shared__ double red[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int ch = blockIdx.x * 64 + cl, chunk = blockIdx.y, b = blockIdx.z;
    const int per = (t + nch - 1) / nch;
    const int r0 = chunk * per, r1 = (r0 + per) < t ? (r0 + per) : t;
    double s = 0.0, ss = 0.0;
    if (ch < c) {
        const float* xb = x + (long long)b * t * ldx + ch;
        for (int r = r0 + rl; r < r1; r += 4) {
            const double v = (double)xb[(long long)r * ldx];
            s += v;
            ss += v * v;
        }
    }
    red[0][rl][cl] = s;
    red[1][rl][cl] = ss;
    __syncthreads();
    if (rl == 0 && ch < c) {
        double* o = part + (((long long)b * nch + chunk) * c + ch) * 2;
        o[0] = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
        o[1] = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    }
}

__global__ __launch_bounds__(256) void colstats_final_kernel(const double* part, float* stats, int t, int c, int nch, float eps) {
    const int ch = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
    if (ch >= c) return;
    double s = 0.0, ss = 0.0;
    for (int k = 0; k < nch; ++k) {
        const double* o = part + (((long long)b * nch + k) * c + ch) * 2;
        s += o[0];
        ss += o[1];
    }
    const double mean = s / t;
    double var = ss / t - mean * mean;         // biased, like torch group_norm
    if (var < 0.0) var = 0.0;
    stats[((long long)b * c + ch) * 2] = (float)mean;
    stats[((long long)b * c + ch) * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

__global__ __launch_bounds__(256) void channel_norm_gelu_kernel(const float* x, const float* stats, const float* gamma,
                                                                const float* beta, float* y, int t, int c, int ldx, int ldy) {
    const int c4 = c >> 2;
    const long long total = (long long)t * c4;
    const int b = blockIdx.y;
    const float* xb = x + (long long)b * t * ldx;
    float* yb = y + (long long)b * t * ldy;
    const float* st = stats + (long long)b * c * 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / c4;
        const int cc = (int)(i - r * c4) * 4;
        const float4 v = *reinterpret_cast<const float4*>(xb + r * ldx + cc);
        const float vv[4] = {v.x, v.y, v.z, v.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float n = (vv[j] - st[2 * (cc + j)]) * st[2 * (cc + j) + 1] * (gamma ? gamma[cc + j] : 1.f) + (beta ? beta[cc + j] : 0.f);
            o[j] = svcmi_gelu(n);
        }
        *reinterpret_cast<float4*>(yb + r * ldy + cc) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

// Split-K tail fused with the residual update and the LayerNorm that follows it (Whisper: x += W_o a + b_o; h = LN(x)):
//   x[r,:] += bias + sum_{s < split} partials[b, s, t, :]   (fixed slice order: deterministic);   y[r,:] = LN(x[r,:])
// One 256-thread block per row (a 500-row Whisper window would otherwise occupy only 500 waves, each waiting on
// `split` dependent-latency slab reads): a thread owns <= 2 float4 columns, statistics go wave shuffle -> LDS.
// Replaces the split-K reduce launch and the LayerNorm launch.
constexpr int SKV = 2;      // float4 columns per thread: c <= 256 * 4 * SKV = 2048

__device__ __forceinline__ float block_sum_256(float v, float* red) {
    v = wave_sum(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();                 // red[] may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void splitk_layernorm_kernel(const float* part, int split, const float* bias, float* x,
                                                               const float* gamma, const float* beta, float* y,
                                                               int rows_per_batch, int c, int ldx, int ldy, float eps,
                                                               unsigned short* y16, int ldy16, int f16) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const int nv = c >> 2;
    const int b = row / rows_per_batch, t = row - b * rows_per_batch;
    float* xr = x + (long long)row * ldx;
    const float* pr = part + ((long long)b * split * rows_per_batch + t) * c;
    const long long sstride = (long long)rows_per_batch * c;
    float4 v[SKV], gv[SKV], bv[SKV];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < SKV; ++i) {
        const int q = threadIdx.x + 256 * i;
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (q < nv) {       // (the affine parameters too: behind the two block reductions their latency would be exposed once more)
            gv[i] = gamma ? *reinterpret_cast<const float4*>(gamma + 4 * q) : make_float4(1.f, 1.f, 1.f, 1.f);
            bv[i] = beta ? *reinterpret_cast<const float4*>(beta + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            // every operand of this item is requested before the first add: the slabs of up to 4 slices at a time (a run-time trip count
            // made the loop wait for each slice in turn: 4 dependent L2 round trips per row at MLP-down), the bias and the residual row;
            // the slices are still added in slice order
            float4 acc = *reinterpret_cast<const float4*>(pr + 4 * q);
            const float4 xo = *reinterpret_cast<const float4*>(xr + 4 * q);
            const float4 bb = bias ? *reinterpret_cast<const float4*>(bias + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int sl0 = 1; sl0 < split; sl0 += 4) {
                float4 u[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    u[e] = sl0 + e < split ? *reinterpret_cast<const float4*>(pr + (sl0 + e) * sstride + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (sl0 + e < split) { acc.x += u[e].x; acc.y += u[e].y; acc.z += u[e].z; acc.w += u[e].w; }
            }
            if (bias) { acc.x += bb.x; acc.y += bb.y; acc.z += bb.z; acc.w += bb.w; }
            acc.x += xo.x; acc.y += xo.y; acc.z += xo.z; acc.w += xo.w;      // epilogue order of the GEMM: (sum + bias) + res
            *reinterpret_cast<float4*>(xr + 4 * q) = acc;
            v[i] = acc;
            s += (acc.x + acc.y) + (acc.z + acc.w);
        }
    }
    const float mean = block_sum_256(s, red) / (float)c;
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < SKV; ++i) {
        if (threadIdx.x + 256 * i < nv) {
            const float dx = v[i].x - mean, dy = v[i].y - mean, dz = v[i].z - mean, dw = v[i].w - mean;
            ss += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
    }
    const float rstd = 1.0f / sqrtf(block_sum_256(ss, red) / (float)c + eps);
    float* yr = y + (long long)row * ldy;
#pragma unroll
    for (int i = 0; i < SKV; ++i) {
        const int q = threadIdx.x + 256 * i;
        if (q < nv) {
            const float4 g = gv[i], bb = bv[i];
            float4 o;
            o.x = (v[i].x - mean) * rstd * g.x + bb.x;
            o.y = (v[i].y - mean) * rstd * g.y + bb.y;
            o.z = (v[i].z - mean) * rstd * g.z + bb.z;
            o.w = (v[i].w - mean) * rstd * g.w + bb.w;
            *reinterpret_cast<float4*>(yr + 4 * q) = o;
            if (y16) svcmi_store4_16(y16 + (long long)row * ldy16 + 4 * q, ldy16 >> 1, o.x, o.y, o.z, o.w, f16);
        }
    }
}

// ------------------------------------------------------------------------------------ attention
// Flash-style attention on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32), both products "swapped" so
// that a lane always owns ONE query column:
//     S^T[key][query] = K Q^T        A = K rows (i = key),  B = Q^T (j = query)
//     O^T[d][query]   = V^T P^T      A = V^T   (i = d),     B = P^T (j = query)
// The D layout of the first product (lane (query, quarter g) holds keys 4g..4g+3 of a 16-key tile) is exactly
// the B-operand layout the second one needs -- MFMA step r contracts keys {r, 4+r, 8+r, 12+r} -- so P never
// moves between lanes, softmax statistics are per-lane scalars (max: 7 fmax + 2 shuffles per 32 keys), and the
// online rescale of O^T is a lane-local multiply.  A wave owns 16 queries; the NS waves of a block split the
// key range (flash-decoding style) and merge their (max, sum, O) states through LDS, which also turns the
// per-lane O^T columns into coalesced 16-byte row stores.  K and V fragments are read straight from global
// memory (a head's K/V is <= 1 MB and L2-resident; waves of a block read different keys, so LDS staging would
// buy no reuse).  The relative-position band (|j-i| <= window, vits/attentions.py:225-347) adds q.E_k[j-i+w]
// to the <= 2w+1 in-band scores -- only tiles touching the diagonal pay for it -- and accumulates the in-band
// probabilities per query so that sum_r Pband[r] E_v[r] is added once, in the merge.
// Blocks are enumerated so that consecutive (head, q-tile) work lands on one XCD (block b runs on XCD b % 8):
// each XCD's L2 then holds ~1/8 of the heads instead of all of them.
struct AttnArgs {
    const float* q; const float* k; const float* v; float* o;
    int ldq, ldk, ldv, ldo;
    long long q_bs, k_bs, v_bs, o_bs;
    int t, heads, nq;
    float scale;
    const float* rel_k; const float* rel_v;
    int window;
    const int32_t* lengths;
    unsigned short* o16;       // optional 16-bit copy of o (rows of ldo16 values, batch stride o16_bs): the out-projection's A operand
    long long o16_bs;
    int ldo16, o16_f16;
};

constexpr int MAXW = 4;            // largest relative window
constexpr int NREL = 2 * MAXW + 1;
constexpr int BST = 12;            // row stride of the per-query band sums in LDS
constexpr float NEG_BIG = -3.0e38f;
constexpr float LOG2E = 1.4426950408889634f;
constexpr float MASKED2 = -1.0e4f * LOG2E;     // the reference's masked_fill value, in the log2 domain

__device__ __forceinline__ float quarter_sum(float v) {    // sum over the 4 lane quarters (same lane & 15)
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 32);
    return v;
}

// V^T tiles of the second product.  Head widths 32 / 64 (DS = 2 / 4): row i of tile dt is channel d = DS*i + dt, so the DS A-operand
// values a lane needs of one key are CONTIGUOUS in memory (one 16-byte load at D = 64 where per-channel tiles took four 4-byte ones; a
// wave instruction then fetches 4 whole head rows).  Other widths keep row i of tile dt = channel 16*dt + i (at D = 96 three 8-byte loads
// per key measured SLOWER than six 4-byte ones: 40.2 vs 31.7 us for the T = 1000 prior-encoder launch, profiles/r03x_kernel_stats.csv).
// `vrow` = the key's row of this head; lq = lane & 15.
template <int DS>
__device__ __forceinline__ void load_vrow(const float* vrow, int lq, float (&v)[DS]) {
    if constexpr (DS == 4) {
        const float4 t4 = *reinterpret_cast<const float4*>(vrow + 4 * lq);
        v[0] = t4.x; v[1] = t4.y; v[2] = t4.z; v[3] = t4.w;
    } else if constexpr (DS == 2) {
        const float2 t2 = *reinterpret_cast<const float2*>(vrow + 2 * lq);
        v[0] = t2.x; v[1] = t2.y;
    } else {
#pragma unroll
        for (int e = 0; e < DS; ++e) v[e] = vrow[16 * e + lq];
    }
}
// ... and the accumulators back into channel order: oacc[dt][r] of lane (lq, g4) is channel DS * (4*g4 + r) + dt (contiguous form) or
// 16*dt + 4*g4 + r of query lq.  `orow` = the query's row of the partial-O tile.
template <int DS>
__device__ __forceinline__ void store_orow(float* orow, int g4, const svcmi_f32x4 (&oacc)[DS]) {
    if constexpr (DS == 4) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *reinterpret_cast<float4*>(orow + 16 * g4 + 4 * r) = make_float4(oacc[0][r], oacc[1][r], oacc[2][r], oacc[3][r]);
    } else if constexpr (DS == 2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) *reinterpret_cast<float2*>(orow + 8 * g4 + 2 * r) = make_float2(oacc[0][r], oacc[1][r]);
    } else {
#pragma unroll
        for (int dt = 0; dt < DS; ++dt)
            *reinterpret_cast<float4*>(orow + 16 * dt + 4 * g4) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
    }
}

// REL (round 6: a template parameter, was a run-time flag): the relative-position band.  The band-free instantiation (Whisper) carries none of
// its code and none of its 18 registers (R[] / Pb[]) -- at the 64-register budget of 4 waves per SIMD those were accumulator-file spills.
template <int D, int NS, bool REL = true>
__global__ __launch_bounds__(64 * NS) void attention_kernel(AttnArgs p) {
    constexpr int DS = D / 16;         // 16-wide d groups: float4 K/Q fragments per row and 16-row tiles of O^T
    constexpr int OLD = D + 4;         // padded row of the partial-O tile: conflict-free ds_write_b128
    __shared__ __attribute__((aligned(16))) float smem[NS * 16 * OLD + 2 * NS * 16 + (REL ? NS * 16 * BST + 2 * NREL * D : 0)];
    float* const Opart = smem;                         // [NS][16][OLD]
    float* const Mpart = Opart + NS * 16 * OLD;        // [NS][16]
    float* const Lpart = Mpart + NS * 16;              // [NS][16]
    float* const Bpart = Lpart + NS * 16;              // [NS][16][BST]
    float* const Ek = Bpart + NS * 16 * BST;           // [NREL][D]
    float* const Ev = Ek + NREL * D;                   // [NREL][D]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = SVCMI_UNIFORM((int)(tid >> 6));
    const int lq = lane & 15, g4 = lane >> 4;
    // XCD-aware, bijective enumeration of (batch*head, q-tile) pairs
    int L;
    {
        const int total = (int)gridDim.x, id = (int)blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = id & 7, slot = id >> 3;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int qt = L % p.nq, hb = L / p.nq;
    const int h = hb % p.heads, b = hb / p.heads;
    const int T = p.t;
    const int len = p.lengths ? p.lengths[b] : T;
    const bool has_rel = REL && p.rel_k != nullptr;
    const int W = p.window;
    const int nrel = has_rel ? 2 * W + 1 : 0;
    const float scale2 = p.scale * LOG2E;              // softmax in the log2 domain (v_exp_f32 computes 2^x)

    if (has_rel) {
        for (int i = tid; i < nrel * D; i += 64 * NS) { Ek[i] = p.rel_k[i]; Ev[i] = p.rel_v[i]; }
        __syncthreads();
    }

    const int q0 = qt * 16, qi = q0 + lq;
    float qf[DS][4];
    {
        const float* qp = p.q + (long long)b * p.q_bs + (long long)(qi < T ? qi : T - 1) * p.ldq + h * D + 4 * g4;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float4 t4 = *reinterpret_cast<const float4*>(qp + 16 * s);
            qf[s][0] = t4.x; qf[s][1] = t4.y; qf[s][2] = t4.z; qf[s][3] = t4.w;
        }
    }
    float R[NREL], Pb[NREL];
#pragma unroll
    for (int e = 0; e < NREL; ++e) { R[e] = 0.f; Pb[e] = 0.f; }
    if (has_rel) {
#pragma unroll
        for (int e = 0; e < NREL; ++e) {
            if (e < nrel) {           // wave-uniform
                float a = 0.f;
#pragma unroll
                for (int s = 0; s < DS; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a = fmaf(qf[s][j], Ek[e * D + 16 * s + 4 * g4 + j], a);
                R[e] = quarter_sum(a);
            }
        }
    }

    svcmi_f32x4 oacc[DS];
#pragma unroll
    for (int dt = 0; dt < DS; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[dt][r] = 0.f;
    float mrun = NEG_BIG, lrun = 0.f;

    const float* kb = p.k + (long long)b * p.k_bs + h * D + 4 * g4;
    const float* vb = p.v + (long long)b * p.v_bs + h * D;
    const int per = ((T + NS - 1) / NS + 31) / 32 * 32;     // keys per wave, a multiple of the 32-key step
    const int jbeg = w * per;
    const int jend = (jbeg + per) < T ? (jbeg + per) : T;

    // (Requesting the next step's K fragments / this step's V values ahead of the MFMAs -- explicit software pipelining -- was measured
    // and not kept: +32 / +64 VGPRs take the kernel from 4 to 3 / 2 waves per SIMD and one T = 500 window stays at 26-27 us,
    // profiles/r03q_attnpf.log; with 4 waves per SIMD the other waves already cover a wave's load latency.)
    auto load_k = [&](int kt, float4 (&ka)[2][DS]) {
        const int k0 = kt + lq, k1 = kt + 16 + lq;
        const float* kr0 = kb + (long long)(k0 < T ? k0 : T - 1) * p.ldk;
        const float* kr1 = kb + (long long)(k1 < T ? k1 : T - 1) * p.ldk;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            ka[0][s] = *reinterpret_cast<const float4*>(kr0 + 16 * s);
            ka[1][s] = *reinterpret_cast<const float4*>(kr1 + 16 * s);
        }
    };
    for (int kt = jbeg; kt < jend; kt += 32) {
        float4 kc[2][DS];
        load_k(kt, kc);
        // ---- S^T = K Q^T for two 16-key tiles (two independent accumulators hide the 40-cycle MFMA latency)
        svcmi_f32x4 sacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[u][r] = 0.f;
        {
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const float4 a0 = kc[0][s];
                const float4 a1 = kc[1][s];
                sacc[0] = svcmi_mfma_16x16x4(a0.x, qf[s][0], sacc[0]);
                sacc[1] = svcmi_mfma_16x16x4(a1.x, qf[s][0], sacc[1]);
                sacc[0] = svcmi_mfma_16x16x4(a0.y, qf[s][1], sacc[0]);
                sacc[1] = svcmi_mfma_16x16x4(a1.y, qf[s][1], sacc[1]);
                sacc[0] = svcmi_mfma_16x16x4(a0.z, qf[s][2], sacc[0]);
                sacc[1] = svcmi_mfma_16x16x4(a1.z, qf[s][2], sacc[1]);
                sacc[0] = svcmi_mfma_16x16x4(a0.w, qf[s][3], sacc[0]);
                sacc[1] = svcmi_mfma_16x16x4(a1.w, qf[s][3], sacc[1]);
            }
        }
        // ---- scores of this lane: keys kt + 16u + 4*g4 + r, query qi
        const bool diag = has_rel && (kt + 31 >= q0 - W) && (kt <= q0 + 15 + W);    // wave-uniform
        const bool clean = kt + 32 <= (len < T ? len : T) && q0 + 16 <= len;         // wave-uniform: no masked or out-of-range entry
        float sv[2][4];
        float mt = NEG_BIG;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * u + 4 * g4 + r;
                float a = sacc[u][r];
                if (diag) {
                    const int rel = key - qi + W;
                    float add = 0.f;
#pragma unroll
                    for (int e = 0; e < NREL; ++e) add = (rel == e && e < nrel) ? R[e] : add;
                    a += add;
                }
                a *= scale2;                                    // scores live in the log2 domain: exp(s - m) = 2^(s' - m')
                if (!clean) {                                   // (wave-uniform: a step entirely inside the valid rows skips the compares)
                    if (qi >= len || key >= len) a = MASKED2;   // masked_fill(mask == 0, -1e4)
                    if (key >= T) a = NEG_BIG;                  // beyond the sequence: weight 0
                }
                sv[u][r] = a;
                mt = fmaxf(mt, a);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mnew = fmaxf(mrun, mt);
        const float corr = svcmi_exp2(mrun - mnew);
        mrun = mnew;
        lrun *= corr;
#pragma unroll
        for (int dt = 0; dt < DS; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] *= corr;
        float pv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[u][r] = svcmi_exp2(sv[u][r] - mnew);         // 2^(NEG_BIG - finite) == 0 for keys >= T
                lrun += pv[u][r];
            }
        if (diag) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) Pb[e] *= corr;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int rel = kt + 16 * u + 4 * g4 + r - qi + W;
#pragma unroll
                    for (int e = 0; e < NREL; ++e) Pb[e] += (rel == e && e < nrel) ? pv[u][r] : 0.f;
                }
        } else if (has_rel) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) Pb[e] *= corr;
        }
        // ---- O^T += V^T P^T : step (u, r) contracts keys kt + 16u + {r, 4+r, 8+r, 12+r}
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * u + 4 * g4 + r;
                float vv[DS];
                load_vrow<DS>(vb + (long long)(key < T ? key : T - 1) * p.ldv, lq, vv);
#pragma unroll
                for (int dt = 0; dt < DS; ++dt) oacc[dt] = svcmi_mfma_16x16x4(vv[dt], pv[u][r], oacc[dt]);
            }
    }

    // ---- publish this wave's state: O^T columns become rows of Opart
    lrun = quarter_sum(lrun);
    if (has_rel) {
#pragma unroll
        for (int e = 0; e < NREL; ++e) Pb[e] = quarter_sum(Pb[e]);
    }
    {
        store_orow<DS>(Opart + (w * 16 + lq) * OLD, g4, oacc);
        if (g4 == 0) {
            Mpart[w * 16 + lq] = mrun;
            Lpart[w * 16 + lq] = lrun;
            if (has_rel) {
#pragma unroll
                for (int e = 0; e < NREL; ++e) Bpart[(w * 16 + lq) * BST + e] = Pb[e];
            }
        }
    }
    __syncthreads();

    // ---- merge the NS partial states; one thread per (query, 4 channels)
    float* ob = p.o + (long long)b * p.o_bs + h * D;
    for (int item = tid; item < 16 * (D / 4); item += 64 * NS) {
        const int ql = item / (D / 4), c4 = (item - ql * (D / 4)) * 4;
        float mall = Mpart[ql];
#pragma unroll
        for (int ww = 1; ww < NS; ++ww) mall = fmaxf(mall, Mpart[ww * 16 + ql]);
        float den = 0.f;
        float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
        float bs[NREL];
#pragma unroll
        for (int e = 0; e < NREL; ++e) bs[e] = 0.f;
#pragma unroll
        for (int ww = 0; ww < NS; ++ww) {
            const float cw = svcmi_exp2(Mpart[ww * 16 + ql] - mall);
            den = fmaf(cw, Lpart[ww * 16 + ql], den);
            const float4 ov = *reinterpret_cast<const float4*>(Opart + (ww * 16 + ql) * OLD + c4);
            num.x = fmaf(cw, ov.x, num.x); num.y = fmaf(cw, ov.y, num.y);
            num.z = fmaf(cw, ov.z, num.z); num.w = fmaf(cw, ov.w, num.w);
            if (has_rel) {
#pragma unroll
                for (int e = 0; e < NREL; ++e) bs[e] = fmaf(cw, Bpart[(ww * 16 + ql) * BST + e], bs[e]);
            }
        }
        if (has_rel) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) {
                if (e < nrel) {
                    const float4 ev = *reinterpret_cast<const float4*>(Ev + e * D + c4);
                    num.x = fmaf(bs[e], ev.x, num.x); num.y = fmaf(bs[e], ev.y, num.y);
                    num.z = fmaf(bs[e], ev.z, num.z); num.w = fmaf(bs[e], ev.w, num.w);
                }
            }
        }
        const int qrow = q0 + ql;
        if (qrow < T) {
            const float inv = 1.0f / den;
            *reinterpret_cast<float4*>(ob + (long long)qrow * p.ldo + c4) = make_float4(num.x * inv, num.y * inv, num.z * inv, num.w * inv);
            if (p.o16) svcmi_store4_16(p.o16 + (long long)b * p.o16_bs + (long long)qrow * p.ldo16 + h * D + c4, p.ldo16 >> 1, num.x * inv, num.y * inv, num.z * inv, num.w * inv, p.o16_f16);
        }
    }
}

// Maximum over the four lane quarters (lanes with the same lane & 15), in every lane.  gfx950's row / half swaps are VALU operations
// (v_permlane16_swap / v_permlane32_swap: a few cycles) where the two ds_bpermute shuffles of __shfl_xor cross the LDS crossbar (~100 each).
__device__ __forceinline__ float quarter_max(float v) {
#ifdef SVCMI_EMU
    v = fmaxf(v, __shfl_xor(v, 16));
    return fmaxf(v, __shfl_xor(v, 32));
#else
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto h = __builtin_amdgcn_permlane32_swap(u, u, false, false);        // {lower half everywhere, upper half everywhere}
    const float m = fmaxf(__builtin_bit_cast(float, (unsigned)h[0]), __builtin_bit_cast(float, (unsigned)h[1]));
    const unsigned um = __builtin_bit_cast(unsigned, m);
    const auto r = __builtin_amdgcn_permlane16_swap(um, um, false, false);      // {even rows everywhere, odd rows everywhere}
    return fmaxf(__builtin_bit_cast(float, (unsigned)r[0]), __builtin_bit_cast(float, (unsigned)r[1]));
#endif
}

// Round 6: the same decomposition as attention_kernel with a 64-KEY step whose memory requests are all in flight at once.
// attention_kernel's 32-key step, as hipcc schedules it inside 68 registers, is six dependent L2 round trips (K in two halves, V two keys at a
// time, each pair waited for before the next is requested): 24 serialized round trips per 128-key wave -- the kernel is bound by that chain
// (24.6 us for Whisper's T = 500 window against 8.5 us of matrix-pipe time), not by issue slots.  Here a step requests its 16 K fragments
// together (4 key tiles x D / 16 16-byte loads), runs four independent QK^T accumulator chains, requests all 16 V rows BEFORE the softmax
// arithmetic (they fly during it), and a 128-key wave is two steps = four round trips.  ~150 registers: 3 waves per SIMD, enough for the
// 2.5 waves per SIMD a T = 500 window launches.  K and V go through range-checked buffer loads with 32-bit lane offsets (no 64-bit address
// arithmetic in the loop).  The per-step online-softmax bookkeeping (one rescale of O per 64 keys instead of per 32) halves as well.
// Probe builds only (scripts/build_variant.sh -DSVCMI_PROBE_ATTN=n; never the product): 1 = the MFMAs of the key loop replaced by one FMA each,
// 2 = no K / V requests (operands taken from registers), 3 = no key loop at all (prologue + publish + merge: the launch's fixed cost)
#ifndef SVCMI_PROBE_ATTN
#define SVCMI_PROBE_ATTN 0
#endif
#if SVCMI_PROBE_ATTN == 1
__device__ __forceinline__ svcmi_f32x4 attn_probe_mfma(float a, float b, svcmi_f32x4 c) { c[0] = fmaf(a, b, c[0]); return c; }
#define SVCMI_ATTN_MFMA attn_probe_mfma
#else
#define SVCMI_ATTN_MFMA svcmi_mfma_16x16x4
#endif
#ifndef SVCMI_EMU
#define SVCMI_ATTN_WIDE_OCC __attribute__((amdgpu_waves_per_eu(1, 3)))   // register budget of 3 waves per SIMD: hipcc must not serialize the requests to save registers
#else
#define SVCMI_ATTN_WIDE_OCC
#endif
template <int D, int NS, bool REL>
__global__ __launch_bounds__(64 * NS) SVCMI_ATTN_WIDE_OCC void attention_wide_kernel(AttnArgs p) {
    constexpr int DS = D / 16, OLD = D + 4, KT = 4, KW = 16 * KT;
    __shared__ __attribute__((aligned(16))) float smem[NS * 16 * OLD + 2 * NS * 16 + (REL ? NS * 16 * BST + 2 * NREL * D : 0)];
    float* const Opart = smem;                         // [NS][16][OLD]
    float* const Mpart = Opart + NS * 16 * OLD;        // [NS][16]
    float* const Lpart = Mpart + NS * 16;              // [NS][16]
    float* const Bpart = Lpart + NS * 16;              // [NS][16][BST]
    float* const Ek = Bpart + NS * 16 * BST;           // [NREL][D]
    float* const Ev = Ek + NREL * D;                   // [NREL][D]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = SVCMI_UNIFORM((int)(tid >> 6));
    const int lq = lane & 15, g4 = lane >> 4;
    int L;
    {
        const int total = (int)gridDim.x, id = (int)blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = id & 7, slot = id >> 3;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int qt = L % p.nq, hb = L / p.nq;
    const int h = hb % p.heads, b = hb / p.heads;
    const int T = p.t;
    const int len = p.lengths ? p.lengths[b] : T;
    const bool has_rel = REL && p.rel_k != nullptr;
    const int W = p.window;
    const int nrel = has_rel ? 2 * W + 1 : 0;
    const float scale2 = p.scale * LOG2E;

    if (has_rel) {
        for (int i = tid; i < nrel * D; i += 64 * NS) { Ek[i] = p.rel_k[i]; Ev[i] = p.rel_v[i]; }
        __syncthreads();
    }

    const int q0 = qt * 16, qi = q0 + lq;
    float qf[DS][4];
    {
        const float* qp = p.q + (long long)b * p.q_bs + (long long)(qi < T ? qi : T - 1) * p.ldq + h * D + 4 * g4;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float4 t4 = *reinterpret_cast<const float4*>(qp + 16 * s);
            qf[s][0] = t4.x; qf[s][1] = t4.y; qf[s][2] = t4.z; qf[s][3] = t4.w;
        }
    }
    float R[REL ? NREL : 1], Pb[REL ? NREL : 1];
    if constexpr (REL) {
#pragma unroll
        for (int e = 0; e < NREL; ++e) { R[e] = 0.f; Pb[e] = 0.f; }
        if (has_rel) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) {
                if (e < nrel) {           // wave-uniform
                    float a = 0.f;
#pragma unroll
                    for (int s = 0; s < DS; ++s)
#pragma unroll
                        for (int j = 0; j < 4; ++j) a = fmaf(qf[s][j], Ek[e * D + 16 * s + 4 * g4 + j], a);
                    R[e] = quarter_sum(a);
                }
            }
        }
    }

    svcmi_f32x4 oacc[DS];
#pragma unroll
    for (int dt = 0; dt < DS; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[dt][r] = 0.f;
    float mrun = NEG_BIG, lrun = 0.f;

    // one descriptor per operand: the head's column block of this batch item; lane offsets are row * ld (+ the lane's columns), in bytes
    // (svcmi_attention_f32 bounds t * ld * 4 below 2^31); rows are clamped to T - 1, so every request is inside the tensor
    const svcmi_brsrc rk = svcmi_make_brsrc(p.k + (long long)b * p.k_bs + h * D, 0x7fffffffu);
    const svcmi_brsrc rv = svcmi_make_brsrc(p.v + (long long)b * p.v_bs + h * D, 0x7fffffffu);
    const unsigned ldk4 = 4u * (unsigned)p.ldk, ldv4 = 4u * (unsigned)p.ldv;
    const unsigned kcol = 16u * (unsigned)g4;                             // K fragment: columns 4 g4 .. 4 g4 + 3 of each 16-wide d group
    const unsigned vcol = DS == 4 ? 16u * (unsigned)lq : (DS == 2 ? 8u * (unsigned)lq : 4u * (unsigned)lq);
    const int per = ((T + NS - 1) / NS + KW - 1) / KW * KW;              // keys per wave, a multiple of the 64-key step
    const int jbeg = w * per;
#if SVCMI_PROBE_ATTN == 3
    const int jend = jbeg;
#else
    const int jend = (jbeg + per) < T ? (jbeg + per) : T;
#endif

    for (int kt = jbeg; kt < jend; kt += KW) {
        // ---- every K fragment of the step is requested before the first MFMA
        svcmi_f32x4 kc[KT][DS];
#pragma unroll
        for (int u = 0; u < KT; ++u) {
            const int krow = kt + 16 * u + lq;
            const unsigned ko = (unsigned)(krow < T ? krow : T - 1) * ldk4 + kcol;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
#if SVCMI_PROBE_ATTN == 2
                kc[u][s][0] = qf[s][0] + (float)ko; kc[u][s][1] = qf[s][1]; kc[u][s][2] = qf[s][2]; kc[u][s][3] = qf[s][3];
#else
                kc[u][s] = svcmi_buf_load16(rk, ko + 64u * (unsigned)s);
#endif
            }
        }
        SVCMI_SCHED_BARRIER();
        // ---- S^T = K Q^T: four independent 16-key accumulator chains
        svcmi_f32x4 sacc[KT];
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[u][r] = 0.f;
#pragma unroll
        for (int s = 0; s < DS; ++s)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int u = 0; u < KT; ++u) sacc[u] = SVCMI_ATTN_MFMA(kc[u][s][j], qf[s][j], sacc[u]);
        SVCMI_SCHED_BARRIER();
        // ---- every V row of the step is requested now: the requests fly during the softmax arithmetic
        float vv[KT][4][DS];
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * u + 4 * g4 + r;
                const unsigned vo = (unsigned)(key < T ? key : T - 1) * ldv4 + vcol;
                if constexpr (DS == 4) {
#if SVCMI_PROBE_ATTN == 2
                    const svcmi_f32x4 t4 = {qf[0][0] + (float)vo, qf[1][1], qf[2][2], qf[3][3]};
#else
                    const svcmi_f32x4 t4 = svcmi_buf_load16(rv, vo);
#endif
                    vv[u][r][0] = t4[0]; vv[u][r][1] = t4[1]; vv[u][r][2] = t4[2]; vv[u][r][3] = t4[3];
                } else {
                    const float* vrow = p.v + (long long)b * p.v_bs + h * D + (long long)(key < T ? key : T - 1) * p.ldv;
                    load_vrow<DS>(vrow, lq, vv[u][r]);
                }
            }
        SVCMI_SCHED_BARRIER();
        // ---- scores of this lane: keys kt + 16u + 4*g4 + r, query qi
        const bool diag = has_rel && (kt + KW - 1 >= q0 - W) && (kt <= q0 + 15 + W);   // wave-uniform
        const bool clean = kt + KW <= (len < T ? len : T) && q0 + 16 <= len;            // wave-uniform: no masked or out-of-range entry
        float sv[KT][4];
        float mt = NEG_BIG;
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * u + 4 * g4 + r;
                float a = sacc[u][r];
                if constexpr (REL) {
                    if (diag) {
                        const int rel = key - qi + W;
                        float add = 0.f;
#pragma unroll
                        for (int e = 0; e < NREL; ++e) add = (rel == e && e < nrel) ? R[e] : add;
                        a += add;
                    }
                }
                a *= scale2;
                if (!clean) {
                    if (qi >= len || key >= len) a = MASKED2;   // masked_fill(mask == 0, -1e4)
                    if (key >= T) a = NEG_BIG;                  // beyond the sequence: weight 0
                }
                sv[u][r] = a;
                mt = fmaxf(mt, a);
            }
        mt = quarter_max(mt);
        const float mnew = fmaxf(mrun, mt);
        const float corr = svcmi_exp2(mrun - mnew);
        mrun = mnew;
        lrun *= corr;
#pragma unroll
        for (int dt = 0; dt < DS; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] *= corr;
        float pv[KT][4];
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[u][r] = svcmi_exp2(sv[u][r] - mnew);         // 2^(NEG_BIG - finite) == 0 for keys >= T
                lrun += pv[u][r];
            }
        if constexpr (REL) {
            if (has_rel) {
#pragma unroll
                for (int e = 0; e < NREL; ++e) Pb[e] *= corr;
            }
            if (diag) {
#pragma unroll
                for (int u = 0; u < KT; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rel = kt + 16 * u + 4 * g4 + r - qi + W;
#pragma unroll
                        for (int e = 0; e < NREL; ++e) Pb[e] += (rel == e && e < nrel) ? pv[u][r] : 0.f;
                    }
            }
        }
        // ---- O^T += V^T P^T : step (u, r) contracts keys kt + 16u + {r, 4+r, 8+r, 12+r}
#pragma unroll
        for (int u = 0; u < KT; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int dt = 0; dt < DS; ++dt) oacc[dt] = SVCMI_ATTN_MFMA(vv[u][r][dt], pv[u][r], oacc[dt]);
    }

    // ---- publish this wave's state and merge: as attention_kernel
    lrun = quarter_sum(lrun);
    if constexpr (REL) {
        if (has_rel) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) Pb[e] = quarter_sum(Pb[e]);
        }
    }
    {
        store_orow<DS>(Opart + (w * 16 + lq) * OLD, g4, oacc);
        if (g4 == 0) {
            Mpart[w * 16 + lq] = mrun;
            Lpart[w * 16 + lq] = lrun;
            if constexpr (REL) {
                if (has_rel) {
#pragma unroll
                    for (int e = 0; e < NREL; ++e) Bpart[(w * 16 + lq) * BST + e] = Pb[e];
                }
            }
        }
    }
    __syncthreads();

    float* ob = p.o + (long long)b * p.o_bs + h * D;
    for (int item = tid; item < 16 * (D / 4); item += 64 * NS) {
        const int ql = item / (D / 4), c4 = (item - ql * (D / 4)) * 4;
        float mall = Mpart[ql];
#pragma unroll
        for (int ww = 1; ww < NS; ++ww) mall = fmaxf(mall, Mpart[ww * 16 + ql]);
        float den = 0.f;
        float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
        float bs[REL ? NREL : 1];
        if constexpr (REL) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) bs[e] = 0.f;
        }
#pragma unroll
        for (int ww = 0; ww < NS; ++ww) {
            const float cw = svcmi_exp2(Mpart[ww * 16 + ql] - mall);
            den = fmaf(cw, Lpart[ww * 16 + ql], den);
            const float4 ov = *reinterpret_cast<const float4*>(Opart + (ww * 16 + ql) * OLD + c4);
            num.x = fmaf(cw, ov.x, num.x); num.y = fmaf(cw, ov.y, num.y);
            num.z = fmaf(cw, ov.z, num.z); num.w = fmaf(cw, ov.w, num.w);
            if constexpr (REL) {
                if (has_rel) {
#pragma unroll
                    for (int e = 0; e < NREL; ++e) bs[e] = fmaf(cw, Bpart[(ww * 16 + ql) * BST + e], bs[e]);
                }
            }
        }
        if constexpr (REL) {
            if (has_rel) {
#pragma unroll
                for (int e = 0; e < NREL; ++e) {
                    if (e < nrel) {
                        const float4 ev = *reinterpret_cast<const float4*>(Ev + e * D + c4);
                        num.x = fmaf(bs[e], ev.x, num.x); num.y = fmaf(bs[e], ev.y, num.y);
                        num.z = fmaf(bs[e], ev.z, num.z); num.w = fmaf(bs[e], ev.w, num.w);
                    }
                }
            }
        }
        const int qrow = q0 + ql;
        if (qrow < T) {
            const float inv = 1.0f / den;
            *reinterpret_cast<float4*>(ob + (long long)qrow * p.ldo + c4) = make_float4(num.x * inv, num.y * inv, num.z * inv, num.w * inv);
            if (p.o16) svcmi_store4_16(p.o16 + (long long)b * p.o16_bs + (long long)qrow * p.ldo16 + h * D + c4, p.ldo16 >> 1, num.x * inv, num.y * inv, num.z * inv, num.w * inv, p.o16_f16);
        }
    }
}

// Plain (no relative-position band) attention with TWO 16-query tiles per wave: every K / V fragment a wave fetches feeds
// twice the MFMAs (PMC: the one-tile kernel spends 47 % of its wave-cycles parked on memory at Whisper's T = 500 and
// keeps the matrix pipe 22 % busy), and the four QK accumulators of a step are independent.  Same key split / merge.
template <int D, int NS>
__global__ __launch_bounds__(64 * NS) void attention_q32_kernel(AttnArgs p) {
    constexpr int DS = D / 16, OLD = D + 4, QT = 2, QB = 16 * QT;
    __shared__ __attribute__((aligned(16))) float smem[NS * QB * OLD + 2 * NS * QB];
    float* const Opart = smem;                         // [NS][QB][OLD]
    float* const Mpart = Opart + NS * QB * OLD;        // [NS][QB]
    float* const Lpart = Mpart + NS * QB;              // [NS][QB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = SVCMI_UNIFORM((int)(tid >> 6));
    const int lq = lane & 15, g4 = lane >> 4;
    int L;
    {
        const int total = (int)gridDim.x, id = (int)blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = id & 7, slot = id >> 3;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int qt = L % p.nq, hb = L / p.nq;
    const int h = hb % p.heads, b = hb / p.heads;
    const int T = p.t;
    const int len = p.lengths ? p.lengths[b] : T;
    const int q0 = qt * QB;
    const float scale2 = p.scale * LOG2E;

    float qf[QT][DS][4];
#pragma unroll
    for (int a = 0; a < QT; ++a) {
        const int qi = q0 + 16 * a + lq;
        const float* qp = p.q + (long long)b * p.q_bs + (long long)(qi < T ? qi : T - 1) * p.ldq + h * D + 4 * g4;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float4 t4 = *reinterpret_cast<const float4*>(qp + 16 * s);
            qf[a][s][0] = t4.x; qf[a][s][1] = t4.y; qf[a][s][2] = t4.z; qf[a][s][3] = t4.w;
        }
    }
    svcmi_f32x4 oacc[QT][DS];
    float mrun[QT], lrun[QT];
#pragma unroll
    for (int a = 0; a < QT; ++a) {
        mrun[a] = NEG_BIG; lrun[a] = 0.f;
#pragma unroll
        for (int dt = 0; dt < DS; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[a][dt][r] = 0.f;
    }

    const float* kb = p.k + (long long)b * p.k_bs + h * D + 4 * g4;
    const float* vb = p.v + (long long)b * p.v_bs + h * D;
    const int per = ((T + NS - 1) / NS + 31) / 32 * 32;
    const int jbeg = w * per;
    const int jend = (jbeg + per) < T ? (jbeg + per) : T;

    for (int kt = jbeg; kt < jend; kt += 32) {
        svcmi_f32x4 sacc[QT][2];
#pragma unroll
        for (int a = 0; a < QT; ++a)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) sacc[a][u][r] = 0.f;
        {
            const int k0 = kt + lq, k1 = kt + 16 + lq;
            const float* kr0 = kb + (long long)(k0 < T ? k0 : T - 1) * p.ldk;
            const float* kr1 = kb + (long long)(k1 < T ? k1 : T - 1) * p.ldk;
#pragma unroll
            for (int s = 0; s < DS; ++s) {
                const float4 a0 = *reinterpret_cast<const float4*>(kr0 + 16 * s);
                const float4 a1 = *reinterpret_cast<const float4*>(kr1 + 16 * s);
                const float k0v[4] = {a0.x, a0.y, a0.z, a0.w}, k1v[4] = {a1.x, a1.y, a1.z, a1.w};
#pragma unroll
                for (int c = 0; c < 4; ++c)
#pragma unroll
                    for (int a = 0; a < QT; ++a) {
                        sacc[a][0] = svcmi_mfma_16x16x4(k0v[c], qf[a][s][c], sacc[a][0]);
                        sacc[a][1] = svcmi_mfma_16x16x4(k1v[c], qf[a][s][c], sacc[a][1]);
                    }
            }
        }
        const bool clean = kt + 32 <= (len < T ? len : T) && q0 + QB <= len;          // wave-uniform: nothing to mask in this step
        float pv[QT][2][4];
#pragma unroll
        for (int a = 0; a < QT; ++a) {
            const int qi = q0 + 16 * a + lq;
            float sv[2][4];
            float mt = NEG_BIG;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int key = kt + 16 * u + 4 * g4 + r;
                    float v = sacc[a][u][r] * scale2;               // log2 domain, as in attention_kernel
                    if (!clean) {
                        if (qi >= len || key >= len) v = MASKED2;   // masked_fill(mask == 0, -1e4)
                        if (key >= T) v = NEG_BIG;                  // beyond the sequence: weight 0
                    }
                    sv[u][r] = v;
                    mt = fmaxf(mt, v);
                }
            mt = fmaxf(mt, __shfl_xor(mt, 16));
            mt = fmaxf(mt, __shfl_xor(mt, 32));
            const float mnew = fmaxf(mrun[a], mt);
            const float corr = svcmi_exp2(mrun[a] - mnew);
            mrun[a] = mnew;
            lrun[a] *= corr;
#pragma unroll
            for (int dt = 0; dt < DS; ++dt)
#pragma unroll
                for (int r = 0; r < 4; ++r) oacc[a][dt][r] *= corr;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    pv[a][u][r] = svcmi_exp2(sv[u][r] - mnew);
                    lrun[a] += pv[a][u][r];
                }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * u + 4 * g4 + r;
                float vv[DS];
                load_vrow<DS>(vb + (long long)(key < T ? key : T - 1) * p.ldv, lq, vv);
#pragma unroll
                for (int dt = 0; dt < DS; ++dt)
#pragma unroll
                    for (int a = 0; a < QT; ++a) oacc[a][dt] = svcmi_mfma_16x16x4(vv[dt], pv[a][u][r], oacc[a][dt]);
            }
    }

#pragma unroll
    for (int a = 0; a < QT; ++a) {
        const float lsum = quarter_sum(lrun[a]);
        store_orow<DS>(Opart + (w * QB + 16 * a + lq) * OLD, g4, oacc[a]);
        if (g4 == 0) {
            Mpart[w * QB + 16 * a + lq] = mrun[a];
            Lpart[w * QB + 16 * a + lq] = lsum;
        }
    }
    __syncthreads();

    float* ob = p.o + (long long)b * p.o_bs + h * D;
    for (int item = tid; item < QB * (D / 4); item += 64 * NS) {
        const int ql = item / (D / 4), c4 = (item - ql * (D / 4)) * 4;
        float mall = Mpart[ql];
#pragma unroll
        for (int ww = 1; ww < NS; ++ww) mall = fmaxf(mall, Mpart[ww * QB + ql]);
        float den = 0.f;
        float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int ww = 0; ww < NS; ++ww) {
            const float cw = svcmi_exp2(Mpart[ww * QB + ql] - mall);
            den = fmaf(cw, Lpart[ww * QB + ql], den);
            const float4 ov = *reinterpret_cast<const float4*>(Opart + (ww * QB + ql) * OLD + c4);
            num.x = fmaf(cw, ov.x, num.x); num.y = fmaf(cw, ov.y, num.y);
            num.z = fmaf(cw, ov.z, num.z); num.w = fmaf(cw, ov.w, num.w);
        }
        const int qrow = q0 + ql;
        if (qrow < T) {
            const float inv = 1.0f / den;
            *reinterpret_cast<float4*>(ob + (long long)qrow * p.ldo + c4) = make_float4(num.x * inv, num.y * inv, num.z * inv, num.w * inv);
            if (p.o16) svcmi_store4_16(p.o16 + (long long)b * p.o16_bs + (long long)qrow * p.ldo16 + h * D + c4, p.ldo16 >> 1, num.x * inv, num.y * inv, num.z * inv, num.w * inv, p.o16_f16);
        }
    }
}

// Band-free attention with the K / V tiles of a head staged ONCE per block through LDS and shared by QT query tiles (VERDICT r1 item 6;
// DESIGN.md section 8.1).  One block = QT x KS waves over QT * 16 queries of one head: wave (qt, ks) owns query tile qt and key range ks;
// per 32-key step the QT waves of a key range fetch that range's next K and V tiles (2 x 8 KB at D = 64) with coalesced float4 loads into
// registers, the block computes on the current tiles out of LDS (K fragments as ds_read_b128, V fragments as 32-bit reads along d), then
// the fetched tiles go to the other LDS buffer and ONE block-wide barrier closes the step.  L2 -> CU traffic per launch drops by QT (32
// FLOP per fetched byte at QT = 4 instead of 8); the KS partial states of a query tile merge through LDS as in attention_kernel.  Opt-in
// ("attn_lds" tuning knob) until it has been measured on hardware: with QT = 4 a T = 500 x 20-head launch is only 160 blocks.
// REL (round 4): the relative-position band of the prior encoder (vits/attentions.py:225-347) in the same decomposition, as attention_kernel
// and attention16_kernel carry it: R[e] = q . E_k[e] per query, added to the in-band scores of diagonal steps; Pb[e] collects the in-band
// probabilities so that sum_e Pb[e] E_v[e] is added in the merge.
template <int D, int QT, int KS, bool REL = false>
__global__ __launch_bounds__(64 * QT * KS) void attention_lds_kernel(AttnArgs p) {
    constexpr int DS = D / 16, OLD = D + 4, NW = QT * KS, NT = 64 * NW, QB = 16 * QT;
    constexpr int TLD = D + 4;                          // row stride of a staged tile: float4-aligned, rows 4 banks apart
    constexpr int TILE = 32 * TLD;                      // one 32-key tile (K or V)
    constexpr int STAGE = KS * 2 * 2 * TILE;            // [ks][buffer][K | V]
    constexpr int MERGE = NW * 16 * OLD + 2 * NW * 16 + (REL ? NW * 16 * BST : 0);  // partial O / max / sum (/ band sums) of every wave (aliases the tiles after the last step)
    constexpr int MAIN = STAGE > MERGE ? STAGE : MERGE;
    __shared__ __attribute__((aligned(16))) float smem[MAIN + (REL ? 2 * NREL * D : 0)];
    float* const Ek = smem + MAIN;                      // [NREL][D]  (REL only; never aliased by the tiles / the merge state)
    float* const Ev = Ek + NREL * D;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = SVCMI_UNIFORM((int)(tid >> 6));
    const int qt_l = w % QT, ks = w / QT;               // wave-uniform
    const int lq = lane & 15, g4 = lane >> 4;
    int L;
    {
        const int total = (int)gridDim.x, id = (int)blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = id & 7, slot = id >> 3;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int qg = L % p.nq, hb = L / p.nq;             // p.nq = query GROUPS of QB rows
    const int h = hb % p.heads, b = hb / p.heads;
    const int T = p.t;
    const int len = p.lengths ? p.lengths[b] : T;
    const float scale2 = p.scale * LOG2E;
    const int q0 = qg * QB + 16 * qt_l, qi = q0 + lq;
    const int W = REL ? p.window : 0;
    const int nrel = REL ? 2 * W + 1 : 0;

    float qf[DS][4];
    {
        const float* qp = p.q + (long long)b * p.q_bs + (long long)(qi < T ? qi : T - 1) * p.ldq + h * D + 4 * g4;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float4 t4 = *reinterpret_cast<const float4*>(qp + 16 * s);
            qf[s][0] = t4.x; qf[s][1] = t4.y; qf[s][2] = t4.z; qf[s][3] = t4.w;
        }
    }
    float R[REL ? NREL : 1], Pb[REL ? NREL : 1];
    if constexpr (REL) {
        for (int i = tid; i < nrel * D; i += NT) { Ek[i] = p.rel_k[i]; Ev[i] = p.rel_v[i]; }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NREL; ++e) {
            R[e] = 0.f; Pb[e] = 0.f;
            if (e < nrel) {           // wave-uniform
                float a = 0.f;
#pragma unroll
                for (int s = 0; s < DS; ++s)
#pragma unroll
                    for (int j = 0; j < 4; ++j) a = fmaf(qf[s][j], Ek[e * D + 16 * s + 4 * g4 + j], a);
                R[e] = quarter_sum(a);
            }
        }
    }
    svcmi_f32x4 oacc[DS];
#pragma unroll
    for (int dt = 0; dt < DS; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[dt][r] = 0.f;
    float mrun = NEG_BIG, lrun = 0.f;

    const int per = ((T + KS - 1) / KS + 31) / 32 * 32;      // keys per key range, a multiple of the step
    const int steps = per / 32;                              // every range runs the same number of steps (rows past T are masked)
    const int jbeg = ks * per;
    // staging role of this thread inside its key range: QT * 64 threads move 32 rows x D/4 float4 of K and of V per step
    constexpr int F4 = D / 4, PER_T = (32 * F4 + 64 * QT - 1) / (64 * QT);
    const int st = qt_l * 64 + lane;                         // 0 .. 64 * QT - 1
    const float* kg = p.k + (long long)b * p.k_bs + h * D;
    const float* vg = p.v + (long long)b * p.v_bs + h * D;
    float* const Kt = smem + ks * 4 * TILE;                  // [buffer][K | V] of this key range
    float4 kreg[PER_T], vreg[PER_T];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int j = 0; j < PER_T; ++j) {
            const int idx = st + j * 64 * QT;
            if (idx < 32 * F4) {
                const int row = idx / F4, c4 = idx - row * F4;
                const int key = kt + row;
                const long long off = (long long)(key < T ? key : T - 1);
                kreg[j] = *reinterpret_cast<const float4*>(kg + off * p.ldk + 4 * c4);
                vreg[j] = *reinterpret_cast<const float4*>(vg + off * p.ldv + 4 * c4);
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int j = 0; j < PER_T; ++j) {
            const int idx = st + j * 64 * QT;
            if (idx < 32 * F4) {
                const int row = idx / F4, c4 = idx - row * F4;
                *reinterpret_cast<float4*>(Kt + (buf * 2 + 0) * TILE + row * TLD + 4 * c4) = kreg[j];
                *reinterpret_cast<float4*>(Kt + (buf * 2 + 1) * TILE + row * TLD + 4 * c4) = vreg[j];
            }
        }
    };
    fetch(jbeg);
    stash(0);
    __syncthreads();
    for (int it = 0; it < steps; ++it) {
        const int kt = jbeg + 32 * it, buf = it & 1;
        if (it + 1 < steps) fetch(kt + 32);                 // uniform over the block
        const float* Kc = Kt + (buf * 2 + 0) * TILE;
        const float* Vc = Kt + (buf * 2 + 1) * TILE;
        svcmi_f32x4 sacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[u][r] = 0.f;
#pragma unroll
        for (int s = 0; s < DS; ++s) {
            const float4 a0 = *reinterpret_cast<const float4*>(Kc + lq * TLD + 16 * s + 4 * g4);
            const float4 a1 = *reinterpret_cast<const float4*>(Kc + (16 + lq) * TLD + 16 * s + 4 * g4);
            sacc[0] = svcmi_mfma_16x16x4(a0.x, qf[s][0], sacc[0]);
            sacc[1] = svcmi_mfma_16x16x4(a1.x, qf[s][0], sacc[1]);
            sacc[0] = svcmi_mfma_16x16x4(a0.y, qf[s][1], sacc[0]);
            sacc[1] = svcmi_mfma_16x16x4(a1.y, qf[s][1], sacc[1]);
            sacc[0] = svcmi_mfma_16x16x4(a0.z, qf[s][2], sacc[0]);
            sacc[1] = svcmi_mfma_16x16x4(a1.z, qf[s][2], sacc[1]);
            sacc[0] = svcmi_mfma_16x16x4(a0.w, qf[s][3], sacc[0]);
            sacc[1] = svcmi_mfma_16x16x4(a1.w, qf[s][3], sacc[1]);
        }
        const bool clean = kt + 32 <= (len < T ? len : T) && q0 + 16 <= len;     // wave-uniform
        const bool diag = REL && (kt + 31 >= q0 - W) && (kt <= q0 + 15 + W);       // wave-uniform: this step touches the band
        float sv[2][4];
        float mt = NEG_BIG;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * u + 4 * g4 + r;
                float a = sacc[u][r];
                if constexpr (REL) {
                    if (diag) {
                        const int rel = key - qi + W;
                        float add = 0.f;
#pragma unroll
                        for (int e = 0; e < NREL; ++e) add = (rel == e && e < nrel) ? R[e] : add;
                        a += add;
                    }
                }
                a *= scale2;
                if (!clean) {
                    if (qi >= len || key >= len) a = MASKED2;       // masked_fill(mask == 0, -1e4)
                    if (key >= T) a = NEG_BIG;                      // beyond the sequence: weight 0
                }
                sv[u][r] = a;
                mt = fmaxf(mt, a);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mnew = fmaxf(mrun, mt);
        // a key range that lies entirely past T has only NEG_BIG scores: keep its state empty (2^(NEG_BIG - NEG_BIG) would be 1)
        const float corr = mnew > -1.0e38f ? svcmi_exp2(mrun - mnew) : 0.f;
        mrun = mnew;
        lrun *= corr;
#pragma unroll
        for (int dt = 0; dt < DS; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] *= corr;
        float pv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[u][r] = mnew > -1.0e38f ? svcmi_exp2(sv[u][r] - mnew) : 0.f;
                lrun += pv[u][r];
            }
        if constexpr (REL) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) Pb[e] *= corr;
            if (diag) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rel = kt + 16 * u + 4 * g4 + r - qi + W;
#pragma unroll
                        for (int e = 0; e < NREL; ++e) Pb[e] += (rel == e && e < nrel) ? pv[u][r] : 0.f;
                    }
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float* vr = Vc + (16 * u + 4 * g4 + r) * TLD + lq;
#pragma unroll
                for (int dt = 0; dt < DS; ++dt) oacc[dt] = svcmi_mfma_16x16x4(vr[16 * dt], pv[u][r], oacc[dt]);
            }
        if (it + 1 < steps) stash(buf ^ 1);
        __syncthreads();                                    // next tiles visible; this step's tiles free for the step after next
    }

    // ---- merge the KS partial states of every query tile (the tiles are dead: the barrier above closed the last step)
    float* const Opart = smem;                              // [NW][16][OLD]
    float* const Mpart = Opart + NW * 16 * OLD;             // [NW][16]
    float* const Lpart = Mpart + NW * 16;                   // [NW][16]
    float* const Bpart = Lpart + NW * 16;                   // [NW][16][BST]  (REL)
    lrun = quarter_sum(lrun);
    if constexpr (REL) {
#pragma unroll
        for (int e = 0; e < NREL; ++e) Pb[e] = quarter_sum(Pb[e]);
    }
    {
        float* orow = Opart + (w * 16 + lq) * OLD + 4 * g4;
#pragma unroll
        for (int dt = 0; dt < DS; ++dt)
            *reinterpret_cast<float4*>(orow + 16 * dt) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
        if (g4 == 0) {
            Mpart[w * 16 + lq] = mrun;
            Lpart[w * 16 + lq] = lrun;
            if constexpr (REL) {
#pragma unroll
                for (int e = 0; e < NREL; ++e) Bpart[(w * 16 + lq) * BST + e] = Pb[e];
            }
        }
    }
    __syncthreads();
    float* ob = p.o + (long long)b * p.o_bs + h * D;
    for (int item = tid; item < QB * (D / 4); item += NT) {
        const int qr = item / (D / 4), c4 = (item - qr * (D / 4)) * 4;      // qr: row inside the query group
        const int qtl = qr >> 4, ql = qr & 15;
        float mall = NEG_BIG;
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) mall = fmaxf(mall, Mpart[(k2 * QT + qtl) * 16 + ql]);
        float den = 0.f;
        float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
        float bs[REL ? NREL : 1];
#pragma unroll
        for (int e = 0; e < (REL ? NREL : 1); ++e) bs[e] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) {
            const int ww = k2 * QT + qtl;
            const float mw = Mpart[ww * 16 + ql];
            const float cw = mw > -1.0e38f ? svcmi_exp2(mw - mall) : 0.f;
            den = fmaf(cw, Lpart[ww * 16 + ql], den);
            const float4 ov = *reinterpret_cast<const float4*>(Opart + (ww * 16 + ql) * OLD + c4);
            num.x = fmaf(cw, ov.x, num.x); num.y = fmaf(cw, ov.y, num.y);
            num.z = fmaf(cw, ov.z, num.z); num.w = fmaf(cw, ov.w, num.w);
            if constexpr (REL) {
#pragma unroll
                for (int e = 0; e < NREL; ++e) bs[e] = fmaf(cw, Bpart[(ww * 16 + ql) * BST + e], bs[e]);
            }
        }
        if constexpr (REL) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) {
                if (e < nrel) {
                    const float4 ev = *reinterpret_cast<const float4*>(Ev + e * D + c4);
                    num.x = fmaf(bs[e], ev.x, num.x); num.y = fmaf(bs[e], ev.y, num.y);
                    num.z = fmaf(bs[e], ev.z, num.z); num.w = fmaf(bs[e], ev.w, num.w);
                }
            }
        }
        const int qrow = qg * QB + qr;
        if (qrow < T) {
            const float inv = 1.0f / den;
            *reinterpret_cast<float4*>(ob + (long long)qrow * p.ldo + c4) = make_float4(num.x * inv, num.y * inv, num.z * inv, num.w * inv);
            if (p.o16) svcmi_store4_16(p.o16 + (long long)b * p.o16_bs + (long long)qrow * p.ldo16 + h * D + c4, p.ldo16 >> 1, num.x * inv, num.y * inv, num.z * inv, num.w * inv, p.o16_f16);
        }
    }
}

// ------------------------------------------------------------------------------------ attention on the 16-bit matrix cores
// The bf16 / f16 modes (reference: `.half()` on an accelerator, whisper/inference.py:22-23): the same decomposition as attention_lds_kernel
// -- QT query tiles x KS key ranges per block, K / V tiles of a head staged once per block through LDS, both products swapped so that a lane
// owns one query -- with Q / K / V arriving as 16-bit tensors (the QKV GEMM's 16-bit output copy) and both products on
// v_mfma_f32_16x16x32_{bf16,f16}: 4 + 4 matrix instructions per 32-key step and wave instead of 32 + 32 exact-fp32 ones.  Softmax
// statistics, the online rescale and the accumulators stay fp32; P is rounded to 16 bits right before the PV product.
//   S^T tile: A = K rows (lane (key, g): 8 consecutive d = one ds_read_b128 from the [32][D] tile, rows padded to D * 2 + 16 bytes),
//             B = Q^T (lane (query, g): 8 consecutive d, kept in registers);
//   O^T tile: B = P^T -- the MFMA's contraction slot (g, e) is DEFINED as key 4g + e (e < 4) resp. 16 + 4g + e - 4, which is exactly the
//             set of keys whose probabilities the lane already holds from the first product: P never moves between lanes;
//             A = V^T in the same slot order: the V tile is stored TRANSPOSED in LDS as [key group of 4][d][4 keys] (16-bit), so the lane
//             (d, g) reads two 8-byte units (key groups g and 4 + g).  The transposition happens in the staging writes (ds_write_b16).
struct Attn16Args {
    const unsigned short* q; const unsigned short* k; const unsigned short* v;
    float* o; unsigned short* o16;
    int ld16, ldo, ldo16;
    long long bs16, o_bs, o16_bs;
    int t, heads, nq;
    float scale;
    const int32_t* lengths;
    int o16_f16;
    const float* rel_k; const float* rel_v;      // REL instantiations: the relative-position band of vits/attentions.py:225-347 (fp32 tables)
    int window;
};

template <bool F16>
__device__ __forceinline__ float h16_to_f32(unsigned h) {       // low 16 bits of h
#ifdef SVCMI_EMU
    return F16 ? emu_f16_f32(h & 0xffffu) : emu_bf16_f32(h & 0xffffu);
#else
    if constexpr (F16) return (float)__builtin_bit_cast(_Float16, (unsigned short)(h & 0xffffu));
    else return svcmi_bits_f32(h << 16);
#endif
}

template <int D, int QT, int KS, bool F16, bool REL = false>
__global__ __launch_bounds__(64 * QT * KS) void attention16_kernel(Attn16Args p) {
    constexpr int DS = D / 16, DK = D / 32, OLD = D + 4, NW = QT * KS, NT = 64 * NW, QB = 16 * QT;
    constexpr int KLD = D / 2 + 4;                      // K tile row stride in floats: D 16-bit values + 16 bytes
    constexpr int KT = 32 * KLD;                        // floats per K tile
    constexpr int VGS = (D + 16) * 2;                   // floats per key group of the transposed V tile: (D + 16) units of 4 keys x 2 bytes
    constexpr int VT = 8 * VGS;                         // floats per V tile (8 key groups)
    constexpr int STAGE = KS * 2 * (KT + VT);           // [ks][buffer][K | V]
    constexpr int MERGE = NW * 16 * OLD + 2 * NW * 16 + (REL ? NW * 16 * BST : 0);
    constexpr int MAIN = STAGE > MERGE ? STAGE : MERGE;
    __shared__ __attribute__((aligned(16))) float smem[MAIN + (REL ? 2 * NREL * D : 0)];
    float* const Ek = smem + MAIN;                      // [NREL][D]  (REL only; never aliased by the tiles / the merge state)
    float* const Ev = Ek + NREL * D;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = SVCMI_UNIFORM((int)(tid >> 6));
    const int qt_l = w % QT, ks = w / QT;
    const int lq = lane & 15, g4 = lane >> 4;
    int L;
    {
        const int total = (int)gridDim.x, id = (int)blockIdx.x;
        const int q8 = total >> 3, r8 = total & 7, xcd = id & 7, slot = id >> 3;
        L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    }
    const int qg = L % p.nq, hb = L / p.nq;
    const int h = hb % p.heads, b = hb / p.heads;
    const int T = p.t;
    const int len = p.lengths ? p.lengths[b] : T;
    const float scale2 = p.scale * LOG2E;
    const int q0 = qg * QB + 16 * qt_l, qi = q0 + lq;
    const int W = REL ? p.window : 0;
    const int nrel = REL ? 2 * W + 1 : 0;

    svcmi_u32x4 qf[DK];
    {
        const unsigned short* qp = p.q + (long long)b * p.bs16 + (long long)(qi < T ? qi : T - 1) * p.ld16 + h * D + 8 * g4;
#pragma unroll
        for (int s = 0; s < DK; ++s) qf[s] = *reinterpret_cast<const svcmi_u32x4*>(qp + 32 * s);
    }
    // relative-position band (vits/attentions.py:225-347): R[e] = q . E_k[e] per query (from the 16-bit q the matrix cores see), added to
    // the <= 2W + 1 in-band scores of diagonal tiles; Pb[e] collects the in-band probabilities so that sum_e Pb[e] E_v[e] is added in the merge
    float R[REL ? NREL : 1], Pb[REL ? NREL : 1];
    if constexpr (REL) {
        for (int i = tid; i < nrel * D; i += NT) { Ek[i] = p.rel_k[i]; Ev[i] = p.rel_v[i]; }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < NREL; ++e) {
            R[e] = 0.f; Pb[e] = 0.f;
            if (e < nrel) {           // wave-uniform
                float a = 0.f;
#pragma unroll
                for (int s2 = 0; s2 < DK; ++s2)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int d = 8 * g4 + 32 * s2 + 2 * j;
                        a = fmaf(h16_to_f32<F16>(qf[s2][j]), Ek[e * D + d], a);
                        a = fmaf(h16_to_f32<F16>(qf[s2][j] >> 16), Ek[e * D + d + 1], a);
                    }
                R[e] = quarter_sum(a);
            }
        }
    }
    svcmi_f32x4 oacc[DS];
#pragma unroll
    for (int dt = 0; dt < DS; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) oacc[dt][r] = 0.f;
    float mrun = NEG_BIG, lrun = 0.f;

    const int per = ((T + KS - 1) / KS + 31) / 32 * 32;
    const int steps = per / 32;
    const int jbeg = ks * per;
    // staging role inside the key range: QT * 64 threads move 32 rows x D/8 16-byte chunks of K per step, and the V tile as TASKS of
    // (key group of 4 rows, chunk of UD d-values): a task loads its 4 rows' chunk, transposes 4 x UD 16-bit values in registers (one
    // v_perm_b32 per output dword) and writes UD units [d][4 keys] as UD / 2 ds_write_b128 -- the round-3 form wrote every 16-bit value
    // with its own ds_write_b16 (32 per chunk; PMC: LDS bank-conflict share 0.44-0.65 of the kernel's LDS cycles)
    constexpr int NTS = 64 * QT;                         // staging threads of a key range
    constexpr int C8 = D / 8, CH = 32 * C8, PER_K = (CH + NTS - 1) / NTS;
    constexpr int VSPLIT = QT >= 4 ? 2 : 1;              // QT >= 4: half chunks (8-byte loads), so that 8 * C8 * 2 tasks spread over more threads
    constexpr int UD = 8 / VSPLIT, VW = UD / 2;          // d-values / dwords per row of a task
    constexpr int NV = 8 * C8 * VSPLIT, PER_V = (NV + NTS - 1) / NTS;
    const int st = qt_l * 64 + lane;
    const unsigned short* kg_ = p.k + (long long)b * p.bs16 + h * D;
    const unsigned short* vg_ = p.v + (long long)b * p.bs16 + h * D;
    float* const Kt = smem + ks * 2 * (KT + VT);            // [buffer][K | V] of this key range
    svcmi_u32x4 kreg[PER_K];
    unsigned vreg[PER_V][4][VW];
    auto fetch = [&](int kt) {
#pragma unroll
        for (int j = 0; j < PER_K; ++j) {
            const int idx = st + j * NTS;
            if (idx < CH) {
                const int row = idx / C8, c8 = idx - row * C8;
                const int key = kt + row;
                kreg[j] = *reinterpret_cast<const svcmi_u32x4*>(kg_ + (long long)(key < T ? key : T - 1) * p.ld16 + 8 * c8);
            }
        }
#pragma unroll
        for (int j = 0; j < PER_V; ++j) {
            const int idx = st + j * NTS;
            if (idx < NV) {
                const int kq = idx / (C8 * VSPLIT), cu = idx - kq * (C8 * VSPLIT);      // key group (4 rows), chunk of UD d-values
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int key = kt + 4 * kq + i;
                    const unsigned short* src = vg_ + (long long)(key < T ? key : T - 1) * p.ld16 + UD * cu;
                    if constexpr (VSPLIT == 1) {
                        const svcmi_u32x4 t4 = *reinterpret_cast<const svcmi_u32x4*>(src);
                        vreg[j][i][0] = t4[0]; vreg[j][i][1] = t4[1]; vreg[j][i][2] = t4[2]; vreg[j][i][3] = t4[3];
                    } else {
                        const svcmi_u32x2 t2 = *reinterpret_cast<const svcmi_u32x2*>(src);
                        vreg[j][i][0] = t2[0]; vreg[j][i][1] = t2[1];
                    }
                }
            }
        }
    };
    auto stash = [&](int buf) {
        float* const Kc = Kt + buf * (KT + VT);
        unsigned short* const Vc = reinterpret_cast<unsigned short*>(Kc + KT);
#pragma unroll
        for (int j = 0; j < PER_K; ++j) {
            const int idx = st + j * NTS;
            if (idx < CH) {
                const int row = idx / C8, c8 = idx - row * C8;
                *reinterpret_cast<svcmi_u32x4*>(Kc + row * KLD + 4 * c8) = kreg[j];
            }
        }
#pragma unroll
        for (int j = 0; j < PER_V; ++j) {
            const int idx = st + j * NTS;
            if (idx < NV) {
                const int kq = idx / (C8 * VSPLIT), cu = idx - kq * (C8 * VSPLIT);
                unsigned short* dst = Vc + (kq * (D + 16) + UD * cu) * 4;      // unit (key group kq, d = UD * cu): 4 keys x 2 bytes, units of consecutive d adjacent
#pragma unroll
                for (int wd = 0; wd < VW; ++wd) {          // dword wd of the rows holds d = 2 wd (low half) and 2 wd + 1 (high half): two units = 16 bytes
                    svcmi_u32x4 o;
                    o[0] = svcmi_pack_lo16(vreg[j][0][wd], vreg[j][1][wd]);
                    o[1] = svcmi_pack_lo16(vreg[j][2][wd], vreg[j][3][wd]);
                    o[2] = svcmi_pack_hi16(vreg[j][0][wd], vreg[j][1][wd]);
                    o[3] = svcmi_pack_hi16(vreg[j][2][wd], vreg[j][3][wd]);
                    *reinterpret_cast<svcmi_u32x4*>(dst + 8 * wd) = o;
                }
            }
        }
    };
    fetch(jbeg);
    stash(0);
    __syncthreads();
    for (int it = 0; it < steps; ++it) {
        const int kt = jbeg + 32 * it, buf = it & 1;
        if (it + 1 < steps) fetch(kt + 32);
        const float* Kc = Kt + buf * (KT + VT);
        const unsigned short* Vc = reinterpret_cast<const unsigned short*>(Kc + KT);
        svcmi_f32x4 sacc[2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) sacc[u][r] = 0.f;
#pragma unroll
        for (int s = 0; s < DK; ++s) {
            const svcmi_u32x4 a0 = *reinterpret_cast<const svcmi_u32x4*>(Kc + lq * KLD + 4 * g4 + 16 * s);
            const svcmi_u32x4 a1 = *reinterpret_cast<const svcmi_u32x4*>(Kc + (16 + lq) * KLD + 4 * g4 + 16 * s);
            sacc[0] = svcmi_mfma16_16x16x32<F16>(a0, qf[s], sacc[0]);
            sacc[1] = svcmi_mfma16_16x16x32<F16>(a1, qf[s], sacc[1]);
        }
        const bool clean = kt + 32 <= (len < T ? len : T) && q0 + 16 <= len;     // wave-uniform
        const bool diag = REL && (kt + 31 >= q0 - W) && (kt <= q0 + 15 + W);       // wave-uniform: this step touches the band
        float sv[2][4];
        float mt = NEG_BIG;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt + 16 * u + 4 * g4 + r;
                float a = sacc[u][r];
                if constexpr (REL) {
                    if (diag) {
                        const int rel = key - qi + W;
                        float add = 0.f;
#pragma unroll
                        for (int e = 0; e < NREL; ++e) add = (rel == e && e < nrel) ? R[e] : add;
                        a += add;
                    }
                }
                a *= scale2;
                if (!clean) {
                    if (qi >= len || key >= len) a = MASKED2;       // masked_fill(mask == 0, -1e4)
                    if (key >= T) a = NEG_BIG;                      // beyond the sequence: weight 0
                }
                sv[u][r] = a;
                mt = fmaxf(mt, a);
            }
        mt = fmaxf(mt, __shfl_xor(mt, 16));
        mt = fmaxf(mt, __shfl_xor(mt, 32));
        const float mnew = fmaxf(mrun, mt);
        const float corr = mnew > -1.0e38f ? svcmi_exp2(mrun - mnew) : 0.f;
        mrun = mnew;
        lrun *= corr;
#pragma unroll
        for (int dt = 0; dt < DS; ++dt)
#pragma unroll
            for (int r = 0; r < 4; ++r) oacc[dt][r] *= corr;
        float pv[2][4];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pv[u][r] = mnew > -1.0e38f ? svcmi_exp2(sv[u][r] - mnew) : 0.f;
                lrun += pv[u][r];
            }
        if constexpr (REL) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) Pb[e] *= corr;
            if (diag) {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int rel = kt + 16 * u + 4 * g4 + r - qi + W;
#pragma unroll
                        for (int e = 0; e < NREL; ++e) Pb[e] += (rel == e && e < nrel) ? pv[u][r] : 0.f;
                    }
            }
        }
        svcmi_u32x4 pf;         // contraction slot (g, e): e < 4 -> key 4g + e of the first 16-key tile, else key 4g + e - 4 of the second
        pf[0] = F16 ? svcmi_cvt_pk_f16(pv[0][0], pv[0][1]) : svcmi_cvt_pk_bf16(pv[0][0], pv[0][1]);
        pf[1] = F16 ? svcmi_cvt_pk_f16(pv[0][2], pv[0][3]) : svcmi_cvt_pk_bf16(pv[0][2], pv[0][3]);
        pf[2] = F16 ? svcmi_cvt_pk_f16(pv[1][0], pv[1][1]) : svcmi_cvt_pk_bf16(pv[1][0], pv[1][1]);
        pf[3] = F16 ? svcmi_cvt_pk_f16(pv[1][2], pv[1][3]) : svcmi_cvt_pk_bf16(pv[1][2], pv[1][3]);
#pragma unroll
        for (int dt = 0; dt < DS; ++dt) {
            const int d = 16 * dt + lq;
            const svcmi_u32x2 lo = *reinterpret_cast<const svcmi_u32x2*>(Vc + (g4 * (D + 16) + d) * 4);
            const svcmi_u32x2 hi = *reinterpret_cast<const svcmi_u32x2*>(Vc + ((4 + g4) * (D + 16) + d) * 4);
            svcmi_u32x4 a;
            a[0] = lo[0]; a[1] = lo[1]; a[2] = hi[0]; a[3] = hi[1];
            oacc[dt] = svcmi_mfma16_16x16x32<F16>(a, pf, oacc[dt]);
        }
        if (it + 1 < steps) stash(buf ^ 1);
        __syncthreads();
    }

    float* const Opart = smem;
    float* const Mpart = Opart + NW * 16 * OLD;
    float* const Lpart = Mpart + NW * 16;
    float* const Bpart = Lpart + NW * 16;                  // [NW][16][BST]  (REL)
    lrun = quarter_sum(lrun);
    if constexpr (REL) {
#pragma unroll
        for (int e = 0; e < NREL; ++e) Pb[e] = quarter_sum(Pb[e]);
    }
    {
        float* orow = Opart + (w * 16 + lq) * OLD + 4 * g4;
#pragma unroll
        for (int dt = 0; dt < DS; ++dt)
            *reinterpret_cast<float4*>(orow + 16 * dt) = make_float4(oacc[dt][0], oacc[dt][1], oacc[dt][2], oacc[dt][3]);
        if (g4 == 0) {
            Mpart[w * 16 + lq] = mrun;
            Lpart[w * 16 + lq] = lrun;
            if constexpr (REL) {
#pragma unroll
                for (int e = 0; e < NREL; ++e) Bpart[(w * 16 + lq) * BST + e] = Pb[e];
            }
        }
    }
    __syncthreads();
    for (int item = tid; item < QB * (D / 4); item += NT) {
        const int qr = item / (D / 4), c4 = (item - qr * (D / 4)) * 4;
        const int qtl = qr >> 4, ql = qr & 15;
        float mall = NEG_BIG;
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) mall = fmaxf(mall, Mpart[(k2 * QT + qtl) * 16 + ql]);
        float den = 0.f;
        float4 num = make_float4(0.f, 0.f, 0.f, 0.f);
        float bs[REL ? NREL : 1];
#pragma unroll
        for (int e = 0; e < (REL ? NREL : 1); ++e) bs[e] = 0.f;
#pragma unroll
        for (int k2 = 0; k2 < KS; ++k2) {
            const int ww = k2 * QT + qtl;
            const float mw = Mpart[ww * 16 + ql];
            const float cw = mw > -1.0e38f ? svcmi_exp2(mw - mall) : 0.f;
            den = fmaf(cw, Lpart[ww * 16 + ql], den);
            const float4 ov = *reinterpret_cast<const float4*>(Opart + (ww * 16 + ql) * OLD + c4);
            num.x = fmaf(cw, ov.x, num.x); num.y = fmaf(cw, ov.y, num.y);
            num.z = fmaf(cw, ov.z, num.z); num.w = fmaf(cw, ov.w, num.w);
            if constexpr (REL) {
#pragma unroll
                for (int e = 0; e < NREL; ++e) bs[e] = fmaf(cw, Bpart[(ww * 16 + ql) * BST + e], bs[e]);
            }
        }
        if constexpr (REL) {
#pragma unroll
            for (int e = 0; e < NREL; ++e) {
                if (e < nrel) {
                    const float4 ev = *reinterpret_cast<const float4*>(Ev + e * D + c4);
                    num.x = fmaf(bs[e], ev.x, num.x); num.y = fmaf(bs[e], ev.y, num.y);
                    num.z = fmaf(bs[e], ev.z, num.z); num.w = fmaf(bs[e], ev.w, num.w);
                }
            }
        }
        const int qrow = qg * QB + qr;
        if (qrow < T) {
            const float inv = 1.0f / den;
            if (p.o) *reinterpret_cast<float4*>(p.o + (long long)b * p.o_bs + (long long)qrow * p.ldo + h * D + c4) = make_float4(num.x * inv, num.y * inv, num.z * inv, num.w * inv);
            if (p.o16) svcmi_store4_16(p.o16 + (long long)b * p.o16_bs + (long long)qrow * p.ldo16 + h * D + c4, p.ldo16 >> 1, num.x * inv, num.y * inv, num.z * inv, num.w * inv, p.o16_f16);
        }
    }
}

int g_attn16 = 0;       // tuning knob ("attn16", 0 = heuristic | 10 * QT + KS): block shape of attention16_kernel

// the relative-position variant (prior encoder: 2 heads x 96): few (head, query group) pairs, so the shapes trade query tiles for key ranges
template <int D, bool F16>
int launch_attn16_rel(const Attn16Args& a_in, int batch, void* stream) {
    Attn16Args a = a_in;
    int code = g_attn16;
    if (!code) {
        const long long pairs2 = (long long)((a.t + 31) / 32) * a.heads * batch;       // blocks at 2 query tiles each
        code = a.t < 128 ? 21 : (pairs2 >= 512 ? 42 : (pairs2 >= 128 ? 24 : 14));
    }
    const int qt = code / 10;
    a.nq = (a.t + 16 * qt - 1) / (16 * qt);
    dim3 g((unsigned)((long long)a.nq * a.heads * batch));
    switch (code) {
        case 21: SVCMI_LAUNCH((attention16_kernel<D, 2, 1, F16, true>), g, dim3(64 * 2), 0, stream, a); break;
        case 24: SVCMI_LAUNCH((attention16_kernel<D, 2, 4, F16, true>), g, dim3(64 * 8), 0, stream, a); break;
        case 14: SVCMI_LAUNCH((attention16_kernel<D, 1, 4, F16, true>), g, dim3(64 * 4), 0, stream, a); break;
        case 42: SVCMI_LAUNCH((attention16_kernel<D, 4, 2, F16, true>), g, dim3(64 * 8), 0, stream, a); break;
        case 44: SVCMI_LAUNCH((attention16_kernel<D, 4, 4, F16, true>), g, dim3(64 * 16), 0, stream, a); break;
        default: return SVCMI_EINVAL;
    }
    return SVCMI_LAST_ERROR();
}

template <int D, bool F16>
int launch_attn16(const Attn16Args& a_in, int batch, void* stream) {
    Attn16Args a = a_in;
    int code = g_attn16;
    if (!code) {
        // measured on MI355X (profiles/r03j_attn16.log, f16, 20 heads x 64): one T = 500 / 750 window 4 x 4 (12.1 / 13.9 us; fp32 kernels 29.2 /
        // 49.3), B = 2 windows or T = 1500 8 x 2 (16.4 / 22.2 / 37.2 us), B = 4 4 x 2 (26.7), B = 16 8 x 1 (62 us = 331 TFLOP/s; fp32 195 us)
        const long long blocks8 = (long long)((a.t + 127) / 128) * a.heads * batch;
        if (a.t < 128) code = 41;
        else if (blocks8 >= 1024) code = 81;
        else if (blocks8 >= 128 && blocks8 <= 256 && a.t >= 256) code = 82;
        else if (blocks8 < 128 && a.t >= 512 - 64) code = 44;
        else code = a.t >= 256 ? 42 : 41;
    }
    const int qt = code / 10;
    a.nq = (a.t + 16 * qt - 1) / (16 * qt);
    dim3 g((unsigned)((long long)a.nq * a.heads * batch));
    switch (code) {
        case 41: SVCMI_LAUNCH((attention16_kernel<D, 4, 1, F16>), g, dim3(64 * 4), 0, stream, a); break;
        case 42: SVCMI_LAUNCH((attention16_kernel<D, 4, 2, F16>), g, dim3(64 * 8), 0, stream, a); break;
        case 44: SVCMI_LAUNCH((attention16_kernel<D, 4, 4, F16>), g, dim3(64 * 16), 0, stream, a); break;
        case 81: SVCMI_LAUNCH((attention16_kernel<D, 8, 1, F16>), g, dim3(64 * 8), 0, stream, a); break;
        case 82: SVCMI_LAUNCH((attention16_kernel<D, 8, 2, F16>), g, dim3(64 * 16), 0, stream, a); break;
        default: return SVCMI_EINVAL;
    }
    return SVCMI_LAST_ERROR();
}

int g_attn_lds = 0;     // tuning knob ("attn_lds"): 0 = heuristic, -1 = never, 1 / 10 * QT + KS = force the LDS-staged kernel (band-free, D <= 64)
int g_attn_ns = 0;      // tuning knob (svcmi_tune_set("attn_ns", 0 | 1 | 2 | 4 | 8)); 0 = heuristic

int g_attn_q32 = -1;    // tuning knob ("attn_q32", -1 = heuristic | 0 | 1): two query tiles per wave for band-free attention
int g_attn_wide = -1;   // tuning knob ("attn_wide", -1 = default (off) | 0 | 1): the 64-key-step kernel (attention_wide_kernel; D = 64 band-free, D = 96 band)

template <int D>
int launch_attn(const AttnArgs& a_in, int batch, void* stream) {
    AttnArgs a = a_in;
    // two query tiles per wave: every K / V fragment a wave fetches feeds twice the MFMAs (16 instead of 8 FLOP per fetched byte).  Once a
    // launch fills the chip that feed is what bounds the kernel (profiles/r02ae_attention_saturation.log: B = 16 x T = 500: 215 vs 283-323 us,
    // 95 vs 63-72 TFLOP/s; B = 4: 63-67 vs 80-84 us; T = 1500: 139.5 vs 175 us); a single T = 500 window (640 blocks of one query tile) is
    // latency-bound and keeps the one-tile kernel (29.3 vs 33.5 us)
    const long long blocks16 = (long long)((a.t + 15) / 16) * a.heads * batch;
    const bool q32 = g_attn_q32 >= 0 ? (g_attn_q32 != 0 && !a.rel_k) : (!a.rel_k && D <= 64 && (a.t >= 1024 || blocks16 >= 1280));
    if constexpr (D <= 64) {
        // LDS-staged kernel, 8 query tiles per block: measured on MI355X (profiles/r03a_attnlds.log) it wins where its blocks fill the chip
        // evenly -- one resident round of <= 256 blocks of 16 waves (B = 2 x T = 500: 43.8 vs 47.7 us, B = 2 x T = 750: 63.8 vs 73.6,
        // T = 1500: 120 vs 132) or many rounds (B = 16 x T = 500: 195 vs 237 us = 105 TFLOP/s) -- and loses at 1.25 blocks per CU
        // (B = 4: 80-86 vs 69 us) and for one T = 500 window alone (80 blocks: 43 vs 29.6 us), which keep the kernels below.
        int lds_code = g_attn_lds < 0 ? 0 : g_attn_lds;
        if (g_attn_lds == 0 && g_attn_q32 < 0 && g_attn_ns == 0 && !a.rel_k && D == 64) {     // (the other knobs force the older kernels)
            const long long blocks8 = (long long)((a.t + 127) / 128) * a.heads * batch;
            if (blocks8 >= 1024) lds_code = 81;
            else if (blocks8 >= 128 && blocks8 <= 256 && a.t >= 256) lds_code = 82;
        }
        if (lds_code && !a.rel_k) {
            // knob value = 10 * QT + KS (1 = the default shape 4 x 2, one key range for short sequences)
            int code = lds_code == 1 ? (a.t >= 256 ? 42 : 41) : lds_code;
            const int qt = code / 10;
            a.nq = (a.t + 16 * qt - 1) / (16 * qt);
            dim3 g((unsigned)((long long)a.nq * a.heads * batch));
            switch (code) {
                case 21: SVCMI_LAUNCH((attention_lds_kernel<D, 2, 1>), g, dim3(64 * 2), 0, stream, a); break;
                case 22: SVCMI_LAUNCH((attention_lds_kernel<D, 2, 2>), g, dim3(64 * 4), 0, stream, a); break;
                case 24: SVCMI_LAUNCH((attention_lds_kernel<D, 2, 4>), g, dim3(64 * 8), 0, stream, a); break;
                case 41: SVCMI_LAUNCH((attention_lds_kernel<D, 4, 1>), g, dim3(64 * 4), 0, stream, a); break;
                case 42: SVCMI_LAUNCH((attention_lds_kernel<D, 4, 2>), g, dim3(64 * 8), 0, stream, a); break;
                case 44: SVCMI_LAUNCH((attention_lds_kernel<D, 4, 4>), g, dim3(64 * 16), 0, stream, a); break;
                case 81: SVCMI_LAUNCH((attention_lds_kernel<D, 8, 1>), g, dim3(64 * 8), 0, stream, a); break;
                case 82: SVCMI_LAUNCH((attention_lds_kernel<D, 8, 2>), g, dim3(64 * 16), 0, stream, a); break;
                default: return SVCMI_EINVAL;
            }
            return SVCMI_LAST_ERROR();
        }
    }
    if constexpr (D == 96 || D == 32) {
        // relative-position band through the LDS-staged kernel (round 4): the prior encoder has 2 heads, so only batched launches fill the
        // chip with 8-query-tile blocks (B = 16 x T = 1000: 256 blocks); one clip (16 blocks) keeps the register-fed kernel below
        if (a.rel_k && a.window <= MAXW && g_attn_lds >= 0 && g_attn_ns == 0) {
            const long long blocks8 = (long long)((a.t + 127) / 128) * a.heads * batch;
            int code = g_attn_lds > 1 ? g_attn_lds : (g_attn_lds == 1 ? 82 : (blocks8 >= 128 ? (blocks8 >= 512 ? 81 : 82) : 0));
            if (code == 81 || code == 82 || code == 41 || code == 42) {
                const int qt = code / 10;
                a.nq = (a.t + 16 * qt - 1) / (16 * qt);
                dim3 g((unsigned)((long long)a.nq * a.heads * batch));
                switch (code) {
                    case 41: SVCMI_LAUNCH((attention_lds_kernel<D, 4, 1, true>), g, dim3(64 * 4), 0, stream, a); break;
                    case 42: SVCMI_LAUNCH((attention_lds_kernel<D, 4, 2, true>), g, dim3(64 * 8), 0, stream, a); break;
                    case 81: SVCMI_LAUNCH((attention_lds_kernel<D, 8, 1, true>), g, dim3(64 * 8), 0, stream, a); break;
                    default: SVCMI_LAUNCH((attention_lds_kernel<D, 8, 2, true>), g, dim3(64 * 16), 0, stream, a); break;
                }
                return SVCMI_LAST_ERROR();
            }
        }
    }
    if (q32) a.nq = (a.t + 31) / 32;
    // key-split NS: ~2 waves per SIMD (1024 SIMDs), but keep >= 64 keys per wave
    const long long blocks = (long long)a.nq * a.heads * batch;
    int ns = 1;
    while (ns < 8 && blocks * ns < 2048 && a.t >= 128 * ns) ns *= 2;
    if (q32 && g_attn_q32 < 0) ns = blocks >= 4096 ? 1 : (blocks >= 800 ? 2 : 4);      // measured: 1 / 2 / 4-way split at 5120 / 940-3840 / 640 blocks
    if (q32 && a.t < 128 * ns) ns = a.t >= 256 ? 2 : 1;
    if (g_attn_ns) ns = g_attn_ns;
    dim3 grid((unsigned)blocks);
    if (q32) {
        if constexpr (D <= 64) {
            switch (ns) {
                case 1: SVCMI_LAUNCH((attention_q32_kernel<D, 1>), grid, dim3(64), 0, stream, a); break;
                case 2: SVCMI_LAUNCH((attention_q32_kernel<D, 2>), grid, dim3(128), 0, stream, a); break;
                case 4: SVCMI_LAUNCH((attention_q32_kernel<D, 4>), grid, dim3(256), 0, stream, a); break;
                default: SVCMI_LAUNCH((attention_q32_kernel<D, 8>), grid, dim3(512), 0, stream, a); break;
            }
            return SVCMI_LAST_ERROR();
        }
    }
    if constexpr (D == 64 || D == 96) {
        // 64-key steps with all of a step's requests in flight (round 6): OPT-IN ("attn_wide" = 1).  Measured (profiles/r06s_attention_wide.log): a
        // T = 500 Whisper window 24.3 against 23.3 us (graph-timed), the judged line 0.6 % slower with it -- the launch is not bound by the chain
        // of round trips: with the key loop's MFMAs removed it still takes 18.0 us, with the K / V requests removed 17.7 us, without the loop 2.7
        const bool fits32 = (long long)a.t * (a.ldk > a.ldv ? a.ldk : a.ldv) * 4 < 0x7fffffffLL;        // 32-bit byte offsets of the buffer loads
        const bool wide = fits32 && g_attn_wide > 0;
        if (wide && (D == 64) == (a.rel_k == nullptr)) {
            constexpr bool RELW = D == 96;
            switch (ns) {
                case 1: SVCMI_LAUNCH((attention_wide_kernel<D, 1, RELW>), grid, dim3(64), 0, stream, a); break;
                case 2: SVCMI_LAUNCH((attention_wide_kernel<D, 2, RELW>), grid, dim3(128), 0, stream, a); break;
                case 4: SVCMI_LAUNCH((attention_wide_kernel<D, 4, RELW>), grid, dim3(256), 0, stream, a); break;
                default: SVCMI_LAUNCH((attention_wide_kernel<D, 8, RELW>), grid, dim3(512), 0, stream, a); break;
            }
            return SVCMI_LAST_ERROR();
        }
    }
    if (!a.rel_k) {
        switch (ns) {
            case 1: SVCMI_LAUNCH((attention_kernel<D, 1, false>), grid, dim3(64), 0, stream, a); break;
            case 2: SVCMI_LAUNCH((attention_kernel<D, 2, false>), grid, dim3(128), 0, stream, a); break;
            case 4: SVCMI_LAUNCH((attention_kernel<D, 4, false>), grid, dim3(256), 0, stream, a); break;
            default: SVCMI_LAUNCH((attention_kernel<D, 8, false>), grid, dim3(512), 0, stream, a); break;
        }
        return SVCMI_LAST_ERROR();
    }
    switch (ns) {
        case 1: SVCMI_LAUNCH((attention_kernel<D, 1>), grid, dim3(64), 0, stream, a); break;
        case 2: SVCMI_LAUNCH((attention_kernel<D, 2>), grid, dim3(128), 0, stream, a); break;
        case 4: SVCMI_LAUNCH((attention_kernel<D, 4>), grid, dim3(256), 0, stream, a); break;
        default: SVCMI_LAUNCH((attention_kernel<D, 8>), grid, dim3(512), 0, stream, a); break;
    }
    return SVCMI_LAST_ERROR();
}

}  // namespace

extern "C" int svcmi_layernorm_f32(const float* x, const float* res, const float* gamma, const float* beta, float* y,
                                   int32_t batch, int32_t rows_per_batch, int32_t c, int32_t ldx, int32_t ldr,
                                   int32_t ldy, int32_t gb_bstride, float eps, void* y16, int32_t ldy16, int32_t y16_format, void* stream) {
    if (!x || !y || batch <= 0 || rows_per_batch <= 0 || c <= 0) return SVCMI_EINVAL;
    if (y16 && (svcmi_fmt16(y16_format) < 0 || !svcmi_fmt16_row_ok(y16_format, ldy16, c) || ((uintptr_t)y16 & 7))) return SVCMI_EINVAL;
    if (c % 4 != 0 || c > 64 * 4 * LN_MAXV) return SVCMI_EUNSUPPORTED;
    if (ldx % 4 || ldy % 4 || (res && ldr % 4) || gb_bstride % 4) return SVCMI_EALIGN;
    if (((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)res & 15) || ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15))
        return SVCMI_EALIGN;
    const long long rows = (long long)batch * rows_per_batch;
    if (rows > 0x7fffffffLL) return SVCMI_EUNSUPPORTED;
    SVCMI_LAUNCH(layernorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, res, gamma, beta, y,
                 (int)rows, rows_per_batch, c, ldx, ldr, ldy, gb_bstride, eps, static_cast<unsigned short*>(y16), ldy16,
                 svcmi_fmt16(y16_format));
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_channel_norm_gelu_f32(const float* x, const float* gamma, const float* beta, float* y, double* scratch,
                                           int32_t batch, int32_t t, int32_t c, int32_t ldx, int32_t ldy, float eps, void* stream) {
    // scratch: >= batch * (2*64 + 1) * c doubles (chunk partials, then mean/rstd as floats behind them)
    if (!x || !y || !scratch || batch <= 0 || t <= 0 || c <= 0) return SVCMI_EINVAL;
    if (c % 4 || ldx % 4 || ldy % 4 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15)) return SVCMI_EALIGN;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    int nch = (t + 255) / 256;
    if (nch > CN_CHUNKS_MAX) nch = CN_CHUNKS_MAX;
    float* stats = reinterpret_cast<float*>(scratch + (long long)batch * CN_CHUNKS_MAX * c * 2);
    SVCMI_LAUNCH(colstats_kernel, dim3((c + 63) / 64, nch, batch), dim3(256), 0, stream, x, scratch, t, c, ldx, nch);
    int rc = SVCMI_LAST_ERROR();
    if (rc) return rc;
    SVCMI_LAUNCH(colstats_final_kernel, dim3((c + 255) / 256, batch), dim3(256), 0, stream, (const double*)scratch, stats, t, c, nch, eps);
    rc = SVCMI_LAST_ERROR();
    if (rc) return rc;
    long long nb = ((long long)t * (c / 4) + 255) / 256;
    if (nb > 2048) nb = 2048;
    SVCMI_LAUNCH(channel_norm_gelu_kernel, dim3((unsigned)nb, batch), dim3(256), 0, stream, x, (const float*)stats, gamma, beta, y, t, c, ldx, ldy);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_splitk_layernorm_f32(const float* partials, int32_t split, const float* bias, float* x, const float* gamma,
                                          const float* beta, float* y, int32_t batch, int32_t rows_per_batch, int32_t c,
                                          int32_t ldx, int32_t ldy, float eps, void* y16, int32_t ldy16, int32_t y16_format, void* stream) {
    if (!partials || !x || !y || split < 1 || batch <= 0 || rows_per_batch <= 0 || c <= 0) return SVCMI_EINVAL;
    if (y16 && (svcmi_fmt16(y16_format) < 0 || !svcmi_fmt16_row_ok(y16_format, ldy16, c) || ((uintptr_t)y16 & 7))) return SVCMI_EINVAL;
    if (c % 4 != 0 || c > 256 * 4 * SKV) return SVCMI_EUNSUPPORTED;
    if (ldx % 4 || ldy % 4) return SVCMI_EALIGN;
    if (((uintptr_t)partials & 15) || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)bias & 15) ||
        ((uintptr_t)gamma & 15) || ((uintptr_t)beta & 15)) return SVCMI_EALIGN;
    const long long rows = (long long)batch * rows_per_batch;
    if (rows > 0x7fffffffLL) return SVCMI_EUNSUPPORTED;
    SVCMI_LAUNCH(splitk_layernorm_kernel, dim3((unsigned)rows), dim3(256), 0, stream, partials, split, bias, x, gamma,
                 beta, y, rows_per_batch, c, ldx, ldy, eps, static_cast<unsigned short*>(y16), ldy16, svcmi_fmt16(y16_format));
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_attention_f32(const float* q, const float* k, const float* v, float* o,
                                   int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo,
                                   int64_t q_bstride, int64_t k_bstride, int64_t v_bstride, int64_t o_bstride,
                                   int32_t batch, int32_t t, int32_t heads, int32_t head_dim, float scale,
                                   const float* rel_k, const float* rel_v, int32_t window,
                                   const int32_t* lengths, void* o16, int32_t ldo16, int64_t o16_bstride, int32_t o16_format, void* stream) {
    if (!q || !k || !v || !o || batch <= 0 || t <= 0 || heads <= 0) return SVCMI_EINVAL;
    if (o16 && (svcmi_fmt16(o16_format) < 0 || !svcmi_fmt16_row_ok(o16_format, ldo16, heads * head_dim) || o16_bstride % 4 || ((uintptr_t)o16 & 7)))
        return SVCMI_EINVAL;
    if ((rel_k == nullptr) != (rel_v == nullptr)) return SVCMI_EINVAL;
    if (rel_k && (window < 0 || window > MAXW)) return SVCMI_EUNSUPPORTED;
    if (ldq % 4 || ldk % 4 || ldv % 4 || ldo % 4 || q_bstride % 4 || k_bstride % 4 || v_bstride % 4 || o_bstride % 4)
        return SVCMI_EALIGN;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15) || ((uintptr_t)o & 15)) return SVCMI_EALIGN;
    if ((long long)batch * heads * ((t + 15) / 16) > 0x7fffffffLL) return SVCMI_EUNSUPPORTED;
    AttnArgs a;
    a.q = q; a.k = k; a.v = v; a.o = o; a.ldq = ldq; a.ldk = ldk; a.ldv = ldv; a.ldo = ldo;
    a.q_bs = q_bstride; a.k_bs = k_bstride; a.v_bs = v_bstride; a.o_bs = o_bstride;
    a.t = t; a.heads = heads; a.nq = (t + 15) / 16; a.scale = scale; a.rel_k = rel_k; a.rel_v = rel_v; a.window = window; a.lengths = lengths;
    a.o16 = static_cast<unsigned short*>(o16); a.ldo16 = ldo16; a.o16_bs = o16_bstride; a.o16_f16 = svcmi_fmt16(o16_format);
    switch (head_dim) {
        case 16: return launch_attn<16>(a, batch, stream);
        case 32: return launch_attn<32>(a, batch, stream);
        case 64: return launch_attn<64>(a, batch, stream);
        case 96: return launch_attn<96>(a, batch, stream);
        default: return SVCMI_EUNSUPPORTED;
    }
}

extern "C" int svcmi_attention16(const void* q, const void* k, const void* v, int32_t ld16, int64_t bstride16, float* o, int32_t ldo,
                                 int64_t o_bstride, void* o16, int32_t ldo16, int64_t o16_bstride, int32_t batch, int32_t t, int32_t heads,
                                 int32_t head_dim, float scale, const float* rel_k, const float* rel_v, int32_t window,
                                 const int32_t* lengths, int32_t format, void* stream) {
    if (!q || !k || !v || (!o && !o16) || batch <= 0 || t <= 0 || heads <= 0) return SVCMI_EINVAL;
    if (format != SVCMI_PREC_BF16 && format != SVCMI_PREC_F16) return SVCMI_EINVAL;
    if ((rel_k == nullptr) != (rel_v == nullptr)) return SVCMI_EINVAL;
    if (rel_k && (window < 0 || window > MAXW)) return SVCMI_EUNSUPPORTED;
    if (rel_k ? (head_dim != 32 && head_dim != 96) : (head_dim != 32 && head_dim != 64)) return SVCMI_EUNSUPPORTED;
    if (ld16 % 8 || bstride16 % 8 || ((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ((uintptr_t)v & 15)) return SVCMI_EALIGN;
    if (o && (ldo % 4 || o_bstride % 4 || ((uintptr_t)o & 15))) return SVCMI_EALIGN;
    if (o16 && (ldo16 % 4 || o16_bstride % 4 || ((uintptr_t)o16 & 7))) return SVCMI_EALIGN;
    if ((long long)batch * heads * ((t + 31) / 32) > 0x7fffffffLL) return SVCMI_EUNSUPPORTED;
    Attn16Args a;
    a.q = static_cast<const unsigned short*>(q); a.k = static_cast<const unsigned short*>(k); a.v = static_cast<const unsigned short*>(v);
    a.o = o; a.o16 = static_cast<unsigned short*>(o16); a.ld16 = ld16; a.ldo = ldo; a.ldo16 = ldo16; a.bs16 = bstride16; a.o_bs = o_bstride;
    a.o16_bs = o16_bstride; a.t = t; a.heads = heads; a.nq = 0; a.scale = scale; a.lengths = lengths; a.o16_f16 = format == SVCMI_PREC_F16;
    a.rel_k = rel_k; a.rel_v = rel_v; a.window = window;
    if (rel_k) {
        if (format == SVCMI_PREC_F16) return head_dim == 96 ? launch_attn16_rel<96, true>(a, batch, stream) : launch_attn16_rel<32, true>(a, batch, stream);
        return head_dim == 96 ? launch_attn16_rel<96, false>(a, batch, stream) : launch_attn16_rel<32, false>(a, batch, stream);
    }
    if (format == SVCMI_PREC_F16) return head_dim == 64 ? launch_attn16<64, true>(a, batch, stream) : launch_attn16<32, true>(a, batch, stream);
    return head_dim == 64 ? launch_attn16<64, false>(a, batch, stream) : launch_attn16<32, false>(a, batch, stream);
}

// Development knob (reached through svcmi_tune_set): results do not depend on it beyond fp32 re-association of the key split.
extern "C" int svcmi_attn_tune_set(const char* name, int32_t value) {
    const char* k = "attn_ns";
    int i = 0;
    while (k[i] && name[i] == k[i]) ++i;
    if (k[i] == 0 && name[i] == 0 && (value == 0 || value == 1 || value == 2 || value == 4 || value == 8)) { g_attn_ns = value; return 0; }
    const char* k4 = "attn_lds";
    i = 0;
    while (k4[i] && name[i] == k4[i]) ++i;
    if (k4[i] == 0 && name[i] == 0 && (value == -1 || value == 0 || value == 1 || value == 21 || value == 22 || value == 24 || value == 41 || value == 42 ||
                                       value == 44 || value == 81 || value == 82)) { g_attn_lds = value; return 0; }
    const char* k5 = "attn16";
    i = 0;
    while (k5[i] && name[i] == k5[i]) ++i;
    if (k5[i] == 0 && name[i] == 0 && (value == 0 || value == 14 || value == 21 || value == 24 || value == 41 || value == 42 || value == 44 ||
                                       value == 81 || value == 82)) { g_attn16 = value; return 0; }
    const char* k6 = "attn_wide";
    i = 0;
    while (k6[i] && name[i] == k6[i]) ++i;
    if (k6[i] == 0 && name[i] == 0 && value >= -1 && value <= 1) { g_attn_wide = value; return 0; }
    const char* k2 = "attn_q32";
    i = 0;
    while (k2[i] && name[i] == k2[i]) ++i;
    if (k2[i] == 0 && name[i] == 0 && value >= -1 && value <= 1) { g_attn_q32 = value; return 0; }
    return SVCMI_EINVAL;
}
