// retrieval.hip -- exact k-nearest-neighbour feature blend (row N4; feature_retrieval/index.py:57-94).
// The reference searches a faiss IVF-Flat index (squared-L2 metric) for the k nearest stored feature vectors of every
// content frame and mixes their 1/d^2-weighted mean into the frame.  Here the candidate scores of ALL stored vectors
// come from one svcmi_conv_gemm_f32 launch (dots = X * Bank^T, the only O(t*n*d) work), and this file holds the
// HBM-bound rest: per-row squared norms of the bank, and one block per query frame that streams its row of dots once
// (coalesced), keeps a per-thread sorted shortlist, merges the shortlists through LDS into 8 candidates, re-measures those with
// exact differences (so neither the choice of the k nearest nor the weights inherit the |x|^2 + |b|^2 - 2xb
// cancellation) and writes the blended frame.
#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

constexpr int KNN_TPB = 256;
constexpr int KNN_WAVES = KNN_TPB / 64;
constexpr int KNN_KMAX = 8;
constexpr int KNN_NONE = 0x7fffffff;

__device__ __forceinline__ float wave_sum_r(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float* x, int ldx, long long rows, int d, float* out) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* xr = x + row * ldx;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        s += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    s = wave_sum_r(s);
    if (lane == 0) out[row] = s;
}

// (score, index) ordering: smaller score first, then smaller index -- a total order, so the result does not depend on
// which thread saw a candidate.
__device__ __forceinline__ bool knn_less(float sa, int ia, float sb, int ib) { return sa < sb || (sa == sb && ia < ib); }

__global__ __launch_bounds__(KNN_TPB) void knn_blend_kernel(const float* x, int ldx, const float* bank, int ldb,
                                                            const float* dots, long long ldd, const float* bank_sq,
                                                            float* out, int ldo, int n, int d, int k, float ratio) {
    __shared__ float cand_s[KNN_KMAX * KNN_TPB];
    __shared__ int cand_i[KNN_KMAX * KNN_TPB];
    __shared__ float red_s[KNN_WAVES * KNN_KMAX];
    __shared__ int red_i[KNN_WAVES];
    __shared__ int sel[KNN_KMAX];
    __shared__ float selw[KNN_KMAX];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row = blockIdx.x;
    const float* xr = x + row * ldx;
    const float* dr = dots + row * ldd;

    // |x|^2 (only shifts every score of this row by the same amount, kept so the scores are distances)
    float xs = 0.f;
    for (int c = tid * 4; c < d; c += KNN_TPB * 4) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        xs += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
    xs = wave_sum_r(xs);
    if (lane == 0) red_s[wave] = xs;
    __syncthreads();
    xs = (red_s[0] + red_s[1]) + (red_s[2] + red_s[3]);
    __syncthreads();

    // stream the row of dot products; per-thread sorted shortlist of the KNN_KMAX best
    float bs[KNN_KMAX];
    int bi[KNN_KMAX];
#pragma unroll
    for (int q = 0; q < KNN_KMAX; ++q) { bs[q] = __builtin_inff(); bi[q] = KNN_NONE; }
    for (int j = tid; j < n; j += KNN_TPB) {
        const float s = (xs + bank_sq[j]) - 2.0f * dr[j];
        if (s < bs[KNN_KMAX - 1]) {       // indices grow within a thread: strict < keeps the smaller index on ties
            bs[KNN_KMAX - 1] = s;
            bi[KNN_KMAX - 1] = j;
#pragma unroll
            for (int q = KNN_KMAX - 1; q >= 1; --q) {
                if (bs[q] < bs[q - 1]) {
                    const float ts = bs[q]; bs[q] = bs[q - 1]; bs[q - 1] = ts;
                    const int ti = bi[q]; bi[q] = bi[q - 1]; bi[q - 1] = ti;
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < KNN_KMAX; ++q) { cand_s[q * KNN_TPB + tid] = bs[q]; cand_i[q * KNN_TPB + tid] = bi[q]; }

    // kc rounds of block arg-min over the shortlist heads; the owner of the winner advances its head.  kc = KNN_KMAX >= k
    // candidates are kept: fp32 scores of near-equidistant neighbours (1e-6 relative apart) can swap the k-th and
    // (k+1)-th, so the final k are chosen below from exact distances.
    const int kc = n < KNN_KMAX ? n : KNN_KMAX;
    int head = 0;
    for (int r = 0; r < kc; ++r) {
        float s = head < KNN_KMAX ? cand_s[head * KNN_TPB + tid] : __builtin_inff();
        int i = head < KNN_KMAX ? cand_i[head * KNN_TPB + tid] : KNN_NONE;
        const float my_s = s;
        const int my_i = i;
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float os = __shfl_xor(s, m);
            const int oi = __shfl_xor(i, m);
            if (knn_less(os, oi, s, i)) { s = os; i = oi; }
        }
        if (lane == 0) { red_s[wave] = s; red_i[wave] = i; }
        __syncthreads();
        s = red_s[0]; i = red_i[0];
#pragma unroll
        for (int w = 1; w < KNN_WAVES; ++w)
            if (knn_less(red_s[w], red_i[w], s, i)) { s = red_s[w]; i = red_i[w]; }
        if (my_i == i && i != KNN_NONE) { ++head; sel[r] = i; }
        if (i == KNN_NONE && tid == 0) sel[r] = 0;      // fewer than k finite candidates (NaN input): stay in bounds
        __syncthreads();
    }

    // exact squared distances of the candidates
    float acc[KNN_KMAX];
#pragma unroll
    for (int q = 0; q < KNN_KMAX; ++q) acc[q] = 0.f;
    for (int c = tid * 4; c < d; c += KNN_TPB * 4) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
#pragma unroll
        for (int q = 0; q < KNN_KMAX; ++q) {
            if (q < kc) {
                const float4 b = *reinterpret_cast<const float4*>(bank + (long long)sel[q] * ldb + c);
                const float dx = v.x - b.x, dy = v.y - b.y, dz = v.z - b.z, dw = v.w - b.w;
                acc[q] += (dx * dx + dy * dy) + (dz * dz + dw * dw);
            }
        }
    }
#pragma unroll
    for (int q = 0; q < KNN_KMAX; ++q) {
        const float t = wave_sum_r(acc[q]);
        if (lane == 0) red_s[wave * KNN_KMAX + q] = t;
    }
    __syncthreads();
    if (tid == 0) {
        float dist[KNN_KMAX];
        int id[KNN_KMAX];
        for (int q = 0; q < kc; ++q) {      // insertion sort by (exact distance, index): the k nearest, ascending like faiss
            float dq = (red_s[q] + red_s[KNN_KMAX + q]) + (red_s[2 * KNN_KMAX + q] + red_s[3 * KNN_KMAX + q]);
            int iq = sel[q], pos = q;
            while (pos > 0 && knn_less(dq, iq, dist[pos - 1], id[pos - 1])) { dist[pos] = dist[pos - 1]; id[pos] = id[pos - 1]; --pos; }
            dist[pos] = dq; id[pos] = iq;
        }
        float w[KNN_KMAX], wsum = 0.f;      // weight = (1/d)^2, normalised over the k neighbours (index.py:86-88)
        for (int q = 0; q < k; ++q) {
            const float inv = 1.0f / dist[q];
            w[q] = inv * inv;
            wsum += w[q];
        }
        for (int q = 0; q < k; ++q) { selw[q] = w[q] / wsum; sel[q] = id[q]; }
    }
    __syncthreads();
    const float keep = 1.0f - ratio;
    float* orow = out + row * ldo;
    for (int c = tid * 4; c < d; c += KNN_TPB * 4) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < KNN_KMAX; ++q) {
            if (q < k) {
                const float4 b = *reinterpret_cast<const float4*>(bank + (long long)sel[q] * ldb + c);
                const float w = selw[q];
                m.x += b.x * w; m.y += b.y * w; m.z += b.z * w; m.w += b.w * w;
            }
        }
        float4 o;
        o.x = keep * v.x + ratio * m.x; o.y = keep * v.y + ratio * m.y;
        o.z = keep * v.z + ratio * m.z; o.w = keep * v.w + ratio * m.w;
        *reinterpret_cast<float4*>(orow + c) = o;
    }
}

inline bool mis16r(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) != 0; }

}  // namespace

extern "C" int svcmi_row_sqnorm_f32(const float* x, int32_t ldx, int64_t rows, int32_t d, float* out, void* stream) {
    if (!x || !out || rows <= 0 || d <= 0 || ldx < d) return SVCMI_EINVAL;
    if (d % 4 || ldx % 4 || mis16r(x)) return SVCMI_EALIGN;
    if ((rows + 3) / 4 > 0x7fffffffLL) return SVCMI_EUNSUPPORTED;
    SVCMI_LAUNCH(row_sqnorm_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, stream, x, ldx, (long long)rows, d, out);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_knn_blend_f32(const float* x, int32_t ldx, const float* bank, int32_t ldb, const float* dots, int64_t ldd,
                                   const float* bank_sq, float* out, int32_t ldo, int32_t t, int32_t n, int32_t d, int32_t k,
                                   float ratio, void* stream) {
    if (!x || !bank || !dots || !bank_sq || !out || t <= 0 || n <= 0 || d <= 0 || ldx < d || ldb < d || ldo < d || ldd < n)
        return SVCMI_EINVAL;
    if (k < 1 || k > n) return SVCMI_EINVAL;
    if (k > KNN_KMAX) return SVCMI_EUNSUPPORTED;
    if (d % 4 || ldx % 4 || ldb % 4 || ldo % 4 || mis16r(x) || mis16r(bank) || mis16r(out)) return SVCMI_EALIGN;
    SVCMI_LAUNCH(knn_blend_kernel, dim3(t), dim3(KNN_TPB), 0, stream, x, ldx, bank, ldb, dots, (long long)ldd, bank_sq, out, ldo,
                 n, d, k, ratio);
    return SVCMI_LAST_ERROR();
}
