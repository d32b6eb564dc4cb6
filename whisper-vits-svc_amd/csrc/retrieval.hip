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

// ---------------------------------------------------------------------------------------------------------------
// IVF-Flat with nprobe = 1 -- what the reference actually builds and searches (feature_retrieval/index.py:145-151:
// index_factory("IVF{n},Flat", METRIC_L2), nprobe = 1; :57-62 search_and_reconstruct).  faiss (1.7.4, pinned by the
// reference's requirements.txt, not vendored) answers a query in two steps: the coarse quantizer (IndexFlatL2 over the nlist
// centroids) picks the nearest centroid from |x|^2 + |c|^2 - 2 x.c clamped at 0 (its BLAS path, utils/distances.cpp
// exhaustive_L2sqr_blas), then the inverted list of that ONE cell is scanned with exact sum (x - b)^2 distances
// (IndexIVFFlat.cpp IVFFlatScanner / fvec_L2sqr) and the k smallest kept, ascending.  ivf_assign is step 1 (also the
// assignment step of the k-means that trains the centroids, Clustering.cpp), ivf_blend step 2 + the RVC weighting.

// one block per query row: argmin_j (|x|^2 + csq[j]) - 2 dots[row][j], ties -> smaller j
__global__ __launch_bounds__(KNN_TPB) void ivf_assign_kernel(const float* x, int ldx, const float* dots, long long ldd, const float* csq,
                                                             int nlist, int d, int* assign, float* dist) {
    __shared__ float red_s[KNN_WAVES];
    __shared__ int red_i[KNN_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row = blockIdx.x;
    const float* xr = x + row * ldx;
    const float* dr = dots + row * ldd;
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
    float s = __builtin_inff();
    int i = KNN_NONE;
    for (int j = tid; j < nlist; j += KNN_TPB) {
        float v = (xs + csq[j]) - 2.0f * dr[j];
        v = v < 0.f ? 0.f : v;                           // faiss clamps the BLAS-form distance at 0
        if (v < s) { s = v; i = j; }                     // indices grow within a thread: strict < keeps the smaller one
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float os = __shfl_xor(s, m);
        const int oi = __shfl_xor(i, m);
        if (knn_less(os, oi, s, i)) { s = os; i = oi; }
    }
    if (lane == 0) { red_s[wave] = s; red_i[wave] = i; }
    __syncthreads();
    if (tid == 0) {
        s = red_s[0]; i = red_i[0];
        for (int w = 1; w < KNN_WAVES; ++w)
            if (knn_less(red_s[w], red_i[w], s, i)) { s = red_s[w]; i = red_i[w]; }
        assign[row] = i == KNN_NONE ? 0 : i;             // all-NaN row: stay in bounds
        if (dist) dist[row] = s;
    }
}

// one block per query row: scan rows [off[a], off[a+1]) of the list-ordered bank (a = assign[row]) with exact distances, keep the
// k nearest (ascending, ties -> earlier list position), blend.  A cell with fewer than k vectors contributes the ones it has (faiss
// pads with label -1 / a NaN reconstruction, which turns the reference's output frame into NaN); an empty cell leaves the frame as is.
__global__ __launch_bounds__(KNN_TPB) void ivf_blend_kernel(const float* x, int ldx, const int* assign, const int* off, const float* bank,
                                                            int ldb, float* out, int ldo, int d, int k, float ratio, int* out_idx,
                                                            float* out_dist) {
    __shared__ float cand_s[KNN_WAVES * KNN_KMAX];
    __shared__ int cand_i[KNN_WAVES * KNN_KMAX];
    __shared__ int sel[KNN_KMAX];
    __shared__ float selw[KNN_KMAX];
    __shared__ int nsel;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long row = blockIdx.x;
    const float* xr = x + row * ldx;
    const int a = assign[row];
    const int lo = off[a], hi = off[a + 1];

    // every lane of a wave carries the same sorted shortlist (the distance is a wave-wide sum)
    float bs[KNN_KMAX];
    int bi[KNN_KMAX];
#pragma unroll
    for (int q = 0; q < KNN_KMAX; ++q) { bs[q] = __builtin_inff(); bi[q] = KNN_NONE; }
    for (int r = lo + wave; r < hi; r += KNN_WAVES) {
        const float* br = bank + (long long)r * ldb;
        float acc = 0.f;
        for (int c = lane * 4; c < d; c += 256) {
            const float4 v = *reinterpret_cast<const float4*>(xr + c);
            const float4 b = *reinterpret_cast<const float4*>(br + c);
            const float dx = v.x - b.x, dy = v.y - b.y, dz = v.z - b.z, dw = v.w - b.w;
            acc += (dx * dx + dy * dy) + (dz * dz + dw * dw);
        }
        acc = wave_sum_r(acc);
        if (acc < bs[KNN_KMAX - 1]) {                    // rows grow within a wave: strict < keeps the earlier one on ties
            bs[KNN_KMAX - 1] = acc;
            bi[KNN_KMAX - 1] = r;
#pragma unroll
            for (int q = KNN_KMAX - 1; q >= 1; --q) {
                if (bs[q] < bs[q - 1]) {
                    const float ts = bs[q]; bs[q] = bs[q - 1]; bs[q - 1] = ts;
                    const int ti = bi[q]; bi[q] = bi[q - 1]; bi[q - 1] = ti;
                }
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < KNN_KMAX; ++q) { cand_s[wave * KNN_KMAX + q] = bs[q]; cand_i[wave * KNN_KMAX + q] = bi[q]; }
    }
    __syncthreads();
    if (tid == 0) {
        float dist[KNN_KMAX];
        int id[KNN_KMAX];
        int cnt = 0;
        for (int c = 0; c < KNN_WAVES * KNN_KMAX; ++c) {     // merge the four sorted shortlists by (distance, position)
            const float dq = cand_s[c];
            const int iq = cand_i[c];
            if (iq == KNN_NONE) continue;
            if (cnt == k && !knn_less(dq, iq, dist[k - 1], id[k - 1])) continue;
            int pos = cnt < k ? cnt : k - 1;
            while (pos > 0 && knn_less(dq, iq, dist[pos - 1], id[pos - 1])) { dist[pos] = dist[pos - 1]; id[pos] = id[pos - 1]; --pos; }
            dist[pos] = dq; id[pos] = iq;
            if (cnt < k) ++cnt;
        }
        float w[KNN_KMAX], wsum = 0.f;                       // weight = (1/d)^2, normalised (index.py:86-88)
        for (int q = 0; q < cnt; ++q) {
            const float inv = 1.0f / dist[q];
            w[q] = inv * inv;
            wsum += w[q];
        }
        for (int q = 0; q < cnt; ++q) { selw[q] = w[q] / wsum; sel[q] = id[q]; }
        nsel = cnt;
        for (int q = 0; q < k; ++q) {
            if (out_idx) out_idx[row * k + q] = q < cnt ? id[q] : -1;
            if (out_dist) out_dist[row * k + q] = q < cnt ? dist[q] : __builtin_inff();
        }
    }
    __syncthreads();
    const int cnt = nsel;
    const float r_eff = cnt > 0 ? ratio : 0.f;
    const float keep = 1.0f - r_eff;
    float* orow = out + row * ldo;
    for (int c = tid * 4; c < d; c += KNN_TPB * 4) {
        const float4 v = *reinterpret_cast<const float4*>(xr + c);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int q = 0; q < KNN_KMAX; ++q) {
            if (q < cnt) {
                const float4 b = *reinterpret_cast<const float4*>(bank + (long long)sel[q] * ldb + c);
                const float w = selw[q];
                m.x += b.x * w; m.y += b.y * w; m.z += b.z * w; m.w += b.w * w;
            }
        }
        float4 o;
        o.x = keep * v.x + r_eff * m.x; o.y = keep * v.y + r_eff * m.y;
        o.z = keep * v.z + r_eff * m.z; o.w = keep * v.w + r_eff * m.w;
        *reinterpret_cast<float4*>(orow + c) = o;
    }
}

// k-means centroid update (faiss Clustering.cpp compute_centroids without weights): out[c] = (sum of the rows x[order[r]],
// r in [off[c], off[c+1])) * (1 / count), rows added in list order (deterministic); an empty segment keeps its old centroid
// (split_clusters on the host re-seeds it).  One block per centroid, a thread owns 4 columns.
__global__ __launch_bounds__(KNN_TPB) void segment_mean_kernel(const float* x, int ldx, const int* order, const int* off, float* out, int ldo,
                                                               int d) {
    const int c0 = blockIdx.x;
    const int lo = off[c0], hi = off[c0 + 1];
    if (hi <= lo) return;
    const float norm = 1.0f / (float)(hi - lo);
    for (int c = threadIdx.x * 4; c < d; c += KNN_TPB * 4) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r = lo; r < hi; ++r) {
            const float4 v = *reinterpret_cast<const float4*>(x + (long long)order[r] * ldx + c);
            s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
        }
        *reinterpret_cast<float4*>(out + (long long)c0 * ldo + c) = make_float4(s.x * norm, s.y * norm, s.z * norm, s.w * norm);
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

extern "C" int svcmi_ivf_assign_f32(const float* x, int32_t ldx, const float* dots, int64_t ldd, const float* cent_sq, int32_t t,
                                    int32_t nlist, int32_t d, int32_t* assign, float* dist, void* stream) {
    if (!x || !dots || !cent_sq || !assign || t <= 0 || nlist <= 0 || d <= 0 || ldx < d || ldd < nlist) return SVCMI_EINVAL;
    if (d % 4 || ldx % 4 || mis16r(x)) return SVCMI_EALIGN;
    SVCMI_LAUNCH(ivf_assign_kernel, dim3(t), dim3(KNN_TPB), 0, stream, x, ldx, dots, (long long)ldd, cent_sq, nlist, d, assign, dist);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_ivf_blend_f32(const float* x, int32_t ldx, const int32_t* assign, const int32_t* list_off, const float* bank,
                                   int32_t ldb, float* out, int32_t ldo, int32_t t, int32_t d, int32_t k, float ratio,
                                   int32_t* out_idx, float* out_dist, void* stream) {
    if (!x || !assign || !list_off || !bank || !out || t <= 0 || d <= 0 || ldx < d || ldb < d || ldo < d) return SVCMI_EINVAL;
    if (k < 1) return SVCMI_EINVAL;
    if (k > KNN_KMAX) return SVCMI_EUNSUPPORTED;
    if (d % 4 || ldx % 4 || ldb % 4 || ldo % 4 || mis16r(x) || mis16r(bank) || mis16r(out)) return SVCMI_EALIGN;
    SVCMI_LAUNCH(ivf_blend_kernel, dim3(t), dim3(KNN_TPB), 0, stream, x, ldx, assign, list_off, bank, ldb, out, ldo, d, k, ratio, out_idx,
                 out_dist);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_segment_mean_f32(const float* x, int32_t ldx, const int32_t* order, const int32_t* seg_off, float* out, int32_t ldo,
                                      int32_t segments, int32_t d, void* stream) {
    if (!x || !order || !seg_off || !out || segments <= 0 || d <= 0 || ldx < d || ldo < d) return SVCMI_EINVAL;
    if (d % 4 || ldx % 4 || ldo % 4 || mis16r(x) || mis16r(out)) return SVCMI_EALIGN;
    SVCMI_LAUNCH(segment_mean_kernel, dim3(segments), dim3(KNN_TPB), 0, stream, x, ldx, order, seg_off, out, ldo, d);
    return SVCMI_LAST_ERROR();
}
