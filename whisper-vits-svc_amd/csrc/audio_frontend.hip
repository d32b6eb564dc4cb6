// audio_frontend.hip -- Whisper log-mel front-end glue (whisper/audio.py:68-100), the step in front of the encoder.
// The windowed DFT and the mel projection are two launches of the implicit-GEMM kernel (conv_gemm.hip):
//   frames x [hann*cos | -hann*sin] basis  = a stride-160, 400-tap, 1-channel convolution of the reflect-padded signal,
//   power x Slaney filterbank              = a plain linear layer;
// this file holds what sits between them: reflect padding (torch.stft center=True), |X|^2, and the
// log10 / (max - 8) clamp / (x + 4) / 4 tail with its global maximum as a two-kernel reduction (max is exactly
// associative, so the result does not depend on the block order).
#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

constexpr int TPB = 256;

__global__ __launch_bounds__(TPB) void reflect_pad_kernel(const float* x, float* y, long long n, int pad) {
    const int b = blockIdx.y;
    const long long m = n + 2LL * pad;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < m; i += (long long)gridDim.x * TPB) {
        long long j = i - pad;
        if (j < 0) j = -j;                        // torch 'reflect': no edge repeat
        if (j >= n) j = 2 * (n - 1) - j;
        y[(long long)b * m + i] = x[(long long)b * n + j];
    }
}

// p[r][f] = re^2 + im^2 with (re | im) = ri[r][f], ri[r][half + f]; columns nbins..ldp-1 of p are written as zero
__global__ __launch_bounds__(TPB) void power_kernel(const float* ri, float* p, long long rows, int nbins, int half, int ldri, int ldp) {
    const long long total = rows * ldp;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long r = i / ldp;
        const int f = (int)(i - r * ldp);
        float v = 0.f;
        if (f < nbins) {
            const float re = ri[r * ldri + f], im = ri[r * ldri + half + f];
            v = fmaf(re, re, im * im);
        }
        p[i] = v;
    }
}

// x <- log10(max(x, 1e-10)) in place; blockmax[b, block] = max over the block's elements (per batch item)
__global__ __launch_bounds__(TPB) void log10_blockmax_kernel(float* x, float* blockmax, long long per_item) {
    __shared__ float red[TPB / 64];
    const int b = blockIdx.y;
    float* xb = x + (long long)b * per_item;
    float m = -3.0e38f;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < per_item; i += (long long)gridDim.x * TPB) {
        const float v = log10f(fmaxf(xb[i], 1e-10f));
        xb[i] = v;
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) m = fmaxf(m, __shfl_xor(m, s));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float mm = red[0];
        for (int i = 1; i < TPB / 64; ++i) mm = fmaxf(mm, red[i]);
        blockmax[(long long)b * gridDim.x + blockIdx.x] = mm;
    }
}

// out[b][c][t] = (max(x[b][t][c], gmax_b - 8) + 4) / 4, gmax_b = max of blockmax[b][:]   (time-major in, NCL out)
__global__ __launch_bounds__(TPB) void logmel_finish_kernel(const float* x, const float* blockmax, int nblk, float* out, int t, int c, int ldx) {
    __shared__ float tile[32][33];
    __shared__ float gm;
    const int b = blockIdx.z;
    if (threadIdx.x == 0) {
        float m = blockmax[(long long)b * nblk];
        for (int i = 1; i < nblk; ++i) m = fmaxf(m, blockmax[(long long)b * nblk + i]);
        gm = m - 8.0f;
    }
    __syncthreads();
    const float floor_v = gm;
    const int t0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;      // 32 x 8 threads
    for (int r = ly; r < 32; r += 8) {
        const int tt = t0 + r, cc = c0 + lx;
        tile[r][lx] = (tt < t && cc < c) ? x[((long long)b * t + tt) * ldx + cc] : 0.f;
    }
    __syncthreads();
    for (int r = ly; r < 32; r += 8) {
        const int cc = c0 + r, tt = t0 + lx;
        if (cc < c && tt < t) out[((long long)b * c + cc) * t + tt] = (fmaxf(tile[lx][r], floor_v) + 4.0f) * 0.25f;
    }
}

// ------------------------------------------------------------------------------------ CREPE glue
constexpr int CREPE_WIN = 1024, CREPE_LEAD = 254;

// one block per frame: 256 threads x 4 samples; mean and unbiased std by wave shuffles + LDS
__global__ __launch_bounds__(TPB) void crepe_frames_kernel(const float* audio, long long n, int hop, int frame0, float* out, int ld) {
    __shared__ float red[TPB / 64];
    const int f = blockIdx.x;
    const long long s0 = (long long)(frame0 + f) * hop - CREPE_WIN / 2;
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const long long i = s0 + threadIdx.x * 4 + j;
        v[j] = (i >= 0 && i < n) ? audio[i] : 0.f;
        s += v[j];
    }
    auto block_sum = [&](float x) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
        __syncthreads();
        return (red[0] + red[1]) + (red[2] + red[3]);
    };
    const float mean = block_sum(s) / (float)CREPE_WIN;
    float ss = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] -= mean; ss += v[j] * v[j]; }
    const float sd = sqrtf(block_sum(ss) / (float)(CREPE_WIN - 1));      // torch.std: unbiased
    const float inv = 1.0f / fmaxf(1e-10f, sd);
    float* o = out + (long long)f * ld;
    for (int i = threadIdx.x; i < ld; i += TPB)
        if (i < CREPE_LEAD || i >= CREPE_LEAD + CREPE_WIN) o[i] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[CREPE_LEAD + threadIdx.x * 4 + j] = v[j] * inv;
}

__global__ __launch_bounds__(TPB) void bn_maxpool2_kernel(const float* x, const float* scale, const float* shift, float* y,
                                                          long long rows_out, int c, int ldx, int ldy, unsigned short* y16, int ldy16, int fmt16) {
    const int c4 = c >> 2;
    const long long total = rows_out * c4;
    for (long long i = (long long)blockIdx.x * TPB + threadIdx.x; i < total; i += (long long)gridDim.x * TPB) {
        const long long r = i / c4;
        const int cc = (int)(i - r * c4) * 4;
        const float4 a = *reinterpret_cast<const float4*>(x + (2 * r) * ldx + cc);
        const float4 b = *reinterpret_cast<const float4*>(x + (2 * r + 1) * ldx + cc);
        const float4 sc = *reinterpret_cast<const float4*>(scale + cc);
        const float4 sh = *reinterpret_cast<const float4*>(shift + cc);
        float4 o;
        o.x = fmaxf(fmaf(a.x, sc.x, sh.x), fmaf(b.x, sc.x, sh.x));
        o.y = fmaxf(fmaf(a.y, sc.y, sh.y), fmaf(b.y, sc.y, sh.y));
        o.z = fmaxf(fmaf(a.z, sc.z, sh.z), fmaf(b.z, sc.z, sh.z));
        o.w = fmaxf(fmaf(a.w, sc.w, sh.w), fmaf(b.w, sc.w, sh.w));
        if (y) *reinterpret_cast<float4*>(y + r * ldy + cc) = o;
        if (y16) svcmi_store4_16(y16 + r * ldy16 + cc, ldy16 >> 1, o.x, o.y, o.z, o.w, fmt16);
    }
}

// Viterbi decoding of a pitch posteriorgram (crepe/decode.py:53-80 -> librosa.sequence.viterbi): per frame, softmax over
// the allowed bins of the network's (sigmoid) outputs as torch does it in fp32, log-likelihood log(p + tiny); then the
// dynamic programme over S = 360 states in fp64 -- one 384-thread block per decoding batch, thread k owns state k,
// the running values live in LDS, back-pointers in global memory, thread 0 walks them back.  Ties resolve to the lowest
// state index like numpy's argmax.
constexpr int VS = 360;

__global__ __launch_bounds__(64) void viterbi_loglik_kernel(const float* prob, float* lp, int minidx, int maxidx) {
    const int t = blockIdx.x, lane = threadIdx.x;
    const float* pr = prob + (long long)t * VS;
    float m = -3.0e38f;
    for (int s = minidx + lane; s < maxidx; s += 64) m = fmaxf(m, pr[s]);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    float sum = 0.f;
    for (int s = minidx + lane; s < maxidx; s += 64) sum += expf(pr[s] - m);
#pragma unroll
    for (int k = 32; k >= 1; k >>= 1) sum += __shfl_xor(sum, k);
    for (int s = lane; s < VS; s += 64) {
        const float p = (s >= minidx && s < maxidx) ? expf(pr[s] - m) / sum : 0.f;
        lp[(long long)t * VS + s] = logf(p + 1.17549435e-38f);
    }
}

__global__ __launch_bounds__(384) void viterbi_dp_kernel(const float* lp, const double* log_trans, short* ptr, int* path, int t_total,
                                                         int batch_frames) {
    __shared__ double val[2][VS];
    __shared__ int best;
    const int k = threadIdx.x;
    const int f0 = blockIdx.x * batch_frames;
    const int T = (t_total - f0) < batch_frames ? (t_total - f0) : batch_frames;
    const float* lpb = lp + (long long)f0 * VS;
    short* pb = ptr + (long long)f0 * VS;
    if (k < VS) val[0][k] = (double)lpb[k] + log(1.0 / VS + 2.2250738585072014e-308);
    __syncthreads();
    for (int t = 1; t < T; ++t) {
        const double* prev = val[(t - 1) & 1];
        if (k < VS) {
            double bv = prev[0] + log_trans[k];
            int bj = 0;
            for (int j = 1; j < VS; ++j) {
                const double c = prev[j] + log_trans[(long long)j * VS + k];
                if (c > bv) { bv = c; bj = j; }
            }
            val[t & 1][k] = (double)lpb[(long long)t * VS + k] + bv;
            pb[(long long)t * VS + k] = (short)bj;
        }
        __syncthreads();
    }
    if (k == 0) {
        const double* last = val[(T - 1) & 1];
        int bj = 0;
        for (int j = 1; j < VS; ++j)
            if (last[j] > last[bj]) bj = j;
        best = bj;
        int cur = bj;
        path[f0 + T - 1] = cur;
        for (int t = T - 2; t >= 0; --t) {
            cur = pb[(long long)(t + 1) * VS + cur];
            path[f0 + t] = cur;
        }
    }
}

// The same dynamic programme for a BANDED transition matrix (CREPE's: weight max(12 - |i - j|, 0), i.e. |i - j| <= 11 in band,
// everything else log(0 + tiny) = one constant).  Per frame a state compares its 2*band + 1 in-band predecessors plus the
// best out-of-band one, which is max(prev[0 .. k-band-1]) or max(prev[k+band+1 ..]) + that constant: the two running maxima
// (value + lowest index attaining it) come from a prefix and a suffix scan done by waves 0 and 1 (6 states per lane, then a
// shuffle scan over the lane totals).  25 candidates per state instead of 360: the dense kernel took 9.0 ms for the 501
// frames of a 10 s clip.  Candidates are visited in ascending predecessor order with strict >, so ties resolve to the
// lowest index exactly like the dense loop (sums that only become equal through rounding of `+ constant` excepted).
constexpr int VBAND_MAX = 15;
constexpr int VBT_THREADS = 512;  // banded kernel: 2 scanning waves + 6 waves of states
constexpr int VBT = 64;             // frames per backtracking chunk (64 x 360 back-pointers = 45 KB of LDS)
constexpr double VNEG = -1.0e300;

__global__ __launch_bounds__(VBT_THREADS) void viterbi_dp_banded_kernel(const float* lp, const double* log_trans, short* ptr, int* path,
                                                                int t_total, int batch_frames, int band) {
    __shared__ double val[2][VS];
    __shared__ double pm_v[VS], sm_v[VS];
    __shared__ short pm_i[VS], sm_i[VS];
    __shared__ short pbuf[VBT][VS];          // back-pointer rows of one backtracking chunk
    __shared__ int cur_s;
    // 8 waves: waves 0 and 1 run the prefix / suffix max-scans of a step, waves 2-7 own the 360 states (thread 128 + k = state k) and
    // compute their in-band maxima BESIDE the scans; after the barrier a state only compares its in-band result with the two out-of-band
    // candidates.  (With 6 waves the two scanning waves also owned states: scan + in-band + combine were one serial chain per step --
    // 1.43 -> 1.00 ms for the 501 frames of a 10 s clip; the lane scans on DPP row moves instead of ds_bpermute shuffles: 0.93 ms,
    // profiles/r03c_crepe_kernel_stats.log.)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k = tid >= 128 ? tid - 128 : VS;             // the state this thread owns (VS = none)
    const int f0 = blockIdx.x * batch_frames;
    const int T = (t_total - f0) < batch_frames ? (t_total - f0) : batch_frames;
    const float* lpb = lp + (long long)f0 * VS;
    short* pb = ptr + (long long)f0 * VS;
    const double c_out = log_trans[VS - 1];                  // row 0, column VS-1: out of band by construction
    double ltb[2 * VBAND_MAX + 1];                           // log_trans[j][k] for j = k - band .. k + band
#pragma unroll
    for (int e = 0; e < 2 * VBAND_MAX + 1; ++e) {
        const int j = k - band + e;
        ltb[e] = (k < VS && e <= 2 * band && j >= 0 && j < VS) ? log_trans[(long long)j * VS + k] : 0.0;
    }
    if (k < VS) val[0][k] = (double)lpb[k] + log(1.0 / VS + 2.2250738585072014e-308);
    // (the log-likelihood of step t is requested one step ahead instead of as a dependent global load inside the step; with the LDS-staged
    //  backtracking below: 1.59 -> 1.50 ms for 501 frames -- most of a step is the two 64-lane fp64 max-scans)
    float lp_next = (k < VS && T > 1) ? lpb[VS + k] : 0.f;
    __syncthreads();
    for (int t = 1; t < T; ++t) {
        const double* prev = val[(t - 1) & 1];
        const float lp_t = lp_next;
        if (k < VS && t + 1 < T) lp_next = lpb[(long long)(t + 1) * VS + k];
        double bv_in = VNEG; int bj_in = 0;     // best in-band predecessor of state k (lowest index on ties)
        if (wave == 0) {                 // prefix maxima: pm[j] = max prev[0..j], lowest index on ties
            const int j0 = 6 * lane;
            double v = VNEG; int vi = 0;
#pragma unroll
            for (int e = 0; e < 6; ++e)
                if (j0 + e < VS && prev[j0 + e] > v) { v = prev[j0 + e]; vi = j0 + e; }
            // inclusive scan over the lanes on DPP row moves (svcmi_rt.h); the moved value is the EARLIER range, which wins ties
#define VSCAN_STEP(MOVE)                                                                   \
            {                                                                              \
                const double ov = svcmi_dpp_f64(v, [](int q) { return MOVE(q); });         \
                const int oi = MOVE(vi);                                                   \
                if (!(v > ov)) { v = ov; vi = oi; }                                        \
            }
            VSCAN_STEP(svcmi_dpp_row_shr<1>) VSCAN_STEP(svcmi_dpp_row_shr<2>) VSCAN_STEP(svcmi_dpp_row_shr<4>) VSCAN_STEP(svcmi_dpp_row_shr<8>)
            VSCAN_STEP(svcmi_dpp_row_bcast15) VSCAN_STEP(svcmi_dpp_row_bcast31)
#undef VSCAN_STEP
            double xv = svcmi_dpp_f64(v, [](int q) { return svcmi_dpp_wave_shr1(q); });     // exclusive prefix of this lane
            int xi = svcmi_dpp_wave_shr1(vi);
            if (lane == 0) { xv = VNEG; xi = 0; }
#pragma unroll
            for (int e = 0; e < 6; ++e)
                if (j0 + e < VS) {
                    if (prev[j0 + e] > xv) { xv = prev[j0 + e]; xi = j0 + e; }
                    pm_v[j0 + e] = xv; pm_i[j0 + e] = (short)xi;
                }
        } else if (wave == 1) {          // suffix maxima: sm[j] = max prev[j..VS-1], lowest index on ties
            // lane l owns the 6 states of block 59 - l, so that a scan towards HIGHER lanes walks towards LOWER states: the suffix scan
            // becomes the same forward lane scan as above (lanes 60-63 own nothing and come last)
            const int j0 = 6 * (59 - lane);
            double v = VNEG; int vi = VS - 1;
#pragma unroll
            for (int e = 5; e >= 0; --e)
                if (j0 >= 0 && prev[j0 + e] >= v) { v = prev[j0 + e]; vi = j0 + e; }
            // the moved value is the range of HIGHER states: this lane's (lower) states win ties
#define VSCAN_STEP(MOVE)                                                                   \
            {                                                                              \
                const double ov = svcmi_dpp_f64(v, [](int q) { return MOVE(q); });         \
                const int oi = MOVE(vi);                                                   \
                if (ov > v) { v = ov; vi = oi; }                                           \
            }
            VSCAN_STEP(svcmi_dpp_row_shr<1>) VSCAN_STEP(svcmi_dpp_row_shr<2>) VSCAN_STEP(svcmi_dpp_row_shr<4>) VSCAN_STEP(svcmi_dpp_row_shr<8>)
            VSCAN_STEP(svcmi_dpp_row_bcast15) VSCAN_STEP(svcmi_dpp_row_bcast31)
#undef VSCAN_STEP
            double xv = svcmi_dpp_f64(v, [](int q) { return svcmi_dpp_wave_shr1(q); });     // maximum over the states above this lane's block
            int xi = svcmi_dpp_wave_shr1(vi);
            if (lane == 0) { xv = VNEG; xi = VS - 1; }
#pragma unroll
            for (int e = 5; e >= 0; --e)
                if (j0 >= 0) {
                    if (prev[j0 + e] >= xv) { xv = prev[j0 + e]; xi = j0 + e; }
                    sm_v[j0 + e] = xv; sm_i[j0 + e] = (short)xi;
                }
        }
        else if (k < VS) {
#pragma unroll
            for (int e = 0; e < 2 * VBAND_MAX + 1; ++e) {
                const int j = k - band + e;
                if (e <= 2 * band && j >= 0 && j < VS) {
                    const double c = prev[j] + ltb[e];
                    if (c > bv_in) { bv_in = c; bj_in = j; }
                }
            }
        }
        __syncthreads();
        if (k < VS) {       // candidates in increasing predecessor order, strict '>' : the lowest predecessor index wins ties
            double bv = VNEG; int bj = 0;
            if (k - band - 1 >= 0) { bv = pm_v[k - band - 1] + c_out; bj = pm_i[k - band - 1]; }
            if (bv_in > bv) { bv = bv_in; bj = bj_in; }
            if (k + band + 1 < VS) {
                const double c = sm_v[k + band + 1] + c_out;
                if (c > bv) { bv = c; bj = sm_i[k + band + 1]; }
            }
            val[t & 1][k] = (double)lp_t + bv;
            pb[(long long)t * VS + k] = (short)bj;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const double* last = val[(T - 1) & 1];
        int bj = 0;
        for (int j = 1; j < VS; ++j)
            if (last[j] > last[bj]) bj = j;
        cur_s = bj;
        path[f0 + T - 1] = bj;
    }
    __syncthreads();
    // Backtracking: the walk is one thread following T - 1 dependent back-pointers; from global memory that is one L2 round trip per
    // frame.  The rows are staged through LDS VBT frames at a time by the whole block (one contiguous copy), the walk reads LDS.
    for (int t_hi = T - 1; t_hi >= 1; t_hi -= VBT) {
        const int t_lo = t_hi - VBT + 1 > 1 ? t_hi - VBT + 1 : 1;          // rows t_lo .. t_hi
        const int n = (t_hi - t_lo + 1) * VS;
        const short* src = pb + (long long)t_lo * VS;
        short* dst = &pbuf[0][0];
        for (int i = tid; i < n; i += VBT_THREADS) dst[i] = src[i];
        __syncthreads();
        if (tid == 0) {
            int cur = cur_s;
            for (int t = t_hi; t >= t_lo; --t) {
                cur = pbuf[t - t_lo][cur];
                path[f0 + t - 1] = cur;
            }
            cur_s = cur;
        }
        __syncthreads();
    }
}

}  // namespace

extern "C" int svcmi_viterbi_decode(const float* prob, const double* log_trans, float* lp_scratch, int16_t* ptr_scratch, int32_t* path,
                                    int32_t frames, int32_t batch_frames, int32_t minidx, int32_t maxidx, int32_t band, void* stream) {
    if (!prob || !log_trans || !lp_scratch || !ptr_scratch || !path || frames <= 0 || batch_frames <= 0) return SVCMI_EINVAL;
    if (minidx < 0 || maxidx > VS || minidx >= maxidx || band < 0) return SVCMI_EINVAL;
    if (band > VBAND_MAX) return SVCMI_EUNSUPPORTED;
    SVCMI_LAUNCH(viterbi_loglik_kernel, dim3((unsigned)frames), dim3(64), 0, stream, prob, lp_scratch, minidx, maxidx);
    int rc = SVCMI_LAST_ERROR();
    if (rc) return rc;
    const dim3 grid((unsigned)((frames + batch_frames - 1) / batch_frames));
    if (band > 0 && 2 * band + 1 < VS)
        SVCMI_LAUNCH(viterbi_dp_banded_kernel, grid, dim3(VBT_THREADS), 0, stream, (const float*)lp_scratch, log_trans, (short*)ptr_scratch, path,
                     frames, batch_frames, band);
    else
        SVCMI_LAUNCH(viterbi_dp_kernel, grid, dim3(384), 0, stream, (const float*)lp_scratch, log_trans, (short*)ptr_scratch, path, frames,
                     batch_frames);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_crepe_frames_f32(const float* audio, int64_t n, int32_t hop, int32_t frame0, int32_t frames, float* out, int32_t ld, void* stream) {
    if (!audio || !out || n <= 0 || hop <= 0 || frame0 < 0 || frames <= 0) return SVCMI_EINVAL;
    if (ld < CREPE_LEAD + CREPE_WIN + CREPE_LEAD || ld % 4) return SVCMI_EINVAL;
    SVCMI_LAUNCH(crepe_frames_kernel, dim3((unsigned)frames), dim3(TPB), 0, stream, audio, (long long)n, hop, frame0, out, ld);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_bn_maxpool2_f32(const float* x, const float* scale, const float* shift, float* y, int64_t rows_out, int32_t c,
                                     int32_t ldx, int32_t ldy, void* y16, int32_t ldy16, int32_t y16_format, void* stream) {
    if (!x || !scale || !shift || (!y && !y16) || rows_out <= 0 || c <= 0) return SVCMI_EINVAL;
    if (y16 && (svcmi_fmt16(y16_format) < 0 || !svcmi_fmt16_row_ok(y16_format, ldy16, c) || ((uintptr_t)y16 & 7))) return SVCMI_EINVAL;
    if (c % 4 || ldx % 4 || ldy % 4 || ((uintptr_t)x & 15) || ((uintptr_t)y & 15) || ((uintptr_t)scale & 15) || ((uintptr_t)shift & 15))
        return SVCMI_EALIGN;
    long long nb = (rows_out * (c / 4) + TPB - 1) / TPB;
    if (nb > 8192) nb = 8192;
    SVCMI_LAUNCH(bn_maxpool2_kernel, dim3((unsigned)nb), dim3(TPB), 0, stream, x, scale, shift, y, (long long)rows_out, c, ldx, ldy,
                 static_cast<unsigned short*>(y16), ldy16, y16 ? svcmi_fmt16(y16_format) : 0);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_reflect_pad_f32(const float* x, float* y, int32_t batch, int64_t n, int32_t pad, void* stream) {
    if (!x || !y || batch <= 0 || n <= 0 || pad < 0 || pad >= n) return SVCMI_EINVAL;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    long long nb = (n + 2LL * pad + TPB - 1) / TPB;
    if (nb > 4096) nb = 4096;
    SVCMI_LAUNCH(reflect_pad_kernel, dim3((unsigned)nb, batch), dim3(TPB), 0, stream, x, y, (long long)n, pad);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_power_spectrum_f32(const float* ri, float* p, int64_t rows, int32_t nbins, int32_t half, int32_t ldri, int32_t ldp, void* stream) {
    if (!ri || !p || rows <= 0 || nbins <= 0 || half < nbins || ldri < half + nbins || ldp < nbins) return SVCMI_EINVAL;
    long long nb = (rows * ldp + TPB - 1) / TPB;
    if (nb > 4096) nb = 4096;
    SVCMI_LAUNCH(power_kernel, dim3((unsigned)nb), dim3(TPB), 0, stream, ri, p, (long long)rows, nbins, half, ldri, ldp);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_logmel_finish_f32(float* mel_power, float* scratch, float* out, int32_t batch, int32_t t, int32_t c, void* stream) {
    // mel_power: [batch][t][c] (contiguous, overwritten with its log10); scratch: >= batch*64 floats; out: [batch][c][t]
    if (!mel_power || !scratch || !out || batch <= 0 || t <= 0 || c <= 0) return SVCMI_EINVAL;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    const long long per_item = (long long)t * c;
    int nblk = (int)((per_item + TPB * 8 - 1) / (TPB * 8));
    if (nblk > 64) nblk = 64;
    if (nblk < 1) nblk = 1;
    SVCMI_LAUNCH(log10_blockmax_kernel, dim3(nblk, batch), dim3(TPB), 0, stream, mel_power, scratch, per_item);
    int rc = SVCMI_LAST_ERROR();
    if (rc) return rc;
    SVCMI_LAUNCH(logmel_finish_kernel, dim3((t + 31) / 32, (c + 31) / 32, batch), dim3(TPB), 0, stream, mel_power, scratch, nblk, out, t, c, c);
    return SVCMI_LAST_ERROR();
}
