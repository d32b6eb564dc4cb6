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

}  // namespace

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
