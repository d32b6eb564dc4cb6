// amp_block.hip -- a whole AMP block of the narrow BigVGAN stages (20 and 10 channels) as ONE kernel:
//     for q in (dilations 1, 3, 5):  x = x + conv2_q( SnakeAlias( conv1_q( SnakeAlias(x) ) ) )          (vits_decoder/bigv.py:50-58)
// with the time tile resident in LDS from the stage input to the block output.  The half-step kernels of amp_fused.hip do one
// `conv(SnakeAlias(x)) + res` per launch: six launches per block, each reading x (+ res) and writing y through HBM -- 115 MB per grouped
// launch at 10 channels, and PMC showed them WAITING (38-43 % of wave-cycles parked on memory), on neither the VALU nor the HBM roofline.
// Here a block's six half-steps run back to back on one tile; HBM sees the stage input once and the block output once.
//
// Tile: R = 64 * NG rows of LDS, row r <-> time t_blk - H + r, where H = sum over the iterations of 10 + (k-1)/2 * (d + 1) is the halo the
// chain consumes per side (k = 11: 20 + 30 + 40 = 90 samples; k = 7: 66; k = 3: 42).  Every phase shrinks the valid range by its own
// halo -- SnakeAlias 5 rows (6-tap polyphase up-sampler + 12-tap decimator at 2x), convolution (k-1)/2 * d -- and the TB = R - 2H rows
// that survive are the block's output: ~1.1-1.3x recomputation instead of 54 HBM round trips per stage.
//   A [R][LS]  the pre-activation tensor of the current phase (x, then conv1's output, then the new x),
//   S [R][LS]  its SnakeAlias image (zero outside the sequence = the convolutions' zero padding),
//   X          the residual x of the thread's own (row, channel) outputs, in REGISTERS across the iteration (fixed ownership).
// Phases (one barrier each): S = act1(A); A = conv1(S); S = act2(A); A = X = conv2(S) + X.   13 barriers per tile.
// Arithmetic per value is EXACTLY that of the half-step kernels (snake_math.h's operation sequence; bias-initialised accumulator, taps
// in order, channels in order, fmaf), so the fused block equals the six launches bit for bit -- tests compare with torch.equal.
//
// Work split: the convolution is the direct VALU form of amp_fused.hip (lane = time row, CO = 10 output channels per thread, weights as
// SGPR operands through the scalar cache, conflict-free ds_read_b128 rows); a wave owns TT row groups of 64: TT = 1 -> group `tsub`,
// TT = 2 -> the MIRROR pair {tsub, NG-1-tsub}, which leaves the valid range together as it shrinks from both ends, so a wave is either
// fully busy or idle (idle waves give their SIMD slots to the co-resident workgroup).  wave -> tsub is rotated by blockIdx so that the
// outer (early-idle) groups of co-resident workgroups sit on different SIMDs.  SnakeAlias phases spread (channel x run of 8 rows) items
// over all threads.
#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

constexpr int RT = 10;       // SnakeAlias outputs per work item: with one thread per tile row, (rows x channels) / RT items are 1 (10 channels) or 2 (20) full rounds of the workgroup
constexpr int CO = 10;       // output channels per thread
constexpr int GUARD = 28;    // rows before / after each LDS tile that out-of-range convolution taps of rows nobody uses may touch (>= 5 * 5, multiple of 4)
constexpr int NBLK = 3;

#include "snake_math.h"

struct BlockArgs {           // one AMP block (device pointers; arrays indexed by the dilation step q)
    const float* x; float* y;
    const float* w1[3]; const float* b1[3]; const float* w2[3]; const float* b2[3];
    const float* a1a[3]; const float* a1b[3]; const float* a2a[3]; const float* a2b[3];
    int ldw1[3], ldw2[3], dil[3];
    int ks, n_dil;
};
struct StageArgs {
    BlockArgs blk[NBLK];
    const float* filt;
    int n, ld;
};

__host__ __device__ inline int block_halo(int ks, const int* dil, int n_dil) {
    int h = 0;
    for (int q = 0; q < n_dil; ++q) h += 10 + (ks - 1) / 2 * (dil[q] + 1);
    return h;
}

// S rows [r_lo, r_hi) = SnakeAlias of the A rows around them; zero for times outside [0, n).  A row r <-> time t_lo + r.
template <int CR, int LS, int NT>
__device__ __forceinline__ void snake_phase(const float* A, float* S, const float* alpha_log, const float* beta_log, const float (&f)[12],
                                            int n, int t_lo, int r_lo, int r_hi, int tid) {
    const int runs = (r_hi - r_lo + RT - 1) / RT;
    const float* At = A - (long long)t_lo * LS;            // indexed by TIME: At[t * LS + ch]
    for (int item = tid; item < runs * CR; item += NT) {
        const int ch = item % CR, run = item / CR;
        const int r0 = r_lo + run * RT;
        const int t0 = t_lo + r0;
        float out[RT];
        if (t0 + RT <= 0 || t0 >= n) {
#pragma unroll
            for (int r = 0; r < RT; ++r) out[r] = 0.f;
        } else {
            const float a = expf(alpha_log[ch]);
            const float inv_b = 1.0f / (expf(beta_log[ch]) + 1e-9f);
            const float* xc = At + ch;
            SnakeWindow<RT + 10> xw;
#pragma unroll
            for (int i = 0; i < RT + 10; ++i) xw.set(i, xc[clampi(t0 - 5 + i, 0, n - 1) * LS]);
            snake_run<RT>(xw, f, a, inv_b, xc, LS, n, t0, out);
#pragma unroll
            for (int r = 0; r < RT; ++r) {
                const int t = t0 + r;
                if (t < 0 || t >= n) out[r] = 0.f;      // the convolution's zero padding
            }
        }
#pragma unroll
        for (int r = 0; r < RT; ++r)
            if (r0 + r < r_hi) S[(r0 + r) * LS + ch] = out[r];
    }
}

// acc[j][c] = bias[co0 + c] + sum_tap sum_ci w[co0 + c][tap * CP + ci] * S[row_j + (tap - hk) * d][ci] for the TT rows of this lane
template <int CP, int CR, int LS, int TT>
__device__ __forceinline__ void conv_rows(const float* S, const float* w, const float* bias, int ldw, int ks, int d, int co0,
                                          const int (&row)[TT], float (&acc)[TT][CO]) {
#pragma unroll
    for (int c = 0; c < CO; ++c) {
        const float bv = (co0 + c < CR && bias) ? svcmi_load_uniform1(bias + co0 + c) : 0.f;
#pragma unroll
        for (int j = 0; j < TT; ++j) acc[j][c] = bv;
    }
    const float* wg = w + (long long)co0 * ldw;
    const int hk = (ks - 1) / 2;
    for (int tap = 0; tap < ks; ++tap) {
        const int off = (tap - hk) * d * LS;
        const float* wt = wg + tap * CP;
#pragma unroll
        for (int c4 = 0; c4 < (CR + 3) / 4; ++c4) {
            float4 xin[TT];
#pragma unroll
            for (int j = 0; j < TT; ++j) xin[j] = *reinterpret_cast<const float4*>(S + row[j] * LS + off + 4 * c4);
            // the quad's base address made opaque: knowing that the quads of a weight row are contiguous, the compiler merges them into
            // s_load_dwordx16 per output channel (160-200 SGPRs per tap) and spills through v_writelane / v_readlane
            const float* wq = svcmi_opaque_uniform(wt + 4 * c4);
#pragma unroll
            for (int c = 0; c < CO; ++c) {
                if (co0 + c < CR) {                    // wave-uniform
                    const svcmi_f32x4 wv = svcmi_load_uniform4(wq + (long long)c * ldw);          // uniform address, read-only: scalar load
#pragma unroll
                    for (int j = 0; j < TT; ++j) {
                        acc[j][c] = fmaf(wv[0], xin[j].x, acc[j][c]);
                        if (4 * c4 + 1 < CR) acc[j][c] = fmaf(wv[1], xin[j].y, acc[j][c]);
                        if (4 * c4 + 2 < CR) acc[j][c] = fmaf(wv[2], xin[j].z, acc[j][c]);
                        if (4 * c4 + 3 < CR) acc[j][c] = fmaf(wv[3], xin[j].w, acc[j][c]);
                    }
                }
            }
            // one input-channel quad at a time: 10 weight quads = 40 SGPRs in flight.  Left alone, the scheduler hoists the scalar loads of
            // a whole tap (30 / 50 quads = 120 / 200 SGPRs) above the FMAs and spills them through v_writelane / v_readlane
            SVCMI_SCHED_BARRIER();
        }
    }
}

// CP = padded channels (row length in HBM and LDS), CR = real channels, G = channel groups of CO, NW = waves per workgroup,
// NG = row groups of 64 per tile, TT = row groups per wave (1: NG == NW / G; 2: NG == 2 * NW / G, mirror pairs).
template <int CP, int CR, int G, int NW, int NG, int TT>
__global__ __launch_bounds__(64 * NW) void amp_block_kernel(StageArgs a) {
    constexpr int NT = 64 * NW, TSUB = NW / G, R = 64 * NG;
    constexpr int LS = (CP / 4) % 2 ? CP : CP + 4;          // 4 * odd floats: conflict-free ds_read_b128 rows
    static_assert(G * CO >= CR && CP % 4 == 0 && CP >= CR && NW % G == 0 && TSUB * TT == NG && (TT == 1 || TT == 2), "tile geometry");
    __shared__ __attribute__((aligned(16))) float smem[2 * (R + 2 * GUARD) * LS];
    float* const A = smem + GUARD * LS;
    float* const S = smem + (R + 2 * GUARD) * LS + GUARD * LS;

    const BlockArgs& p = a.blk[blockIdx.z];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = SVCMI_UNIFORM((int)(tid >> 6));
    const int n = a.n, ld = a.ld, b = blockIdx.y;
    const int ks = p.ks, hk = (ks - 1) / 2;
    const int H = block_halo(ks, p.dil, p.n_dil);
    const int TB = R - 2 * H;                              // output rows of a tile of THIS block (the grid is sized for the longest kernel)
    const int t_blk = blockIdx.x * TB;
    if (t_blk >= n) return;
    const int t_lo = t_blk - H;                            // time of LDS row 0
    const float* xb = p.x + (long long)b * n * ld;

    float f[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) f[k] = a.filt[k];

    // ---- load: A row r = x[t_lo + r] (zero outside the sequence: never read -- the SnakeAlias clamps its taps to [0, n))
    for (int e = tid; e < R * (CP / 4); e += NT) {
        const int r = e / (CP / 4), c4 = e - r * (CP / 4);
        const int t = t_lo + r;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (t >= 0 && t < n) v = *reinterpret_cast<const float4*>(xb + (long long)t * ld + 4 * c4);
        *reinterpret_cast<float4*>(A + r * LS + 4 * c4) = v;
    }
    __syncthreads();

    // ---- ownership of convolution outputs: channel group g, row groups grp[j]
    const int g = wave % G;
    const int tsub = (wave / G + (int)blockIdx.x) % TSUB;  // rotated: co-resident workgroups keep their early-idle waves on different SIMDs
    const int co0 = g * CO;
    int row[TT];
    row[0] = 64 * tsub + lane;
    if constexpr (TT == 2) row[1] = 64 * (NG - 1 - tsub) + lane;
    float X[TT][CO];
#pragma unroll
    for (int j = 0; j < TT; ++j)
#pragma unroll
        for (int c = 0; c < CO; ++c) X[j][c] = (co0 + c < CP) ? A[row[j] * LS + co0 + c] : 0.f;

    int h = 0;
    for (int q = 0; q < p.n_dil; ++q) {
        const int d = p.dil[q];
        // S = act1(A)
        h += 5;
        snake_phase<CR, LS, NT>(A, S, p.a1a[q], p.a1b[q], f, n, t_lo, h, R - h, tid);
        __syncthreads();
        // A = conv1(S), dilation d
        h += hk * d;
        {
            // a wave whose row groups lie outside [h, R - h) has nothing anyone will read (TT == 2: the mirror pair leaves together)
            const bool live = 64 * tsub + 64 > h && 64 * tsub < R - h;
            if (live) {
                float acc[TT][CO];
                conv_rows<CP, CR, LS, TT>(S, p.w1[q], p.b1[q], p.ldw1[q], ks, d, co0, row, acc);
#pragma unroll
                for (int j = 0; j < TT; ++j)
#pragma unroll
                    for (int c = 0; c < CO; c += 2)
                        if (co0 + c < CR) *reinterpret_cast<float2*>(A + row[j] * LS + co0 + c) = make_float2(acc[j][c], acc[j][c + 1]);
            }
        }
        __syncthreads();
        // S = act2(A)
        h += 5;
        snake_phase<CR, LS, NT>(A, S, p.a2a[q], p.a2b[q], f, n, t_lo, h, R - h, tid);
        __syncthreads();
        // A = X = conv2(S) + X, dilation 1 (the last iteration's X is the block output: stored after the loop)
        h += hk;
        {
            const bool live = 64 * tsub + 64 > h && 64 * tsub < R - h;
            const bool last = q == p.n_dil - 1;
            if (live) {
                float acc[TT][CO];
                conv_rows<CP, CR, LS, TT>(S, p.w2[q], p.b2[q], p.ldw2[q], ks, 1, co0, row, acc);
#pragma unroll
                for (int j = 0; j < TT; ++j) {
#pragma unroll
                    for (int c = 0; c < CO; ++c) X[j][c] = (co0 + c < CR) ? acc[j][c] + X[j][c] : 0.f;      // (+ res, alpha = 1: the half-step's epilogue)
                    if (!last) {
#pragma unroll
                        for (int c = 0; c < CO; c += 2)
                            if (co0 + c < CR) *reinterpret_cast<float2*>(A + row[j] * LS + co0 + c) = make_float2(X[j][c], X[j][c + 1]);
                    }
                }
            }
        }
        if (q + 1 < p.n_dil) __syncthreads();
    }
    // ---- block output: the rows [H, R - H) of the tile.  The ONLY global store of the kernel, and it comes after every load: inside the
    // loop it would make the compiler treat the (uniform) weight loads as clobberable and fetch them through the vector memory path
    // (global_load + vmcnt(0) per tap: measured 2x slower than the half-step kernels) instead of the scalar cache.
#pragma unroll
    for (int j = 0; j < TT; ++j) {
        const int t = t_lo + row[j];
        if (row[j] >= H && row[j] < R - H && t < n) {
            float* yr = p.y + ((long long)b * n + t) * ld + co0;
#pragma unroll
            for (int c = 0; c < CO; c += 2) *reinterpret_cast<float2*>(yr + c) = make_float2(X[j][c], X[j][c + 1]);
            if (g == G - 1) {              // pad channels [G*CO, CP) stay exactly zero
#pragma unroll
                for (int c = G * CO; c < CP; c += 2) *reinterpret_cast<float2*>(p.y + ((long long)b * n + t) * ld + c) = make_float2(0.f, 0.f);
            }
        }
    }
}

int g_variant = 0;      // tuning knob "amp_block_variant": 0 = default per width, 1.. = the instantiations below

}  // namespace

extern "C" int svcmi_amp_block_group_supported(int32_t c, int32_t ld) { return (c == 10 && ld == 12) || (c == 20 && ld == 20); }

extern "C" int svcmi_amp_block_tune_set(const char* name, int32_t value) {
    const char* k = "amp_block_variant";
    int i = 0;
    while (k[i] && name[i] == k[i]) ++i;
    if (k[i] == 0 && name[i] == 0 && value >= 0 && value <= 4) { g_variant = value; return 0; }
    return SVCMI_EINVAL;
}

#define AMP_LAUNCH(CP, CR, G, NW, NG, TT)                                                                                     \
    do {                                                                                                                      \
        constexpr int R_ = 64 * (NG);                                                                                         \
        const int tb = R_ - 2 * hmax;                                                                                         \
        if (tb < 32) return SVCMI_EUNSUPPORTED;                                                                               \
        dim3 grid((unsigned)((len + tb - 1) / tb), (unsigned)batch, (unsigned)count);                                         \
        SVCMI_LAUNCH((amp_block_kernel<CP, CR, G, NW, NG, TT>), grid, dim3(64 * (NW)), 0, stream, a);                         \
        return SVCMI_LAST_ERROR();                                                                                            \
    } while (0)

extern "C" int svcmi_amp_block_group_f32(const svcmi_amp_block_desc* descs, int32_t count, const float* filt, int32_t batch, int32_t len,
                                         int32_t c, int32_t ld, void* stream) {
    if (!descs || !filt || count < 1 || count > NBLK || batch <= 0 || len <= 0) return SVCMI_EINVAL;
    if (!svcmi_amp_block_group_supported(c, ld)) return SVCMI_EUNSUPPORTED;
    if (batch > 65535) return SVCMI_EUNSUPPORTED;
    int order[NBLK] = {0, 1, 2};
    for (int i = 0; i < count; ++i)           // most taps first (blockIdx.z = 0 is dispatched first)
        for (int j = i + 1; j < count; ++j)
            if (descs[order[j]].ksize > descs[order[i]].ksize) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    StageArgs a;
    int hmax = 0;
    for (int i = 0; i < count; ++i) {
        const svcmi_amp_block_desc& d = descs[order[i]];
        BlockArgs& p = a.blk[i];
        if (!d.x || !d.y || d.x == d.y || d.n_dil < 1 || d.n_dil > 3) return SVCMI_EINVAL;
        if (d.ksize != 3 && d.ksize != 7 && d.ksize != 11) return SVCMI_EUNSUPPORTED;
        if (((uintptr_t)d.x & 15) || ((uintptr_t)d.y & 7)) return SVCMI_EALIGN;
        for (int q = 0; q < 3; ++q) {
            const int qq = q < d.n_dil ? q : 0;
            if (!d.w1[qq] || !d.w2[qq] || !d.a1_alpha[qq] || !d.a1_beta[qq] || !d.a2_alpha[qq] || !d.a2_beta[qq]) return SVCMI_EINVAL;
            if (d.dil[qq] < 1 || d.dil[qq] > 5) return SVCMI_EUNSUPPORTED;
            if (d.ldw1[qq] < d.ksize * ld || d.ldw1[qq] % 4 || d.ldw2[qq] < d.ksize * ld || d.ldw2[qq] % 4) return SVCMI_EINVAL;
            if (((uintptr_t)d.w1[qq] & 15) || ((uintptr_t)d.w2[qq] & 15)) return SVCMI_EALIGN;
            p.w1[q] = d.w1[qq]; p.b1[q] = d.b1[qq]; p.w2[q] = d.w2[qq]; p.b2[q] = d.b2[qq];
            p.a1a[q] = d.a1_alpha[qq]; p.a1b[q] = d.a1_beta[qq]; p.a2a[q] = d.a2_alpha[qq]; p.a2b[q] = d.a2_beta[qq];
            p.ldw1[q] = d.ldw1[qq]; p.ldw2[q] = d.ldw2[qq]; p.dil[q] = d.dil[qq];
        }
        p.x = d.x; p.y = d.y; p.ks = d.ksize; p.n_dil = d.n_dil;
        const int h = block_halo(p.ks, p.dil, p.n_dil);
        if (h > hmax) hmax = h;
    }
    for (int i = count; i < NBLK; ++i) a.blk[i] = a.blk[0];
    a.filt = filt; a.n = len; a.ld = ld;
    // Instantiations (scripts/microbench.py ampblock picks the defaults on hardware):
    //   10 channels (48-byte rows, G = 1):  v1 = 8 waves x 64 rows (R = 512, 54 KB: 2 workgroups per CU), v2 = 16 waves (R = 1024, 104 KB),
    //                                      v3 = 4 waves x mirror pairs (R = 512), v4 = 12 waves (R = 768, 79 KB: 2 per CU)
    //   20 channels (80-byte rows, G = 2):  v1 = 8 waves x mirror pairs (R = 512, 91 KB), v2 = 14 waves x mirror pairs (R = 896, 152 KB),
    //                                      v3 = 16 waves x one group (R = 512), v4 = 12 waves x one group (R = 384, 70 KB: 2 per CU)
    const int v = g_variant;
    const long long rows = (long long)batch * len;
    if (c == 10) {
        const int pick = v ? v : (rows >= 2000000 ? 2 : 1);
        if (pick == 2) AMP_LAUNCH(12, 10, 1, 16, 16, 1);
        if (pick == 3) AMP_LAUNCH(12, 10, 1, 4, 8, 2);
        if (pick == 4) AMP_LAUNCH(12, 10, 1, 12, 12, 1);
        AMP_LAUNCH(12, 10, 1, 8, 8, 1);
    }
    const int pick = v ? v : (rows >= 1000000 ? 2 : 1);
    if (pick == 2) AMP_LAUNCH(20, 20, 2, 14, 14, 2);
    if (pick == 3) AMP_LAUNCH(20, 20, 2, 16, 8, 1);
    if (pick == 4) AMP_LAUNCH(20, 20, 2, 12, 6, 1);
    AMP_LAUNCH(20, 20, 2, 8, 8, 2);
}
