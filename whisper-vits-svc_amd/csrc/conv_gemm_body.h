// conv_gemm_body.h -- implicit-GEMM 1-D convolution / linear layer on the gfx950 matrix cores: the kernel body shared by
// conv_gemm.hip (fp32 operands, v_mfma_f32_32x32x2_f32 / 16x16x4_f32) and conv_gemm_lp.hip (16-bit operands, see "Reduced
// precision" below).
//
// One kernel family serves every Conv1d / Linear / (polyphase) ConvTranspose1d of the path (svcmi.h).
// Time-major activations make the im2col matrix free: row t of the A operand is the contiguous
// span x[t*stride - pad ...][0:c_in] for dilation 1, and a gather of `ksize` row segments otherwise.
//
// Tiling (wave64, 256 threads = 2x2 waves):  block tile (64*WM) x (64*WN), wave tile (32*WM) x (32*WN)
// as WM x WN accumulators of v_mfma_f32_32x32x2_f32 (16 VGPR each).  K is walked in steps of 32:
// both operands are K-contiguous in HBM and go straight to LDS with 16-byte LDS-DMA
// (buffer_load_dwordx4 ... lds: no VGPR round trip, nothing to wait for until the tile is consumed), double
// buffered so the next tile's DMA flies under the current tile's MFMAs.  The LDS image is [row][32]
// with the eight 16-byte chunks of a row XOR-swizzled by ((row >> 1) & 7) -- applied on the SOURCE offset of the
// DMA and again on the fragment read -- so the 16-lane groups of ds_read_b128 hit 16 distinct 4-bank slots.
// K-order trick: a lane's ds_read_b128 returns 4 consecutive k; lanes 0-31 take k = 8s+0..3 and
// lanes 32-63 take k = 8s+4..7, so MFMA #j of sub-step s multiplies k-pairs (8s+j, 8s+4+j) -- a
// permutation of the K summation shared by A and B, i.e. the same dot product.
//
// Addressing: each operand is a raw buffer (x of this batch item, bounded at the first masked row; the weight
// matrix) and a lane's source is a 32-bit byte offset = row part (hoisted out of the K loop) + K part (wave
// uniform in CHUNK mode) -- one v_add per 1-KiB DMA piece per K-step.  Zero fill (padding taps, masked rows,
// ragged tile edges, the K tail) is the hardware's buffer range check: rows before the tensor give a negative
// (huge unsigned) offset, rows past its end run over num_records, and everything else that must read as zero
// adds a 2^30 sentinel; an out-of-range DMA lane writes zeros to LDS.  The staging code has no branches and no
// selects on loaded data, and its issue slots are spread over the sub-steps of the previous tile's MFMAs.
// Three instantiations of the A-gather keep the loop free of per-element integer division:
//   CHUNK : c_in % 32 == 0 -- a K-step lies inside one tap (scalar tap/channel bookkeeping);
//   VEC   : c_in % 4 == 0  -- one magic-number division per thread per K-step;
//   SCALAR: anything else (the 1-channel source convolutions) -- 4-byte DMA.
// (CHUNK_RS is CHUNK with the np.repeat(x, 2, 0) row shift fused into the gather; it recomputes rows per piece.)
//
// Blocks are enumerated M-tile fastest inside contiguous per-XCD ranges (block b runs on XCD b % 8), so all the
// M tiles that consume one weight tile share an L2 instead of pulling it into all eight.
//
// Split-K (deterministic): when the tile grid cannot fill the 256 CUs (M = 500 Whisper rows against
// N = 1280, or the N = 192 prior-encoder convs with K = 6400) the grid also enumerates S slices of the
// K range; slices write raw partial tiles to a caller-provided workspace and a second small kernel adds
// them in fixed order and applies the epilogue.  No atomics anywhere: results are run-to-run identical.
//
// Reduced precision (PREC != 0, svcmi_conv_gemm_lp): activations stay fp32 in HBM and travel to LDS exactly as above; only the
// weights change format.  They are packed once (svcmi_pack_weights_lp) to a 16-bit image [n_out][ldw16] -- bf16, bf16 hi + lo
// (bf16x3) or fp16 -- whose K order inside every block of 32 is permuted so that 16-byte chunk q (8 values) holds
// k = 4q..4q+3 and 16+4q..16+4q+3: a lane's MFMA fragment (8 consecutive-in-LDS weights = ONE ds_read_b128) then pairs with
// the fp32 A chunks q and q+4 (two ds_read_b128, both conflict-free under the fp32 swizzle), which are rounded to 16 bit
// in registers (v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32) right before the MFMA:
//   bf16x3: x = hi + lo with hi = bf16(x), lo = bf16(x - hi) on BOTH operands and acc += hi*hi + lo*hi + hi*lo (fp32
//           accumulate; the dropped lo*lo term is 2^-18 relative) -- three v_mfma_f32_32x32x16_bf16 at 16x the fp32 rate;
//   bf16 / f16: one MFMA on the rounded operands.
// The B tile is [rows][32 k] x 2 bytes = 64-byte rows, four 16-byte chunks XOR-swizzled by fB(row) = {0,2,3,1}[(row>>2)&3]
// (conflict-free for the 16-lane groups of ds_read_b128 under both fragment patterns), 16 rows per 1-KiB DMA piece.
// Tile geometry, A gather, split-K, grouping and the epilogue are the fp32 kernel's; 32x32x2 becomes 32x32x16 (2 sub-steps
// per K-step of 32), 16x16x4 becomes 16x16x32 (1 sub-step).
#pragma once
#include <type_traits>

#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

// Build switch of the fp32 K loop: 1 (default since round 5: judged line 1429-1433 -> 1446-1447 audio-s/s, one clip alone 9.48 -> 9.41 ms,
// profiles/r05p_midbar.log) = the mid-barrier pipeline (conv_gemm_body, "MIDBAR"), 0 = barrier at the top of every K-step (what the
// 16-bit kernels keep)
#ifndef SVCMI_GEMM_MIDBAR
#define SVCMI_GEMM_MIDBAR 1
#endif
// Build switch of the mid-barrier loop's instruction placement ("SPREAD", round 5): 1 = every LDS fragment request and every LDS-DMA issue of
// a K-step is pinned between two MFMAs by register ties (svcmi_lds_read16 / svcmi_bdma16_at name the A fragment the MFMAs consume), one
// request per MFMA and the refill's DMA pieces spread over the last sub-step, so that no gap between consecutive matrix instructions holds
// more issue slots than an MFMA covers (the ISA of the 64x80 tile: gaps of 23 / 15 / 27 / 14 instructions -> at most 11; an in-order wave
// issues nothing to the matrix pipe inside such a gap).  0 = requests in front of a sub-step's MFMAs, the refill DMAs wherever the
// scheduler leaves them.  Same MFMAs in the same order either way: bit-identical results.  conv_gemm.hip is built with its accumulators in
// architectural registers (build.py FILE_FLAGS): with the ties in place the accumulator-file form rotates three accumulator tuples
// through v_accvgpr copies at the top of every K-step.
#ifndef SVCMI_GEMM_SPREAD
#define SVCMI_GEMM_SPREAD 1
#endif
#ifndef SVCMI_PROBE_NOMFMA      // timing probe (svcmi_rt.h): the K loop without its matrix instructions
#define SVCMI_PROBE_NOMFMA 0
#endif
// Timing probe (scripts/build_variant.sh ktrace -DSVCMI_PROBE_KTRACE=1; never in the product build): every wave of the mid-barrier K loop
// stamps s_memtime before the DMA wait, after it and after the barrier, sums the differences in scalar registers and leaves
// {cycles between barrier exits, cycles in the vmcnt wait, cycles in the barrier, K-steps} in g_ktrace[4 * (block * 4 + wave)]
// (read back with svcmi_probe_ktrace_read, scripts/microbench.py ktrace).
#ifndef SVCMI_PROBE_KTRACE
#define SVCMI_PROBE_KTRACE 0
#endif
#if SVCMI_PROBE_KTRACE
#ifndef SVCMI_PROBE_KTRACE_N      // record only launches with this n_out (0 = every launch: the last one stays)
#define SVCMI_PROBE_KTRACE_N 0
#endif
// In-flight timeline (round 6): rocprofv3 serialises the four HIP queues of the judged launch regime, so what overlaps
// with what while clips are in flight has to be recorded by the kernels themselves.  EVERY block of every implicit-GEMM launch appends one
// record {s_memrealtime at entry, at the end of its epilogue, output pointer (= which lane and layer), n_out << 32 | K, rows} to g_tl
// (one atomic add + 40 bytes per block); scripts/inflight_timeline.py groups the blocks into launches and reconstructs the timeline.
constexpr unsigned TL_CAP = 1u << 22;            // records (168 MB)
__device__ unsigned long long g_tl[5ull * TL_CAP];
__device__ unsigned g_tl_n;
__device__ unsigned long long g_ktrace[4 * 4 * 8192 + 8];      // + block 0 wave 0: {loop s_memtime ticks, loop s_memrealtime ticks (100 MHz), entry -> first barrier exit, last barrier exit -> end of the epilogue, launches recorded}
#endif
// Build switch (round 6): the refill's DMA pieces of the pinned loop spread over a WHOLE K-step instead of its last sub-step -- the first half
// behind the mid-step barrier of K-step it (as before), the second half between the MFMAs of K-step it+1's earlier sub-steps (the slot is
// free from that barrier on and its tile is not needed before the mid-step barrier of K-step it+NST-1, so this needs a ring of >= 3 tiles).
// Measured (profiles/r06a_kprobe.log): the K-step of the 64x80 tile costs 2500 cycles without its DMA issues and 3070 with them, while
// bare MFMA streams with the same number of evenly spaced pieces lose a third of that (scripts/probes/dma_issue_cost.hip).
// Default since round 6: bit-identical over 307 launches on hardware (scripts/spread_check.py, profiles/r06f_spread2_m0raw.log); one clip
// alone 9.31 -> 9.20 ms together with the raw-M0 DMA issue (svcmi_rt.h), the judged line +0.3 % (its MLP GEMMs run the 2-deep ring).
#ifndef SVCMI_GEMM_SPREAD2
#define SVCMI_GEMM_SPREAD2 1
#endif
constexpr int BK = 32;
enum { MODE_CHUNK = 0, MODE_VEC = 1, MODE_SCALAR = 2, MODE_CHUNK_RS = 3 };   // _RS: CHUNK with x_row_shift != 0

enum { PREC_F32 = 0, PREC_BF16X3 = 1, PREC_BF16 = 2, PREC_F16 = 3, PREC_BF16_A16 = 4, PREC_F16_A16 = 5, PREC_BF16X3_A16 = 6, PREC_F16W2_A16 = 9 };   // == enum svcmi_precision
// F16W2_A16 (round 4): fp16 activation rows as F16_A16, but the weight image holds (hi, lo) fp16 pairs -- rows [hi | lo] -- and a fragment pair
// costs two MFMAs (a*hi + a*lo): the weight is exact to 2^-22, only the activation rounding is left in the product.
// _A16: the ACTIVATIONS arrive as a 16-bit tensor too (written by the producing kernel's epilogue) and the weight image is in natural
// k order (svcmi_pack_weights_lp with an _A16 precision).  Both tiles then use the fp32 kernel's data movement unchanged -- rows of 128
// bytes, eight 16-byte chunks XOR-swizzled by swz(row), 8 rows per 1-KiB DMA piece -- on 16-bit data: a K-step covers 64 k instead of
// 32, a lane's ds_read_b128 fragment (8 consecutive k) feeds ONE 16-bit MFMA where the fp32 kernel issues four, and nothing is rounded
// in registers.  Half the barriers / DMA issues / LDS bytes per FLOP of the plain 16-bit modes, and the same products (the same fp32
// values rounded the same way by the producer).
// BF16X3_A16: the same with BOTH operands as (hi, lo) bf16 pairs -- activation rows are [hi: ldx/2 values | lo: ldx/2 values] (a producer's
// 16-bit output in format SVCMI_PREC_BF16X3), weight rows [hi | lo] in natural k order -- and three MFMAs per fragment pair
// (hi*hi + lo*hi + hi*lo, fp32 accumulation): the products of PREC_BF16X3 without its in-register splitting, on 64-k K-steps.

struct ConvArgs {
    const float* x; const float* w; const float* bias; const float* res; float* y; const int32_t* lengths;
    float* ws;                   // split-K partials [batch][split][t_out][n_out]
    int* cnt;                    // one arrival counter per (batch, n-tile, m-tile), all zero between launches; NULL = two-kernel reduce
    long long x_bs, y_bs, r_bs;
    int t_in, t_out, c_in, ldx, n_out, ldw, ldy, ldr;
    int ksize, stride, dil, pad, rshift, act, flags;
    const unsigned short* w16;   // reduced precision: 16-bit weight image, row n = [hi: ldw16 values][lo: ldw16 values, bf16x3 only]
    int ldw16;
    unsigned short* y16;         // optional 16-bit copy of the output (bf16 or f16, y16_f16), rows of ldy16 values: the next GEMM's A operand
    long long y16_bs;
    int ldy16, y16_f16;          // y16_f16: 0 bf16, 1 f16, 2 split bf16 (hi at [n], lo at [ldy16/2 + n])
    int vec;                     // 1: y / res / bias / workspace rows are 16-byte aligned multiples of 4 floats -> float4 epilogue
    int ktot;                    // ksize * c_in
    int split;                   // K slices (1 = none)
    int mt, nt;                  // tile grid (time x channels)
    unsigned magic;              // ceil(2^32 / c_in) for the VEC / SCALAR gathers (0 when c_in == 1)
    float alpha;
};

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case SVCMI_ACT_RELU: return v > 0.f ? v : 0.f;
        case SVCMI_ACT_GELU: return svcmi_gelu(v);
        case SVCMI_ACT_MISH: {   // x * tanh(softplus(x)), softplus with torch's threshold 20
            // log1p(e), e = exp(v) > 0, as log(u) * e / (u - 1) with u = fl(1 + e) (exact where u == 1: the result is e): within 2 ulp of libm's
            // log1pf.  Not the libm call: its double-float arithmetic is vectorised into v_pk_add_f32 ... op_sel:[0,1], the packed form MI355X
            // computes wrongly in lanes 48..63 beside another wave's v_mfma_f32_16x16x32_{f16,bf16} (see svcmi_hsum2) -- and this epilogue is
            // compiled into every GEMM kernel of the library
            float sp = v;
            if (!(v > 20.f)) {
                const float e = expf(v), u = 1.0f + e;
                sp = u == 1.0f ? e : logf(u) * (e / (u - 1.0f));
            }
            return v * tanhf(sp);
        }
        case SVCMI_ACT_TANH: return tanhf(v);
        case SVCMI_ACT_SIGMOID: return 1.0f / (1.0f + expf(-v));
        default: return v;
    }
}

__device__ __forceinline__ float epilogue(const ConvArgs& p, float v, float bias, const float* res_row, float* dst,
                                          int n, bool masked) {
    v = act_apply(v + bias, p.act);
    if (res_row) v += res_row[n];
    v *= p.alpha;
    if (p.flags & SVCMI_CONV_ACCUMULATE) v += *dst;
    return masked ? 0.f : v;
}

__device__ __forceinline__ int div_magic(int q, unsigned magic) {   // q / c_in, exact while q * c_in < 2^32
    return magic ? (int)__umulhi((unsigned)q, magic) : q;
}

// LDS swizzle: the 16-byte chunk c of tile row r is stored at position c ^ swz(r).  A 16-lane ds_read_b128 group
// reads one chunk column from 16 rows {4 consecutive, 4 consecutive, 8 consecutive}; a 256-byte bank row holds 2
// tile rows x 8 positions, so the group is conflict-free iff rows of equal parity get distinct positions:
// swz(r) = (r >> 1) & 7 does that for every group (r & 7 left 2-way conflicts: PMC SQ_LDS_BANK_CONFLICT ~ 50 %).
__device__ __forceinline__ int swz(int row) { return (row >> 1) & 7; }
// 16-bit B tile (64-byte rows, 4 chunks): chunk q of row r is stored at position q ^ swz16(r), swz16 = {0,2,3,1}[(r>>2)&3].
// A ds_read_b128 lane group covers 16 rows; rows of equal r&3 share a 64-byte quarter of the bank row and must land on
// distinct positions: {0,12,20,24} / {4,8,16,28} (32x32x16: one chunk per group) and {0,12 | 4,8 with the next chunk}
// (16x16x32: lanes 16-31 read chunk q+1) all do.
__device__ __forceinline__ int swz16(int row) { return (0x78 >> (((row >> 2) & 3) * 2)) & 3; }

constexpr unsigned OOB = 0x40000000u;   // added to an offset that must read as zero; buffers are < 2^29 bytes

// Two micro-kernel policies share everything but the fragment / MFMA / accumulator code:
//   P16 = false: waves 2 x 2, wave tile (32*WM) x (32*WN), v_mfma_f32_32x32x2_f32   -> block (64*WM) x (64*WN);
//   P16 = true : waves 4 x 1, wave tile (16*WM) x (16*WN), v_mfma_f32_16x16x4_f32   -> block (64*WM) x (16*WN):
//                right-sized N for the 40 / 80 / 160-channel generator stages (a 64-multiple pads them by 60 / 60 / 20 %),
//                and since fp32 MFMA time is proportional to the padded tile, that padding is pure loss.
// `grid_blocks` / `block_id`: the launch geometry of THIS problem (a grouped launch runs several problems back to back in one grid).
//   P16 = true, NW = 8 (round 6): the same wave tile on EIGHT waves -> block (128*WM) x (16*WN), 512 threads.  M = 500 x N = 5120 is then 256
//                blocks = one per CU with today's two waves per SIMD, and the CU fills 128 + 80 rows per K-step instead of 2 x (64 + 80):
//                26 instead of 36 KB -- the K-step of the 64x80 tile costs 2500 cycles without its LDS-DMA issues and 3040 with them
//                (profiles/r06a_kprobe.log, r06e_ktrace_in_pipeline.log), so the pieces per FLOP are what is left to cut.  Same wave tile, same
//                K order: the same bits as the four-wave tile.
template <int WM, int WN, int MODE, bool P16, int NSTO = 0, int PREC = PREC_F32, int NW = 4>
__device__ __forceinline__ void conv_gemm_body(const ConvArgs& p, const int grid_blocks, const int block_id) {
    static_assert(NW == 4 || (NW == 8 && P16 && PREC == PREC_F32), "eight-wave blocks: fp32 16x16x4 policy only");
    constexpr int NT = 64 * NW;                       // threads per block
    constexpr bool A16 = PREC >= PREC_BF16_A16;       // 16-bit activations AND weights in the fp32 kernel's tile geometry, K-step 64
    constexpr bool LP = PREC != PREC_F32 && !A16;     // 16-bit weight image(s) in 64-byte rows, fp32 activations rounded in registers
    constexpr int KS = A16 ? 2 * BK : BK;             // k per K-step
    constexpr unsigned ESZ = A16 ? 2u : 4u;           // bytes per activation element
    constexpr bool F16OP = PREC == PREC_F16 || PREC == PREC_F16_A16 || PREC == PREC_F16W2_A16;
    constexpr bool X3A = PREC == PREC_BF16X3_A16;     // ... both as (hi, lo) bf16 pairs
    constexpr bool W2A = PREC == PREC_F16W2_A16;      // ... fp16 activations x (hi, lo) fp16 weights
    constexpr int NB = (PREC == PREC_BF16X3 || X3A || W2A) ? 2 : 1;   // B images per stage (hi, lo)
    constexpr int NA = X3A ? 2 : 1;                   // A images per stage
    constexpr int BM = P16 ? 16 * WM * NW : 64 * WM, BN = P16 ? 16 * WN : 64 * WN;
    constexpr int BROW = LP ? BK / 2 : BK;            // floats per B row in LDS (LP: 32 x 2 bytes)
    constexpr int BNL = LP ? (BN + 63) / 64 * 64 : (BN + 8 * NW - 1) / (8 * NW) * (8 * NW);   // B rows held in LDS (whole NW-wave DMA rounds)
    constexpr int A_PER = BM / (8 * NW), B_PER = LP ? BNL / 64 : BNL / (8 * NW);      // 1-KiB LDS-DMA pieces (8 x 128 B or 16 x 64 B rows) per wave per K-step (per image)
    constexpr int BTILE = BNL * BROW;                 // floats per B image per stage
    constexpr int CLD = BN;                           // epilogue staging tile [BM][BN] reuses the operand buffers
    // LDS ring of NST operand tiles (48 / 72 / 64 KiB per block): tile it+NST-1 is in flight while tile
    // `it` is consumed, so a K-step never waits a full HBM/L2 round trip -- what short K ranges (split-K slices,
    // the k=1 projections of the prior encoder / flow, k=3 convolutions) would otherwise pay on every step.
    constexpr int NST = NSTO ? NSTO : NW == 8 ? 3 : (X3A || (W2A && P16)) ? 2 : (LP ? 3 : (P16 ? ((BM + BNL) > 192 ? 2 : 3) : ((WM * WN == 1) ? 3 : (WM * WN == 2 ? 3 : 2))));
    constexpr int RING = NST * (NA * BM * BK + NB * BTILE);
    static_assert(LP || A16 || BM * CLD <= RING, "C tile must fit in the operand buffers");
    // + the split-K ticket word (4 floats) + the tile's bias values (round 6: one 4-byte LDS-DMA per 64 columns in front of the first
    // operand tile -- the epilogue reads them from LDS instead of waiting for a global load per iteration; no bias = all lanes out of range = zeros)
    constexpr int BIAS0 = (RING > BM * CLD ? RING : BM * CLD) + 4, BIASN = (BN + 63) / 64 * 64;
    __shared__ __attribute__((aligned(16))) float smem[BIAS0 + BIASN];
    float* const As0 = smem;                          // As[slot][image] = As0 + (slot*NA + image)*BM*BK, rows of 32 floats, chunk-swizzled
    float* const Bs0 = smem + NST * NA * BM * BK;          // Bs[slot][image] = Bs0 + (slot*NB + image)*BTILE

#if SVCMI_PROBE_KTRACE
    unsigned long long kt_entry, tl_entry;
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(kt_entry), "=s"(tl_entry));
#endif
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = SVCMI_UNIFORM((int)(tid >> 6));
    const int wm = P16 ? wave : (wave >> 1), wn = P16 ? 0 : (wave & 1);
    // XCD-aware bijective enumeration: XCD g owns a contiguous range of the (z, n-tile, m-tile) order, m fastest
    int bx, by, bz;
    {
        const int total = grid_blocks, id = block_id;
        const int q8 = total >> 3, r8 = total & 7, xcd = id & 7, slot = id >> 3;
        const int L = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        bx = L % p.mt;
        const int rest = L / p.mt;
        by = rest % p.nt;
        bz = rest / p.nt;
    }
    const int b = bz / p.split, slice = bz - b * p.split;
    const int m0 = bx * BM, n0 = by * BN;
    const int len = p.lengths ? p.lengths[b] : 0x7fffffff;
    const int t_lim = (p.flags & SVCMI_CONV_MASK_IN) ? (len < p.t_in ? len : p.t_in) : p.t_in;
    // operand buffers: x of this batch item up to the first row that must read as zero; the whole weight matrix
    const int x_rows = (t_lim + (1 << p.rshift) - 1) >> p.rshift;
    const svcmi_rsrc xr = A16 ? svcmi_make_rsrc(reinterpret_cast<const unsigned short*>(p.x) + (long long)b * p.x_bs, (unsigned)x_rows * (unsigned)p.ldx * 2u)
                              : svcmi_make_rsrc(p.x + (long long)b * p.x_bs, (unsigned)x_rows * (unsigned)p.ldx * 4u);
    const svcmi_rsrc wr = (LP || A16) ? svcmi_make_rsrc(p.w16, (unsigned)p.n_out * (unsigned)(NB * p.ldw16) * 2u)
                             : svcmi_make_rsrc(p.w, (unsigned)p.n_out * (unsigned)p.ldw * 4u);

    const int nk_all = (p.ktot + KS - 1) / KS;
    const int it_beg = (int)((long long)nk_all * slice / p.split);
    const int it_end = (int)((long long)nk_all * (slice + 1) / p.split);

    // LDS image: row r holds its 32 k-values as 8 chunks of 16 B, chunk c stored at position c ^ swz(r).
    // A 1-KiB piece = 8 consecutive rows; the DMA writes lane l at byte 16*l of the piece, i.e. row l>>3,
    // position l&7, so lane l must FETCH logical chunk (l&7) ^ swz(row) of that row (swizzle on the source).
    const int prow = lane >> 3;                        // row within a piece
    const int lkq = ((lane & 7) ^ swz(prow + 8 * wave)) * 4;   // this lane's k offset within the K-step (pieces start at multiples of 8 rows: swz(row) only needs row mod 16)
    // K-invariant per-piece state: piece i of this wave covers rows (wave + 4*i)*8 .. +8
    int a_tb[A_PER];                  // first input row of the piece's output row: t*stride - pad
    unsigned a_row[A_PER];            // its byte offset (x_row_shift == 0), or the OOB sentinel past t_out
    unsigned b_row[B_PER];
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int t = m0 + (wave + NW * i) * 8 + prow;
        a_tb[i] = t * p.stride - p.pad;
        a_row[i] = t < p.t_out ? (unsigned)a_tb[i] * (unsigned)p.ldx * ESZ : OOB;
        if (t >= p.t_out) a_tb[i] = -0x40000000;
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        // LP: a piece is 16 rows of 64 bytes, lane l -> row l>>2, position l&3
        const int nl = (wave + NW * i) * (LP ? 16 : 8) + (LP ? (lane >> 2) : prow), n = n0 + nl;
        // (rows of the DMA round behind the tile's last column are never fetched: they used to pull the NEXT tile's weights through L2
        //  into LDS rows nobody reads -- 2 of the 12 B pieces of the 64x80 tile, 2 of 8 at 64x48)
        b_row[i] = (n < p.n_out && nl < BN) ? ((LP || A16) ? (unsigned)(n * NB * p.ldw16) * 2u : (unsigned)(n * p.ldw) * 4u) : OOB;
    }
    const int lkq16 = ((lane & 3) ^ swz16(lane >> 2)) * 16;    // LP: this lane's byte offset within the 64-byte K-step of a B row
    // CHUNK mode: wave-uniform (tap, first channel) of the K-step, advanced incrementally
    int tap_u = 0, ci_u = 0;
    if (MODE == MODE_CHUNK || MODE == MODE_CHUNK_RS) {
        const int k0 = it_beg * KS;
        tap_u = k0 / p.c_in;
        ci_u = k0 - tap_u * p.c_in;
    }
    const svcmi_ldsaddr lds_a = svcmi_lds_advance(svcmi_lds_addr(As0), wave * 8 * BK);
    const svcmi_ldsaddr lds_b = svcmi_lds_advance(svcmi_lds_addr(Bs0), wave * 8 * BK);
    static_assert(WM <= 2, "the pinned placement names at most two A fragments");
    static_assert(!P16 || (MODE == MODE_CHUNK || MODE == MODE_VEC), "16x16x4 policy: vector gathers only");

    using acc_t = typename std::conditional<P16, svcmi_f32x4, svcmi_f32x16>::type;
    constexpr int ACC_N = P16 ? 4 : 16;
    acc_t acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < ACC_N; ++r) acc[i][j][r] = 0.f;

    // Per-K-step source offsets of tile `it`, then the DMA pieces (none lands in registers, none is waited for here).
    unsigned a_koff = 0, b_koff = 0;     // K part of this lane's byte offset (or OOB), valid between prep and issue
    int tap_v = 0, ci_v = 0;             // x_row_shift path: this K-step's (tap, channel)
    auto stage_prep = [&](int it) {
        const int lkk = A16 ? 2 * lkq : lkq;         // this lane's k offset within the K-step (a 16-byte chunk = 4 fp32 or 8 16-bit values)
        const int kk = it * KS + lkk;
        if (LP) b_koff = (it * BK * 2 + lkq16) < p.ldw16 * 2 ? (unsigned)(it * BK * 2 + lkq16) : OOB;
        else if (A16) b_koff = kk < p.ldw16 ? (unsigned)kk * 2u : OOB;
        else b_koff = kk < p.ldw ? (unsigned)kk * 4u : OOB;
        if (MODE == MODE_CHUNK || MODE == MODE_CHUNK_RS) {
            tap_v = tap_u; ci_v = ci_u + lkk;
            a_koff = (unsigned)(tap_u * p.dil * p.ldx + ci_u + lkk) * ESZ;
            ci_u += KS;
            if (ci_u >= p.c_in) { ci_u -= p.c_in; ++tap_u; }
        } else if (MODE == MODE_VEC) {      // (A16: c_in % 8 == 0, so a lane's 8 consecutive k lie inside one tap)
            tap_v = div_magic(kk, p.magic); ci_v = kk - tap_v * p.c_in;
            a_koff = kk < p.ktot ? (unsigned)(tap_v * p.dil * p.ldx + ci_v) * ESZ : OOB;
        }
    };
    auto stage_a = [&](int it, int buf, int i, auto&... tie) {          // piece i of the A tile (X3A: i >= A_PER = the lo image); tie: svcmi_bdma16_at
        if constexpr (X3A) {
            const int im = i >= A_PER, ii = i - im * A_PER;
            svcmi_bdma16(a_row[ii] + a_koff + (im ? (unsigned)p.ldx : 0u), svcmi_lds_advance(lds_a, (buf * NA + im) * BM * BK + NW * ii * 8 * BK), xr);
            return;
        }
        const svcmi_ldsaddr dst = svcmi_lds_advance(lds_a, buf * BM * BK + NW * i * 8 * BK);
        if (MODE == MODE_SCALAR) {
            // 4-byte DMA: a wave-instruction fills 2 rows (64 floats); the piece needs 4 of them.
#pragma unroll
            for (int h = 0; h < 4; ++h) {
                const int rr = 2 * h + (lane >> 5);                  // row within the piece
                const int pc = (lane & 31) >> 2, e = lane & 3;       // physical chunk, element
                const int q = it * BK + ((pc ^ swz(rr + 8 * wave)) << 2) + e;   // logical k of this LDS word
                const int t = m0 + (wave + NW * i) * 8 + rr;
                const int k = div_magic(q, p.magic), ci = q - k * p.c_in;
                const int tin = t * p.stride - p.pad + k * p.dil;
                const bool ok = t < p.t_out && q < p.ktot && (unsigned)tin < (unsigned)t_lim;
                svcmi_bdma4(ok ? (unsigned)((tin >> p.rshift) * p.ldx + ci) * 4u : OOB, svcmi_lds_advance(dst, 2 * h * BK), xr);
            }
        } else if (MODE == MODE_CHUNK_RS) {               // fused np.repeat(x, 2, 0): rows are tin >> 1
            const int tin = a_tb[i] + tap_v * p.dil;
            const bool ok = (unsigned)tin < (unsigned)t_lim;
            svcmi_bdma16_at(ok ? (unsigned)((tin >> p.rshift) * p.ldx + ci_v) * 4u : OOB, dst, xr, tie...);
        } else {
            svcmi_bdma16_at(a_row[i] + a_koff, dst, xr, tie...);
        }
    };
    auto stage_b = [&](int buf, int i, auto&... tie) {                  // piece i of the B tile(s): i < B_PER hi (or fp32), then the lo image
        if (i < B_PER) svcmi_bdma16_at(b_row[i] + b_koff, svcmi_lds_advance(lds_b, buf * NB * BTILE + NW * i * 8 * BK), wr, tie...);
        else svcmi_bdma16(b_row[i - B_PER] + b_koff + (unsigned)p.ldw16 * 2u, svcmi_lds_advance(lds_b, (buf * NB + 1) * BTILE + NW * (i - B_PER) * 8 * BK), wr);
    };

    // fragment addresses.  32x32x2: row (lane&31) of the wave tile, logical chunk 2s + (lane>>5), 4 sub-steps of 8 k;
    // 16x16x4: row (lane&15), chunk 4s + (lane>>4), 2 sub-steps of 16 k (MFMA #c contracts k = 16s + c + {0,4,8,12}).
    // Either way the chunk sits at position chunk ^ swz(row), and both access patterns are bank-conflict free.
    constexpr int FR = P16 ? 16 : 32;                  // rows per MFMA tile
    const int frow = lane & (FR - 1), fhi = P16 ? (lane >> 4) : (lane >> 5);
    const int a_off = (wm * FR * WM + frow) * BK, b_off = (wn * FR * WN + frow) * BROW;
    // Fragment reads of sub-step s into (a4, b4).  `tie` is a register the MFMAs issued next consume.
    constexpr int FA = NA * WM, FB = ((X3A || W2A) ? 2 : 1) * WN;      // fragments per sub-step (X3A / W2A: hi images first, then lo)
    auto load_frags = [&](const float* Ab, const float* Bb, int s, svcmi_f32x4 (&a4)[FA], svcmi_f32x4 (&b4)[FB], svcmi_f32x4& tie) {
        const int pos = ((((P16 ? 4 : 2) * s + fhi) ^ swz(frow)) << 2);
#pragma unroll
        for (int i = 0; i < FA; ++i) svcmi_lds_read16(a4[i], Ab + (i / WM) * BM * BK + (i % WM) * FR * BK + pos, tie);
#pragma unroll
        for (int j = 0; j < FB; ++j) svcmi_lds_read16(b4[j], Bb + (j / WN) * BTILE + (j % WN) * FR * BK + pos, tie);
    };
    // fragment request k of sub-step s (k < FA: the A fragments, then the B fragments) as ONE statement pinned by its ties (svcmi_lds_read16)
    auto frag_src = [&](const float* Ab, const float* Bb, int s, int k) {
        const int pos = ((((P16 ? 4 : 2) * s + fhi) ^ swz(frow)) << 2);
        return k < FA ? Ab + (k / WM) * BM * BK + (k % WM) * FR * BK + pos : Bb + ((k - FA) / WN) * BTILE + ((k - FA) % WN) * FR * BK + pos;
    };
    auto load_frag_at = [&](const float* Ab, const float* Bb, int s, int k, svcmi_f32x4 (&a4)[FA], svcmi_f32x4 (&b4)[FB], auto&... tie) {
        svcmi_lds_read16(k < FA ? a4[k < FA ? k : 0] : b4[k < FA ? 0 : k - FA], frag_src(Ab, Bb, s, k), tie...);
    };
    auto mma = [&](acc_t& c, float a, float b) {
        if (SVCMI_PROBE_NOMFMA) return;
        if constexpr (P16) c = svcmi_mfma_16x16x4(a, b, c);
        else c = svcmi_mfma_32x32x2(a, b, c);
    };
    auto frags_arrive = [&](svcmi_f32x4 (&a4)[FA], svcmi_f32x4 (&b4)[FB]) {      // one s_waitcnt, every fragment pinned behind it
        svcmi_lds_arrive(a4[0]);
#pragma unroll
        for (int i = 1; i < FA; ++i) svcmi_lds_landed(a4[i]);
#pragma unroll
        for (int j = 0; j < FB; ++j) svcmi_lds_landed(b4[j]);
    };
    // Reduced precision.  Sub-step s of a K-step contracts the 16 (32x32x16: s = 0, 1) or all 32 (16x16x32) k of the step: the
    // lane reads 16-byte chunk q = 2s + (lane>>5) resp. lane>>4 of its 16-bit B row(s) and the fp32 A chunks q and q + 4.
    auto load_frags_lp = [&](const float* Ab, const float* Bb, int s, svcmi_f32x4 (&a8)[WM][2], svcmi_f32x4 (&b8)[NB][WN], svcmi_f32x4& tie) {
        const int q = (P16 ? 0 : 2 * s) + fhi;
        const int pa = (q ^ swz(frow)) << 2, pb = (q ^ swz16(frow)) << 2;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            svcmi_lds_read16(a8[i][0], Ab + i * FR * BK + pa, tie);
            svcmi_lds_read16(a8[i][1], Ab + i * FR * BK + (pa ^ 16), tie);
        }
#pragma unroll
        for (int h = 0; h < NB; ++h)
#pragma unroll
            for (int j = 0; j < WN; ++j) svcmi_lds_read16(b8[h][j], Bb + h * BTILE + j * FR * BROW + pb, tie);
    };
    auto frags_arrive_lp = [&](svcmi_f32x4 (&a8)[WM][2], svcmi_f32x4 (&b8)[NB][WN]) {
        svcmi_lds_arrive(a8[0][0]);
        svcmi_lds_landed(a8[0][1]);
#pragma unroll
        for (int i = 1; i < WM; ++i) { svcmi_lds_landed(a8[i][0]); svcmi_lds_landed(a8[i][1]); }
#pragma unroll
        for (int h = 0; h < NB; ++h)
#pragma unroll
            for (int j = 0; j < WN; ++j) svcmi_lds_landed(b8[h][j]);
    };
    // fp32 fragment (chunks q | q+4) -> packed 16-bit operand(s): hi = round(x), lo = round(x - hi) (bf16x3 only)
    auto round_frag = [&](const svcmi_f32x4 (&r)[2], svcmi_u32x4& hi, svcmi_u32x4& lo) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float x0 = r[e >> 1][2 * (e & 1)], x1 = r[e >> 1][2 * (e & 1) + 1];
            if constexpr (F16OP) {
                hi[e] = svcmi_cvt_pk_f16(x0, x1);
            } else {
                const unsigned h = svcmi_cvt_pk_bf16(x0, x1);
                hi[e] = h;
                if constexpr (NB == 2)
                    lo[e] = svcmi_cvt_pk_bf16(x0 - svcmi_bits_f32(h << 16), x1 - svcmi_bits_f32(h & 0xffff0000u));
            }
        }
    };
    auto mma16 = [&](acc_t& c, svcmi_u32x4 a, svcmi_u32x4 b) {
        if constexpr (P16) c = svcmi_mfma16_16x16x32<F16OP>(a, b, c);
        else c = svcmi_mfma16_32x32x16<F16OP>(a, b, c);
    };

    // sub-steps per tile: fp32 4 x 8 k (32x32x2) or 2 x 16 k (16x16x4); 16-bit 2 x 16 k (32x32x16) or 1 x 32 k (16x16x32)
    constexpr int NSUB = LP ? (P16 ? 1 : 2) : (P16 ? BK / 16 : BK / 8);
    constexpr int APCS = NA * A_PER;
    constexpr int PIECES = APCS + NB * B_PER;             // pieces per tile per wave ...
    constexpr int DMAS = (MODE == MODE_SCALAR ? 4 * A_PER : APCS) + NB * B_PER;   // ... and the DMA instructions they take
    // prologue: tiles it_beg .. it_beg+NST-2 into slots 0 .. NST-2
    // (the mid-barrier loop below -- fp32, MIDBAR -- fills ALL NST slots up front: it refills a slot right after the barrier that retires it)
    constexpr bool MIDBAR = SVCMI_GEMM_MIDBAR != 0 && !LP && !A16;
    constexpr bool SPREAD = SVCMI_GEMM_SPREAD != 0 && MIDBAR && MODE != MODE_SCALAR;      // (the 4-byte gathers keep the plain placement)
    if (wave < BIASN / 64) {        // (the oldest DMA of the wave: every counted wait below covers it)
        const int n = n0 + 64 * wave + lane;
        const svcmi_rsrc br = svcmi_make_rsrc(p.bias, p.bias ? (unsigned)p.n_out * 4u : 0u);
        svcmi_bdma4(p.bias && n < p.n_out ? (unsigned)n * 4u : OOB, svcmi_lds_advance(svcmi_lds_addr(smem + BIAS0), 64 * wave), br);
    }
#pragma unroll
    for (int s0 = 0; s0 < (MIDBAR ? NST : NST - 1); ++s0) {
        if (it_beg + s0 < it_end) {
            stage_prep(it_beg + s0);
#pragma unroll
            for (int i = 0; i < APCS; ++i) stage_a(it_beg + s0, s0, i);
#pragma unroll
            for (int i = 0; i < NB * B_PER; ++i) stage_b(s0, i);
        }
    }
    // One K-step.  ISSUE (compile time) = tile it+NST-1 exists: its DMA pieces are issued here, spread over the
    // sub-steps so that their issue slots sit between this tile's MFMAs.  `inflight` = tiles issued after `it`
    // that may still be in flight when tile `it` is needed (vmcnt counts this wave's DMAs in issue order).
    auto k_step = [&](int it, int slot, int inflight, auto issue_tag) {
        constexpr bool ISSUE = decltype(issue_tag)::value;
        if (ISSUE || inflight == NST - 2) svcmi_dma_wait_n<(NST - 2) * DMAS>();
        else if (inflight == 0) svcmi_dma_wait_n<0>();
        else if (inflight == 1) svcmi_dma_wait_n<DMAS>();
        else svcmi_dma_wait_n<2 * DMAS>();
        __syncthreads();     // tile `it` has landed for every wave; all reads of the slot refilled below are done
        int nslot = slot + NST - 1;
        if (nslot >= NST) nslot -= NST;
        if (ISSUE) stage_prep(it + NST - 1);
        const float* Ab = As0 + slot * NA * BM * BK + a_off;
        const float* Bb = Bs0 + slot * NB * BTILE + b_off;
        svcmi_f32x4 tie0 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (LP) {
            svcmi_f32x4 a8[2][WM][2], b8[2][NB][WN];
            load_frags_lp(Ab, Bb, 0, a8[0], b8[0], tie0);
            frags_arrive_lp(a8[0], b8[0]);
#pragma unroll
            for (int s = 0; s < NSUB; ++s) {
                if (s + 1 < NSUB) load_frags_lp(Ab, Bb, s + 1, a8[(s + 1) & 1], b8[(s + 1) & 1], a8[s & 1][0][0]);
                if (ISSUE) {
#pragma unroll
                    for (int q = s * PIECES / NSUB; q < (s + 1) * PIECES / NSUB; ++q) {
                        if (q < A_PER) stage_a(it + NST - 1, nslot, q);
                        else stage_b(nslot, q - A_PER);
                    }
                }
                svcmi_u32x4 ahi[WM], alo[WM];
#pragma unroll
                for (int i = 0; i < WM; ++i) round_frag(a8[s & 1][i], ahi[i], alo[i]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma16(acc[i][j], ahi[i], svcmi_as_u32x4(b8[s & 1][0][j]));
                if constexpr (NB == 2) {
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) mma16(acc[i][j], alo[i], svcmi_as_u32x4(b8[s & 1][0][j]));
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) mma16(acc[i][j], ahi[i], svcmi_as_u32x4(b8[s & 1][1][j]));
                }
                if (s + 1 < NSUB) {
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) svcmi_pin(acc[i][j]);
                    frags_arrive_lp(a8[(s + 1) & 1], b8[(s + 1) & 1]);
                }
            }
            return;
        }
        svcmi_f32x4 a4[2][FA], b4[2][FB];
        load_frags(Ab, Bb, 0, a4[0], b4[0], tie0);
        frags_arrive(a4[0], b4[0]);
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            svcmi_f32x4(&af)[FA] = a4[s & 1];
            svcmi_f32x4(&bf)[FB] = b4[s & 1];
            // next sub-step's fragments are requested before this sub-step's MFMAs are issued ...
            if (s + 1 < NSUB) load_frags(Ab, Bb, s + 1, a4[(s + 1) & 1], b4[(s + 1) & 1], af[0]);
            if (ISSUE) {
#pragma unroll
                for (int q = s * PIECES / NSUB; q < (s + 1) * PIECES / NSUB; ++q) {
                    if (q < APCS) stage_a(it + NST - 1, nslot, q);
                    else stage_b(nslot, q - APCS);
                }
            }
            if constexpr (X3A) {         // hi*hi + lo*hi + hi*lo; the lo fragments sit behind the hi ones
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma16(acc[i][j], svcmi_as_u32x4(af[i]), svcmi_as_u32x4(bf[j]));
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma16(acc[i][j], svcmi_as_u32x4(af[WM + i]), svcmi_as_u32x4(bf[j]));
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma16(acc[i][j], svcmi_as_u32x4(af[i]), svcmi_as_u32x4(bf[WN + j]));
            } else if constexpr (W2A) {         // a*hi + a*lo
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma16(acc[i][j], svcmi_as_u32x4(af[i]), svcmi_as_u32x4(bf[j]));
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma16(acc[i][j], svcmi_as_u32x4(af[i]), svcmi_as_u32x4(bf[WN + j]));
            } else if constexpr (A16) {         // the fragment's 16 bytes are 8 consecutive 16-bit k: one MFMA per (i, j)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma16(acc[i][j], svcmi_as_u32x4(af[i]), svcmi_as_u32x4(bf[j]));
            } else {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i][0], bf[j][0]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i][1], bf[j][1]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i][2], bf[j][2]);
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i][3], bf[j][3]);
            }
            // ... and waited for after them (the pins keep this sub-step's MFMAs above the wait)
            if (s + 1 < NSUB) {
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < WN; ++j) svcmi_pin(acc[i][j]);
                frags_arrive(a4[(s + 1) & 1], b4[(s + 1) & 1]);
            }
        }
    };
#if SVCMI_PROBE_KTRACE
    unsigned long long kt0 = 0, kt1 = 0, kt2 = 0, kt_prev = 0, kt_step = 0, kt_vm = 0, kt_bar = 0, kt_n = 0, kt_r0 = 0, kt_r1 = 0, kt_first = 0;
#endif
    if constexpr (MIDBAR) {
        // Mid-barrier pipeline (round 5).  The loop above opens every K-step with [wait DMA, barrier, fragment reads of sub-step 0,
        // s_waitcnt lgkmcnt(0)]: an exposed LDS round trip per K-step in which the wave issues no MFMA (the ISA of the 64x80 tile: 6
        // ds_read_b128 and their wait between the barrier and the first of 40 MFMAs).  Here the barrier that publishes tile it+1 sits
        // BEFORE THE LAST SUB-STEP of tile it -- by then every wave holds all of tile it's fragments in registers -- and the first
        // fragments of tile it+1 are requested right behind it, so they land under the last sub-step's MFMAs; the slot of tile it is
        // refilled (tile it+NST) right after that barrier, a whole NST-1 K-steps before it is needed.  Same MFMAs in the same order:
        // bit-identical results.
        auto wait_tiles = [&](int k) {            // at most k later tiles' DMAs of this wave may still be in flight
            if (k <= 0) svcmi_dma_wait_n<0>();
            else if (k == 1) svcmi_dma_wait_n<DMAS>();
            else svcmi_dma_wait_n<(NST >= 3 ? 2 : 1) * DMAS>();
            static_assert(NST <= 4, "wait_tiles counts at most two later tiles (steady state: NST - 2)");
        };
        if (it_beg < it_end) {
            svcmi_f32x4 a4[2][FA], b4[2][FB];
            svcmi_f32x4 tie0 = {0.f, 0.f, 0.f, 0.f};
            {
                const int issued = it_end - it_beg < NST ? it_end - it_beg : NST;
                wait_tiles(issued - 1);
                __syncthreads();
                load_frags(As0 + a_off, Bs0 + b_off, 0, a4[0], b4[0], tie0);
                frags_arrive(a4[0], b4[0]);
            }
            // one K-step; NEXT / REFILL are compile-time so that the steady state is ONE basic block (three loops below: steady, drain, last)
            constexpr bool SPREAD2 = SVCMI_GEMM_SPREAD2 != 0 && SPREAD && NST >= 3 && NSUB >= 2;
            auto step = [&](int it, int slot, int nslot, auto next_tag, auto refill_tag, auto prev_tag) {
                constexpr bool NEXT = decltype(next_tag)::value, REFILL = decltype(refill_tag)::value;
                constexpr bool PREV = decltype(prev_tag)::value && SPREAD2;      // the previous K-step issued the first half of a refill: its second half goes here
                const float* Ab = As0 + slot * NA * BM * BK + a_off;
                const float* Bb = Bs0 + slot * NB * BTILE + b_off;
                if constexpr (SPREAD) {
                    // Pinned placement.  MFMA n of a sub-step is (c, i, j) = (n / (WM*WN), (n / WN) % WM, n % WN); a request "after MFMA n" names
                    // the sub-step's A fragment(s) read-write: every MFMA consumes one of them, so the MFMAs issued before the statement
                    // stay above it and the later ones below -- an exact position in the stream, at no cost in instructions or registers.
                    // Fragment request k of the NEXT sub-step goes in front of MFMA k (k = 0: above the sub-step, as before); in the last
                    // sub-step the refill's DMA pieces follow, spread evenly over the remaining MFMAs.  Nothing moves across the mid-step
                    // barrier or across a wait: the requests of a barrier interval are the plain loop's, in the plain loop's order.
                    constexpr int NACC = WM * WN, NM = 4 * NACC, NF = FA + FB;
                    static_assert(NF < NM, "one fragment request per MFMA, and room for the refill behind them");
                    static_assert(NSUB % 2 == 0, "a4 / b4 are double-buffered across K-steps by sub-step parity");
                    constexpr int P1 = SPREAD2 ? PIECES / 2 : PIECES;            // pieces issued behind the mid-step barrier; the rest one K-step later
                    constexpr int EARLY = (NSUB - 1) * NM;                      // MFMAs of the sub-steps in front of the barrier
                    const int pslot = slot == 0 ? NST - 1 : slot - 1;          // the slot the previous K-step retired
#pragma unroll
                    for (int sb = 0; sb < NSUB; ++sb) {
                        svcmi_f32x4(&af)[FA] = a4[sb & 1];
                        svcmi_f32x4(&bf)[FB] = b4[sb & 1];
                        svcmi_f32x4(&na)[FA] = a4[(sb + 1) & 1];
                        svcmi_f32x4(&nb)[FB] = b4[(sb + 1) & 1];
                        const bool last = sb + 1 == NSUB;
                        const bool want = !last || NEXT;                          // fragments to request during this sub-step?
                        const float* nAb = Ab;
                        const float* nBb = Bb;
                        int ns = sb + 1;
                        if (last) {
                            if constexpr (NEXT) {
#if SVCMI_PROBE_KTRACE
                                asm volatile("s_memtime %0" : "=s"(kt0));
#endif
                                if constexpr (REFILL) {
                                    wait_tiles(NST - 2);                     // steady state: tiles it+2 .. it+NST-1 may still be in flight
                                } else {
                                    const int lastt = it + NST - 1 < it_end - 1 ? it + NST - 1 : it_end - 1;      // newest tile issued
                                    wait_tiles(lastt - (it + 1));
                                }
#if SVCMI_PROBE_KTRACE
                                asm volatile("s_memtime %0" : "=s"(kt1));
#endif
                                __syncthreads();         // tile it+1 has landed for every wave; every wave holds all of tile it in registers
#if SVCMI_PROBE_KTRACE
                                asm volatile("s_memtime %0\n\ts_memrealtime %1" : "=s"(kt2), "=s"(kt_r1));
#endif
                                nAb = As0 + nslot * NA * BM * BK + a_off;
                                nBb = Bs0 + nslot * NB * BTILE + b_off;
                                ns = 0;
                                if constexpr (REFILL) stage_prep(it + NST);
                            }
                        }
                        if (want) svcmi_lds_read16(na[0], frag_src(nAb, nBb, ns, 0), af[0]);
#pragma unroll
                        for (int n = 0; n < NM; ++n) {
                            const int c = n / NACC, i = (n / WN) % WM, j = n % WN;
                            mma(acc[i][j], af[i][c], bf[j][c]);
                            if (n + 1 < NM) {
                                if (want && n + 1 < NF) {
                                    if constexpr (WM == 1) load_frag_at(nAb, nBb, ns, n + 1, na, nb, af[0]);
                                    else load_frag_at(nAb, nBb, ns, n + 1, na, nb, af[0], af[1]);
                                }
                                if constexpr (PREV) {
                                    if (!last) {
#pragma unroll
                                        for (int q = P1; q < PIECES; ++q) {
                                            const int g = ((2 * (q - P1) + 1) * EARLY) / (2 * (PIECES - P1));
                                            if (g / NM != sb || (g % NM < NM - 1 ? g % NM : NM - 2) != n) continue;
                                            if constexpr (WM == 1) {
                                                if (q < APCS) stage_a(it + NST - 1, pslot, q, af[0]);
                                                else stage_b(pslot, q - APCS, af[0]);
                                            } else {
                                                if (q < APCS) stage_a(it + NST - 1, pslot, q, af[0], af[1]);
                                                else stage_b(pslot, q - APCS, af[0], af[1]);
                                            }
                                        }
                                    }
                                }
                                if constexpr (NEXT && REFILL) {
                                    if (last) {
#pragma unroll
                                        for (int q = 0; q < P1; ++q) {
                                            if ((NF - 1) + q * (NM - NF) / P1 != n) continue;
                                            if constexpr (WM == 1) {
                                                if (q < APCS) stage_a(it + NST, slot, q, af[0]);
                                                else stage_b(slot, q - APCS, af[0]);
                                            } else {
                                                if (q < APCS) stage_a(it + NST, slot, q, af[0], af[1]);
                                                else stage_b(slot, q - APCS, af[0], af[1]);
                                            }
                                        }
                                    }
                                }
                            }
                        }
                        if (want) {             // one wait below this sub-step's MFMAs (they read af), every requested fragment pinned behind it
                            if constexpr (WM == 1) svcmi_lds_arrive(na[0], af[0]);
                            else svcmi_lds_arrive(na[0], af[0], af[1]);
#pragma unroll
                            for (int i = 1; i < FA; ++i) svcmi_lds_landed(na[i]);
#pragma unroll
                            for (int j = 0; j < FB; ++j) svcmi_lds_landed(nb[j]);
                        }
#if SVCMI_PROBE_KTRACE
                        if (last && NEXT) {          // (the s_waitcnt lgkmcnt(0) above also covers the three s_memtime results)
                            asm volatile("" : "+s"(kt0), "+s"(kt1), "+s"(kt2));
                            kt_vm += kt1 - kt0; kt_bar += kt2 - kt1;
                            asm volatile("" : "+s"(kt_r1));
                            if (kt_prev) { kt_step += kt2 - kt_prev; ++kt_n; } else { kt_r0 = kt_r1; kt_first = kt2; }
                            kt_prev = kt2;
                        }
#endif
                    }
                    return;
                }
#pragma unroll
                for (int sb = 0; sb < NSUB; ++sb) {
                    svcmi_f32x4(&af)[FA] = a4[sb & 1];
                    svcmi_f32x4(&bf)[FB] = b4[sb & 1];
                    if (sb + 1 < NSUB) {
                        load_frags(Ab, Bb, sb + 1, a4[(sb + 1) & 1], b4[(sb + 1) & 1], af[0]);
                    } else if constexpr (NEXT) {
                        if constexpr (REFILL) {
                            wait_tiles(NST - 2);                     // steady state: tiles it+2 .. it+NST-1 may still be in flight
                        } else {
                            const int last = it + NST - 1 < it_end - 1 ? it + NST - 1 : it_end - 1;      // newest tile issued
                            wait_tiles(last - (it + 1));
                        }
                        __syncthreads();         // tile it+1 has landed for every wave; every wave holds all of tile it in registers
                        load_frags(As0 + nslot * NA * BM * BK + a_off, Bs0 + nslot * NB * BTILE + b_off, 0, a4[0], b4[0], af[0]);
                        if constexpr (REFILL) {
                            stage_prep(it + NST);
#pragma unroll
                            for (int q = 0; q < PIECES; ++q) {
                                if (q < APCS) stage_a(it + NST, slot, q);
                                else stage_b(slot, q - APCS);
                            }
                        }
                    }
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i][0], bf[j][0]);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i][1], bf[j][1]);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i][2], bf[j][2]);
#pragma unroll
                    for (int i = 0; i < WM; ++i)
#pragma unroll
                        for (int j = 0; j < WN; ++j) mma(acc[i][j], af[i][3], bf[j][3]);
                    if (sb + 1 < NSUB || NEXT) {
#pragma unroll
                        for (int i = 0; i < WM; ++i)
#pragma unroll
                            for (int j = 0; j < WN; ++j) svcmi_pin(acc[i][j]);
                        frags_arrive(a4[(sb + 1) & 1], b4[(sb + 1) & 1]);
                    }
                }
            };
            int it = it_beg, slot = 0;
            auto advance = [&] { ++it; if (++slot == NST) slot = 0; };
            auto nxt = [&] { return slot + 1 == NST ? 0 : slot + 1; };
            if constexpr (SPREAD2) {
                // (a refill's second half is issued by the K-step AFTER the one that began it: first steady step and the steps behind the
                //  first drain step have none to issue; >= 1 drain step follows the steady state, so the last tile never has one either)
                bool steady = false;
                if (it + NST < it_end) { step(it, slot, nxt(), std::true_type(), std::true_type(), std::false_type()); advance(); steady = true; }
                for (; it + NST < it_end; advance()) step(it, slot, nxt(), std::true_type(), std::true_type(), std::true_type());
                if (it + 1 < it_end) {
                    if (steady) step(it, slot, nxt(), std::true_type(), std::false_type(), std::true_type());
                    else step(it, slot, nxt(), std::true_type(), std::false_type(), std::false_type());
                    advance();
                }
                for (; it + 1 < it_end; advance()) step(it, slot, nxt(), std::true_type(), std::false_type(), std::false_type());
                step(it, slot, nxt(), std::false_type(), std::false_type(), std::false_type());
            } else {
            for (; it + NST < it_end; advance()) step(it, slot, nxt(), std::true_type(), std::true_type(), std::false_type());       // steady state
            for (; it + 1 < it_end; advance()) step(it, slot, nxt(), std::true_type(), std::false_type(), std::false_type());       // drain: nothing left to request
            step(it, slot, nxt(), std::false_type(), std::false_type(), std::false_type());                                          // last tile
            }
        }
    } else {
        int it = it_beg, slot = 0;
        for (; it + NST - 1 < it_end; ++it) {             // steady state: NST-2 later tiles in flight
            k_step(it, slot, NST - 2, std::true_type());
            if (++slot == NST) slot = 0;
        }
        for (; it < it_end; ++it) {                       // drain
            const int rem = it_end - 1 - it;
            k_step(it, slot, rem < NST - 2 ? rem : NST - 2, std::false_type());
            if (++slot == NST) slot = 0;
        }
    }
    svcmi_dma_wait();        // (nothing is outstanding after the last tile; a K range of zero steps still waits for the bias here)
    __syncthreads();         // last tile fully consumed before the buffers are reused below
#if SVCMI_PROBE_KTRACE
    const bool kt_on = SVCMI_PROBE_KTRACE_N == 0 || p.n_out == SVCMI_PROBE_KTRACE_N;
    if (kt_on && lane == 0 && block_id < 8192 && wave < 4) {
        unsigned long long* kt = g_ktrace + 4 * (block_id * 4 + wave);
        kt[0] = kt_step; kt[1] = kt_vm; kt[2] = kt_bar; kt[3] = kt_n;
        if (block_id == 0 && wave == 0) { g_ktrace[4 * 4 * 8192] = kt_prev - kt_first; g_ktrace[4 * 4 * 8192 + 1] = kt_r1 - kt_r0; g_ktrace[4 * 4 * 8192 + 2] = kt_first - kt_entry; }
    }
    unsigned long long kt_a, kt_b = 0;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(kt_a)::"memory");
    auto kt_finish = [&] {
        if (tid == 0) {       // timeline record of this block (every launch, every block)
            unsigned long long te;
            asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(te)::"memory");
            const unsigned idx = atomicAdd(&g_tl_n, 1u);
            if (idx < TL_CAP) {
                unsigned long long* r = g_tl + 5ull * idx;
                r[0] = tl_entry; r[1] = te; r[2] = (unsigned long long)(size_t)p.y ^ ((unsigned long long)(size_t)p.ws << 1);
                r[3] = ((unsigned long long)(unsigned)p.n_out << 32) | (unsigned)p.ktot; r[4] = ((unsigned long long)(unsigned)grid_blocks << 32) | (unsigned)p.t_out;
            }
        }
        if (kt_on && block_id == 0 && tid == 0) {
            unsigned long long t, tc;
            asm volatile("s_memtime %1\n\ts_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t), "=s"(tc)::"memory");
            g_ktrace[4 * 4 * 8192 + 3] = t - kt_prev;
            g_ktrace[4 * 4 * 8192 + 4] += 1;
            g_ktrace[4 * 4 * 8192 + 5] = kt_a - kt_prev;        // the K-steps behind the last stamped barrier
            g_ktrace[4 * 4 * 8192 + 6] = kt_b - kt_a;           // accumulators -> LDS + barrier
            g_ktrace[4 * 4 * 8192 + 7] = tc - kt_b;             // the epilogue loop (loads, activation, stores issued)
        }
    };
#else
    auto kt_finish = [] {};
#endif

    // Epilogue through LDS: the accumulators (D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)) are
    // laid out as a [BM][BN] tile so that the (rolled, single-copy) epilogue loop walks n fastest and
    // every store instruction writes 256 contiguous bytes.  The loop's last barrier already retired all
    // operand reads, so the buffers can be reused.
    float* Cs = smem;
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int r = 0; r < ACC_N; ++r) {
                // D layouts: 32x32 -> col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5);  16x16 -> col = lane&15, row = 4*(lane>>4) + r
                const int ml = P16 ? (wm * 16 * WM + i * 16 + 4 * (lane >> 4) + r)
                                   : (wm * 32 * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5));
                const int nl = P16 ? (j * 16 + (lane & 15)) : (wn * 32 * WN + j * 32 + (lane & 31));
                Cs[ml * CLD + nl] = acc[i][j][r];
            }
    __syncthreads();
#if SVCMI_PROBE_KTRACE
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(kt_b)::"memory");
#endif
    const int nvalid = (p.n_out - n0) < BN ? (p.n_out - n0) : BN;
    const int mvalid = (p.t_out - m0) < BM ? (p.t_out - m0) : BM;
    if (p.split > 1 || (p.flags & SVCMI_CONV_PARTIALS)) {   // raw partial tile into this slice's slab
        float* wsb = p.ws + ((long long)b * p.split + slice) * p.t_out * p.n_out;
        if (!p.cnt) {        // no ticket counters: splitk_reduce_kernel sums the slabs in a second launch
            if (p.vec) {
                for (int e = tid; e < BM * (BN / 4); e += NT) {
                    const int ml = e / (BN / 4), nl = (e - ml * (BN / 4)) * 4;
                    if (ml < mvalid && nl < nvalid)
                        *reinterpret_cast<float4*>(wsb + (long long)(m0 + ml) * p.n_out + n0 + nl) = *reinterpret_cast<const float4*>(Cs + ml * CLD + nl);
                }
                kt_finish();
                return;
            }
            for (int e = tid; e < BM * BN; e += NT) {
                const int ml = e / BN, nl = e - ml * BN;
                if (ml < mvalid && nl < nvalid) wsb[(long long)(m0 + ml) * p.n_out + n0 + nl] = Cs[ml * CLD + nl];
            }
            kt_finish();
            return;
        }
        // In-launch combine (cdna_hip_programming.md section 5, "in-launch split-K reduction", write-through form): the
        // slab goes out with 16-byte sc1 (write-through) stores, which need no release fence -- a per-block
        // `buffer_wbl2` made this 2.6x slower than the two-kernel path -- then every wave drains its stores and ONE
        // relaxed agent-scope ticket is drawn.  The block that draws the last ticket of its tile acquires once and
        // sums all slabs IN SLICE ORDER (its own included, from memory): the result does not depend on who is last.
        {
            const svcmi_rsrc sr = svcmi_make_rsrc(wsb, (unsigned)p.t_out * (unsigned)p.n_out * 4u);   // n_out % 4 == 0 here
            for (int e = tid; e < BM * BN / 4; e += NT) {
                const int ml = e / (BN / 4), nl = (e - ml * (BN / 4)) * 4;
                if (ml < mvalid && nl < nvalid)
                    svcmi_store16_sc1(*reinterpret_cast<const svcmi_f32x4*>(Cs + ml * CLD + nl), sr,
                                      (unsigned)((m0 + ml) * p.n_out + n0 + nl) * 4u);
            }
        }
        svcmi_dma_wait();                                   // every wave: its slab stores have left
        __syncthreads();
        int* const flag = reinterpret_cast<int*>(smem + BM * CLD);      // spare word behind the C tile (one LDS object only)
        const int tile_id = (b * p.nt + by) * p.mt + bx;
        if (tid == 0) *flag = svcmi_ticket(p.cnt + tile_id);
        __syncthreads();
        if (*flag != p.split - 1) { kt_finish(); return; }
        if (tid == 0) {
            SVCMI_ACQUIRE_AGENT();
            p.cnt[tile_id] = 0;                             // leave the counter ready for the next launch
        }
        __syncthreads();
        float* yb2 = p.y + (long long)b * p.y_bs;
        const float* rb2 = p.res ? p.res + (long long)b * p.r_bs : nullptr;
        const float* ws0 = p.ws + (long long)b * p.split * p.t_out * p.n_out;
        const long long sstride = (long long)p.t_out * p.n_out;
        for (int e = tid; e < BM * BN; e += NT) {
            const int ml = e / BN, nl = e - ml * BN;
            if (ml >= mvalid || nl >= nvalid) continue;
            const int t = m0 + ml, n = n0 + nl;
            const float* src = ws0 + (long long)t * p.n_out + n;
            float v = 0.f;
            for (int sl = 0; sl < p.split; ++sl) v += src[sl * sstride];
            float* dst = yb2 + (long long)t * p.ldy + n;
            *dst = epilogue(p, v, smem[BIAS0 + nl], rb2 ? rb2 + (long long)t * p.ldr : nullptr, dst, n,
                            (p.flags & SVCMI_CONV_MASK_OUT) && t >= len);
        }
        kt_finish();
        return;
    }
    float* yb = p.y + (long long)b * p.y_bs;
    const float* rbp = p.res ? p.res + (long long)b * p.r_bs : nullptr;
    const bool mask_out = (p.flags & SVCMI_CONV_MASK_OUT) != 0;
    if (p.vec) {
        // Round 6 (s_memtime stamps in the kernel, profiles/r06h_ktrace_epilogue.log: this loop was 18 000 of the 150 000 cycles of a Whisper
        // MLP-up launch and ended with no store outstanding -- it ran at the pace of the store ACKNOWLEDGEMENTS):
        //  * gfx9 counts loads and stores in one counter, and with a conditional load in the loop hipcc opened every iteration with
        //    s_waitcnt vmcnt(0), i.e. a wait for the previous iteration's store.  Now every memory operation is unconditional: 16-byte
        //    buffer accesses whose per-lane offset the hardware range-checks (an absent residual = zero records, an invalid lane or an
        //    unused accumulate operand = offset 2^31), the operands of iteration i + 1 requested before iteration i is computed -- the
        //    compiler counts vmcnt exactly, and a launch without residual / accumulate has no load and no wait at all;
        //  * the bias comes from the tile's LDS copy (one 4-byte LDS-DMA per 64 columns in front of the first operand tile);
        //  * the activation is chosen OUTSIDE the loop: with the switch inside, every element walked ~10 scalar branches to its arm.
        const bool acc_on = (p.flags & SVCMI_CONV_ACCUMULATE) != 0;
        constexpr int EP_IT = BM * (BN / 4) / NT;
        static_assert(BM * (BN / 4) % NT == 0, "whole epilogue iterations");
        constexpr unsigned NOWHERE = 0x80000000u;
        const svcmi_brsrc ry = svcmi_make_brsrc(yb + (long long)m0 * p.ldy + n0, 0x7fffffffu);           // tile origin; lane offsets < 128 rows x ld (prepare: ld < 2^22)
        const svcmi_brsrc rr = svcmi_make_brsrc(rbp ? rbp + (long long)m0 * p.ldr + n0 : nullptr, rbp ? 0x7fffffffu : 0u);
        auto loop = [&](auto act_tag, auto loads_tag) {
            constexpr int ACT = decltype(act_tag)::value;
            constexpr bool LOADS = decltype(loads_tag)::value;
            svcmi_f32x4 rv_n = {0.f, 0.f, 0.f, 0.f}, yo_n = {0.f, 0.f, 0.f, 0.f};
            auto where = [&](int e, unsigned& oy, unsigned& orr) {
                const int ml = e / (BN / 4), nl = (e - ml * (BN / 4)) * 4;
                const bool ok = ml < mvalid && nl < nvalid;
                oy = ok ? (unsigned)(ml * p.ldy + nl) * 4u : NOWHERE;
                orr = ok ? (unsigned)(ml * p.ldr + nl) * 4u : NOWHERE;
            };
            unsigned oy_n, or_n;
            where(tid, oy_n, or_n);
            if constexpr (LOADS) {
                rv_n = svcmi_buf_load16(rr, or_n);
                yo_n = svcmi_buf_load16(ry, acc_on ? oy_n : NOWHERE);
            }
#pragma nounroll
            for (int i = 0; i < EP_IT; ++i) {
                const int e = tid + NT * i;
                const int ml = e / (BN / 4), nl = (e - ml * (BN / 4)) * 4;
                const unsigned oy = oy_n;
                const svcmi_f32x4 rv = rv_n, yo = yo_n;
                if constexpr (LOADS) {
                    where(e + NT, oy_n, or_n);              // (past the last iteration: rows >= BM >= mvalid, nothing is fetched)
                    rv_n = svcmi_buf_load16(rr, or_n);
                    yo_n = svcmi_buf_load16(ry, acc_on ? oy_n : NOWHERE);
                } else {
                    where(e + NT, oy_n, or_n);
                }
                const int t = m0 + ml;
                const float4 c = *reinterpret_cast<const float4*>(Cs + ml * CLD + nl), bs = *reinterpret_cast<const float4*>(smem + BIAS0 + nl);
                const bool masked = mask_out && t >= len;
                const float cv[4] = {c.x, c.y, c.z, c.w}, bv[4] = {bs.x, bs.y, bs.z, bs.w};
                svcmi_f32x4 o;
#pragma unroll
                for (int k = 0; k < 4; ++k) {       // the arithmetic of `epilogue`, per component
                    float q = act_apply(cv[k] + bv[k], ACT);
                    if (LOADS && rbp) q += rv[k];
                    q *= p.alpha;
                    if (LOADS && acc_on) q += yo[k];
                    o[k] = masked ? 0.f : q;
                }
                svcmi_buf_store16(o, ry, oy);
                if (p.y16 && oy != NOWHERE)
                    svcmi_store4_16(p.y16 + (long long)b * p.y16_bs + (long long)t * p.ldy16 + n0 + nl, p.ldy16 >> 1, o[0], o[1], o[2], o[3], p.y16_f16);
            }
        };
        auto by_act = [&](auto loads_tag) {
            switch (p.act) {
                case SVCMI_ACT_NONE: loop(std::integral_constant<int, SVCMI_ACT_NONE>(), loads_tag); break;
                case SVCMI_ACT_RELU: loop(std::integral_constant<int, SVCMI_ACT_RELU>(), loads_tag); break;
                case SVCMI_ACT_GELU: loop(std::integral_constant<int, SVCMI_ACT_GELU>(), loads_tag); break;
                case SVCMI_ACT_MISH: loop(std::integral_constant<int, SVCMI_ACT_MISH>(), loads_tag); break;
                case SVCMI_ACT_TANH: loop(std::integral_constant<int, SVCMI_ACT_TANH>(), loads_tag); break;
                default: loop(std::integral_constant<int, SVCMI_ACT_SIGMOID>(), loads_tag); break;
            }
        };
        if (rbp || acc_on) by_act(std::true_type());
        else by_act(std::false_type());
        kt_finish();
        return;
    }
    for (int e = tid; e < BM * BN; e += NT) {
        const int ml = e / BN, nl = e - ml * BN;
        if (ml >= mvalid || nl >= nvalid) continue;
        const int t = m0 + ml, n = n0 + nl;
        float* dst = yb + (long long)t * p.ldy + n;
        *dst = epilogue(p, Cs[ml * CLD + nl], smem[BIAS0 + nl], rbp ? rbp + (long long)t * p.ldr : nullptr, dst, n,
                        mask_out && t >= len);
    }
    kt_finish();
}

// Build experiment (scripts/build_variant.sh nst2 -DSVCMI_GEMM_NST=2): ring depth of the single-launch fp32 kernels.  The default 3-deep
// ring of the 64x80 / 64x64 tiles is 61 / 49 KB of LDS = 2 / 3 resident blocks per CU; 2-deep is 41 / 33 KB = 3 / 4 blocks, and leaves
// room for another lane's vector-ALU blocks beside two GEMM blocks when clips are in flight.
#ifndef SVCMI_GEMM_NST
#define SVCMI_GEMM_NST 0
#endif
template <int WM, int WN, int MODE, bool P16, int PREC = PREC_F32, int NSTO = 0, int NW = 4>
__global__ __launch_bounds__(64 * NW) void conv_gemm_kernel(ConvArgs p) {
    conv_gemm_body<WM, WN, MODE, P16, (NSTO ? NSTO : (PREC == PREC_F32 && WM * WN <= 5 ? SVCMI_GEMM_NST : 0)), PREC, NW>(p, (int)gridDim.x, (int)blockIdx.x);
}

// Grouped launch: up to GROUP_MAX problems of identical tile policy / gather mode in ONE grid, blocks of problem 0 first.  The
// generator runs the three AMP blocks of a stage (same shapes, 3 / 7 / 11 taps) this way: one launch carries 3x the blocks
// of a single convolution, so the 20000 x 80 and 80000 x 40 problems fill the 256 CUs several blocks deep without relying
// on multi-stream concurrency, and the long-K problem goes first so the short ones fill the tail.
constexpr int GROUP_MAX = 3;
struct GroupArgs {
    ConvArgs p[GROUP_MAX];
    int first[GROUP_MAX + 1];      // first[i] = first block of problem i; first[count..] = grid size
};

// NSTO: ring depth override -- the grouped problems have 4 .. 55 K-steps and thousands of blocks, where a 2-deep ring
// (one more resident block per CU) can beat the 3-deep one tuned for the long-K Whisper GEMMs.
template <int WM, int WN, int MODE, bool P16, int NSTO, int PREC = PREC_F32>
__global__ __launch_bounds__(256) void conv_gemm_group_kernel(GroupArgs g) {
    const int id = (int)blockIdx.x;
    const int gi = id >= g.first[2] ? 2 : (id >= g.first[1] ? 1 : 0);      // block-uniform: the arguments stay scalar loads
    conv_gemm_body<WM, WN, MODE, P16, NSTO, PREC>(g.p[gi], g.first[gi + 1] - g.first[gi], id - g.first[gi]);
}

// y = epilogue(sum over slices, fixed order).  One thread per 4 consecutive n (n_out % 4 handled by a scalar tail).
__global__ __launch_bounds__(256) void splitk_reduce_kernel(ConvArgs p, int batch) {
    const long long total = (long long)batch * p.t_out * p.n_out;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int n = (int)(e % p.n_out);
        const long long bt = e / p.n_out;
        const int t = (int)(bt % p.t_out), b = (int)(bt / p.t_out);
        const float* src = p.ws + ((long long)b * p.split * p.t_out + t) * p.n_out + n;
        float v = 0.f;
        for (int s = 0; s < p.split; ++s) v += src[(long long)s * p.t_out * p.n_out];
        const int len = p.lengths ? p.lengths[b] : 0x7fffffff;
        float* dst = p.y + (long long)b * p.y_bs + (long long)t * p.ldy + n;
        const float* rr = p.res ? p.res + (long long)b * p.r_bs + (long long)t * p.ldr : nullptr;
        *dst = epilogue(p, v, p.bias ? p.bias[n] : 0.f, rr, dst, n, (p.flags & SVCMI_CONV_MASK_OUT) && t >= len);
    }
}

static int g_last_ring = 0;        // ring depth of the last single fp32 launch's instantiation (svcmi_tune_get("last_conv_ring"): tests assert the dispatch)

template <int WM, int WN, bool P16, int PREC = PREC_F32, int NW = 4>
int launch(const ConvArgs& a_in, int batch, int mode, void* stream) {
    constexpr int BM = P16 ? 16 * WM * NW : 64 * WM, BN = P16 ? 16 * WN : 64 * WN;
    ConvArgs a = a_in;
    a.mt = (a.t_out + BM - 1) / BM;
    a.nt = (a.n_out + BN - 1) / BN;
    const long long blocks = (long long)a.mt * a.nt * batch * a.split;
    if (blocks > 0x7fffffffLL) return SVCMI_EUNSUPPORTED;
    dim3 grid((unsigned)blocks);
    // SVCMI_CONV_RING2: the 2-deep ring of the 64-row fp32 tiles (one more resident block per CU; for launches that share the chip)
    constexpr bool HAS_RING2 = PREC == PREC_F32 && WM == 1 && (P16 ? (WN == 3 || WN == 5) : WN == 1);
    if constexpr (NW == 8) {          // eight-wave blocks (fp32, 16x16x4 policy, vector gathers): 3-deep ring, or 2-deep with SVCMI_CONV_RING2
        if (mode != MODE_CHUNK && mode != MODE_VEC) return SVCMI_EUNSUPPORTED;
        const bool r2 = (a.flags & SVCMI_CONV_RING2) != 0;
        g_last_ring = r2 ? 2 : 3;
        if (mode == MODE_CHUNK) {
            if (r2) SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_CHUNK, P16, PREC, 2, 8>), grid, dim3(512), 0, stream, a);
            else SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_CHUNK, P16, PREC, 3, 8>), grid, dim3(512), 0, stream, a);
        } else {
            if (r2) SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_VEC, P16, PREC, 2, 8>), grid, dim3(512), 0, stream, a);
            else SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_VEC, P16, PREC, 3, 8>), grid, dim3(512), 0, stream, a);
        }
        int rc8 = SVCMI_LAST_ERROR();
        if (rc8 == 0 && a.split > 1 && !a.cnt && !(a.flags & SVCMI_CONV_PARTIALS)) {
            const long long total = (long long)batch * a.t_out * a.n_out;
            long long nb = (total + 255) / 256;
            if (nb > 2048) nb = 2048;
            SVCMI_LAUNCH(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a, batch);
            rc8 = SVCMI_LAST_ERROR();
        }
        return rc8;
    }
    bool ring2 = false;
    if constexpr (HAS_RING2) ring2 = (a.flags & SVCMI_CONV_RING2) && (mode == MODE_CHUNK || mode == MODE_VEC);
    if constexpr (PREC == PREC_F32) g_last_ring = ring2 ? 2 : 3;       // (the default depth of the 128-row tiles is 2 as well: only the 64-row tiles have both)
    if (ring2) {
        if constexpr (HAS_RING2) {
            if (mode == MODE_CHUNK) SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_CHUNK, P16, PREC, 2>), grid, dim3(256), 0, stream, a);
            else SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_VEC, P16, PREC, 2>), grid, dim3(256), 0, stream, a);
        }
    }
    else if (mode == MODE_CHUNK) SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_CHUNK, P16, PREC>), grid, dim3(256), 0, stream, a);
    else if (mode == MODE_VEC) SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_VEC, P16, PREC>), grid, dim3(256), 0, stream, a);
    else if constexpr (!P16) {
        if (mode == MODE_CHUNK_RS) {
            if constexpr (PREC < PREC_BF16_A16) SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_CHUNK_RS, false, PREC>), grid, dim3(256), 0, stream, a);
            else return SVCMI_EUNSUPPORTED;      // the fused row repeat reads fp32 rows only
        } else if constexpr (PREC == PREC_F32) SVCMI_LAUNCH((conv_gemm_kernel<WM, WN, MODE_SCALAR, false>), grid, dim3(256), 0, stream, a);
        else return SVCMI_EUNSUPPORTED;     // per-element gathers (c_in % 4 != 0) stay on the fp32 kernel
    } else {
        return SVCMI_EUNSUPPORTED;
    }
    int rc = SVCMI_LAST_ERROR();
    if (rc == 0 && a.split > 1 && !a.cnt && !(a.flags & SVCMI_CONV_PARTIALS)) {
        const long long total = (long long)batch * a.t_out * a.n_out;
        long long nb = (total + 255) / 256;
        if (nb > 2048) nb = 2048;
        SVCMI_LAUNCH(splitk_reduce_kernel, dim3((unsigned)nb), dim3(256), 0, stream, a, batch);
        rc = SVCMI_LAST_ERROR();
    }
    return rc;
}

static int g_group_nst = 0;        // tuning knob (svcmi_tune_set("group_nst", 0 | 2 | 3)): ring depth of the grouped launches, 0 = default

template <int WM, int WN, bool P16, int PREC = PREC_F32>
int launch_group(GroupArgs& g, int count, int batch, int mode, void* stream) {
    constexpr int BM = 64 * WM, BN = P16 ? 16 * WN : 64 * WN;
    long long blocks = 0;
    for (int i = 0; i < GROUP_MAX; ++i) {
        g.first[i] = (int)blocks;
        if (i < count) {
            ConvArgs& a = g.p[i];
            a.mt = (a.t_out + BM - 1) / BM;
            a.nt = (a.n_out + BN - 1) / BN;
            blocks += (long long)a.mt * a.nt * batch;
            if (blocks > 0x7fffffffLL) return SVCMI_EUNSUPPORTED;
        }
    }
    g.first[GROUP_MAX] = (int)blocks;
    dim3 grid((unsigned)blocks);
    // measured (scripts/microbench.py group): the 16x16x4 tiles gain 7 % (80 channels) / 11 % (40) from the 2-deep ring (3 resp.
    // 4 resident blocks per CU), the 64 x 64 tile does not
    const int nst = g_group_nst ? g_group_nst : (P16 ? 2 : 3);
    if (mode != MODE_CHUNK && mode != MODE_VEC) return SVCMI_EUNSUPPORTED;
    if constexpr (PREC != PREC_F32) {       // one ring depth per policy (the knob is an fp32 tuning aid)
        constexpr int NSTL = P16 ? 2 : 3;
        if (mode == MODE_CHUNK) SVCMI_LAUNCH((conv_gemm_group_kernel<WM, WN, MODE_CHUNK, P16, NSTL, PREC>), grid, dim3(256), 0, stream, g);
        else SVCMI_LAUNCH((conv_gemm_group_kernel<WM, WN, MODE_VEC, P16, NSTL, PREC>), grid, dim3(256), 0, stream, g);
    } else if (nst == 2) {
        if (mode == MODE_CHUNK) SVCMI_LAUNCH((conv_gemm_group_kernel<WM, WN, MODE_CHUNK, P16, 2>), grid, dim3(256), 0, stream, g);
        else SVCMI_LAUNCH((conv_gemm_group_kernel<WM, WN, MODE_VEC, P16, 2>), grid, dim3(256), 0, stream, g);
    } else {
        if (mode == MODE_CHUNK) SVCMI_LAUNCH((conv_gemm_group_kernel<WM, WN, MODE_CHUNK, P16, 3>), grid, dim3(256), 0, stream, g);
        else SVCMI_LAUNCH((conv_gemm_group_kernel<WM, WN, MODE_VEC, P16, 3>), grid, dim3(256), 0, stream, g);
    }
    return SVCMI_LAST_ERROR();
}

// Validation + argument block + gather mode of one convolution (shared by the single and the grouped entry points).
// lp: d->w is the 16-bit image of svcmi_pack_weights_lp and d->ldw its leading dimension in 16-bit values.
int prepare(const svcmi_conv_desc* d, ConvArgs& a, int& mode, int prec = PREC_F32) {
    const bool lp = prec != PREC_F32, a16 = prec >= PREC_BF16_A16, x3a = prec == PREC_BF16X3_A16;
    if (!d || !d->x || !d->w || !d->y) return SVCMI_EINVAL;
    if (a16 && (d->c_in % 8 || d->ldx % 8 || d->x_bstride % 8 || ((uintptr_t)d->x & 15) || d->x_row_shift)) return SVCMI_EUNSUPPORTED;
    if (x3a && (d->ldx % 16 || d->ldx / 2 < d->c_in)) return SVCMI_EUNSUPPORTED;     // rows [hi: ldx/2 | lo: ldx/2]
    if (d->y16 && (svcmi_fmt16(d->y16_format) < 0 || !svcmi_fmt16_row_ok(d->y16_format, d->ldy16, d->n_out) || d->y16_bstride % 4 ||
                   ((uintptr_t)d->y16 & 7) || (d->flags & SVCMI_CONV_PARTIALS) || d->split_k > 1))
        return SVCMI_EINVAL;
    /* 32-bit buffer offsets with a 2^30 out-of-range sentinel: each operand buffer stays below 2^29 bytes */
    if ((long long)d->t_in * d->ldx >= (1LL << 27) || (long long)d->n_out * d->ldw * ((prec == PREC_BF16X3 || x3a || prec == PREC_F16W2_A16) ? 2 : 1) >= (1LL << (lp ? 28 : 27))) return SVCMI_EUNSUPPORTED;
    if (((long long)d->ksize * d->dilation + d->pad) * d->ldx >= (1LL << 27)) return SVCMI_EUNSUPPORTED;
    if (d->batch <= 0 || d->t_in <= 0 || d->t_out <= 0 || d->c_in <= 0 || d->n_out <= 0 || d->ksize <= 0) return SVCMI_EINVAL;
    if (d->stride <= 0 || d->dilation <= 0 || d->x_row_shift < 0 || d->x_row_shift > 1) return SVCMI_EINVAL;
    if (d->ldw % (lp ? 32 : 4) != 0 || d->ldw < d->ksize * d->c_in) return SVCMI_EINVAL;
    if (d->ldy < d->n_out || (d->res && d->ldr < d->n_out) || d->ldx < d->c_in) return SVCMI_EINVAL;
    if ((d->flags & (SVCMI_CONV_MASK_IN | SVCMI_CONV_MASK_OUT)) && !d->lengths) return SVCMI_EINVAL;
    if (d->act < SVCMI_ACT_NONE || d->act > SVCMI_ACT_SIGMOID) return SVCMI_EINVAL;
    if (d->split_k < 0 || (d->split_k > 1 && !d->workspace)) return SVCMI_EINVAL;
    if (((uintptr_t)d->w & 15) != 0) return SVCMI_EALIGN;
    // magic-number division q / c_in is exact while q * c_in < 2^32 (q < ksize*c_in)
    if ((long long)d->ksize * d->c_in * d->c_in >= 0x100000000LL || (long long)d->ksize * d->c_in >= 0x7fffffffLL) return SVCMI_EUNSUPPORTED;

    a.w16 = lp ? reinterpret_cast<const unsigned short*>(d->w) : nullptr;
    a.ldw16 = lp ? d->ldw : 0;
    a.y16 = reinterpret_cast<unsigned short*>(d->y16); a.y16_bs = d->y16_bstride; a.ldy16 = d->ldy16; a.y16_f16 = svcmi_fmt16(d->y16_format);
    a.x = d->x; a.w = d->w; a.bias = d->bias; a.res = d->res; a.y = d->y; a.lengths = d->lengths;
    a.ws = d->workspace;
    a.cnt = nullptr;
    a.x_bs = d->x_bstride; a.y_bs = d->y_bstride; a.r_bs = d->res_bstride;
    a.t_in = d->t_in; a.t_out = d->t_out; a.c_in = d->c_in; a.ldx = d->ldx; a.n_out = d->n_out;
    a.ldw = d->ldw; a.ldy = d->ldy; a.ldr = d->ldr;
    a.ksize = d->ksize; a.stride = d->stride; a.dil = d->dilation; a.pad = d->pad; a.rshift = d->x_row_shift;
    a.act = d->act; a.flags = d->flags; a.alpha = d->alpha;
    {
        const auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
        a.vec = d->n_out % 4 == 0 && d->ldy % 4 == 0 && d->y_bstride % 4 == 0 && al16(d->y) && d->ldy < (1 << 22) && d->ldr < (1 << 22) &&      // (tile-local 32-bit buffer offsets)
                (!d->bias || al16(d->bias)) && (!d->res || (d->ldr % 4 == 0 && d->res_bstride % 4 == 0 && al16(d->res))) &&
                (!d->workspace || al16(d->workspace));
    }
    a.ktot = d->ksize * d->c_in;
    a.magic = d->c_in == 1 ? 0u : (unsigned)((0x100000000ULL + (unsigned)d->c_in - 1) / (unsigned)d->c_in);
    if (d->y16 && !a.vec) return SVCMI_EALIGN;      // the 16-bit copy is written by the float4 epilogue only
    const bool vec = (d->c_in % 4 == 0) && (d->ldx % 4 == 0) && (d->x_bstride % 4 == 0) && (((uintptr_t)d->x & 15) == 0);
    mode = !vec ? MODE_SCALAR : (d->c_in % (a16 ? 2 * BK : BK) == 0 ? MODE_CHUNK : MODE_VEC);      // CHUNK: a K-step (32 k; _A16: 64 k) lies inside one tap
    if (d->x_row_shift) mode = mode == MODE_CHUNK ? MODE_CHUNK_RS : MODE_SCALAR;   // the fused row repeat: CHUNK_RS or per-element
    return SVCMI_OK;
}

}  // namespace
