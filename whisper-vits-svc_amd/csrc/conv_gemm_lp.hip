// conv_gemm_lp.hip -- reduced-precision entry points of the implicit-GEMM convolution: 16-bit weight images, fp32 activations
// rounded in registers, fp32 accumulate, on v_mfma_f32_32x32x16_{bf16,f16} / v_mfma_f32_16x16x32_{bf16,f16} (kernel body and
// the description of the scheme: conv_gemm_body.h).  The reference itself drops to fp16 on an accelerator
// (whisper/inference.py:22-23,43-44: `.half()` model and mel); fp32 stays the parity default of this library and these modes
// are opt-in, each with its measured error against the fp32 oracle (tests/test_gpu_precision.py).
#include "conv_gemm_body.h"

namespace {

// One thread per 16-bit value of the image.  Position pos = 32*blk + 8*q + e of a row holds logical
// k = 32*blk + (e < 4 ? 4q + e : 16 + 4q + e - 4)  (chunk q pairs with the fp32 A chunks q and q + 4, conv_gemm_body.h).
__global__ __launch_bounds__(256) void pack_lp_kernel(const float* w, int n, int ldw, unsigned short* out, int ldw16, int prec) {
    const long long total = (long long)n * ldw16;
    const int nb = (prec == PREC_BF16X3 || prec == PREC_BF16X3_A16 || prec == PREC_F16W2_A16) ? 2 : 1;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int row = (int)(idx / ldw16), pos = (int)(idx - (long long)row * ldw16);
        const int blk = pos >> 5, q = (pos >> 3) & 3, e = pos & 7;
        const int k = prec >= PREC_BF16_A16 ? pos : blk * 32 + (e < 4 ? 4 * q + e : 16 + 4 * q + e - 4);      // _A16: natural order
        const float v = k < ldw ? w[(long long)row * ldw + k] : 0.f;
        unsigned short* dst = out + (long long)row * nb * ldw16 + pos;
        if (prec == PREC_F16 || prec == PREC_F16_A16) {
            dst[0] = (unsigned short)(svcmi_cvt_pk_f16(v, 0.f) & 0xffffu);
        } else if (prec == PREC_F16W2_A16) {        // hi = fp16(w), lo = fp16(w - hi): lo is mostly SUBNORMAL fp16, which the matrix cores honour
            const unsigned h = svcmi_cvt_pk_f16(v, 0.f) & 0xffffu;
            dst[0] = (unsigned short)h;
            dst[ldw16] = (unsigned short)(svcmi_cvt_pk_f16(v - svcmi_f16_bits_f32(h), 0.f) & 0xffffu);
        } else {
            const unsigned h = svcmi_cvt_pk_bf16(v, 0.f) & 0xffffu;
            dst[0] = (unsigned short)h;
            if (nb == 2) dst[ldw16] = (unsigned short)(svcmi_cvt_pk_bf16(v - svcmi_bits_f32(h << 16), 0.f) & 0xffffu);
        }
    }
}

struct TileChoice { int wm, wn; bool p16; };

// Tile of one problem: explicit override bits, else right-sized 16x16x32 tiles for the narrow generator stages, else the
// largest 32x32x16 tile that still gives the 256 CUs a block each (64x128 amortises the in-register rounding of an A
// fragment over two B fragments; 64x64 is for narrow N and split-K).
int choose_tile(const svcmi_conv_desc* d, int mode, TileChoice& t) {
    const int tile = d->flags & SVCMI_CONV_TILE_MASK;
    const int n16 = (d->n_out + 15) / 16;
    const bool p16_ok = mode == MODE_CHUNK || mode == MODE_VEC;
    t = TileChoice{1, 1, false};
    switch (tile) {
        case 0: break;
        case SVCMI_CONV_TILE_64x64: return SVCMI_OK;
        case SVCMI_CONV_TILE_64x128: t.wn = 2; return SVCMI_OK;
        case SVCMI_CONV_TILE_128x128: t.wm = t.wn = 2; return SVCMI_OK;
        case SVCMI_CONV_TILE_P16_64x48: if (!p16_ok) return SVCMI_EUNSUPPORTED; t = TileChoice{1, 3, true}; return SVCMI_OK;
        case SVCMI_CONV_TILE_P16_64x80: if (!p16_ok) return SVCMI_EUNSUPPORTED; t = TileChoice{1, 5, true}; return SVCMI_OK;
        default: return SVCMI_EUNSUPPORTED;
    }
    if (p16_ok && (n16 == 3 || n16 == 5) && d->t_out >= 1024) { t = TileChoice{1, n16, true}; return SVCMI_OK; }
    const long long mt64 = (d->t_out + 63) / 64, mt128 = (d->t_out + 127) / 128, nt128 = (d->n_out + 127) / 128;
    if (d->n_out > 64 && mt128 * nt128 * d->batch >= 1024 && d->t_out >= 128) { t.wm = t.wn = 2; return SVCMI_OK; }
    // long K (CREPE's layer 2: K = 64 taps x 1024 channels; a 10 s clip = 501 frames x 128 rows x 128 channels = 501 such tiles, two per
    // CU): 3.41 vs 4.10 ms for 1024 tiles of 64x128 at bf16x3 and 512 frames, 1.14 vs 1.35 ms at f16 with 16-bit activations
    // (profiles/r03p_microbench_x3a.log, r03t_*)
    if (d->n_out > 64 && mt128 * nt128 * d->batch >= 256 && (long long)d->ksize * d->c_in >= 16384 && d->t_out >= 128) { t.wm = t.wn = 2; return SVCMI_OK; }
    // measured (profiles/r02a_microbench_lp.log, bf16x3): M = 500, N = 3840 / 5120 -> 240 / 320 tiles of 64x128 lose to 480 / 640 of
    // 64x64 (33.6 vs 31.7 us, 48.7 vs 40.0 us: one block per CU hides nothing); M = 750, N = 5120 -> 480 tiles win (49.9 vs 58.9 us)
    if (d->n_out > 64 && mt64 * nt128 * d->batch >= 448) { t.wn = 2; return SVCMI_OK; }
    return SVCMI_OK;
}

template <int PREC>
int dispatch(const ConvArgs& a, const TileChoice& t, int batch, int mode, void* stream) {
    if (t.p16) return t.wn == 3 ? launch<1, 3, true, PREC>(a, batch, mode, stream) : launch<1, 5, true, PREC>(a, batch, mode, stream);
    if (t.wm == 2) return launch<2, 2, false, PREC>(a, batch, mode, stream);
    return t.wn == 2 ? launch<1, 2, false, PREC>(a, batch, mode, stream) : launch<1, 1, false, PREC>(a, batch, mode, stream);
}

template <int PREC>
int dispatch_group(GroupArgs& g, const TileChoice& t, int count, int batch, int mode, void* stream) {
    if (t.p16) return t.wn == 3 ? launch_group<1, 3, true, PREC>(g, count, batch, mode, stream) : launch_group<1, 5, true, PREC>(g, count, batch, mode, stream);
    if (t.wm != 1 || t.wn != 1) return SVCMI_EUNSUPPORTED;
    return launch_group<1, 1, false, PREC>(g, count, batch, mode, stream);
}

}  // namespace

extern "C" int svcmi_pack_weights_lp(const float* w, int32_t n, int32_t ldw, int32_t precision, void* out, int32_t ldw16, void* stream) {
    if (!w || !out || n <= 0 || ldw <= 0) return SVCMI_EINVAL;
    if ((precision < SVCMI_PREC_BF16X3 || precision > SVCMI_PREC_BF16X3_A16) && precision != SVCMI_PREC_F16W2_A16) return SVCMI_EINVAL;
    if (ldw16 % 32 != 0 || ldw16 < ldw) return SVCMI_EINVAL;
    if (((uintptr_t)out & 15) != 0) return SVCMI_EALIGN;
    const long long total = (long long)n * ldw16;
    long long nb = (total + 255) / 256;
    if (nb > 65536) nb = 65536;
    SVCMI_LAUNCH(pack_lp_kernel, dim3((unsigned)nb), dim3(256), 0, stream, w, n, ldw, reinterpret_cast<unsigned short*>(out), ldw16, precision);
    return SVCMI_LAST_ERROR();
}

extern "C" int svcmi_conv_gemm_lp(const svcmi_conv_desc* d, int32_t precision, void* stream) {
    if ((precision < SVCMI_PREC_BF16X3 || precision > SVCMI_PREC_BF16X3_A16) && precision != SVCMI_PREC_F16W2_A16) return SVCMI_EINVAL;
    ConvArgs a;
    int mode;
    if (int rc = prepare(d, a, mode, precision)) return rc;
    if (mode == MODE_SCALAR) return SVCMI_EUNSUPPORTED;
    TileChoice t;
    if (int rc = choose_tile(d, mode, t)) return rc;
    const int bm = 64 * t.wm, bn = t.p16 ? 16 * t.wn : 64 * t.wn;
    const long long blocks = (long long)((d->t_out + bm - 1) / bm) * ((d->n_out + bn - 1) / bn) * d->batch;
    // Split-K as in the fp32 entry point: raw slabs for a consumer kernel (PARTIALS), or slices summed by the reduce kernel
    a.split = 1;
    const int kstep = precision >= SVCMI_PREC_BF16_A16 ? 2 * BK : BK;
    const int nk = (a.ktot + kstep - 1) / kstep;
    if (d->flags & SVCMI_CONV_PARTIALS) {
        if (!d->workspace || d->split_k < 1 || d->split_k > nk) return SVCMI_EINVAL;
        if ((long long)d->batch * d->split_k * d->t_out * d->n_out > d->workspace_floats) return SVCMI_EINVAL;
        a.split = d->split_k;
    } else if (t.wm == 1 && !t.p16 && d->workspace && d->split_k != 1 && !d->y16) {      // (the 16-bit output copy comes out of the non-split epilogue only)
        int s = d->split_k;
        if (s == 0) {       // a K-step is ~5x shorter than the fp32 kernel's: keep >= 16 of them per slice, aim at ~2 blocks per CU
            s = blocks >= 256 ? 1 : (int)(512 / blocks);
            if (s > nk / 16) s = nk / 16;
            if (s > 16) s = 16;
        }
        if (s > nk) s = nk;
        while (s > 1 && (long long)d->batch * s * d->t_out * d->n_out > d->workspace_floats) --s;
        if (s > 1) a.split = s;
    }
    switch (precision) {
        case SVCMI_PREC_BF16X3: return dispatch<PREC_BF16X3>(a, t, d->batch, mode, stream);
        case SVCMI_PREC_BF16: return dispatch<PREC_BF16>(a, t, d->batch, mode, stream);
        case SVCMI_PREC_BF16_A16: return dispatch<PREC_BF16_A16>(a, t, d->batch, mode, stream);
        case SVCMI_PREC_F16_A16: return dispatch<PREC_F16_A16>(a, t, d->batch, mode, stream);
        case SVCMI_PREC_BF16X3_A16: return dispatch<PREC_BF16X3_A16>(a, t, d->batch, mode, stream);
        case SVCMI_PREC_F16W2_A16: return dispatch<PREC_F16W2_A16>(a, t, d->batch, mode, stream);
        default: return dispatch<PREC_F16>(a, t, d->batch, mode, stream);
    }
}

extern "C" int svcmi_conv_gemm_group_lp(const svcmi_conv_desc* descs, int32_t count, int32_t precision, void* stream) {
    if ((precision < SVCMI_PREC_BF16X3 || precision > SVCMI_PREC_BF16X3_A16) && precision != SVCMI_PREC_F16W2_A16) return SVCMI_EINVAL;
    if (!descs || count < 1 || count > GROUP_MAX) return SVCMI_EINVAL;
    GroupArgs g;
    int order[GROUP_MAX] = {0, 1, 2};
    for (int i = 0; i < count; ++i)           // longest K first: the hardware starts blocks in grid order
        for (int j = i + 1; j < count; ++j)
            if (descs[order[j]].ksize > descs[order[i]].ksize) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    int mode = -1;
    const svcmi_conv_desc& d0 = descs[0];
    for (int i = 0; i < count; ++i) {
        const svcmi_conv_desc& d = descs[order[i]];
        int m;
        if (int rc = prepare(&d, g.p[i], m, precision)) return rc;
        if (mode >= 0 && m != mode) return SVCMI_EUNSUPPORTED;
        mode = m;
        if (d.batch != d0.batch || d.t_out != d0.t_out || d.n_out != d0.n_out || d.c_in != d0.c_in) return SVCMI_EINVAL;
        if ((d.flags & SVCMI_CONV_PARTIALS) || d.split_k > 1 || d.x_row_shift) return SVCMI_EUNSUPPORTED;
        g.p[i].split = 1;
    }
    for (int i = count; i < GROUP_MAX; ++i) g.p[i] = g.p[0];
    if (mode != MODE_CHUNK && mode != MODE_VEC) return SVCMI_EUNSUPPORTED;
    TileChoice t;
    const int n16 = (d0.n_out + 15) / 16;
    const int tile = d0.flags & SVCMI_CONV_TILE_MASK;
    if (tile == SVCMI_CONV_TILE_P16_64x48 || (!tile && n16 == 3)) t = TileChoice{1, 3, true};
    else if (tile == SVCMI_CONV_TILE_P16_64x80 || (!tile && n16 == 5)) t = TileChoice{1, 5, true};
    else if (!tile || tile == SVCMI_CONV_TILE_64x64) t = TileChoice{1, 1, false};
    else return SVCMI_EUNSUPPORTED;
    switch (precision) {
        case SVCMI_PREC_BF16X3: return dispatch_group<PREC_BF16X3>(g, t, count, d0.batch, mode, stream);
        case SVCMI_PREC_BF16: return dispatch_group<PREC_BF16>(g, t, count, d0.batch, mode, stream);
        case SVCMI_PREC_BF16_A16: return dispatch_group<PREC_BF16_A16>(g, t, count, d0.batch, mode, stream);
        case SVCMI_PREC_F16_A16: return dispatch_group<PREC_F16_A16>(g, t, count, d0.batch, mode, stream);
        case SVCMI_PREC_BF16X3_A16: return dispatch_group<PREC_BF16X3_A16>(g, t, count, d0.batch, mode, stream);
        case SVCMI_PREC_F16W2_A16: return dispatch_group<PREC_F16W2_A16>(g, t, count, d0.batch, mode, stream);
        default: return dispatch_group<PREC_F16>(g, t, count, d0.batch, mode, stream);
    }
}
