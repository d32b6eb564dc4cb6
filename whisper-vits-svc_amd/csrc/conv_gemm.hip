// conv_gemm.hip -- fp32 entry points of the implicit-GEMM convolution (kernel body: conv_gemm_body.h).
#include <string.h>

#include "conv_gemm_body.h"

extern "C" int svcmi_conv_gemm_f32(const svcmi_conv_desc* d, void* stream) {
    ConvArgs a;
    int mode;
    if (int rc = prepare(d, a, mode)) return rc;
    // Tile: fill the 256 CUs first, then grow the tile for operand reuse (or honour the override).
    const long long mt64 = (d->t_out + 63) / 64, nt64 = (d->n_out + 63) / 64;
    long long blocks64 = mt64 * nt64 * d->batch;
    int tile = (d->flags & SVCMI_CONV_TILE_MASK);
    // narrow outputs (the 40 / 80 / 160-channel generator stages): the 16x16x4 policy with N right-sized to 48 / 80 / 160
    int p16_nt = 0, p16_wm = 1;
    const int n16 = (d->n_out + 15) / 16;
    const bool p16_ok = mode == MODE_CHUNK || mode == MODE_VEC;
    if (tile > SVCMI_CONV_TILE_P16_64x160 && tile != SVCMI_CONV_TILE_P16W8_128x80) return SVCMI_EUNSUPPORTED;      // (64x128 exists on the reduced-precision kernels only)
    int p16_nw = 4;                                                // waves per block of the 16x16x4 policy
    if (tile >= SVCMI_CONV_TILE_P16_64x48) {                       // explicit override (tuning / tests)
        if (!p16_ok) return SVCMI_EUNSUPPORTED;
        p16_nt = tile == SVCMI_CONV_TILE_P16_64x160 ? 10 : (tile >= SVCMI_CONV_TILE_P16_64x80 ? 5 : 3);
        p16_wm = (tile == SVCMI_CONV_TILE_P16_128x48 || tile == SVCMI_CONV_TILE_P16_128x80) ? 2 : 1;
        if (tile == SVCMI_CONV_TILE_P16W8_128x80) p16_nw = 8;
    } else if (!tile && p16_ok && (n16 == 3 || n16 == 5) && d->t_out >= 1024) {
        // measured on MI355X (scripts/microbench.py p16): 64x80 beats 2 x (64x64) by 18 % at 80 channels, 64x48 beats 64x64
        // by 13 % at 40; the 128-row variants lose (half the blocks), and at 160 channels / 5000 rows 64x160 only ties
        p16_nt = n16;
    }
    if (p16_nt) {
        const int bm16 = 16 * p16_wm * p16_nw;
        blocks64 = (long long)((d->t_out + bm16 - 1) / bm16) * d->batch;
        tile = SVCMI_CONV_TILE_64x64;       // (only steers the split-K branch below)
    }
    if (!tile) {
        if (d->n_out > 64 && blocks64 >= 4 * 1024 && d->t_out >= 128) tile = SVCMI_CONV_TILE_128x128;
        else if (blocks64 >= 2 * 1024 && d->t_out >= 128) tile = SVCMI_CONV_TILE_128x64;
        else tile = SVCMI_CONV_TILE_64x64;
    }
    // Split-K: only for 64x64 tiles that leave CUs idle; slices keep >= 4 K-steps each.
    a.split = 1;
    const int nk = (a.ktot + BK - 1) / BK;
    if (d->flags & SVCMI_CONV_PARTIALS) {       // caller consumes the slabs itself: exactly split_k of them, no epilogue
        if (!d->workspace || d->split_k < 1 || d->split_k > nk) return SVCMI_EINVAL;
        if ((long long)d->batch * d->split_k * d->t_out * d->n_out > d->workspace_floats) return SVCMI_EINVAL;
        a.split = d->split_k;
    } else if (tile == SVCMI_CONV_TILE_64x64 && d->workspace && d->split_k != 1 && !d->y16) {      // (the 16-bit output copy comes out of the non-split epilogue only)
        int s = d->split_k;
        if (s == 0) {   // heuristic fitted to sweeps on MI355X (scripts/microbench.py gemm / small): aim at ~5 blocks per CU, keep
                        // >= 10 K-steps per slice (shorter slices are all prologue), and leave grids of >= 1.5 blocks per CU alone --
                        // there the reduce pass costs more than the idle CUs
            s = blocks64 >= 384 ? 1 : (int)(1280 / blocks64);
            if (s > nk / 10) s = nk / 10;
            if (blocks64 >= 200 && nk < 32) s = 1;
            if (s > 16) s = 16;
        }
        if (s > nk) s = nk;
        while (s > 1 && (long long)d->batch * s * d->t_out * d->n_out > d->workspace_floats) --s;
        if (s > 1) a.split = s;
    }

    if (a.split > 1 && d->counters && !(d->flags & SVCMI_CONV_PARTIALS)) {       // in-launch combine needs one zeroed counter per output tile
        const int bm = p16_nt ? 16 * p16_wm * p16_nw : (tile == SVCMI_CONV_TILE_64x64 ? 64 : 128);
        const int bn = p16_nt ? 16 * p16_nt : (tile == SVCMI_CONV_TILE_128x128 ? 128 : 64);
        const long long tiles = (long long)d->batch * ((d->t_out + bm - 1) / bm) * ((d->n_out + bn - 1) / bn);
        if (tiles <= d->counters_len && d->n_out % 4 == 0 && (((uintptr_t)d->workspace & 15) == 0) &&
            (long long)d->t_out * d->n_out < (1LL << 29)) a.cnt = d->counters;      // 16-byte write-through slab stores
    }
    if (p16_nt == 3) return p16_wm == 2 ? launch<2, 3, true>(a, d->batch, mode, stream) : launch<1, 3, true>(a, d->batch, mode, stream);
    if (p16_nt == 5 && p16_nw == 8) return launch<1, 5, true, PREC_F32, 8>(a, d->batch, mode, stream);
    if (p16_nt == 5) return p16_wm == 2 ? launch<2, 5, true>(a, d->batch, mode, stream) : launch<1, 5, true>(a, d->batch, mode, stream);
    if (p16_nt == 10) return launch<1, 10, true>(a, d->batch, mode, stream);
    switch (tile) {
        case SVCMI_CONV_TILE_128x128: return launch<2, 2, false>(a, d->batch, mode, stream);
        case SVCMI_CONV_TILE_128x64: return launch<2, 1, false>(a, d->batch, mode, stream);
        default: return launch<1, 1, false>(a, d->batch, mode, stream);
    }
}

extern "C" int svcmi_conv_gemm_group_f32(const svcmi_conv_desc* descs, int32_t count, void* stream) {
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
        if (int rc = prepare(&d, g.p[i], m)) return rc;
        if (mode >= 0 && m != mode) return SVCMI_EUNSUPPORTED;
        mode = m;
        // one tile policy for the whole grid: same geometry, no split-K / raw partials (the group itself supplies the blocks)
        if (d.batch != d0.batch || d.t_out != d0.t_out || d.n_out != d0.n_out || d.c_in != d0.c_in) return SVCMI_EINVAL;
        if ((d.flags & SVCMI_CONV_PARTIALS) || d.split_k > 1 || d.x_row_shift) return SVCMI_EUNSUPPORTED;
        g.p[i].split = 1;
    }
    for (int i = count; i < GROUP_MAX; ++i) g.p[i] = g.p[0];
    if (mode != MODE_CHUNK && mode != MODE_VEC) return SVCMI_EUNSUPPORTED;
    const int tile = d0.flags & SVCMI_CONV_TILE_MASK;
    const int n16 = (d0.n_out + 15) / 16;
    // 16x16x4 tiles right-sized to the channel count (40 -> 48, 80 -> 80 columns); everything else 64 x 64
    if (tile == SVCMI_CONV_TILE_P16_128x48) return launch_group<2, 3, true>(g, count, d0.batch, mode, stream);
    if (tile == SVCMI_CONV_TILE_P16_128x80) return launch_group<2, 5, true>(g, count, d0.batch, mode, stream);
    if (tile == SVCMI_CONV_TILE_P16_64x48 || (!tile && n16 == 3)) return launch_group<1, 3, true>(g, count, d0.batch, mode, stream);
    // (160 channels: two 64x80 tiles lose to three 64x64 ones, 84 vs 75.5 us per grouped launch)
    if (tile == SVCMI_CONV_TILE_P16_64x80 || (!tile && n16 == 5)) return launch_group<1, 5, true>(g, count, d0.batch, mode, stream);
    if (tile && tile != SVCMI_CONV_TILE_64x64) return SVCMI_EUNSUPPORTED;
    return launch_group<1, 1, false>(g, count, d0.batch, mode, stream);
}

// Development knob of the grouped launches (reached through svcmi_tune_set): results never depend on it.
extern "C" int svcmi_conv_tune_set(const char* name, int32_t value) {
    const char* k = "group_nst";
    int i = 0;
    while (k[i] && name[i] == k[i]) ++i;
    if (k[i] == 0 && name[i] == 0 && (value == 0 || value == 2 || value == 3)) { g_group_nst = value; return 0; }
    return SVCMI_EINVAL;
}

extern "C" int svcmi_conv_tune_get(const char* name, int32_t* value) {
    if (strcmp(name, "group_nst") == 0) { *value = g_group_nst; return 0; }
    if (strcmp(name, "last_conv_ring") == 0) { *value = g_last_ring; return 0; }
    return SVCMI_EINVAL;
}

#if SVCMI_PROBE_KTRACE
// timing-probe builds only (conv_gemm_body.h, SVCMI_PROBE_KTRACE): the per-wave K-loop cycle sums of the LAST fp32 launch
extern "C" int svcmi_probe_ktrace_read(unsigned long long* dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ktrace), (size_t)n * 8, 0, hipMemcpyDeviceToHost);
}
// ... and the in-flight timeline: reset the record counter / read the first `cap` records (5 x u64 each); returns the number recorded
extern "C" int svcmi_probe_timeline_reset(void) {
    const unsigned zero = 0;
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tl_n), &zero, 4, 0, hipMemcpyHostToDevice);
}
extern "C" long long svcmi_probe_timeline_read(unsigned long long* dst, long long cap) {
    unsigned n = 0;
    if (hipMemcpyFromSymbol(&n, HIP_SYMBOL(g_tl_n), 4, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    long long m = n < TL_CAP ? n : TL_CAP;
    if (m > cap) m = cap;
    if (m > 0 && hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tl), (size_t)m * 40, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return m;
}
#endif
