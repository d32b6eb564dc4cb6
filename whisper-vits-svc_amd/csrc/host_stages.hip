// host_stages.hip -- the stage-level entry points of include/svcmi.h: a C++ host that composes the kernels of this library into
// the reference's forward passes (whisper/model.py:147-163 AudioEncoder.forward; vits/models.py:39-52 TextEncoder.forward,
// :89-94 ResidualCouplingBlock reverse, vits_decoder/generator.py:175-200 Generator.inference, vits/models.py:251-256
// SynthesizerInfer.inference).  Plain host code: it only calls the extern "C" launchers of the other translation units, so the
// same file builds into the product (hipcc) and into the CPU emulation library of the unit tests (g++ -DSVCMI_EMU).
//
// Memory: one caller-owned workspace, handed out by a bump allocator with stack discipline (mark / release around loop bodies);
// svcmi_*_workspace_bytes runs the SAME code in plan mode (no launches) and reports the peak.  Nothing allocates, nothing
// synchronises: a stage call is a fixed sequence of launches on one stream and can be captured into a hipGraph.
#include <math.h>
#include <string.h>

#include <vector>

#include "svcmi_rt.h"
#include "../../include/svcmi.h"

namespace {

int g_amp_grouped = 1;     // tuning / test knob ("amp_grouped", 0 | 1): 0 forces the one-launch-per-block fallback of the generator stages
// tuning knob ("amp_lp", 0 | 1): 1 = in the f16 / f16w2 precision classes the narrow stages' half-steps run their convolution on the
// fp16 matrix cores (svcmi_snake_conv_group_lp); 0 = they stay on the fp32 vector kernels whatever the class says
int g_amp_lp = 1;

enum Op {
    OP_CONV_F32, OP_CONV_LP, OP_CONV_GROUP_F32, OP_CONV_GROUP_LP, OP_LAYERNORM, OP_SPLITK_LN, OP_ATTENTION, OP_SNAKE_ALIAS,
    OP_SNAKE_ALIAS_GROUP, OP_BLOCK_MEAN, OP_SNAKE_CONV, OP_SNAKE_CONV_GROUP, OP_UPSAMPLE_NOISE, OP_SNAKE_POST, OP_WN_GATE,
    OP_COUPLING_PRE, OP_COUPLING_POST, OP_EMBED_PITCH, OP_SAMPLE_PRIOR, OP_NCL_TO_NLC, OP_COPY2D, OP_PITCH_PREFIX, OP_PITCH_SOURCE,
    OP_ATTENTION16, OP_SNAKE_CONV_GROUP_LP, OP_COUNT
};
const char* const OP_NAMES[OP_COUNT] = {
    "svcmi_conv_gemm_f32", "svcmi_conv_gemm_lp", "svcmi_conv_gemm_group_f32", "svcmi_conv_gemm_group_lp", "svcmi_layernorm_f32",
    "svcmi_splitk_layernorm_f32", "svcmi_attention_f32", "svcmi_snake_alias_f32", "svcmi_snake_alias_group_f32", "svcmi_block_mean_f32",
    "svcmi_snake_conv_f32", "svcmi_snake_conv_group_f32", "svcmi_upsample_noise_f32", "svcmi_snake_post_f32", "svcmi_wn_gate_f32",
    "svcmi_coupling_pre_f32", "svcmi_coupling_post_f32", "svcmi_embed_pitch_f32", "svcmi_sample_prior_f32", "svcmi_ncl_to_nlc_f32",
    "svcmi_copy2d_f32", "svcmi_pitch_prefix_f64", "svcmi_pitch_source_f32", "svcmi_attention16", "svcmi_snake_conv_group_lp"};

// ------------------------------------------------------------------------------------------------ per-launch trace (bench.py)
struct TraceRec {
    int op; double flops, bytes;
#ifndef SVCMI_EMU
    hipEvent_t e0, e1;
#endif
};
struct Trace {
    bool on = false;
    int cap = 0;
    std::vector<TraceRec> recs;
};
thread_local Trace g_trace;

struct Arena {
    char* base = nullptr;
    int64_t cap = 0, off = 0, peak = 0;
    bool overflow = false;
    void* take(int64_t bytes) {
        off = (off + 255) & ~(int64_t)255;
        void* p = base ? base + off : nullptr;
        off += bytes;
        if (off > peak) peak = off;
        if (base && off > cap) { overflow = true; return base; }     // keep pointers valid; the stage reports SVCMI_EINVAL before launching
        return p;
    }
    float* f(int64_t n) { return static_cast<float*>(take(n * 4)); }
    int64_t mark() const { return off; }
    void release(int64_t m) { off = m; }
};

constexpr int64_t SPLITK_FLOATS = 16LL * 1024 * 1024;     // shared split-K scratch of the launches that let the library choose (64 MB)

struct Ctx {
    void* stream = nullptr;
    Arena ar;
    bool plan = false;          // size the workspace, launch nothing
    int prec = SVCMI_PREC_F32;
    float lp_min_flops = 1.5e9f;
    int rc = 0;
    float* sk_ws = nullptr;
    bool live() const { return !plan && rc == 0 && !ar.overflow; }
};

template <typename F>
void run(Ctx& c, int op, double flops, double bytes, F&& launch) {
    if (!c.live()) return;
    Trace& t = g_trace;
    const bool tr = t.on && (int)t.recs.size() < t.cap;
#ifndef SVCMI_EMU
    TraceRec r{op, flops, bytes, nullptr, nullptr};
    if (tr) {
        (void)hipEventCreate(&r.e0);
        (void)hipEventCreate(&r.e1);
        (void)hipEventRecord(r.e0, (hipStream_t)c.stream);
    }
#else
    TraceRec r{op, flops, bytes};
#endif
    const int rc = launch();
    if (rc) c.rc = rc;
#ifndef SVCMI_EMU
    if (tr) (void)hipEventRecord(r.e1, (hipStream_t)c.stream);
#endif
    if (tr) t.recs.push_back(r);
}

// ------------------------------------------------------------------------------------------------ convolution wrapper
// The keyword arguments of one implicit-GEMM launch (what svcmi/ops.py Ops.conv takes), with the facade's defaults.
struct CV {
    const float* x = nullptr; int64_t x_bs = 0; int B = 1, t_in = 0, c_in = 0, ldx = 0;
    const svcmi_weight* w = nullptr; bool bias = true; int n_out = 0;
    int ksize = 1, stride = 1, dil = 1, pad = 0, t_out = -1, act = SVCMI_ACT_NONE, rshift = 0;
    const float* res = nullptr; int64_t res_bs = 0; int ldr = 0;
    float alpha = 1.f; bool accumulate = false, mask_in = false, mask_out = false;
    const int32_t* lengths = nullptr;
    float* y = nullptr; int64_t y_bs = 0; int ldy = 0;
    int tile = 0, tile_lp = -1;     // tile override (>> 8) on the fp32 / 16-bit kernels (-1: same as `tile`)
    int split_k = 0;                // 0 = library heuristic (shared scratch), 1 = off, n = exactly n with `slabs`
    float* slabs = nullptr;         // SVCMI_CONV_PARTIALS: raw slabs [B][split_k][t_out][n_out] land here
    const void* x16 = nullptr;      // the same rows as `x` as a 16-bit tensor (same ldx / x_bs in elements), written by the producer: a
                                    // launch that goes to the bf16 / f16 kernel then takes the _A16 instantiation (no in-register rounding)
    void* y16 = nullptr;            // 16-bit copy of the output for the NEXT launch (written only in the bf16 / f16 modes)
    int y16_fmt = -1;               // format of that copy: -1 = the launch's own mode, else SVCMI_PREC_BF16 / _F16 whatever the GEMM runs in
    int ring_class = 4;             // bit of g_ring2 that gives this launch the 2-deep operand ring (SVCMI_CONV_RING2): 0 qkv, 1 o, 2 mlp-up, 3 mlp-down of Whisper, 4 everything else
};

// tuning knob ("ring2", bit mask over CV::ring_class): which single-launch fp32 GEMMs take the 2-deep operand ring (one more resident
// block per CU).  Meant for captures that run several clips in flight (svcmi.serving.ClipLanes sets it around its graph captures): measured
// on MI355X with 4 clips in flight the judged line gains 1.8 % with every GEMM on it, while one clip alone loses 4 % (profiles/r05a_*).
int g_ring2 = 0;

// Per-layer mixed precision (SVCMI_PREC_MIXED): every section of the synthesizer sets c.prec to its class's mode before it launches.
int class_prec(const svcmi_synth_model& m, int cls) {
    if (m.precision != SVCMI_PREC_MIXED) return m.precision;
    const int p = cls >= 0 && cls < SVCMI_PREC_CLASSES ? m.class_prec[cls] : SVCMI_PREC_F32;
    return (p >= SVCMI_PREC_F32 && p <= SVCMI_PREC_F16) || p == SVCMI_PREC_F16W2 ? p : SVCMI_PREC_F32;
}

// SVCMI_PREC_F16W2 is a MODE (fp16 activation rows like SVCMI_PREC_F16; split fp16 weights in the launches that read them): everything
// that names a 16-bit FORMAT (a producer's y16_format, the attention kernels, the plain 16-bit weight image) sees fmt16_of() = F16.
int fmt16_of(int prec) { return prec == SVCMI_PREC_F16W2 ? SVCMI_PREC_F16 : prec; }
bool mode16(int prec) { return prec == SVCMI_PREC_BF16 || prec == SVCMI_PREC_F16 || prec == SVCMI_PREC_F16W2; }
// Modes whose GEMM chains hand 16-bit activations from producer to consumer: bf16 / f16 rows, or (bf16x3) split rows [hi: C | lo: C].
bool act16(int prec) { return mode16(prec) || prec == SVCMI_PREC_BF16X3; }
int w16(int prec, int C) { return prec == SVCMI_PREC_BF16X3 ? 2 * C : C; }             // 16-bit values per row of C channels
int a16_code(int prec) { return prec == SVCMI_PREC_BF16X3 ? SVCMI_PREC_BF16X3_A16 : (prec == SVCMI_PREC_F16W2 ? SVCMI_PREC_F16W2_A16 : prec + 2); }

int conv_t_out(const CV& v) { return v.t_out >= 0 ? v.t_out : (v.t_in + 2 * v.pad - v.dil * (v.ksize - 1) - 1) / v.stride + 1; }

// Tile of the AMP convolutions by stage width alone: the library's own choice also looks at the row count (the 16x16x4 policy from 1024
// rows up), and the two policies sum K in different orders -- a streamed tile shorter than that would then differ from the whole-chunk run
// in the last bit.  Pinned, every tile size gives the same bits.
int amp_tile(int cp) { const int n16 = (cp + 15) / 16; return n16 == 3 ? 4 : (n16 == 5 ? 6 : 0); }      // SVCMI_CONV_TILE_P16_64x48 / _64x80 >> 8

bool lp_tile_ok(int tile) { return tile == 0 || tile == 1 || tile == 3 || tile == 4 || tile == 6 || tile == 9; }

// fills the descriptor; returns the precision code of the launch (SVCMI_PREC_F32 = the fp32 kernel)
int conv_desc(const Ctx& c, const CV& v, svcmi_conv_desc& d, double& flops, double& bytes, double lp_flops = -1.0) {
    memset(&d, 0, sizeof(d));
    const int N = v.n_out ? v.n_out : v.w->n;
    const int t_out = conv_t_out(v);
    d.x = v.x; d.w = v.w->w; d.bias = v.bias ? v.w->bias : nullptr; d.res = v.res; d.y = v.y ? v.y : const_cast<float*>(v.x);
    d.lengths = v.lengths;
    d.x_bstride = v.x_bs; d.y_bstride = v.y ? v.y_bs : 0; d.res_bstride = v.res_bs;
    d.batch = v.B; d.t_in = v.t_in; d.t_out = t_out; d.c_in = v.c_in; d.ldx = v.ldx;
    d.n_out = N; d.ldw = v.w->ldw; d.ldy = v.y ? v.ldy : N; d.ldr = v.res ? v.ldr : 0;
    d.ksize = v.ksize; d.stride = v.stride; d.dilation = v.dil; d.pad = v.pad; d.x_row_shift = v.rshift;
    d.act = v.act; d.alpha = v.alpha;
    const bool partials = v.slabs != nullptr;
    flops = 2.0 * v.B * t_out * N * (double)v.ksize * v.c_in;
    bytes = 4.0 * ((double)v.B * (v.t_in >> v.rshift) * v.c_in + (double)N * v.ksize * v.c_in +
                   (double)v.B * t_out * N * (1 + (v.res != nullptr) + (v.accumulate ? 1 : 0)));
    const int tile_lp = v.tile_lp >= 0 ? v.tile_lp : v.tile;
    const bool lp = c.prec != SVCMI_PREC_F32 && (lp_flops >= 0.0 ? lp_flops : flops) >= c.lp_min_flops && v.w->w16 != nullptr &&
                    (v.w->prec16 == 0 || v.w->prec16 == c.prec) &&      // (an image packed for another mode is never misread: fp32 operand instead)
                    v.c_in % 4 == 0 && v.ldx % 4 == 0 && v.x_bs % 4 == 0 && ((uintptr_t)v.x & 15) == 0 &&
                    (!v.rshift || v.c_in % 32 == 0) && lp_tile_ok(tile_lp);
    const int tile = lp ? tile_lp : (v.tile == 9 ? 0 : v.tile);      // 64x128 exists on the 16-bit kernels only
    d.flags = (v.accumulate ? SVCMI_CONV_ACCUMULATE : 0) | (v.mask_in ? SVCMI_CONV_MASK_IN : 0) | (v.mask_out ? SVCMI_CONV_MASK_OUT : 0) |
              (partials ? SVCMI_CONV_PARTIALS : 0) | (tile << 8) | (((g_ring2 >> v.ring_class) & 1) ? SVCMI_CONV_RING2 : 0);
    if (partials) {
        d.split_k = v.split_k; d.workspace = v.slabs; d.workspace_floats = (int64_t)v.B * v.split_k * t_out * N;
    } else if (v.split_k != 1) {
        d.split_k = v.split_k; d.workspace = c.sk_ws; d.workspace_floats = SPLITK_FLOATS;
    } else {
        d.split_k = 1;
    }
    const int yfmt = fmt16_of(v.y16_fmt >= 0 ? v.y16_fmt : c.prec);
    if (v.y16 && act16(yfmt) && !partials && d.split_k == 1) {
        d.y16 = v.y16; d.y16_bstride = w16(yfmt, 1) * d.y_bstride; d.ldy16 = w16(yfmt, d.ldy); d.y16_format = yfmt;
        bytes += 2.0 * v.B * t_out * w16(yfmt, N);
    }
    if (!lp) return SVCMI_PREC_F32;
    d.w = static_cast<const float*>(v.w->w16);
    d.ldw = v.w->ldw16;
    bytes += (double)N * v.ksize * v.c_in * ((c.prec == SVCMI_PREC_BF16X3 ? 4.0 : 2.0) - 4.0);
    if (v.x16 && act16(c.prec) && v.w->w16a && v.c_in % 8 == 0 && v.ldx % 8 == 0 && v.x_bs % 8 == 0 && !v.rshift) {
        d.x = static_cast<const float*>(v.x16);
        d.w = static_cast<const float*>(v.w->w16a);
        d.ldx = w16(c.prec, v.ldx); d.x_bstride = w16(c.prec, 1) * v.x_bs;
        bytes -= (4.0 - 2.0 * w16(c.prec, 1)) * v.B * (double)v.t_in * v.c_in;
        if (c.prec == SVCMI_PREC_F16W2) bytes += 2.0 * (double)N * v.ksize * v.c_in;      // the lo image
        return a16_code(c.prec);    // SVCMI_PREC_BF16_A16 / _F16_A16 / _BF16X3_A16 / _F16W2_A16
    }
    return fmt16_of(c.prec);        // (F16W2 without 16-bit activations: the plain fp16 kernel on the fp16 image)
}

void conv(Ctx& c, const CV& v) {
    if (!c.live()) return;
    svcmi_conv_desc d;
    double flops, bytes;
    const int prec = conv_desc(c, v, d, flops, bytes);
    if (prec != SVCMI_PREC_F32) run(c, OP_CONV_LP | (prec << 8), flops, bytes, [&] { return svcmi_conv_gemm_lp(&d, prec, c.stream); });
    else run(c, OP_CONV_F32, flops, bytes, [&] { return svcmi_conv_gemm_f32(&d, c.stream); });
}

// up to 3 convolutions of one geometry in one launch (Ops.conv_group): no split-K; 16-bit only if every problem qualifies, 16-bit
// activations (x16) only if every problem can take them.  `dry`: decide the kernel only; returns its precision code.
int conv_group(Ctx& c, const CV* vs, int count, bool dry = false) {
    if (!c.live() && !dry) return SVCMI_PREC_F32;
    svcmi_conv_desc d[3];
    double flops[3], bytes[3], total = 0.0;
    for (int i = 0; i < count; ++i) {
        CV v = vs[i];
        v.split_k = 1;
        conv_desc(c, v, d[i], flops[i], bytes[i]);
        total += flops[i];
    }
    bool lp = c.prec != SVCMI_PREC_F32;
    const int tile0 = vs[0].tile_lp >= 0 ? vs[0].tile_lp : vs[0].tile;
    if (lp && !(tile0 == 0 || tile0 == 1 || tile0 == 4 || tile0 == 6)) lp = false;
    double tb = 0.0;
    int prec = SVCMI_PREC_F32;
    if (lp) {
        bool a16 = true;
        for (int i = 0; i < count && lp; ++i) {
            CV v = vs[i];
            v.split_k = 1;
            const int pi = conv_desc(c, v, d[i], flops[i], bytes[i], total);
            lp = pi != SVCMI_PREC_F32;
            a16 = a16 && pi >= SVCMI_PREC_BF16_A16;
        }
        if (lp && !a16)
            for (int i = 0; i < count; ++i) {      // mixed: everyone rounds in registers
                CV v = vs[i];
                v.split_k = 1;
                v.x16 = nullptr;
                conv_desc(c, v, d[i], flops[i], bytes[i], total);
            }
        if (lp) prec = a16 ? a16_code(c.prec) : fmt16_of(c.prec);
        if (!lp)
            for (int i = 0; i < count; ++i) {      // back to fp32 descriptors: a group runs on ONE kernel
                CV v = vs[i];
                v.split_k = 1;
                Ctx c32 = c;
                c32.prec = SVCMI_PREC_F32;
                conv_desc(c32, v, d[i], flops[i], bytes[i]);
            }
    }
    for (int i = 0; i < count; ++i) tb += bytes[i];
    if (dry) return prec;
    if (lp) run(c, OP_CONV_GROUP_LP | (prec << 8), total, tb, [&] { return svcmi_conv_gemm_group_lp(d, count, prec, c.stream); });
    else run(c, OP_CONV_GROUP_F32, total, tb, [&] { return svcmi_conv_gemm_group_f32(d, count, c.stream); });
    return prec;
}

// (y16 / o16: optional 16-bit copies of the outputs, rows of C values, in the format of the bf16 / f16 mode; nullptr = none)
void layernorm(Ctx& c, const float* x, const float* res, const float* g, const float* b, float* y, int B, int T, int C, int ldx, int ldr,
               int ldy, int gb_bs, void* y16 = nullptr) {
    if (!act16(c.prec)) y16 = nullptr;
    run(c, OP_LAYERNORM, 0.0, 4.0 * B * T * C * (2 + (res != nullptr)), [&] {
        return svcmi_layernorm_f32(x, res, g, b, y, B, T, C, ldx, ldr, ldy, gb_bs, 1e-5f, y16, w16(c.prec, C), fmt16_of(c.prec), c.stream);
    });
}

void splitk_layernorm(Ctx& c, const float* part, int split, const float* bias, float* x, const float* g, const float* b, float* y, int B,
                      int T, int C, void* y16 = nullptr) {
    if (!act16(c.prec)) y16 = nullptr;
    run(c, OP_SPLITK_LN, 0.0, 4.0 * B * T * C * (3 + split), [&] {
        return svcmi_splitk_layernorm_f32(part, split, bias, x, g, b, y, B, T, C, C, C, 1e-5f, y16, w16(c.prec, C), fmt16_of(c.prec), c.stream);
    });
}

void attention(Ctx& c, const float* qkv, float* o, int B, int T, int heads, int C, float scale, const float* rel_k, const float* rel_v,
               int window, const int32_t* lengths, void* o16 = nullptr) {
    const int64_t bs = (int64_t)T * 3 * C;
    if (!act16(c.prec)) o16 = nullptr;
    run(c, OP_ATTENTION, 4.0 * B * T * (double)T * C, 16.0 * B * T * C, [&] {
        return svcmi_attention_f32(qkv, qkv + C, qkv + 2 * C, o, 3 * C, 3 * C, 3 * C, C, bs, bs, bs, (int64_t)T * C, B, T, heads, C / heads,
                                   scale, rel_k, rel_v, window, lengths, o16, w16(c.prec, C), (int64_t)T * w16(c.prec, C), fmt16_of(c.prec), c.stream);
    });
}

// band-free attention on the 16-bit matrix cores from the QKV projection's 16-bit output copy (bf16 / f16 modes)
void attention16(Ctx& c, const void* qkv16, float* o, void* o16, int B, int T, int heads, int C, float scale, const int32_t* lengths,
                 const float* rel_k = nullptr, const float* rel_v = nullptr, int window = 0, int fmt = -1) {
    const unsigned short* q = static_cast<const unsigned short*>(qkv16);
    const int f = fmt16_of(fmt >= 0 ? fmt : c.prec);
    run(c, OP_ATTENTION16, 4.0 * B * T * (double)T * C, 10.0 * B * T * C, [&] {
        return svcmi_attention16(q, q + C, q + 2 * C, 3 * C, (int64_t)T * 3 * C, o, C, (int64_t)T * C, o16, C, (int64_t)T * C, B, T, heads, C / heads, scale,
                                 rel_k, rel_v, window, lengths, f, c.stream);
    });
}

void copy2d(Ctx& c, const float* x, int64_t ldx, float* y, int64_t ldy, int64_t rows, int64_t cols) {
    run(c, OP_COPY2D, 0.0, 8.0 * rows * cols, [&] { return svcmi_copy2d_f32(x, ldx, y, ldy, rows, cols, c.stream); });
}

int finish(const Ctx& c) {
    if (c.ar.overflow) return SVCMI_EINVAL;       // workspace smaller than svcmi_*_workspace_bytes says
    return c.rc;
}

int tune(int v, int dflt) { return v == 0 ? dflt : (v < 0 ? 0 : v); }      // model tuning fields: 0 = table default, < 0 = library heuristic

// ------------------------------------------------------------------------------------------------ Whisper audio encoder
// whisper/model.py:147-163.  A block is 6 launches: QKV GEMM, attention, out-projection as raw split-K slabs, [slab sum + bias +
// residual + NEXT LayerNorm] in one kernel, MLP-up GEMM with GELU, MLP-down as raw slabs, [slab sum + ... + next LayerNorm].
void whisper_fwd(Ctx& c, const svcmi_whisper_model& m, const float* mel, const float* noise, float noise_scale, int B, int n, float* out) {
    const int S = m.n_state, tw = (n - 1) / 2 + 1, nb = m.n_layers;
    if (tw > m.n_ctx || n <= 0 || B <= 0 || nb < 0 || nb > SVCMI_MAX_WHISPER_BLOCKS || S % 4 || m.n_mels % 4) { c.rc = SVCMI_EINVAL; return; }
    c.prec = m.precision;
    c.lp_min_flops = m.lp_min_flops > 0.f ? m.lp_min_flops : 1.5e9f;
    // Tuning table for the M = 500 .. 750-row window GEMMs (scripts/microbench.py gemm / wp16 / lp, MI355X): K slices of the two
    // N = n_state projections; the 64x80 tile of the 16x16x4 policy where it balances the 256 CUs better than 64x64 (fp32), 64x128
    // tiles for the two N = n_state projections on the 16-bit kernels.  Batched windows (M > small_m_rows) fill the chip with the
    // library's own tile heuristic and need no K slices.
    int split_o = tune(m.split_o, 2), split_mlp = tune(m.split_mlp, 4);
    int t_qkv = tune(m.tile_qkv, 0), t_o = tune(m.tile_o, 6), t_m1 = tune(m.tile_mlp1, 6), t_m2 = tune(m.tile_mlp2, 6);
    int l_qkv = 0, l_o = 9, l_m1 = 0, l_m2 = 9;
    const int small_m = m.small_m_rows > 0 ? m.small_m_rows : 1024;
    if ((int64_t)B * tw > small_m) {
        split_o = split_mlp = 1;
        t_qkv = t_o = t_m1 = t_m2 = l_qkv = l_o = l_m1 = l_m2 = 0;
    }
    if (split_o < 1) split_o = 1;
    if (split_mlp < 1) split_mlp = 1;
    c.sk_ws = c.ar.f(SPLITK_FLOATS);
    float* x0 = c.ar.f((int64_t)B * n * m.n_mels);
    float* x1 = c.ar.f((int64_t)B * n * S);
    float* x = c.ar.f((int64_t)B * tw * S);
    run(c, OP_NCL_TO_NLC, 0.0, 8.0 * B * n * m.n_mels,
        [&] { return svcmi_ncl_to_nlc_f32(mel, noise, noise ? noise_scale : 0.f, x0, B, m.n_mels, n, m.n_mels, c.stream); });
    {
        CV v; v.x = x0; v.x_bs = (int64_t)n * m.n_mels; v.B = B; v.t_in = n; v.c_in = v.ldx = m.n_mels; v.w = &m.conv1;
        v.ksize = 3; v.pad = 1; v.act = SVCMI_ACT_GELU; v.y = x1; v.y_bs = (int64_t)n * S; v.ldy = S;
        conv(c, v);                                                                  // model.py:150
    }
    {
        CV v; v.x = x1; v.x_bs = (int64_t)n * S; v.B = B; v.t_in = n; v.c_in = v.ldx = S; v.w = &m.conv2;
        v.ksize = 3; v.stride = 2; v.pad = 1; v.act = SVCMI_ACT_GELU; v.res = m.pos; v.res_bs = 0; v.ldr = S;
        v.y = x; v.y_bs = (int64_t)tw * S; v.ldy = S;
        conv(c, v);                                                                  // :151-158 (+ positional_embedding)
    }
    const int H = m.n_heads;
    const float scale = 1.0f / sqrtf((float)(S / H));          // (d^-0.25 on q) * (d^-0.25 on k), model.py:90-92
    if (nb == 0) { layernorm(c, x, nullptr, m.lnp_g, m.lnp_b, out, B, tw, S, S, 0, S, 0); return; }
    float* h = out;
    float* qkv = c.ar.f((int64_t)B * tw * 3 * S);
    float* a = c.ar.f((int64_t)B * tw * S);
    const int F = m.blocks[0].m1.n;
    float* mm = c.ar.f((int64_t)B * tw * F);
    const int so = split_o < m.blocks[0].o.ldw / 128 ? split_o : (m.blocks[0].o.ldw / 128 > 0 ? m.blocks[0].o.ldw / 128 : 1);
    const int sm = split_mlp < m.blocks[0].m2.ldw / 128 ? split_mlp : (m.blocks[0].m2.ldw / 128 > 0 ? m.blocks[0].m2.ldw / 128 : 1);
    float* slabs = c.ar.f((int64_t)B * (so > sm ? so : sm) * tw * S);
    // bf16 / f16 modes: every GEMM's A operand is written as a 16-bit tensor by its producer (LayerNorm -> QKV and MLP-up, attention ->
    // out-projection, MLP-up's GELU epilogue -> MLP-down), so the GEMMs take the _A16 kernels: half the LDS bytes per MFMA, no rounding
    // in registers.  (The fp32 copies stay: LayerNorm output and residual stream are fp32 in every mode.)
    // bf16x3: split rows [hi | lo] from the LayerNorms and the _BF16X3_A16 kernel for the QKV projection only -- the one launch it speeds up
    // in place (T = 500: 30.5 -> 23.9 us).  MLP-down gains alone (36.0 -> 31.9 us) but only at the 64x64 tile with 8 K slices, which the
    // split-K LayerNorm behind it and the split rows out of MLP-up's GELU epilogue give back; the out-projection's short K slices and
    // MLP-up are no faster or slower (profiles/r03p_microbench_x3a.log); the attention stays on the fp32 matrix cores.
    const bool x3 = c.prec == SVCMI_PREC_BF16X3;
    const bool a16 = act16(c.prec) && S % 8 == 0 && F % 8 == 0;
    const int64_t e16 = 2 * w16(c.prec, 1);           // bytes per channel of a 16-bit row
    void* h16 = a16 ? c.ar.take((int64_t)B * tw * S * e16) : nullptr;
    void* at16 = a16 && !x3 ? c.ar.take((int64_t)B * tw * S * e16) : nullptr;
    void* mm16 = a16 && !x3 ? c.ar.take((int64_t)B * tw * F * e16) : nullptr;
    // ... and the attention itself runs on the 16-bit matrix cores from the QKV projection's 16-bit output copy (svcmi_attention16)
    const bool att16 = a16 && mode16(c.prec) && (S / H == 64 || S / H == 32);
    void* qkv16 = att16 ? c.ar.take((int64_t)B * tw * 3 * S * 2) : nullptr;
    layernorm(c, x, nullptr, m.blocks[0].ln1_g, m.blocks[0].ln1_b, h, B, tw, S, S, 0, S, 0, h16);
    // Batched windows (round 6): the four k = 1 projections of a block see ONE matrix of B * tw contiguous rows instead of B
    // items of tw rows -- M tiles then span batch items (16 x 500 rows = 63 tiles of 128 rows instead of 16 x 4 with 12 padded rows each:
    // -1.6 % matrix-pipe time).  Rows are independent and keep their K order: the same bits.  (Every tensor of the block is [B][tw][C]
    // without gaps; the row count stays inside the 2^27-element window of the kernels' 32-bit buffer offsets.)
    const bool flat = B > 1 && (int64_t)B * tw > small_m && (int64_t)B * tw * (F > 3 * S ? F : 3 * S) < (1LL << 27);
    const int rb = flat ? 1 : B, rt = flat ? B * tw : tw;
    for (int i = 0; i < nb; ++i) {
        const svcmi_whisper_block& blk = m.blocks[i];
        CV v; v.B = rb; v.t_in = rt; v.c_in = v.ldx = S; v.x_bs = (int64_t)tw * S;
        {
            CV q = v; q.x = h; q.x16 = h16; q.w = &blk.qkv; q.y = qkv; q.y_bs = (int64_t)tw * 3 * S; q.ldy = 3 * S; q.tile = t_qkv; q.tile_lp = l_qkv; q.ring_class = 0;
            if (att16) { q.y16 = qkv16; q.split_k = 1; }       // (the 16-bit copy comes out of the float4 epilogue: no K slices)
            conv(c, q);
        }
        if (att16) attention16(c, qkv16, a, at16, B, tw, H, S, scale, nullptr);
        else attention(c, qkv, a, B, tw, H, S, scale, nullptr, nullptr, 0, nullptr, at16);
        {
            CV o = v; o.x = a; o.x16 = at16; o.w = &blk.o; o.bias = false; o.slabs = slabs; o.split_k = so; o.tile = t_o; o.tile_lp = l_o; o.ring_class = 1;
            conv(c, o);
        }
        splitk_layernorm(c, slabs, so, blk.o.bias, x, blk.ln2_g, blk.ln2_b, h, B, tw, S, h16);
        {
            CV u = v; u.x = h; u.x16 = x3 ? nullptr : h16; u.w = &blk.m1; u.act = SVCMI_ACT_GELU; u.y = mm; u.y_bs = (int64_t)tw * F; u.ldy = F; u.split_k = 1;
            u.tile = t_m1; u.tile_lp = l_m1; u.y16 = mm16; u.ring_class = 2;
            conv(c, u);
        }
        {
            CV dn = v; dn.x = mm; dn.x16 = mm16; dn.c_in = dn.ldx = F; dn.x_bs = (int64_t)tw * F; dn.w = &blk.m2; dn.bias = false; dn.slabs = slabs;
            dn.split_k = sm; dn.tile = t_m2; dn.tile_lp = l_m2; dn.ring_class = 3;
            conv(c, dn);
        }
        const float* g = i + 1 < nb ? m.blocks[i + 1].ln1_g : m.lnp_g;
        const float* b = i + 1 < nb ? m.blocks[i + 1].ln1_b : m.lnp_b;
        splitk_layernorm(c, slabs, sm, blk.m2.bias, x, g, b, h, B, tw, S, h16);
    }
}

// ------------------------------------------------------------------------------------------------ prior encoder
// TextEncoder.forward, vits/models.py:39-52 + attentions.Encoder.forward, attentions.py:60-72.  z_p: [B][T][inter]
void prior_fwd(Ctx& c, const svcmi_synth_model& m, const svcmi_synth_io& io, float* z_p) {
    const int B = io.batch, T = io.t, H = m.hidden, I = m.inter, sh = io.ppg_row_shift;
    const int64_t mark = c.ar.mark();
    c.prec = class_prec(m, SVCMI_CLASS_ENC);
    float* xa = c.ar.f((int64_t)B * T * H);
    float* xb = c.ar.f((int64_t)B * T * H);
    const int ppg_rows = (T + (1 << sh) - 1) >> sh;
    {
        CV v; v.x = io.ppg; v.x_bs = io.ppg_bstride ? io.ppg_bstride : (int64_t)ppg_rows * m.ppg_dim; v.B = B; v.t_in = T; v.t_out = T;
        v.c_in = v.ldx = m.ppg_dim; v.w = &m.pre; v.ksize = 5; v.pad = 2; v.lengths = io.lengths; v.mask_out = true; v.rshift = sh;
        v.y = xa; v.y_bs = (int64_t)T * H; v.ldy = H;
        conv(c, v);
    }
    {
        CV v; v.x = io.vec; v.x_bs = (int64_t)T * m.vec_dim; v.B = B; v.t_in = T; v.c_in = v.ldx = m.vec_dim; v.w = &m.hub; v.ksize = 5;
        v.pad = 2; v.res = xa; v.res_bs = (int64_t)T * H; v.ldr = H; v.lengths = io.lengths; v.mask_out = true;
        v.y = xa; v.y_bs = (int64_t)T * H; v.ldy = H;
        conv(c, v);
    }
    run(c, OP_EMBED_PITCH, 0.0, 8.0 * B * T * H,
        [&] { return svcmi_embed_pitch_f32(xa, H, io.pit, m.pit_emb, io.lengths, B, T, H, c.stream); });
    const float scale = 1.0f / sqrtf((float)(H / m.n_heads));
    const int F = m.enc[0].f1.n, kf = m.enc_ffn_kernel, pl = (kf - 1) / 2;
    const int f2_nk = (m.enc[0].f2.ldw + 31) / 32;                     // K-steps of the second FFN convolution
    const int f2_blocks = B * ((T + 63) / 64) * ((H + 63) / 64);
    int f2_split = 1;
    if (f2_blocks < 256) {
        const int a = f2_nk / 10, b = (288 + f2_blocks / 2) / f2_blocks;
        f2_split = a < b ? a : b;
        if (f2_split < 1) f2_split = 1;
    }
    float* qkv = c.ar.f((int64_t)B * T * 3 * H);
    float* att = c.ar.f((int64_t)B * T * H);
    float* yo = c.ar.f((int64_t)B * T * H);
    float* hf = c.ar.f((int64_t)B * T * F);
    float* slabs = c.ar.f((int64_t)B * f2_split * T * H);
    // bf16 / f16 modes: 16-bit activation copies from the producers (the LayerNorms, the attention, the ReLU epilogue) for the _A16
    // GEMMs -- which only take over where a launch is large enough (B = 16: all of them; one clip: none) -- and the attention itself on
    // the 16-bit matrix cores from the QKV projection's 16-bit output copy (head widths the kernel has: 32 and 96)
    const int hd = H / m.n_heads;
    const bool a16 = mode16(c.prec) && H % 8 == 0 && F % 8 == 0;
    // the attention kernel's own mode: the encoder's in a single-mode model; under SVCMI_PREC_MIXED its own class, so that a split-bf16
    // encoder (fp32-class GEMMs) can still run QK^T / PV on the 16-bit matrix cores from a 16-bit copy of the QKV output
    const int att_prec = m.precision == SVCMI_PREC_MIXED ? class_prec(m, SVCMI_CLASS_ENC_ATTN) : c.prec;
    const bool att16 = mode16(att_prec) && H % 8 == 0 && (hd == 96 || hd == 32) && m.enc_window <= 4;
    void* x16 = a16 ? c.ar.take((int64_t)B * T * H * 2) : nullptr;
    void* at16 = a16 ? c.ar.take((int64_t)B * T * H * 2) : nullptr;
    void* hf16 = a16 ? c.ar.take((int64_t)B * T * F * 2) : nullptr;
    void* qkv16 = att16 ? c.ar.take((int64_t)B * T * 3 * H * 2) : nullptr;
    float *x = xa, *x2 = xb;
    for (int i = 0; i < m.n_enc; ++i) {
        const svcmi_enc_layer& L = m.enc[i];
        CV v; v.B = B; v.t_in = T; v.c_in = v.ldx = H; v.x_bs = (int64_t)T * H;
        {
            CV q = v; q.x = x; q.x16 = i > 0 ? x16 : nullptr; q.w = &L.qkv; q.y = qkv; q.y_bs = (int64_t)T * 3 * H; q.ldy = 3 * H;
            if (att16) { q.y16 = qkv16; q.y16_fmt = att_prec; q.split_k = 1; }
            conv(c, q);
        }
        if (att16) attention16(c, qkv16, att, fmt16_of(att_prec) == fmt16_of(c.prec) ? at16 : nullptr, B, T, m.n_heads, H, scale, io.lengths, L.rel_k, L.rel_v, m.enc_window, att_prec);
        else attention(c, qkv, att, B, T, m.n_heads, H, scale, L.rel_k, L.rel_v, m.enc_window, io.lengths, at16);
        { CV o = v; o.x = att; o.x16 = (att16 && fmt16_of(att_prec) != fmt16_of(c.prec)) ? nullptr : at16; o.w = &L.o; o.y = yo; o.y_bs = (int64_t)T * H; o.ldy = H; conv(c, o); }
        layernorm(c, x, yo, L.g1, L.b1, x2, B, T, H, H, H, H, 0, x16);
        {
            CV f = v; f.x = x2; f.x16 = x16; f.w = &L.f1; f.ksize = kf; f.pad = pl; f.act = SVCMI_ACT_RELU; f.lengths = io.lengths; f.mask_in = f.mask_out = true;
            f.y = hf; f.y_bs = (int64_t)T * F; f.ldy = F; f.y16 = hf16;
            if (a16) f.split_k = 1;
            conv(c, f);
        }
        // second FFN convolution: raw split-K slabs -> one launch that sums them with the bias and the residual and applies
        // norm_layers_2 (the `* x_mask` of attentions.py:209 only affects rows past the length, which no valid row ever reads:
        // keys are masked in the attention, inputs in the convolutions)
        {
            CV f = v; f.x = hf; f.x16 = hf16; f.c_in = f.ldx = F; f.x_bs = (int64_t)T * F; f.w = &L.f2; f.bias = false; f.ksize = kf; f.pad = pl;
            f.slabs = slabs; f.split_k = f2_split;
            conv(c, f);
        }
        splitk_layernorm(c, slabs, f2_split, L.f2.bias, x2, L.g2, L.b2, x, B, T, H, x16);
    }
    float* stats = qkv;            // [B][T][2I] (2I <= 3H checked by the caller)
    {
        CV v; v.x = x; v.x_bs = (int64_t)T * H; v.B = B; v.t_in = T; v.c_in = v.ldx = H; v.w = &m.proj; v.lengths = io.lengths;
        v.mask_in = v.mask_out = true; v.y = stats; v.y_bs = (int64_t)T * 2 * I; v.ldy = 2 * I;
        conv(c, v);
    }
    run(c, OP_SAMPLE_PRIOR, 0.0, 16.0 * B * T * I,
        [&] { return svcmi_sample_prior_f32(stats, 2 * I, io.noise, io.lengths, z_p, I, B, T, I, c.stream); });
    c.ar.release(mark);
}

// ------------------------------------------------------------------------------------------------ reverse flow
// ResidualCouplingBlock.forward(reverse=True), vits/models.py:89-94; layers vits/modules.py:288-321,178-203.  x [B][T][I] in place.
// WN state as rows of (h | skip): the res_skip convolution does `h = (h + rs[:, :H]) * mask; skip += rs[:, H:]` (modules.py:196-203)
// in its own epilogue (ACCUMULATE | MASK_OUT into the 2H-wide row), and the in_layer convolution hands its raw split-K slabs to the
// gate kernel, which adds them and the bias and applies tanh * sigmoid: 3 launches per WN layer.
void flow_fwd(Ctx& c, const svcmi_synth_model& m, const svcmi_synth_io& io, float* x) {
    const int B = io.batch, T = io.t, H = m.hidden, I = m.inter, half = I / 2, kf = m.flow_kernel;
    const int64_t mark = c.ar.mark();
    c.prec = class_prec(m, SVCMI_CLASS_FLOW);
    float* hs = c.ar.f((int64_t)B * T * 2 * H);
    float* skip = hs + H;
    const int nk = (kf * H + 31) / 32;
    const int blocks = B * ((T + 63) / 64) * ((2 * H + 63) / 64);
    int split = 1;
    if (blocks < 256) {
        const int a = nk / 8, b = (288 + blocks / 2) / blocks;
        split = a < b ? a : b;
        if (split < 1) split = 1;
    }
    float* msvs = c.ar.f((int64_t)B * 2 * half);
    float* x0n = c.ar.f((int64_t)B * T * half);
    float* acts = c.ar.f((int64_t)B * T * H);
    float* mpost = c.ar.f((int64_t)B * T * half);
    float* slabs = c.ar.f((int64_t)B * split * T * 2 * H);
    for (int fl = 0; fl < m.n_flow; ++fl) {
        const svcmi_flow_layer& L = m.flow[fl];
        {
            CV v; v.x = io.spk; v.x_bs = m.spk_dim; v.B = B; v.t_in = 1; v.c_in = v.ldx = m.spk_dim; v.w = &L.snac;
            v.y = msvs; v.y_bs = 2 * half; v.ldy = 2 * half;
            conv(c, v);
        }
        run(c, OP_COUPLING_PRE, 0.0, 8.0 * B * T * half,
            [&] { return svcmi_coupling_pre_f32(x, I, L.x0_off, msvs, x0n, half, io.lengths, B, T, half, c.stream); });
        {
            CV v; v.x = x0n; v.x_bs = (int64_t)T * half; v.B = B; v.t_in = T; v.c_in = v.ldx = half; v.w = &L.pre; v.lengths = io.lengths;
            v.mask_out = true; v.y = hs; v.y_bs = (int64_t)T * 2 * H; v.ldy = 2 * H;
            conv(c, v);
        }
        for (int l = 0; l < L.n_wn; ++l) {
            const svcmi_wn_layer& W = L.wn[l];
            {
                CV v; v.x = hs; v.x_bs = (int64_t)T * 2 * H; v.B = B; v.t_in = T; v.c_in = H; v.ldx = 2 * H; v.w = &W.in; v.bias = false;
                v.ksize = kf; v.pad = (kf - 1) / 2; v.slabs = slabs; v.split_k = split;
                conv(c, v);
            }
            run(c, OP_WN_GATE, 0.0, 4.0 * B * T * H * (2 * split + 1),
                [&] { return svcmi_wn_gate_f32(slabs, W.in.bias, acts, B, T, H, 2 * H, H, split, c.stream); });
            {
                CV v; v.x = acts; v.x_bs = (int64_t)T * H; v.B = B; v.t_in = T; v.c_in = v.ldx = H; v.w = &W.rs; v.lengths = io.lengths;
                v.mask_out = true; v.accumulate = true; v.y = l == L.n_wn - 1 ? skip : hs; v.y_bs = (int64_t)T * 2 * H; v.ldy = 2 * H;
                conv(c, v);
            }
        }
        {
            CV v; v.x = skip; v.x_bs = (int64_t)T * 2 * H; v.B = B; v.t_in = T; v.c_in = H; v.ldx = 2 * H; v.w = &L.post;
            v.lengths = io.lengths; v.mask_out = true; v.y = mpost; v.y_bs = (int64_t)T * half; v.ldy = half;
            conv(c, v);
        }
        run(c, OP_COUPLING_POST, 0.0, 12.0 * B * T * half,
            [&] { return svcmi_coupling_post_f32(x, I, L.x1_off, mpost, half, msvs, io.lengths, B, T, half, c.stream); });
    }
    c.ar.release(mark);
}

// ------------------------------------------------------------------------------------------------ generator
struct Snk { const float* a; const float* b; };

void snake_alias(Ctx& c, const svcmi_synth_model& m, const float* x, float* y, Snk s, int B, int64_t L, int C, int ld) {
    run(c, OP_SNAKE_ALIAS, 0.0, 8.0 * B * L * ld,
        [&] { return svcmi_snake_alias_f32(x, y, s.a, s.b, m.filt, B, (int32_t)L, C, ld, c.stream); });
}

// AMPBlock.forward (vits_decoder/bigv.py:50-58) of ONE block, launches in sequence: `for d: x = x + conv2(act(conv1_d(act(x))))`; the
// last iteration lands in acc as (x_new)/nb (+= for j > 0): the (sum of blocks)/nb of generator.py:188-194 in block order.
// Only for stages the grouped scheme below does not fit.
void amp_block_seq(Ctx& c, const svcmi_synth_model& m, const svcmi_gen_stage& st, const svcmi_amp_block& blk, const float* y, float* acc,
                   float* xj, float* tmp, float* tmp2, int j, int nb, int B, int64_t L) {
    const int cp = st.cp, k = blk.k;
    const int64_t bs = L * cp;
    bool fused = true;
    for (int q = 0; q < blk.n_dil; ++q) fused = fused && svcmi_snake_conv_preferred(st.c, cp, k, blk.dil[q]);
    const float* xc = y;
    for (int q = 0; q < blk.n_dil; ++q) {
        const bool last = q == blk.n_dil - 1;
        const int d = blk.dil[q];
        float* outp = last ? acc : xj;
        const float alpha = last ? 1.0f / nb : 1.0f;
        const bool accum = last && j > 0;
        if (fused) {
            run(c, OP_SNAKE_CONV, 2.0 * B * L * st.c * st.c * k, 8.0 * B * L * st.c, [&] {
                return svcmi_snake_conv_f32(xc, blk.c1[q].w, blk.c1[q].bias, nullptr, tmp, blk.a1_alpha[q], blk.a1_beta[q], m.filt, B,
                                            (int32_t)L, st.c, cp, blk.c1[q].ldw, k, d, 1.0f, 0, c.stream);
            });
            run(c, OP_SNAKE_CONV, 2.0 * B * L * st.c * st.c * k, 8.0 * B * L * st.c, [&] {
                return svcmi_snake_conv_f32(tmp, blk.c2[q].w, blk.c2[q].bias, xc, outp, blk.a2_alpha[q], blk.a2_beta[q], m.filt, B,
                                            (int32_t)L, st.c, cp, blk.c2[q].ldw, k, 1, alpha, accum ? 1 : 0, c.stream);
            });
        } else {
            snake_alias(c, m, xc, tmp, Snk{blk.a1_alpha[q], blk.a1_beta[q]}, B, L, cp, cp);
            CV v; v.B = B; v.t_in = (int)L; v.c_in = v.ldx = cp; v.x_bs = bs; v.y_bs = bs; v.ldy = cp; v.ksize = k; v.tile = amp_tile(cp);
            { CV a = v; a.x = tmp; a.w = &blk.c1[q]; a.dil = d; a.pad = (k * d - d) / 2; a.y = tmp2; conv(c, a); }
            snake_alias(c, m, tmp2, tmp, Snk{blk.a2_alpha[q], blk.a2_beta[q]}, B, L, cp, cp);
            {
                CV a = v; a.x = tmp; a.w = &blk.c2[q]; a.pad = (k - 1) / 2; a.res = xc; a.res_bs = bs; a.ldr = cp; a.alpha = alpha;
                a.accumulate = accum; a.y = outp;
                conv(c, a);
            }
        }
        xc = xj;
    }
}

// The nb AMP blocks of a stage in lock-step: at every step the blocks' activations go out as ONE grouped SnakeAlias launch and
// their convolutions as ONE grouped GEMM launch (3x the blocks per grid, longest K first) -- or, on the narrow stages, ONE grouped
// fused SnakeAlias+conv launch -- then acc = ((o_0 + o_1) + o_2) / nb exactly as generator.py:188-194 sums them.
bool amp_stage_grouped(Ctx& c, const svcmi_synth_model& m, const svcmi_gen_stage& st, const float* y, float* acc, int B, int64_t L) {
    const int nb = st.n_blocks, cp = st.cp;
    if (!g_amp_grouped || nb < 2 || nb > 3 || cp % 4) return false;
    const int nd = st.blocks[0].n_dil;
    int nf = 0, nall = 0;
    for (int j = 0; j < nb; ++j) {
        if (st.blocks[j].n_dil != nd) return false;
        for (int q = 0; q < nd; ++q) { nf += svcmi_snake_conv_preferred(st.c, cp, st.blocks[j].k, st.blocks[j].dil[q]) ? 1 : 0; ++nall; }
    }
    if (nf != 0 && nf != nall) return false;
    const bool fused = nf == nall;
    const int64_t n = (int64_t)B * L * cp, bs = L * cp;
    float *xj[3], *t1[3], *t2[3];
    void* t1h[3] = {nullptr, nullptr, nullptr};
    for (int j = 0; j < nb; ++j) xj[j] = c.ar.f(n);
    for (int j = 0; j < nb; ++j) t1[j] = c.ar.f(n);
    for (int j = 0; j < nb; ++j) t2[j] = c.ar.f(n);
    // bf16 / f16 modes, wide stages: SnakeAlias writes its output as 16-bit rows ONLY (it feeds nothing but the convolution) and
    // the grouped GEMM takes the _A16 kernel -- a quarter less activation traffic per half-step, half the operand bytes through LDS
    // (not in bf16x3: split rows + the _BF16X3_A16 kernel are no faster than the in-register split here, profiles/r03p_*)
    if (!fused && mode16(c.prec) && cp % 8 == 0)
        for (int j = 0; j < nb; ++j) t1h[j] = c.ar.take(n * 2);
    const float* xc[3] = {y, y, y};
    // f16 / f16w2 classes, narrow stages: the half-step with its convolution on the fp16 matrix cores
    bool lp = fused && g_amp_lp && (c.prec == SVCMI_PREC_F16 || c.prec == SVCMI_PREC_F16W2);
    for (int j = 0; lp && j < nb; ++j)
        for (int q = 0; q < nd; ++q) lp = lp && svcmi_snake_conv_lp_supported(st.c, cp, st.blocks[j].k, st.blocks[j].dil[q], SVCMI_PREC_F16W2);
    for (int q = 0; q < nd; ++q) {
        float** outs = q == nd - 1 ? t2 : xj;
        if (fused) {
            svcmi_snake_conv_desc d[3];
            double fl = 0.0;
            for (int j = 0; j < nb; ++j) {
                const svcmi_amp_block& b = st.blocks[j];
                d[j] = svcmi_snake_conv_desc{xc[j], b.c1[q].w, b.c1[q].bias, nullptr, t1[j], b.a1_alpha[q], b.a1_beta[q], b.c1[q].ldw, b.k,
                                             b.dil[q], 0, 1.0f};
                fl += 2.0 * B * L * st.c * st.c * b.k;
            }
            auto half_step = [&] {
                // always with split weights (hi + lo fragments, whatever the class: "f16" or "f16w2"): the second MFMA per tile costs 1.5 % of
                // the launch (the convolution is a tenth of it) and takes a third off the error these stages add (profiles/r04q_*)
                if (lp) run(c, OP_SNAKE_CONV_GROUP_LP, fl, 8.0 * nb * B * L * st.c,
                            [&] { return svcmi_snake_conv_group_lp(d, nb, m.filt, B, (int32_t)L, st.c, cp, SVCMI_PREC_F16W2, c.stream); });
                else run(c, OP_SNAKE_CONV_GROUP, fl, 8.0 * nb * B * L * st.c,
                         [&] { return svcmi_snake_conv_group_f32(d, nb, m.filt, B, (int32_t)L, st.c, cp, c.stream); });
            };
            half_step();
            for (int j = 0; j < nb; ++j) {
                const svcmi_amp_block& b = st.blocks[j];
                d[j] = svcmi_snake_conv_desc{t1[j], b.c2[q].w, b.c2[q].bias, xc[j], outs[j], b.a2_alpha[q], b.a2_beta[q], b.c2[q].ldw, b.k,
                                             1, 0, 1.0f};
            }
            half_step();
        } else {
            const float* px[3]; float* py[3]; const float *pa[3], *pb[3];
            CV vs[3];
            // activation -> 16-bit rows when the convolution that reads them will run on the _A16 kernel (decided from the same descriptors)
            auto snake = [&](bool second) {
                const bool h16 = t1h[0] && conv_group(c, vs, nb, true) >= SVCMI_PREC_BF16_A16;
                for (int j = 0; j < nb; ++j) {
                    px[j] = second ? t2[j] : xc[j]; py[j] = t1[j];
                    pa[j] = second ? st.blocks[j].a2_alpha[q] : st.blocks[j].a1_alpha[q];
                    pb[j] = second ? st.blocks[j].a2_beta[q] : st.blocks[j].a1_beta[q];
                    if (!h16) vs[j].x16 = nullptr;
                }
                run(c, OP_SNAKE_ALIAS_GROUP, 0.0, (h16 ? 4.0 + 2.0 * w16(c.prec, 1) : 8.0) * nb * B * L * cp, [&] {
                    return svcmi_snake_alias_group_f32(px, h16 ? nullptr : py, pa, pb, m.filt, nb, B, (int32_t)L, cp, cp, h16 ? t1h : nullptr, fmt16_of(c.prec),
                                                       c.stream);
                });
            };
            for (int j = 0; j < nb; ++j) {
                const svcmi_amp_block& b = st.blocks[j];
                CV& v = vs[j]; v = CV();
                v.x = t1[j]; v.x16 = t1h[j]; v.x_bs = bs; v.B = B; v.t_in = (int)L; v.c_in = v.ldx = cp; v.w = &b.c1[q]; v.ksize = b.k; v.dil = b.dil[q];
                v.tile = amp_tile(cp);
                v.pad = (b.k * b.dil[q] - b.dil[q]) / 2; v.y = t2[j]; v.y_bs = bs; v.ldy = cp;
            }
            snake(false);
            conv_group(c, vs, nb);
            for (int j = 0; j < nb; ++j) {          // t2 is free again once the second activation has read it
                const svcmi_amp_block& b = st.blocks[j];
                CV& v = vs[j]; v = CV();
                v.x = t1[j]; v.x16 = t1h[j]; v.x_bs = bs; v.B = B; v.t_in = (int)L; v.c_in = v.ldx = cp; v.w = &b.c2[q]; v.ksize = b.k; v.pad = (b.k - 1) / 2;
                v.tile = amp_tile(cp);
                v.res = xc[j]; v.res_bs = bs; v.ldr = cp; v.y = outs[j]; v.y_bs = bs; v.ldy = cp;
            }
            snake(true);
            conv_group(c, vs, nb);
        }
        for (int j = 0; j < nb; ++j) xc[j] = outs[j];
    }
    run(c, OP_BLOCK_MEAN, 0.0, 4.0 * (nb + 1) * n, [&] { return svcmi_block_mean_f32(xc, nb, acc, n, c.stream); });
    return true;
}

// Generator.inference, vits_decoder/generator.py:175-200 (+ SpeakerAdapter :36-47, AMPBlock bigv.py:50-58).
// z [B][T][U] time-major, source [B][hop*T] -> wave [B][hop*T].  split_k: 0 = library heuristic, 1 = off (streaming tiles: the slice
// count would otherwise follow the tile size and break the bit-identity of tiled and untiled runs).
void generator_tile(Ctx& c, const svcmi_synth_model& m, const float* z, const float* spk, const float* source, float* wave, int B, int T,
                    int split_k, int stop_after) {
    const int U = m.upsample_input;
    const int64_t mark = c.ar.mark();
    c.prec = class_prec(m, SVCMI_CLASS_UPS);
    float* sb = c.ar.f((int64_t)B * 2 * U);
    {
        CV v; v.x = spk; v.x_bs = m.spk_dim; v.B = B; v.t_in = 1; v.c_in = v.ldx = m.spk_dim; v.w = &m.adapter; v.y = sb; v.y_bs = 2 * U; v.ldy = 2 * U;
        conv(c, v);
    }
    float* xn = c.ar.f((int64_t)B * T * U);
    layernorm(c, z, nullptr, sb, sb + U, xn, B, T, U, U, 0, U, 2 * U);
    int cin = m.conv_pre.n;
    float* x = c.ar.f((int64_t)B * T * cin);
    {
        CV v; v.x = xn; v.x_bs = (int64_t)T * U; v.B = B; v.t_in = T; v.c_in = v.ldx = U; v.w = &m.conv_pre; v.ksize = 7; v.pad = 3;
        v.act = SVCMI_ACT_MISH; v.split_k = split_k; v.y = x; v.y_bs = (int64_t)T * cin; v.ldy = cin;
        conv(c, v);
    }
    const int64_t Lsrc = (int64_t)T * m.hop;
    if (stop_after == SVCMI_STOP_GEN_PRE) { c.ar.release(mark); return; }        // (truncated pipelines are timing aids: `wave` stays untouched)
    int64_t t_in = T;
    for (int i = 0; i < m.n_stages; ++i) {
        const svcmi_gen_stage& st = m.stages[i];
        const int cp = st.cp, u = st.u;
        const int64_t L = t_in * u;
        float* y = c.ar.f((int64_t)B * L * cp);
        float* acc = c.ar.f((int64_t)B * L * cp);
        const int64_t smark = c.ar.mark();
        c.prec = class_prec(m, SVCMI_CLASS_UPS);
        CV up; up.x = x; up.x_bs = t_in * cin; up.B = B; up.t_in = (int)t_in; up.t_out = (int)t_in; up.c_in = up.ldx = cin; up.w = &st.up;
        up.ksize = st.up_taps; up.pad = st.up_pad; up.split_k = split_k; up.y = y; up.y_bs = L * cp; up.ldy = u * cp;
        if (svcmi_upsample_noise_supported(u, cp, cin)) {
            // narrowest stages: the source convolution (and at 10 channels the transposed convolution too) is a pure stream -- one
            // VALU kernel instead of padded GEMM launches (105 / 63 us for < 0.1 GFLOP)
            const bool fuse_up = cp <= 12;
            if (!fuse_up) conv(c, up);
            run(c, OP_UPSAMPLE_NOISE, 0.0, 4.0 * B * t_in * (cin + u * cp), [&] {
                return svcmi_upsample_noise_f32(fuse_up ? x : nullptr, st.up.w, st.up.bias, source, st.nz.w, st.nz.bias, y, B, (int32_t)t_in,
                                                fuse_up ? cin : 4, st.up.ldw, st.up_taps, st.up_pad, u, cp, Lsrc, st.nz_k, st.nz_stride,
                                                st.nz_pad, st.nz.ldw, c.stream);
            });
        } else {
            conv(c, up);
            CV nz; nz.x = source; nz.x_bs = Lsrc; nz.B = B; nz.t_in = (int)Lsrc; nz.t_out = (int)L; nz.c_in = 1; nz.ldx = 1; nz.w = &st.nz;
            nz.ksize = st.nz_k; nz.stride = st.nz_stride; nz.pad = st.nz_pad; nz.accumulate = true; nz.split_k = split_k;
            nz.y = y; nz.y_bs = L * cp; nz.ldy = cp;
            conv(c, nz);
        }
        c.prec = class_prec(m, SVCMI_CLASS_AMP0 + (i < SVCMI_AMP_CLASSES ? i : SVCMI_AMP_CLASSES - 1));   // a sixth stage shares the fifth stage's class
        if (!amp_stage_grouped(c, m, st, y, acc, B, L)) {
            c.ar.release(smark);
            float* xj = c.ar.f((int64_t)B * L * cp);
            float* tmp = c.ar.f((int64_t)B * L * cp);
            float* tmp2 = c.ar.f((int64_t)B * L * cp);
            for (int j = 0; j < st.n_blocks; ++j) amp_block_seq(c, m, st, st.blocks[j], y, acc, xj, tmp, tmp2, j, st.n_blocks, B, L);
        }
        c.ar.release(smark);
        // (x of the previous stage and y stay allocated below acc: the arena is released as a whole at the end of the tile)
        x = acc; cin = cp; t_in = L;
        if (stop_after == SVCMI_STOP_STAGE0 + i) { c.ar.release(mark); return; }
    }
    const svcmi_gen_stage& last = m.stages[m.n_stages - 1];
    c.prec = class_prec(m, SVCMI_CLASS_UPS);
    if (svcmi_snake_post_supported(last.c, cin, 7) && m.post.ldw >= 7 * cin) {
        run(c, OP_SNAKE_POST, 2.0 * B * t_in * last.c * 7, 4.0 * B * t_in * (last.c + 1), [&] {
            return svcmi_snake_post_f32(x, m.post.w, wave, m.post_alpha, m.post_beta, m.filt, B, (int32_t)t_in, last.c, cin, 7, c.stream);
        });
    } else {
        float* a = c.ar.f((int64_t)B * t_in * cin);
        snake_alias(c, m, x, a, Snk{m.post_alpha, m.post_beta}, B, t_in, cin, cin);
        CV v; v.x = a; v.x_bs = t_in * cin; v.B = B; v.t_in = (int)t_in; v.c_in = v.ldx = cin; v.w = &m.post; v.bias = false; v.n_out = 1;
        v.ksize = 7; v.pad = 3; v.act = SVCMI_ACT_TANH; v.y = wave; v.y_bs = t_in; v.ldy = 1;
        conv(c, v);
    }
    c.ar.release(mark);
}

// Halo of the streaming decoder, in frames.  SURVEY.md A.4 measured an EFFECTIVE receptive field of -23.3 .. +23.7 frames; the exact
// support of the FIR chain is wider (the Kaiser tails it ignores are ~1e-3): per side, in output samples: conv_pre 3 frames = 960;
// ups 480 + 64 + 16 + 4 + 2; an AMP block with k = 11 adds (5 + 15 + 25) + 3 * 5 + 6 * 6 = 96 samples at its stage's rate =
// 96 * (64 + 16 + 4 + 2 + 1) = 8352; output layer 9: 9887 samples = 30.9 frames.  24 frames left a 9e-8 leak (measured); 32 is exact.
constexpr int STREAM_HALO = 32;

void generator_fwd(Ctx& c, const svcmi_synth_model& m, const svcmi_synth_io& io, const float* z) {
    const int B = io.batch, T = io.t, S = io.stream_frames, U = m.upsample_input, hop = m.hop;
    if (S <= 0) { generator_tile(c, m, z, io.spk, io.source, io.wave, B, T, 0, io.stop_after); return; }
    const int64_t mark = c.ar.mark();
    const int tmax = S + 2 * STREAM_HALO < T ? S + 2 * STREAM_HALO : T;
    float* zt = c.ar.f((int64_t)B * tmax * U);
    float* st = c.ar.f((int64_t)B * tmax * hop);
    float* ot = c.ar.f((int64_t)B * tmax * hop);
    for (int t0 = 0; t0 < T; t0 += S) {
        const int a = t0 - STREAM_HALO > 0 ? t0 - STREAM_HALO : 0, b = t0 + S + STREAM_HALO < T ? t0 + S + STREAM_HALO : T;
        const int n = S < T - t0 ? S : T - t0, tt = b - a;
        copy2d(c, z + (int64_t)a * U, (int64_t)T * U, zt, (int64_t)tt * U, B, (int64_t)tt * U);
        copy2d(c, io.source + (int64_t)a * hop, (int64_t)T * hop, st, (int64_t)tt * hop, B, (int64_t)tt * hop);
        generator_tile(c, m, zt, io.spk, st, ot, B, tt, 1, SVCMI_STOP_NONE);
        copy2d(c, ot + (int64_t)(t0 - a) * hop, (int64_t)tt * hop, io.wave + (int64_t)t0 * hop, (int64_t)T * hop, B, (int64_t)n * hop);
    }
    c.ar.release(mark);
}

bool synth_shapes_ok(const svcmi_synth_model& m, const svcmi_synth_io& io) {
    if (m.precision < SVCMI_PREC_F32 || (m.precision > SVCMI_PREC_F16 && m.precision != SVCMI_PREC_MIXED && m.precision != SVCMI_PREC_F16W2)) return false;
    if (m.precision == SVCMI_PREC_MIXED)
        for (int i = 0; i < SVCMI_PREC_CLASSES; ++i)
            if ((m.class_prec[i] < SVCMI_PREC_F32 || m.class_prec[i] > SVCMI_PREC_F16) && m.class_prec[i] != SVCMI_PREC_F16W2) return false;
    return io.batch > 0 && io.t > 0 && m.n_enc >= 1 && m.n_enc <= SVCMI_MAX_ENC_LAYERS && m.n_flow >= 0 && m.n_flow <= SVCMI_MAX_FLOWS &&
           m.n_stages >= 1 && m.n_stages <= SVCMI_MAX_STAGES && m.hidden % 4 == 0 && m.inter % 8 == 0 && 2 * m.inter <= 3 * m.hidden &&
           m.upsample_input % 4 == 0 && (io.ppg_row_shift == 0 || io.ppg_row_shift == 1) && io.ppg_bstride % 4 == 0;
}

void synth_fwd(Ctx& c, const svcmi_synth_model& m, const svcmi_synth_io& io) {
    if (!synth_shapes_ok(m, io)) { c.rc = SVCMI_EINVAL; return; }
    const int B = io.batch, T = io.t, I = m.inter;
    c.lp_min_flops = m.lp_min_flops > 0.f ? m.lp_min_flops : 1.5e9f;
    c.sk_ws = c.ar.f(SPLITK_FLOATS);
    float* zp = c.ar.f((int64_t)B * T * I);
    prior_fwd(c, m, io, zp);
    if (io.z_p) copy2d(c, zp, (int64_t)T * I, io.z_p, (int64_t)T * I, B, (int64_t)T * I);
    if (io.stop_after == SVCMI_STOP_PRIOR) return;
    flow_fwd(c, m, io, zp);
    if (io.z) copy2d(c, zp, (int64_t)T * I, io.z, (int64_t)T * I, B, (int64_t)T * I);
    if (io.stop_after == SVCMI_STOP_FLOW) return;
    generator_fwd(c, m, io, zp);
}

Ctx make_ctx(void* ws, int64_t bytes, void* stream, bool plan) {
    Ctx c;
    c.stream = stream;
    c.plan = plan;
    c.ar.base = plan ? nullptr : static_cast<char*>(ws);
    c.ar.cap = plan ? 0 : bytes;
    return c;
}

}  // namespace

extern "C" int64_t svcmi_whisper_workspace_bytes(const svcmi_whisper_model* m, int32_t batch, int32_t n_frames) {
    if (!m) return SVCMI_EINVAL;
    Ctx c = make_ctx(nullptr, 0, nullptr, true);
    whisper_fwd(c, *m, nullptr, nullptr, 0.f, batch, n_frames, nullptr);
    return c.rc ? (int64_t)c.rc : c.ar.peak + 256;
}

extern "C" int svcmi_whisper_encoder_fwd(const svcmi_whisper_model* m, const float* mel, const float* noise, float noise_scale,
                                         int32_t batch, int32_t n_frames, float* out, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!m || !mel || !out || !workspace || ((uintptr_t)workspace & 255)) return SVCMI_EINVAL;
    if (workspace_bytes < svcmi_whisper_workspace_bytes(m, batch, n_frames)) return SVCMI_EINVAL;
    Ctx c = make_ctx(workspace, workspace_bytes, stream, false);
    whisper_fwd(c, *m, mel, noise, noise_scale, batch, n_frames, out);
    return finish(c);
}

extern "C" int64_t svcmi_synth_workspace_bytes(const svcmi_synth_model* m, int32_t batch, int32_t t, int32_t stream_frames) {
    if (!m) return SVCMI_EINVAL;
    svcmi_synth_io io;
    memset(&io, 0, sizeof(io));
    io.batch = batch; io.t = t; io.stream_frames = stream_frames;
    Ctx c = make_ctx(nullptr, 0, nullptr, true);
    synth_fwd(c, *m, io);
    const int64_t a = c.ar.peak;
    const int64_t p2s = (int64_t)batch * t * 11 * 8 + 256;          // svcmi_pitch2source_fwd on the same workspace
    return c.rc ? (int64_t)c.rc : (a > p2s ? a : p2s) + 256;
}

extern "C" int svcmi_pitch2source_fwd(const svcmi_synth_model* m, const float* f0, const float* rand_ini, const float* noise, int32_t batch,
                                      int32_t t, float* source, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!m || !f0 || !rand_ini || !noise || !source || !workspace || batch <= 0 || t <= 0) return SVCMI_EINVAL;
    if (((uintptr_t)workspace & 255) || workspace_bytes < (int64_t)batch * t * 11 * 8) return SVCMI_EINVAL;
    Ctx c = make_ctx(workspace, workspace_bytes, stream, false);
    double* prefix = static_cast<double*>(c.ar.take((int64_t)batch * t * 11 * 8));
    run(c, OP_PITCH_PREFIX, 0.0, 0.0, [&] { return svcmi_pitch_prefix_f64(f0, rand_ini, prefix, batch, t, m->hop, m->sampling_rate, stream); });
    run(c, OP_PITCH_SOURCE, 0.0, 48.0 * batch * t * m->hop, [&] {
        return svcmi_pitch_source_f32(f0, prefix, noise, m->merge_w, m->merge_b, source, batch, t, m->hop, m->sampling_rate, stream);
    });
    return finish(c);
}

namespace {
template <typename F>
int synth_entry(const svcmi_synth_model* m, const svcmi_synth_io* io, void* workspace, int64_t workspace_bytes, void* stream, F&& body) {
    if (!m || !io || !workspace || ((uintptr_t)workspace & 255)) return SVCMI_EINVAL;
    if (!synth_shapes_ok(*m, *io)) return SVCMI_EINVAL;
    if (workspace_bytes < svcmi_synth_workspace_bytes(m, io->batch, io->t, io->stream_frames)) return SVCMI_EINVAL;
    Ctx c = make_ctx(workspace, workspace_bytes, stream, false);
    c.lp_min_flops = m->lp_min_flops > 0.f ? m->lp_min_flops : 1.5e9f;       // (c.prec: set per layer class by the sections themselves)
    body(c);
    return finish(c);
}
}  // namespace

extern "C" int svcmi_text_encoder_fwd(const svcmi_synth_model* m, const svcmi_synth_io* io, float* z_p, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
    if (!z_p || !io || !io->ppg || !io->vec || !io->pit || !io->lengths || !io->noise) return SVCMI_EINVAL;
    return synth_entry(m, io, workspace, workspace_bytes, stream, [&](Ctx& c) {
        c.sk_ws = c.ar.f(SPLITK_FLOATS);
        prior_fwd(c, *m, *io, z_p);
    });
}

extern "C" int svcmi_flow_reverse_fwd(const svcmi_synth_model* m, const svcmi_synth_io* io, float* x, void* workspace, int64_t workspace_bytes,
                                      void* stream) {
    if (!x || !io || !io->spk || !io->lengths) return SVCMI_EINVAL;
    return synth_entry(m, io, workspace, workspace_bytes, stream, [&](Ctx& c) {
        c.sk_ws = c.ar.f(SPLITK_FLOATS);
        flow_fwd(c, *m, *io, x);
    });
}

extern "C" int svcmi_generator_fwd(const svcmi_synth_model* m, const svcmi_synth_io* io, const float* z, void* workspace, int64_t workspace_bytes,
                                   void* stream) {
    if (!z || !io || !io->spk || !io->source || !io->wave) return SVCMI_EINVAL;
    return synth_entry(m, io, workspace, workspace_bytes, stream, [&](Ctx& c) {
        c.sk_ws = c.ar.f(SPLITK_FLOATS);
        generator_fwd(c, *m, *io, z);
    });
}

extern "C" int svcmi_synth_infer_fwd(const svcmi_synth_model* m, const svcmi_synth_io* io, void* workspace, int64_t workspace_bytes, void* stream) {
    if (!io || !io->ppg || !io->vec || !io->pit || !io->spk || !io->lengths || !io->source || !io->noise || !io->wave) return SVCMI_EINVAL;
    return synth_entry(m, io, workspace, workspace_bytes, stream, [&](Ctx& c) { synth_fwd(c, *m, *io); });
}

extern "C" int svcmi_trace_begin(int32_t max_records) {
    if (max_records <= 0) return SVCMI_EINVAL;
    g_trace.recs.clear();
    g_trace.recs.reserve(max_records);
    g_trace.cap = max_records;
    g_trace.on = true;
    return 0;
}

extern "C" int svcmi_trace_end(svcmi_trace_record* out, int32_t cap) {
    Trace& t = g_trace;
    t.on = false;
    const int n = (int)t.recs.size();
#ifndef SVCMI_EMU
    if (n) (void)hipEventSynchronize(t.recs[n - 1].e1);
#endif
    for (int i = 0; i < n; ++i) {
        float ms = 0.f;
#ifndef SVCMI_EMU
        (void)hipEventElapsedTime(&ms, t.recs[i].e0, t.recs[i].e1);
        (void)hipEventDestroy(t.recs[i].e0);
        (void)hipEventDestroy(t.recs[i].e1);
#endif
        if (out && i < cap) out[i] = svcmi_trace_record{t.recs[i].op, ms, t.recs[i].flops, t.recs[i].bytes};
    }
    t.recs.clear();
    return n;
}

extern "C" const char* svcmi_trace_op_name(int32_t op) { return op >= 0 && (op & 255) < OP_COUNT ? OP_NAMES[op & 255] : ""; }

// layout check for FFI bindings: sizeof of the structs a caller fills, in the order svcmi/_lib.py lists them
extern "C" int svcmi_struct_sizes(int64_t* out, int32_t cap) {
    const int64_t v[] = {(int64_t)sizeof(svcmi_weight), (int64_t)sizeof(svcmi_whisper_model), (int64_t)sizeof(svcmi_synth_model),
                         (int64_t)sizeof(svcmi_synth_io), (int64_t)sizeof(svcmi_trace_record), (int64_t)sizeof(svcmi_conv_desc),
                         (int64_t)sizeof(svcmi_snake_conv_desc)};
    const int n = (int)(sizeof(v) / sizeof(v[0]));
    for (int i = 0; i < n && i < cap; ++i) out[i] = v[i];
    return n < cap ? n : cap;
}

extern "C" int svcmi_conv_tune_get(const char* name, int32_t* value);      // conv_gemm.hip
// The current value of a knob (ABI 22): a caller that changes one for a while restores what it found (svcmi.serving.ClipLanes).
extern "C" int svcmi_tune_get(const char* name, int32_t* value) {
    if (!name || !value) return SVCMI_EINVAL;
    if (strcmp(name, "ring2") == 0) { *value = g_ring2; return 0; }
    if (strcmp(name, "amp_grouped") == 0) { *value = g_amp_grouped; return 0; }
    if (strcmp(name, "amp_lp") == 0) { *value = g_amp_lp; return 0; }
    return svcmi_conv_tune_get(name, value);
}

extern "C" int svcmi_host_tune_set(const char* name, int32_t value) {
    if (strcmp(name, "amp_grouped") == 0 && (value == 0 || value == 1)) { g_amp_grouped = value; return 0; }
    if (strcmp(name, "amp_lp") == 0 && (value == 0 || value == 1)) { g_amp_lp = value; return 0; }
    if (strcmp(name, "ring2") == 0 && value >= 0 && value < 32) { g_ring2 = value; return 0; }
    return SVCMI_EINVAL;
}

// ------------------------------------------------------------------------------------------------ packed-model files
// A model for a host that has no Python: `python -m svcmi.tools pack` (svcmi/packed.py) writes ONE file = header + relocation table +
// the model struct with every pointer replaced by (byte offset into the arena + 1, 0 = NULL) + the flat weight arena (fp32, packed
// layouts, 256-byte aligned tensors).  The host reads the file, uploads the arena bytes to the device and calls
// svcmi_packed_model_bind, which copies the struct image and turns the offsets into device pointers.  Nothing else to parse.
namespace {
struct PackedHeader {
    char magic[8];              // "SVCMIPK1"
    uint32_t kind, abi;         // 1 = svcmi_synth_model, 2 = svcmi_whisper_model; SVCMI_ABI_VERSION of the writer
    uint64_t struct_bytes, n_reloc, arena_offset, arena_bytes;
};
const PackedHeader* packed_header(const void* file, int64_t file_bytes) {
    if (!file || file_bytes < (int64_t)sizeof(PackedHeader)) return nullptr;
    const PackedHeader* h = static_cast<const PackedHeader*>(file);
    if (memcmp(h->magic, "SVCMIPK1", 8) != 0 || h->abi != SVCMI_ABI_VERSION) return nullptr;
    const uint64_t want = h->kind == 1 ? sizeof(svcmi_synth_model) : (h->kind == 2 ? sizeof(svcmi_whisper_model) : 0);
    if (!want || h->struct_bytes != want) return nullptr;
    // every field below is file-controlled: compare by subtraction from a bound already established, never by a sum that can wrap
    const uint64_t fb = (uint64_t)file_bytes, fixed = sizeof(PackedHeader) + h->struct_bytes;
    if (h->arena_offset > fb || h->arena_offset < fixed) return nullptr;
    if (h->n_reloc > (h->arena_offset - fixed) / 8) return nullptr;
    if (h->arena_bytes > fb - h->arena_offset) return nullptr;
    return h;
}
}  // namespace

extern "C" int svcmi_packed_model_info(const void* file, int64_t file_bytes, int32_t* kind, int64_t* arena_offset, int64_t* arena_bytes) {
    const PackedHeader* h = packed_header(file, file_bytes);
    if (!h) return SVCMI_EINVAL;
    if (kind) *kind = (int32_t)h->kind;
    if (arena_offset) *arena_offset = (int64_t)h->arena_offset;
    if (arena_bytes) *arena_bytes = (int64_t)h->arena_bytes;
    return 0;
}

extern "C" int svcmi_packed_model_bind(const void* file, int64_t file_bytes, const void* device_arena, void* model_out, int64_t model_bytes) {
    const PackedHeader* h = packed_header(file, file_bytes);
    if (!h || !device_arena || !model_out || (uint64_t)model_bytes != h->struct_bytes || ((uintptr_t)device_arena & 255)) return SVCMI_EINVAL;
    const char* p = static_cast<const char*>(file) + sizeof(PackedHeader);
    const uint64_t* reloc = reinterpret_cast<const uint64_t*>(p);
    const char* image = p + 8 * h->n_reloc;
    char* out = static_cast<char*>(model_out);
    memcpy(out, image, h->struct_bytes);
    for (uint64_t i = 0; i < h->n_reloc; ++i) {
        if (reloc[i] > h->struct_bytes - sizeof(void*) || reloc[i] % sizeof(void*)) return SVCMI_EINVAL;
        uint64_t v;
        memcpy(&v, out + reloc[i], 8);
        const void* ptr = nullptr;
        if (v) {
            if (v - 1 >= h->arena_bytes) return SVCMI_EINVAL;
            ptr = static_cast<const char*>(device_arena) + (v - 1);
        }
        memcpy(out + reloc[i], &ptr, sizeof(void*));
    }
    return 0;
}
