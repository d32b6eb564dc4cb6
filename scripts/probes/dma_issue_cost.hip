// Probe (round 6): what does ONE LDS-DMA issue cost the fp32 matrix pipe, and WHO pays for it -- the issuing wave or its SIMD?
// The GEMM K loop of conv_gemm_body.h measures 1.15 us per K-step with no DMA and no LDS reads, 1.24 with the reads, 1.52 with both
// (profiles/r06a_kprobe.log): the fill alone takes 0.44 us and does NOT overlap.  Here: bare v_mfma_f32_16x16x4_f32 streams (40 per
// K-step on 5 accumulators = the 64x80 tile's wave) and the same number of 1-KiB `buffer_load_dwordx4 ... lds` pieces per CU and
// K-step, issued either by the MFMA waves themselves (between their MFMAs) or by extra loader waves that issue nothing else.
//   SELF = pieces per MFMA wave per K-step, LOAD = pieces per loader wave per K-step, BAR = one s_barrier per K-step for everybody.
// Source addresses mimic the M = 500 x N = 5120 x K = 1280 launch (A rows from a 2.5 MB matrix, B rows from a 26 MB one).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int rsrc_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
    const size_t a = (size_t)base;
    rsrc_t r; r.x = (int)(unsigned)a; r.y = (int)(unsigned)(a >> 32); r.z = (int)bytes; r.w = 0x00020000;
    return r;
}
__device__ __forceinline__ void dma16(unsigned voff, unsigned lds_base, rsrc_t rsrc) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(lds_base), "s"(rsrc) : "memory");
}
template <int N> __device__ __forceinline__ void vm_wait() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

constexpr int KDIM = 1280, NROWS_A = 512, NROWS_B = 5120;

template <int NMW, int NLW, int SELF, int LOAD, bool BAR, int FORCE_KB = 0, int UNROLL = 1, int DELAY = 0, int PAD = 0, int PLAIN = 0>
__global__ __launch_bounds__(64 * (NMW + NLW)) void kern(const float* A, const float* W, float* out, int ksteps, unsigned* where, unsigned long long* stamps) {
    constexpr int PIECES = NMW * SELF + NLW * LOAD;          // per block per K-step
    constexpr int RING = 3;
    constexpr int LDSF = (PIECES * RING > FORCE_KB ? PIECES * RING : FORCE_KB) * 256;
    __shared__ __attribute__((aligned(16))) float lds[LDSF > 0 ? LDSF : 256];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds);
    if (lane == 0) {       // where this wave runs: (XCC id << 16) | HW_ID[15:0] (wave, simd, pipe, cu, sh, se)
        unsigned hw, xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
        where[blockIdx.x * (NMW + NLW) + wave] = ((xcc & 0xf) << 16) | (hw & 0xffff);
    }
    if (FORCE_KB > 0 && ksteps < 0) lds[tid] = 1.f;      // (keeps the forced allocation alive)
    const rsrc_t ra = make_rsrc(A, NROWS_A * KDIM * 4u), rw = make_rsrc(W, NROWS_B * KDIM * 4u);
    constexpr int MT = NMW == 4 ? 8 : 4;
    const int mt = blockIdx.x % MT, nt = blockIdx.x / MT;
    // piece q of the block: the first 40 % are A rows, the rest B rows (8 rows x 128 B per piece)
    auto voff = [&](int q, int it) -> unsigned {
        const int row8 = q * 8 + (lane >> 3);
        const unsigned k = (unsigned)((it % (KDIM / 32)) * 128 + (lane & 7) * 16);
        return (unsigned)(row8 * KDIM * 4) + k;
    };
    unsigned long long t_begin, c_begin;
    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin), "=s"(c_begin));
    if (DELAY > 0 && wave >= NMW / 2 && wave < NMW)
        for (int d = 0; d < DELAY; ++d) __builtin_amdgcn_s_sleep(16);
    if (wave < NMW) {
        f32x4 c[5];
        for (int k = 0; k < 5; ++k) for (int r = 0; r < 4; ++r) c[k][r] = 0.f;
        float a[4], b[4];
        for (int k = 0; k < 4; ++k) { a[k] = 0.001f * (float)(lane + k + 1); b[k] = 0.002f * (float)(lane * 3 + k + 1); }
        f32x4 pl[2][PLAIN > 0 ? PLAIN : 1];          // PLAIN of the SELF pieces go straight to registers (16 rows x 64 B per instruction = an A fragment), used one K-step later
        for (int u = 0; u < 2; ++u) for (int q = 0; q < (PLAIN > 0 ? PLAIN : 1); ++q) pl[u][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (PAD) asm volatile("s_nop 0");       // shifts the loop by 4 bytes: 8-byte instructions at 0 or 4 mod 8
        for (int it0 = 0; it0 < ksteps; it0 += UNROLL)
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const int it = it0 + u;
            const unsigned slot = (unsigned)(it % RING) * PIECES * 1024u;
#pragma unroll
            for (int m = 0; m < 40; ++m) {
                c[m % 5] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(m / 5) & 3], b[(m + m / 5) & 3], c[m % 5], 0, 0, 0);
                if (SELF > 0) {
#pragma unroll
                    for (int q = 0; q < SELF; ++q)
                        if (m == (40 / SELF) * q + (40 / SELF) / 2) {
                            const int piece = wave * SELF + q;
                            const bool isA = piece * 5 < PIECES * 2;
                            const int rowbase = isA ? mt * 16 * NMW : nt * 80 - (PIECES * 2 / 5) * 8;
                            asm volatile("" : "+v"(c[m % 5]));        // keep the DMA at this position of the stream
                            if (q < PLAIN) {        // fragment-shaped: lane l -> row l & 15, 16 bytes at 16 * (l >> 4) (+ 64 for the second half)
                                const unsigned fo = (unsigned)((mt * 16 * NMW + wave * 16 + (lane & 15)) * KDIM * 4) + (unsigned)((it % (KDIM / 32)) * 128 + (q & 1) * 64 + (lane >> 4) * 16);
                                asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "=v"(pl[u & 1][q]) : "v"(fo), "s"(ra) : "memory");
                            } else
                            dma16(voff(piece, it) + (unsigned)(rowbase * KDIM * 4), lds0 + slot + (unsigned)piece * 1024u, isA ? ra : rw);
                            asm volatile("" : "+v"(c[(m + 1) % 5]));
                        }
                }
            }
            if (SELF > 0) vm_wait<SELF>();                  // the previous K-step's pieces have landed
            if (PLAIN > 0) {                                // ... and its register fragments are consumed
#pragma unroll
                for (int q = 0; q < PLAIN; ++q) { asm volatile("" : "+v"(pl[(u + 1) & 1][q])); a[q & 3] += pl[(u + 1) & 1][q][0] * 1e-30f; }
            }
            if (BAR) __builtin_amdgcn_s_barrier();
        }
        float s = 0.f;
        for (int k = 0; k < 5; ++k) s += c[k][0] + c[k][3];
        out[(size_t)blockIdx.x * blockDim.x + tid] = s;
        unsigned long long t_end, c_end;
        asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end), "=s"(c_end));
        if (lane == 0) { stamps[2 * (blockIdx.x * (NMW + NLW) + wave)] = t_begin; stamps[2 * (blockIdx.x * (NMW + NLW) + wave) + 1] = t_end; }
        if (lane == 0 && blockIdx.x == 0 && wave == 0) { stamps[2 * 4096 * 16 - 2] = c_end - c_begin; stamps[2 * 4096 * 16 - 1] = t_end - t_begin; }
    } else {
        if (LOAD > 0) {
            const int lw = wave - NMW;
            for (int it = 0; it < ksteps; ++it) {
                const unsigned slot = (unsigned)(it % RING) * PIECES * 1024u;
#pragma unroll
                for (int q = 0; q < LOAD; ++q) {
                    const int piece = NMW * SELF + lw * LOAD + q;
                    const bool isA = piece * 5 < PIECES * 2;
                    const int rowbase = isA ? mt * 16 * NMW : nt * 80 - (PIECES * 2 / 5) * 8;
                    dma16(voff(piece, it) + (unsigned)(rowbase * KDIM * 4), lds0 + slot + (unsigned)piece * 1024u, isA ? ra : rw);
                }
                vm_wait<LOAD>();
                if (BAR) __builtin_amdgcn_s_barrier();
            }
        }
        out[(size_t)blockIdx.x * blockDim.x + tid] = 0.f;
    }
}

static double waves_cu_f(int blocks, int nmw) { return (double)blocks * nmw / 256.0; }
#include <map>
#include <vector>
template <int NMW, int NLW, int SELF, int LOAD, bool BAR, int FORCE_KB = 0, int UNROLL = 1, int DELAY = 0, int PAD = 0, int PLAIN = 0>
static void run(const float* A, const float* W, float* out, int blocks_override = 0) {
    const int blocks = blocks_override ? blocks_override : (NMW == 4 ? 512 : 256), ksteps = 400, WV = NMW + NLW;
    static unsigned* where = nullptr;
    static unsigned long long* stamps = nullptr;
    if (!where) { hipMalloc(&where, 4096 * 16 * 4); hipMalloc(&stamps, 4096 * 16 * 16 + 64); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<NMW, NLW, SELF, LOAD, BAR, FORCE_KB, UNROLL, DELAY, PAD, PLAIN><<<blocks, 64 * WV>>>(A, W, out, ksteps, where, stamps);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        kern<NMW, NLW, SELF, LOAD, BAR, FORCE_KB, UNROLL, DELAY, PAD, PLAIN><<<blocks, 64 * WV>>>(A, W, out, ksteps, where, stamps);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    std::vector<unsigned> h((size_t)blocks * WV);
    hipMemcpy(h.data(), where, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<unsigned, int> per_cu, per_simd, mf_simd;      // blocks per CU, waves per SIMD, MFMA waves per SIMD
    for (int b = 0; b < blocks; ++b) {
        per_cu[h[(size_t)b * WV] >> 8]++;
        for (int w = 0; w < WV; ++w) {
            const unsigned key = h[(size_t)b * WV + w] >> 4;
            per_simd[key]++;
            if (w < NMW) mf_simd[key]++;
        }
    }
    // do the MFMA waves that share a SIMD run at the same time?  overlap of their [begin, end] intervals (100 MHz ticks)
    std::vector<unsigned long long> st((size_t)blocks * WV * 2);
    hipMemcpy(st.data(), stamps, st.size() * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::vector<int>> on_simd;
    for (int b = 0; b < blocks; ++b) for (int w = 0; w < NMW; ++w) on_simd[h[(size_t)b * WV + w] >> 4].push_back(b * WV + w);
    double ov = 0, len = 0; int pairs = 0;
    for (auto& kv : on_simd) if (kv.second.size() == 2) {
        const int i = kv.second[0], j = kv.second[1];
        const double b0 = (double)st[2 * i], e0 = (double)st[2 * i + 1], b1 = (double)st[2 * j], e1 = (double)st[2 * j + 1];
        const double o = (e0 < e1 ? e0 : e1) - (b0 > b1 ? b0 : b1);
        ov += o > 0 ? o : 0; len += 0.5 * ((e0 - b0) + (e1 - b1)); ++pairs;
    }
    int hist_cu[9] = {0}, hist_simd[17] = {0}, hist_mf[17] = {0};
    for (auto& kv : per_cu) hist_cu[kv.second > 8 ? 8 : kv.second]++;
    for (auto& kv : per_simd) hist_simd[kv.second > 16 ? 16 : kv.second]++;
    for (auto& kv : mf_simd) hist_mf[kv.second > 16 ? 16 : kv.second]++;
    const double us_step = best * 1e3 / ksteps;
    const double waves_cu = (double)blocks * NMW / 256.0;          // MFMA waves per CU
    const double ideal = 40.0 * 32.0 * (waves_cu / 4.0) / 2400.0;      // us per K-step at 2.4 GHz
    const double pieces_cu = (double)blocks * (NMW * SELF + NLW * LOAD) / 256.0;
    printf("%d blocks, unroll %d, pad %d, plain %d: mfma waves/block %d  loader waves %d  self %d  load %d  barrier %d : %.3f us per K-step (pipe floor %.3f = %.2f)  %.0f pieces per CU per K-step  LDS %d KB/block  err %d\n",
           blocks, UNROLL, PAD, PLAIN, NMW, NLW, SELF, LOAD, (int)BAR, us_step, ideal, ideal / us_step, pieces_cu, (NMW * SELF + NLW * LOAD) * 3 > FORCE_KB ? (NMW * SELF + NLW * LOAD) * 3 : FORCE_KB, (int)hipGetLastError());
    {
        unsigned long long cal[2];
        hipMemcpy(cal, stamps + 2 * 4096 * 16 - 2, 16, hipMemcpyDeviceToHost);
        printf("     s_memtime: %llu ticks in %.1f us of s_memrealtime (100 MHz) -> %.3f GHz; per K-step %.0f ticks (the MFMAs of a SIMD need %.0f cycles)\n", cal[0], cal[1] / 100.0,
               cal[0] / (cal[1] * 10.0), (double)cal[0] / ksteps, 40.0 * 32.0 * waves_cu_f(blocks, NMW) / 4.0);
    }
    if (pairs) printf("     SIMD pairs %d: mean wave lifetime %.1f us, mean overlap of the two waves %.1f us\n", pairs, len / pairs / 100.0, ov / pairs / 100.0);
    printf("     CUs seen %d; CUs with k blocks:", (int)per_cu.size());
    for (int k = 1; k <= 8; ++k) if (hist_cu[k]) printf(" %d:%d", k, hist_cu[k]);
    printf("   SIMDs with k waves:");
    for (int k = 1; k <= 16; ++k) if (hist_simd[k]) printf(" %d:%d", k, hist_simd[k]);
    printf("   SIMDs with k MFMA waves:");
    for (int k = 1; k <= 16; ++k) if (hist_mf[k]) printf(" %d:%d", k, hist_mf[k]);
    printf("\n");
    fflush(stdout);
}

int main(int argc, char** argv) {
    float *A, *W, *out;
    hipMalloc(&A, (size_t)NROWS_A * KDIM * 4); hipMalloc(&W, (size_t)NROWS_B * KDIM * 4); hipMalloc(&out, (size_t)512 * 1024 * 4);
    hipMemset(A, 0, (size_t)NROWS_A * KDIM * 4); hipMemset(W, 0, (size_t)NROWS_B * KDIM * 4);
    if (argc > 4) {        // A fragments straight to registers: 2 of the 5 pieces per wave as plain fragment-shaped loads (unroll 2: static register sets)
        run<4, 0, 0, 0, true, 60, 2>(A, W, out);
        run<4, 0, 5, 0, true, 0, 2>(A, W, out);
        run<4, 0, 5, 0, true, 0, 2, 0, 0, 2>(A, W, out);
        run<4, 0, 3, 0, true, 0, 2>(A, W, out);
        run<4, 0, 2, 0, true, 0, 2, 0, 0, 2>(A, W, out);
        run<4, 0, 5, 0, true, 0, 2, 0, 0, 5>(A, W, out);
        return 0;
    }
    if (argc > 3) {        // the same loops shifted by 4 bytes
        run<4, 0, 0, 0, false, 60, 1, 0, 0>(A, W, out);
        run<4, 0, 0, 0, false, 60, 1, 0, 1>(A, W, out);
        run<4, 0, 0, 0, false, 120, 1, 0, 0>(A, W, out, 256);
        run<4, 0, 0, 0, false, 120, 1, 0, 1>(A, W, out, 256);
        run<8, 0, 0, 0, false, 0, 1, 0, 0>(A, W, out);
        run<8, 0, 0, 0, false, 0, 1, 0, 1>(A, W, out);
        run<4, 0, 5, 0, true, 0, 1, 0, 0>(A, W, out);
        run<4, 0, 5, 0, true, 0, 1, 0, 1>(A, W, out);
        run<8, 0, 4, 0, true, 0, 1, 0, 0>(A, W, out);
        run<8, 0, 4, 0, true, 0, 1, 0, 1>(A, W, out);
        return 0;
    }
    if (argc > 2) {        // lockstep: the second half of an 8-wave block's MFMA waves starts late; and who overlaps whom
        run<4, 0, 0, 0, false, 60, 1>(A, W, out);
        run<8, 0, 0, 0, false, 0, 1>(A, W, out);
        run<8, 0, 0, 0, false, 0, 1, 3>(A, W, out);
        run<8, 0, 0, 0, false, 0, 1, 50>(A, W, out);
        run<8, 0, 4, 0, true, 0, 1>(A, W, out);
        run<8, 0, 4, 0, false, 0, 1, 50>(A, W, out);
        return 0;
    }
    if (argc > 1) {        // the taken branch per K-step: bare MFMA loops with 40 / 80 / 160 MFMAs per branch
        run<4, 0, 0, 0, false, 60, 1>(A, W, out);
        run<4, 0, 0, 0, false, 60, 2>(A, W, out);
        run<4, 0, 0, 0, false, 60, 4>(A, W, out);
        run<4, 0, 0, 0, false, 120, 1>(A, W, out, 256);
        run<4, 0, 0, 0, false, 120, 4>(A, W, out, 256);
        run<4, 0, 5, 0, true, 0, 1>(A, W, out);
        run<4, 0, 5, 0, true, 0, 2>(A, W, out);
        run<4, 0, 5, 0, true, 0, 4>(A, W, out);
        run<8, 0, 0, 0, false, 0, 1>(A, W, out);
        run<8, 0, 0, 0, false, 0, 4>(A, W, out);
        run<8, 0, 4, 0, true, 0, 1>(A, W, out);
        run<8, 0, 4, 0, true, 0, 4>(A, W, out);
        return 0;
    }
    printf("-- 4 MFMA waves per block, 2 blocks per CU (today's 64x80 launch: 20 pieces per block)\n");
    run<4, 0, 0, 0, false>(A, W, out);
    run<4, 0, 0, 0, false, 60>(A, W, out);
    run<4, 0, 0, 0, true, 60>(A, W, out);
    run<4, 0, 0, 0, false, 120>(A, W, out, 256);
    run<4, 0, 5, 0, false>(A, W, out);
    run<4, 0, 5, 0, true>(A, W, out);
    run<4, 0, 4, 0, true>(A, W, out);
    run<4, 0, 2, 0, true>(A, W, out);
    run<4, 2, 0, 10, true>(A, W, out);
    run<4, 4, 0, 5, true>(A, W, out);
    printf("-- 8 MFMA waves per block, 1 block per CU (128x80: 26 pieces per block)\n");
    run<8, 0, 0, 0, false>(A, W, out);
    run<8, 0, 0, 0, true>(A, W, out);
    run<8, 0, 5, 0, true>(A, W, out);
    run<8, 0, 5, 0, false>(A, W, out);
    run<8, 0, 4, 0, true>(A, W, out);
    run<8, 0, 4, 0, false>(A, W, out);
    run<8, 0, 3, 0, true>(A, W, out);
    run<8, 2, 0, 13, true>(A, W, out);
    run<8, 4, 0, 7, true>(A, W, out);
    printf("-- 8 MFMA waves per block, 2 blocks per CU (4 MFMA waves per SIMD)\n");
    run<8, 0, 0, 0, false, 60>(A, W, out, 512);
    run<8, 0, 4, 0, true, 60>(A, W, out, 512);
    return 0;
}
