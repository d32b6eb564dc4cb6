// Probe: does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` write zeros to LDS (or leave it alone)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const float* src, int nbytes, float* out) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 256; i += 64) lds[i] = 7.0f;
    __syncthreads();
    v4i rsrc;
    rsrc.x = (int)(unsigned)(size_t)src;
    rsrc.y = (int)(unsigned)((size_t)src >> 32);
    rsrc.z = nbytes;
    rsrc.w = 0x00020000;
    // lanes 0..31 in range, lanes 32..47 past the end, lanes 48..63 "negative" offsets
    unsigned voff = lane < 32 ? lane * 16u : (lane < 48 ? (unsigned)nbytes + (lane - 32) * 16u : (unsigned)(-16 * (lane - 47)));
    unsigned ldsaddr = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)"
                 :: "v"(voff), "s"(ldsaddr), "s"(rsrc) : "memory", "m0");
    __syncthreads();
    for (int i = lane; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    float *src, *out; float h[256], hs[128];
    for (int i = 0; i < 128; ++i) hs[i] = 100.f + i;
    hipMalloc(&src, 4096); hipMalloc(&out, 1024);
    hipMemset(src, 0x7f, 4096);
    hipMemcpy(src, hs, 512, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(src, 512, out);
    hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; l += 8) printf("lane %2d: %g %g %g %g\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    printf("lane 31: %g, lane 32: %g, lane 47: %g, lane 48: %g lane 63: %g\n", h[4*31], h[4*32], h[4*47], h[4*48], h[4*63]);
    return 0;
}
