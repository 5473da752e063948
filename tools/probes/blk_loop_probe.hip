// Probe: the pair loop of render_bwd_blk.hip in isolation — no LDS, no global memory, synthetic splats — at 1..6 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -I das3r_amd/csrc -o /tmp/blk_loop_probe tools/probes/blk_loop_probe.hip && /tmp/blk_loop_probe
// Prints SIMD cycles per pixel step (64 pairs) = elapsed cycles x SIMDs / (waves x batches x 16): what the loop costs when nothing
// but the vector ALU is in the way.  VARIANT 0: block_row as shipped; 1: without the two DPP row scans; 2: without the state hand-off;
// 3: plain independent v_fma of the same count (the issue-rate yardstick).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "render_blk.h"
using namespace das3r;

template <int VARIANT, int PIX = 0>
__global__ void __launch_bounds__(256) probe(float *out, int batches, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63, s = lane & 15, row = lane >> 4, wave = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) char lds_cst[16 * PIX_CST_ROW];
    __shared__ __attribute__((aligned(16))) char lds_st[16 * PIX_ST_ROW];
    char *const cst = lds_cst + (wave * 4 + row) * PIX_CST_ROW, *const st = lds_st + (wave * 4 + row) * PIX_ST_ROW;
    PixelRegs px;
    px.pxf = (float)(8 + (PIX ? 0 : (s & 3)) + 4 * (row & 1)); px.pyf = (float)(16 + (PIX ? 0 : (s >> 2)) + 4 * (row >> 1));
    px.d0 = 0.001f * (s + 1); px.d1 = -0.002f * (s + 2); px.d2 = 0.0015f * (s + 3);
    px.T = 0.3f + 0.01f * s; px.R = 0.0001f * s; px.lastrel = 128.f;
    if (PIX > 0) *reinterpret_cast<float4 *>(cst + s * 16) = make_float4(px.d0, px.d1, px.d2, px.lastrel);
    if (PIX == 2) *reinterpret_cast<float2 *>(st + s * 8) = make_float2(px.T, px.R);
    __syncthreads();
    Sums a = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int b = 0; b < batches; b++) {
        SplatRegs sp;
        sp.x = 9.3f + 0.37f * s + 0.01f * b; sp.y = 17.1f + 0.21f * s; sp.A = 0.21f; sp.B = 0.03f; sp.C = 0.17f; sp.o = 0.05f + 0.01f * s;
        sp.c0 = 0.3f; sp.c1 = 0.5f; sp.c2 = 0.7f; sp.posrel = (float)(s + (b & 63));
        if (VARIANT == 3) {
            float f0 = sp.x, f1 = sp.y, f2 = sp.A, f3 = sp.B, f4 = sp.C, f5 = sp.o, f6 = sp.c0, f7 = sp.c1;
#pragma unroll
            for (int i = 0; i < 100; i++) {   // 800 independent-ish fma per "batch"
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                             : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7));
            }
            a.C0 += f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
        } else {
            block_row<0, PIX, VARIANT>(sp, px, a, cst, st);
            block_row<1, PIX, VARIANT>(sp, px, a, cst, st);
            block_row<2, PIX, VARIANT>(sp, px, a, cst, st);
            block_row<3, PIX, VARIANT>(sp, px, a, cst, st);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 256 + threadIdx.x] = a.C0 + a.C1 + a.C2 + a.M0 + a.Mu + a.Mv + a.Muu + a.Muv + a.Mvv + px.T + px.R;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int V, int PIX = 0>
static void run(const char *what, int wgs_per_cu) {
    const int cus = 256, blocks = cus * wgs_per_cu, batches = 2000;
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256); (void)hipMalloc(&cyc, 8 * blocks);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<V, PIX><<<blocks, 256>>>(out, 10, cyc);
    (void)hipEventRecord(e0);
    probe<V, PIX><<<blocks, 256>>>(out, batches, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    // every SIMD hosts wgs_per_cu waves; each wave does `batches` x 16 steps
    const double steps_per_simd = (double)wgs_per_cu * batches * 16;
    // (the oldest wave of a SIMD runs nearly unimpeded whatever its neighbours do: block 0's own s_memtime is NOT the SIMD's cost)
    printf("%-34s %d waves/SIMD: %6.1f ns per step per SIMD   (kernel %.3f ms; block 0 alone: %.1f cycles per step)\n", what, wgs_per_cu,
           ms * 1e6 / steps_per_simd, ms, (double)h[0] / (batches * 16.0));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main() {
    for (int w : {1, 2, 4, 5, 6, 8}) run<0, 0>("block_row, registers + DPP", w);
    for (int w : {1, 2, 4, 5, 6, 8}) run<0, 1>("block_row, constants in LDS", w);
    for (int w : {1, 2, 4, 5, 6, 8}) run<0, 2>("block_row, constants + state LDS", w);
    for (int w : {5}) run<1, 0>("  registers, without the scans", w);
    for (int w : {5}) run<1, 2>("  LDS, without the scans", w);
    for (int w : {1, 2, 4, 5}) run<3, 0>("800 v_fma per batch", w);
    return 0;
}
