// Probe: SATURATED issue cost of one wave64 instruction on a gfx950 SIMD, per instruction class: kernel time (HIP events) of
// W waves per SIMD x N instructions, W = 4 and 8 — cycles per instruction per SIMD = time x clock x SIMDs / (waves x N).
// (tools/probes/issue_probe.hip reads s_memtime inside one wave: under the oldest-first arbitration the first wave of a SIMD
//  runs nearly unimpeded whatever its neighbours do, which hides what an instruction costs the SIMD.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate_probe tools/probes/valu_rate_probe.hip && /tmp/valu_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define R8(x) x x x x x x x x
#define BODY8(I0, I1, I2, I3, I4, I5, I6, I7) R8(R8(I0 I1 I2 I3 I4 I5 I6 I7))   /* 512 instructions */

#define KERNEL(NAME, ASM, CLOB...)                                                                                   \
    __global__ void __launch_bounds__(256) NAME(float *out, int reps) {                                               \
        float a = threadIdx.x * 0.001f + 1.0f, b = a + 0.5f, c = a * 0.7f, d = b * 0.3f, e = 1.1f * a, f = 0.9f * b, g = c + d, h = e - f; \
        for (int r = 0; r < reps; r++) {                                                                              \
            asm volatile(ASM : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::CLOB);          \
        }                                                                                                             \
        out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;                                          \
    }

KERNEL(k_fma, BODY8("v_fma_f32 %0, %0, %1, %0\n", "v_fma_f32 %1, %1, %2, %1\n", "v_fma_f32 %2, %2, %3, %2\n", "v_fma_f32 %3, %3, %4, %3\n",
                    "v_fma_f32 %4, %4, %5, %4\n", "v_fma_f32 %5, %5, %6, %5\n", "v_fma_f32 %6, %6, %7, %6\n", "v_fma_f32 %7, %7, %0, %7\n"), "memory")
KERNEL(k_mul, BODY8("v_mul_f32 %0, %0, %1\n", "v_mul_f32 %1, %1, %2\n", "v_mul_f32 %2, %2, %3\n", "v_mul_f32 %3, %3, %4\n",
                    "v_mul_f32 %4, %4, %5\n", "v_mul_f32 %5, %5, %6\n", "v_mul_f32 %6, %6, %7\n", "v_mul_f32 %7, %7, %0\n"), "memory")
KERNEL(k_fmac, BODY8("v_fmac_f32 %0, %1, %2\n", "v_fmac_f32 %1, %2, %3\n", "v_fmac_f32 %2, %3, %4\n", "v_fmac_f32 %3, %4, %5\n",
                     "v_fmac_f32 %4, %5, %6\n", "v_fmac_f32 %5, %6, %7\n", "v_fmac_f32 %6, %7, %0\n", "v_fmac_f32 %7, %0, %1\n"), "memory")
KERNEL(k_dpp_shr, BODY8("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n", "v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n",
                        "v_mul_f32_dpp %2, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n", "v_mul_f32_dpp %3, %3, %3 row_shr:2 row_mask:0xf bank_mask:0xf\n",
                        "v_add_f32_dpp %4, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xf\n", "v_add_f32_dpp %5, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xf\n",
                        "v_add_f32_dpp %6, %6, %6 row_shr:8 row_mask:0xf bank_mask:0xf\n", "v_add_f32_dpp %7, %7, %7 row_shr:8 row_mask:0xf bank_mask:0xf\n"), "memory")
KERNEL(k_dpp_bcast, BODY8("v_fmac_f32_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n", "v_fmac_f32_dpp %1, %2, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf\n",
                          "v_mul_f32_dpp %2, %3, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf\n", "v_subrev_f32_dpp %3, %4, %5 row_newbcast:9 row_mask:0xf bank_mask:0xf\n",
                          "v_fmac_f32_dpp %4, %5, %6 row_newbcast:12 row_mask:0xf bank_mask:0xf\n", "v_add_f32_dpp %5, %6, %7 row_newbcast:15 row_mask:0xf bank_mask:0xf\n",
                          "v_mul_f32_dpp %6, %7, %0 row_newbcast:0 row_mask:0xf bank_mask:0xf\n", "v_fmac_f32_dpp %7, %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf\n"), "memory")
KERNEL(k_mov_dpp, BODY8("v_mov_b32_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n", "v_mov_b32_dpp %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf\n",
                        "v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n", "v_mov_b32_dpp %3, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n",
                        "v_mov_b32_dpp %4, %5 row_newbcast:12 row_mask:0xf bank_mask:0xf\n", "v_mov_b32_dpp %5, %6 row_ror:3 row_mask:0xf bank_mask:0xf\n",
                        "v_mov_b32_dpp %6, %7 row_newbcast:0 row_mask:0xf bank_mask:0xf\n", "v_mov_b32_dpp %7, %0 row_mirror row_mask:0xf bank_mask:0xf\n"), "memory")
KERNEL(k_exp, BODY8("v_exp_f32 %0, %0\n", "v_exp_f32 %1, %1\n", "v_exp_f32 %2, %2\n", "v_exp_f32 %3, %3\n", "v_exp_f32 %4, %4\n", "v_exp_f32 %5, %5\n",
                    "v_exp_f32 %6, %6\n", "v_exp_f32 %7, %7\n"), "memory")
KERNEL(k_rcp, BODY8("v_rcp_f32 %0, %0\n", "v_rcp_f32 %1, %1\n", "v_rcp_f32 %2, %2\n", "v_rcp_f32 %3, %3\n", "v_rcp_f32 %4, %4\n", "v_rcp_f32 %5, %5\n",
                    "v_rcp_f32 %6, %6\n", "v_rcp_f32 %7, %7\n"), "memory")
KERNEL(k_cnd_vcc, BODY8("v_cndmask_b32 %0, %0, %1, vcc\n", "v_cndmask_b32 %1, %1, %2, vcc\n", "v_cndmask_b32 %2, %2, %3, vcc\n", "v_cndmask_b32 %3, %3, %4, vcc\n",
                        "v_cndmask_b32 %4, %4, %5, vcc\n", "v_cndmask_b32 %5, %5, %6, vcc\n", "v_cndmask_b32 %6, %6, %7, vcc\n", "v_cndmask_b32 %7, %7, %0, vcc\n"), "vcc")
KERNEL(k_cmp_sgpr, BODY8("v_cmp_lt_f32 s[40:41], %0, %1\n", "v_cmp_lt_f32 s[42:43], %1, %2\n", "v_cmp_lt_f32 s[44:45], %2, %3\n", "v_cmp_lt_f32 s[46:47], %3, %4\n",
                         "v_cmp_lt_f32 s[40:41], %4, %5\n", "v_cmp_lt_f32 s[42:43], %5, %6\n", "v_cmp_lt_f32 s[44:45], %6, %7\n", "v_cmp_lt_f32 s[46:47], %7, %0\n"),
       "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47")
KERNEL(k_cmp_vcc, BODY8("v_cmp_lt_f32 vcc, %0, %1\n", "v_cmp_lt_f32 vcc, %1, %2\n", "v_cmp_lt_f32 vcc, %2, %3\n", "v_cmp_lt_f32 vcc, %3, %4\n",
                        "v_cmp_lt_f32 vcc, %4, %5\n", "v_cmp_lt_f32 vcc, %5, %6\n", "v_cmp_lt_f32 vcc, %6, %7\n", "v_cmp_lt_f32 vcc, %7, %0\n"), "vcc")
KERNEL(k_cmp_cnd, BODY8("v_cmp_lt_f32 vcc, %0, %1\n", "v_cndmask_b32 %0, %0, %2, vcc\n", "v_cmp_lt_f32 vcc, %3, %4\n", "v_cndmask_b32 %3, %3, %5, vcc\n",
                        "v_cmp_lt_f32 vcc, %6, %7\n", "v_cndmask_b32 %6, %6, %1, vcc\n", "v_cmp_lt_f32 vcc, %2, %5\n", "v_cndmask_b32 %2, %2, %4, vcc\n"), "vcc")
KERNEL(k_cmp_and_cnd, BODY8("v_cmp_lt_f32 vcc, %0, %1\n", "v_cmp_lt_f32 s[40:41], %1, %2\n", "s_and_b64 vcc, vcc, s[40:41]\n", "v_cndmask_b32 %0, %0, %2, vcc\n",
                            "v_cndmask_b32 %3, %3, %5, vcc\n", "v_cmp_lt_f32 vcc, %6, %7\n", "v_cmp_lt_f32 s[42:43], %2, %5\n", "v_cndmask_b32 %6, %6, %1, vcc\n"),
       "vcc", "s40", "s41", "s42", "s43")
KERNEL(k_min3, BODY8("v_min3_f32 %0, %0, %1, %2\n", "v_min3_f32 %1, %1, %2, %3\n", "v_med3_f32 %2, %2, %3, %4\n", "v_min3_f32 %3, %3, %4, %5\n",
                     "v_min3_f32 %4, %4, %5, %6\n", "v_med3_f32 %5, %5, %6, %7\n", "v_min3_f32 %6, %6, %7, %0\n", "v_min3_f32 %7, %7, %0, %1\n"), "memory")
KERNEL(k_fma_lit, BODY8("v_fmac_f32 %0, 0x40400000, %1\n", "v_fmac_f32 %1, 0x41100000, %2\n", "v_fmac_f32 %2, 0x40400000, %3\n", "v_fmac_f32 %3, 0x41100000, %4\n",
                        "v_mul_f32 %4, 0x3fb8aa3b, %5\n", "v_mul_f32 %5, 0x3fb8aa3b, %6\n", "v_fmac_f32 %6, 0x40400000, %7\n", "v_fmac_f32 %7, 0x41100000, %0\n"), "memory")
KERNEL(k_nop_mix, BODY8("v_fma_f32 %0, %0, %1, %0\n", "s_nop 1\n", "v_fma_f32 %2, %2, %3, %2\n", "v_fma_f32 %3, %3, %4, %3\n",
                        "v_fma_f32 %4, %4, %5, %4\n", "s_nop 0\n", "v_fma_f32 %6, %6, %7, %6\n", "v_fma_f32 %7, %7, %0, %7\n"), "memory")
KERNEL(k_salu_mix, BODY8("v_fma_f32 %0, %0, %1, %0\n", "s_mov_b64 vcc, s[40:41]\n", "v_fma_f32 %2, %2, %3, %2\n", "v_fma_f32 %3, %3, %4, %3\n",
                         "v_fma_f32 %4, %4, %5, %4\n", "s_and_b64 s[42:43], vcc, s[40:41]\n", "v_fma_f32 %6, %6, %7, %6\n", "v_fma_f32 %7, %7, %0, %7\n"),
       "vcc", "s40", "s41", "s42", "s43")
KERNEL(k_trans_mix, BODY8("v_fma_f32 %0, %0, %1, %0\n", "v_exp_f32 %1, %1\n", "v_fma_f32 %2, %2, %3, %2\n", "v_fma_f32 %3, %3, %4, %3\n",
                          "v_fma_f32 %4, %4, %5, %4\n", "v_fma_f32 %5, %5, %6, %5\n", "v_fma_f32 %6, %6, %7, %6\n", "v_fma_f32 %7, %7, %0, %7\n"), "memory")
KERNEL(k_dpp_mix, BODY8("v_fma_f32 %0, %0, %1, %0\n", "v_fmac_f32_dpp %1, %2, %3 row_newbcast:7 row_mask:0xf bank_mask:0xf\n", "v_fma_f32 %2, %2, %3, %2\n", "v_fma_f32 %3, %3, %4, %3\n",
                        "v_fma_f32 %4, %4, %5, %4\n", "v_mul_f32_dpp %5, %6, %7 row_newbcast:2 row_mask:0xf bank_mask:0xf\n", "v_fma_f32 %6, %6, %7, %6\n", "v_fma_f32 %7, %7, %0, %7\n"), "memory")

typedef void (*kern_t)(float *, int);
static double run(kern_t k, int wgs_per_cu, int reps) {
    const int blocks = 256 * wgs_per_cu;
    float *out; (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, 4);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}
int main() {
    struct { const char *name; kern_t k; } ks[] = {
        {"v_fma_f32 (VOP3)", k_fma}, {"v_mul_f32 (VOP2)", k_mul}, {"v_fmac_f32 (VOP2)", k_fmac}, {"DPP row_shr mul/add", k_dpp_shr},
        {"DPP row_newbcast fmac/mul/sub", k_dpp_bcast}, {"v_mov_b32_dpp (mixed controls)", k_mov_dpp}, {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp},
        {"v_cndmask_b32 vcc", k_cnd_vcc}, {"v_cmp -> sgpr pair", k_cmp_sgpr}, {"v_cmp -> vcc", k_cmp_vcc}, {"v_cmp + v_cndmask pairs", k_cmp_cnd},
        {"2 cmp + s_and + 2 cndmask (8 = 5 valu + 1 salu + 2)", k_cmp_and_cnd}, {"v_min3 / v_med3", k_min3}, {"VOP2 with 32-bit literal", k_fma_lit},
        {"6 fma + s_nop 1 + s_nop 0", k_nop_mix}, {"6 fma + 2 salu", k_salu_mix}, {"7 fma + 1 exp", k_trans_mix}, {"6 fma + 2 dpp bcast", k_dpp_mix}};
    const int reps = 200;   // x 512 instructions
    setvbuf(stdout, nullptr, _IONBF, 0);
    printf("%-52s %12s %12s %12s   (ns per wave-instruction per SIMD; x clock GHz = cycles)\n", "instruction class", "1 wave/SIMD", "4 waves", "8 waves");
    for (auto &e : ks) {
        double t[3]; int w[3] = {1, 4, 8};
        for (int i = 0; i < 3; i++) t[i] = run(e.k, w[i], reps) * 1e6 / ((double)w[i] * reps * 512.0);
        printf("%-52s %12.3f %12.3f %12.3f\n", e.name, t[0], t[1], t[2]);
        fflush(stdout);
    }
    return 0;
}
