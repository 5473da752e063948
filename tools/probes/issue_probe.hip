// Probe: what one wave64 instruction costs on a gfx950 SIMD, and what overlaps with what (run on the GPU box).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/issue_probe tools/probes/issue_probe.hip && /tmp/issue_probe
// Every test runs REPS x 64 copies of one instruction pattern in a wave and reads s_memtime around it.  Rows:
//   1 wave/SIMD  : one 256-thread workgroup (a wave on each SIMD), cycles per instruction of wave 0
//   2, 4 /SIMD   : 512 / 1024 threads: the waves of a SIMD share its issue port — cycles per instruction PER SIMD (time / all
//                  instructions issued on that SIMD) tell whether another wave fills the gaps
//   mixes        : MFMA + VALU in one wave, and an MFMA-only wave beside a VALU-only wave on the same SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>

#define REPS 200
typedef float v4f __attribute__((ext_vector_type(4)));
typedef short v4s __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned long long now() { return __builtin_amdgcn_s_memtime(); }

#define R8(x) x x x x x x x x
#define R64(x) R8(R8(x))

enum { T_FMA, T_MUL, T_ADD, T_CNDMASK, T_CMP, T_DPP_MUL, T_DPP_ADD_CHAIN, T_EXP, T_RCP, T_MOV, T_MFMA_F32, T_MFMA_F32_DEP, T_MFMA_BF16,
       T_MIX_F32, T_MIX_BF16, T_PERMLANE, T_FMA_DEP, T_CND_SGPR, T_CND_SPARSE, T_BFI, T_AND, T_ASHR, T_MED3, T_MAX, T_CMP_VCC,
       T_CMP_CND, T_SIGNSEL, T_LDS_ADD64, T_LDS_ADD12, T_LDS_WRITE, T_LDS_READ128, T_SWIZZLE, T_MFMA_BF16_K32, T_BPERMUTE, T_COUNT };

template <int T>
__device__ __forceinline__ void body(float &a, float &b, float &c, float &d, float &e, float &f, float &g, float &h, v4f &m0, v4f &m1,
                                     v4f &m2, v4f &m3) {
    if constexpr (T == T_FMA) {
        asm volatile(R8("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                        "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_FMA_DEP) {
        asm volatile(R64("v_fma_f32 %0, %0, %0, %0\n") : "+v"(a));
    } else if constexpr (T == T_MUL) {
        asm volatile(R8("v_mul_f32 %0, %0, %0\n v_mul_f32 %1, %1, %1\n v_mul_f32 %2, %2, %2\n v_mul_f32 %3, %3, %3\n"
                        "v_mul_f32 %4, %4, %4\n v_mul_f32 %5, %5, %5\n v_mul_f32 %6, %6, %6\n v_mul_f32 %7, %7, %7\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_ADD) {
        asm volatile(R8("v_add_f32 %0, %0, %0\n v_add_f32 %1, %1, %1\n v_add_f32 %2, %2, %2\n v_add_f32 %3, %3, %3\n"
                        "v_add_f32 %4, %4, %4\n v_add_f32 %5, %5, %5\n v_add_f32 %6, %6, %6\n v_add_f32 %7, %7, %7\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_CNDMASK) {
        asm volatile(R8("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n"
                        "v_cndmask_b32 %4, %4, %5, vcc\n v_cndmask_b32 %5, %5, %6, vcc\n v_cndmask_b32 %6, %6, %7, vcc\n v_cndmask_b32 %7, %7, %0, vcc\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"vcc");
    } else if constexpr (T == T_CMP) {
        asm volatile(R8("v_cmp_lt_f32 s[40:41], %0, %1\n v_cmp_lt_f32 s[42:43], %1, %2\n v_cmp_lt_f32 s[44:45], %2, %3\n v_cmp_lt_f32 s[46:47], %3, %4\n"
                        "v_cmp_lt_f32 s[40:41], %4, %5\n v_cmp_lt_f32 s[42:43], %5, %6\n v_cmp_lt_f32 s[44:45], %6, %7\n v_cmp_lt_f32 s[46:47], %7, %0\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47");
    } else if constexpr (T == T_DPP_MUL) {   // eight independent chains: the 2-wait-state DPP hazard is covered
        asm volatile(R8("v_mul_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mul_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mul_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                        "v_mul_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mul_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_DPP_ADD_CHAIN) {   // one dependent chain with the s_nop the hazard needs
        asm volatile(R64("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a));
    } else if constexpr (T == T_EXP) {
        asm volatile(R8("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_RCP) {
        asm volatile(R8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_MOV) {
        asm volatile(R8("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_PERMLANE) {
        asm volatile(R8("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                        "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_CND_SGPR) {
        asm volatile("s_mov_b64 s[40:41], 0x5555\n" R8("v_cndmask_b32_e64 %0, %0, %1, s[40:41]\n v_cndmask_b32_e64 %1, %1, %2, s[40:41]\n v_cndmask_b32_e64 %2, %2, %3, s[40:41]\n"
                        "v_cndmask_b32_e64 %3, %3, %4, s[40:41]\n v_cndmask_b32_e64 %4, %4, %5, s[40:41]\n v_cndmask_b32_e64 %5, %5, %6, s[40:41]\n"
                        "v_cndmask_b32_e64 %6, %6, %7, s[40:41]\n v_cndmask_b32_e64 %7, %7, %0, s[40:41]\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"s40", "s41");
    } else if constexpr (T == T_CND_SPARSE) {   // one select among seven v_fma
        asm volatile(R8("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_cndmask_b32 %7, %7, %3, vcc\n"
                        "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %3, %3, %3, %3\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"vcc");
    } else if constexpr (T == T_BFI) {
        asm volatile(R8("v_bfi_b32 %0, %0, %1, %2\n v_bfi_b32 %1, %1, %2, %3\n v_bfi_b32 %2, %2, %3, %4\n v_bfi_b32 %3, %3, %4, %5\n"
                        "v_bfi_b32 %4, %4, %5, %6\n v_bfi_b32 %5, %5, %6, %7\n v_bfi_b32 %6, %6, %7, %0\n v_bfi_b32 %7, %7, %0, %1\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_AND) {
        asm volatile(R8("v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %4\n"
                        "v_and_b32 %4, %4, %5\n v_and_b32 %5, %5, %6\n v_and_b32 %6, %6, %7\n v_and_b32 %7, %7, %0\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_ASHR) {
        asm volatile(R8("v_ashrrev_i32 %0, 31, %0\n v_ashrrev_i32 %1, 31, %1\n v_ashrrev_i32 %2, 31, %2\n v_ashrrev_i32 %3, 31, %3\n"
                        "v_ashrrev_i32 %4, 31, %4\n v_ashrrev_i32 %5, 31, %5\n v_ashrrev_i32 %6, 31, %6\n v_ashrrev_i32 %7, 31, %7\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_MED3) {
        asm volatile(R8("v_med3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_med3_f32 %2, %2, %3, %4\n v_med3_f32 %3, %3, %4, %5\n"
                        "v_med3_f32 %4, %4, %5, %6\n v_med3_f32 %5, %5, %6, %7\n v_med3_f32 %6, %6, %7, %0\n v_med3_f32 %7, %7, %0, %1\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_MAX) {
        asm volatile(R8("v_max_f32 %0, %0, %1\n v_min_f32 %1, %1, %2\n v_max_f32 %2, %2, %3\n v_min_f32 %3, %3, %4\n"
                        "v_max_f32 %4, %4, %5\n v_min_f32 %5, %5, %6\n v_max_f32 %6, %6, %7\n v_min_f32 %7, %7, %0\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_CMP_VCC) {
        asm volatile(R8("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %4\n"
                        "v_cmp_lt_f32 vcc, %4, %5\n v_cmp_lt_f32 vcc, %5, %6\n v_cmp_lt_f32 vcc, %6, %7\n v_cmp_lt_f32 vcc, %7, %0\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"vcc");
    } else if constexpr (T == T_CMP_CND) {   // compare + select pairs (what a ?: on floats compiles to), four independent pairs
        asm volatile(R8("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc\n v_cmp_lt_f32 vcc, %3, %4\n v_cndmask_b32 %3, %3, %5, vcc\n"
                        "v_cmp_lt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %1, vcc\n v_cmp_lt_f32 vcc, %2, %5\n v_cndmask_b32 %2, %2, %4, vcc\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h)::"vcc");
    } else if constexpr (T == T_SIGNSEL) {   // the same selects on the vector ALU alone: sub, arithmetic shift, and (3 instructions per select; 64 = 21 selects + 1)
        asm volatile(R8("v_sub_f32 %7, %0, %1\n v_ashrrev_i32 %7, 31, %7\n v_and_b32 %0, %0, %7\n v_sub_f32 %6, %3, %4\n v_ashrrev_i32 %6, 31, %6\n v_and_b32 %3, %3, %6\n"
                        "v_sub_f32 %5, %2, %1\n v_ashrrev_i32 %5, 31, %5\n")
                     : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
    } else if constexpr (T == T_LDS_ADD64 || T == T_LDS_ADD12 || T == T_LDS_WRITE || T == T_LDS_READ128 || T == T_SWIZZLE || T == T_BPERMUTE) {
        extern __shared__ float lds_probe[];
        const int lane = threadIdx.x & 63;
        float *p = lds_probe + (threadIdx.x >> 6) * 1024 + lane * 9;   // 36-byte rows, as the backward kernel's accumulators
        if constexpr (T == T_LDS_ADD64) {
#pragma unroll
            for (int i = 0; i < 64; i++) __hip_atomic_fetch_add(p + (i & 7), a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if constexpr (T == T_LDS_ADD12) {
            if (lane < 12) {
#pragma unroll
                for (int i = 0; i < 64; i++) __hip_atomic_fetch_add(p + (i & 7), a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        } else if constexpr (T == T_LDS_WRITE) {
#pragma unroll
            for (int i = 0; i < 64; i++) { p[i & 7] = a; asm volatile("" ::: "memory"); }
        } else if constexpr (T == T_LDS_READ128) {
            const float4 *q = reinterpret_cast<const float4 *>(lds_probe + (threadIdx.x >> 6) * 1024) + (lane >> 4) * 2;   // 4 addresses per instruction
            float4 acc4 = {0, 0, 0, 0};
#pragma unroll
            for (int i = 0; i < 64; i++) { const float4 v = q[(i & 7) * 8]; acc4.x += v.x; asm volatile("" ::: "memory"); }
            a += acc4.x;
        } else if constexpr (T == T_SWIZZLE) {
#pragma unroll
            for (int i = 0; i < 64; i++) a = __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(a), 0x01EF));   // and 0x0F... pattern: broadcast within 32
        } else {
#pragma unroll
            for (int i = 0; i < 64; i++) a = __int_as_float(__builtin_amdgcn_ds_bpermute((lane ^ 1) << 2, __float_as_int(a)));
        }
    } else if constexpr (T == T_MFMA_BF16_K32) {
        typedef short v8s __attribute__((ext_vector_type(8)));
        typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
        v8bf x, y;
        for (int i = 0; i < 8; i++) { x[i] = (__bf16)(a + i); y[i] = (__bf16)(b - i); }
#pragma unroll
        for (int i = 0; i < 16; i++) {
            m0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, y, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(x, x, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, y, m3, 0, 0, 0);
        }
    } else if constexpr (T == T_MFMA_F32) {   // 64 MFMAs on four independent accumulators
#pragma unroll
        for (int i = 0; i < 16; i++) {
            m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, d, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(e, f, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(g, h, m3, 0, 0, 0);
        }
    } else if constexpr (T == T_MFMA_F32_DEP) {   // one accumulator chain
#pragma unroll
        for (int i = 0; i < 64; i++) m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m0, 0, 0, 0);
    } else if constexpr (T == T_MFMA_BF16) {
        v4s x = {(short)__float_as_uint(a), 1, 2, 3}, y = {(short)__float_as_uint(b), 3, 2, 1};
#pragma unroll
        for (int i = 0; i < 16; i++) {
            m0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y, x, m1, 0, 0, 0);
            m2 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, x, m2, 0, 0, 0);
            m3 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y, y, m3, 0, 0, 0);
        }
    } else if constexpr (T == T_MIX_F32) {   // per repetition: 1 f32 MFMA + 8 independent v_fma (64 x: 8 MFMA + 64 fma ... x 8)
#pragma unroll
        for (int i = 0; i < 8; i++) {
            m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, m0, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h));
            m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(c, d, m1, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        }
    } else if constexpr (T == T_MIX_BF16) {
        v4s x = {(short)__float_as_uint(a), 1, 2, 3}, y = {(short)__float_as_uint(b), 3, 2, 1};
#pragma unroll
        for (int i = 0; i < 8; i++) {
            m0 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(x, y, m0, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h));
            m1 = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(y, x, m1, 0, 0, 0);
            asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                         : "+v"(e), "+v"(f), "+v"(g), "+v"(h));
        }
    }
}

// role of a wave: test TA for waves whose (wave index / 4) is even, TB for the others (the two share a SIMD in a 512-thread block)
template <int TA, int TB>
__global__ void __launch_bounds__(1024) probe(unsigned long long *cycles, float *sink) {
    const int wave = threadIdx.x >> 6;
    float a = 1.0f + threadIdx.x * 1e-9f, b = 0.999f, c = 1.0001f, d = 0.9999f, e = 1.0f, f = 0.5f, g = 0.25f, h = 0.75f;
    v4f m0 = {0, 0, 0, 0}, m1 = m0, m2 = m0, m3 = m0;
    __syncthreads();
    const unsigned long long t0 = now();
    if (((wave >> 2) & 1) == 0) {
        for (int r = 0; r < REPS; r++) body<TA>(a, b, c, d, e, f, g, h, m0, m1, m2, m3);
    } else {
        for (int r = 0; r < REPS; r++) body<TB>(a, b, c, d, e, f, g, h, m0, m1, m2, m3);
    }
    const unsigned long long t1 = now();
    if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 16 + wave] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = a + b + c + d + e + f + g + h + m0[0] + m1[1] + m2[2] + m3[3];
}

struct Row { const char *name; int ta, tb, insts_a, insts_b; };

template <int TA, int TB>
static void run(const char *name, int insts_a, int insts_b, unsigned long long *dc, float *ds) {
    for (int threads : {256, 512, 1024}) {
        if (TA != TB && threads == 256) continue;
        probe<TA, TB><<<1, threads, 16 * 4096>>>(dc, ds);
        hipDeviceSynchronize();
        unsigned long long h[16];
        hipMemcpy(h, dc, sizeof(h), hipMemcpyDeviceToHost);
        const int waves = threads / 64;
        // waves w, w + 4, w + 8, ... sit on one SIMD (observed placement; the numbers below tell if that holds)
        double worst_a = 0, worst_b = 0;
        for (int w = 0; w < waves; w++) {
            if (((w >> 2) & 1) == 0) worst_a = worst_a > (double)h[w] ? worst_a : (double)h[w];
            else worst_b = worst_b > (double)h[w] ? worst_b : (double)h[w];
        }
        const double ia = (double)REPS * insts_a, ib = (double)REPS * insts_b;
        if (TA == TB) {
            const double per_simd_insts = ia * (waves / 4);
            printf("%-28s %2d wave(s)/SIMD: %7.2f cycles per instruction of a wave, %6.2f per instruction on the SIMD\n", name, waves / 4,
                   worst_a / ia, worst_a / per_simd_insts);
        } else {
            printf("%-28s %2d wave(s)/SIMD: A-waves %7.2f cycles per their instruction, B-waves %7.2f (alone: see rows above)\n", name, waves / 4,
                   worst_a / ia, worst_b / ib);
        }
    }
}

int main() {
    unsigned long long *dc;
    float *ds;
    hipMalloc(&dc, 16 * sizeof(unsigned long long));
    hipMalloc(&ds, 1024 * sizeof(float));
    printf("s_memtime ticks; REPS=%d x 64 instructions per wave (MFMA mixes: 16 MFMA + 128 v_fma per repetition)\n", REPS);
    run<T_FMA, T_FMA>("v_fma_f32 (8 chains)", 64, 64, dc, ds);
    run<T_FMA_DEP, T_FMA_DEP>("v_fma_f32 (1 dependent chain)", 64, 64, dc, ds);
    run<T_MUL, T_MUL>("v_mul_f32", 64, 64, dc, ds);
    run<T_ADD, T_ADD>("v_add_f32", 64, 64, dc, ds);
    run<T_MOV, T_MOV>("v_mov_b32", 64, 64, dc, ds);
    run<T_CNDMASK, T_CNDMASK>("v_cndmask_b32 (vcc)", 64, 64, dc, ds);
    run<T_CMP, T_CMP>("v_cmp_lt_f32 -> sgpr pair", 64, 64, dc, ds);
    run<T_DPP_MUL, T_DPP_MUL>("v_mul_f32_dpp row_shr (8 ch)", 64, 64, dc, ds);
    run<T_DPP_ADD_CHAIN, T_DPP_ADD_CHAIN>("s_nop 1 + v_add_f32_dpp chain", 64, 64, dc, ds);
    run<T_PERMLANE, T_PERMLANE>("v_permlane32/16_swap", 64, 64, dc, ds);
    run<T_EXP, T_EXP>("v_exp_f32", 64, 64, dc, ds);
    run<T_RCP, T_RCP>("v_rcp_f32", 64, 64, dc, ds);
    run<T_MFMA_F32, T_MFMA_F32>("mfma_f32_16x16x4_f32 (4 acc)", 64, 64, dc, ds);
    run<T_MFMA_F32_DEP, T_MFMA_F32_DEP>("mfma_f32_16x16x4_f32 (1 acc)", 64, 64, dc, ds);
    run<T_MFMA_BF16, T_MFMA_BF16>("mfma_f32_16x16x16_bf16 (4 acc)", 64, 64, dc, ds);
    run<T_CND_SGPR, T_CND_SGPR>("v_cndmask_b32_e64 (sgpr pair)", 64, 64, dc, ds);
    run<T_CND_SPARSE, T_CND_SPARSE>("7 v_fma + 1 v_cndmask", 64, 64, dc, ds);
    run<T_CMP_VCC, T_CMP_VCC>("v_cmp_lt_f32 -> vcc", 64, 64, dc, ds);
    run<T_CMP_CND, T_CMP_CND>("v_cmp + v_cndmask pairs", 64, 64, dc, ds);
    run<T_SIGNSEL, T_SIGNSEL>("sub + ashr + and selects", 64, 64, dc, ds);
    run<T_BFI, T_BFI>("v_bfi_b32", 64, 64, dc, ds);
    run<T_AND, T_AND>("v_and_b32", 64, 64, dc, ds);
    run<T_ASHR, T_ASHR>("v_ashrrev_i32", 64, 64, dc, ds);
    run<T_MED3, T_MED3>("v_med3_f32", 64, 64, dc, ds);
    run<T_MAX, T_MAX>("v_max_f32 / v_min_f32", 64, 64, dc, ds);
    run<T_LDS_ADD64, T_LDS_ADD64>("ds_add_f32, 64 lanes", 64, 64, dc, ds);
    run<T_LDS_ADD12, T_LDS_ADD12>("ds_add_f32, 12 lanes", 64, 64, dc, ds);
    run<T_LDS_WRITE, T_LDS_WRITE>("ds_write_b32, 64 lanes", 64, 64, dc, ds);
    run<T_LDS_READ128, T_LDS_READ128>("ds_read_b128, 4 addresses", 64, 64, dc, ds);
    run<T_SWIZZLE, T_SWIZZLE>("ds_swizzle_b32 (dependent)", 64, 64, dc, ds);
    run<T_BPERMUTE, T_BPERMUTE>("ds_bpermute_b32 (dependent)", 64, 64, dc, ds);
    run<T_MFMA_BF16_K32, T_MFMA_BF16_K32>("mfma_f32_16x16x32_bf16 (4 acc)", 64, 64, dc, ds);
    run<T_MFMA_BF16_K32, T_FMA>("A bf16 K=32 MFMA | B v_fma", 64, 64, dc, ds);
    printf("-- one wave issuing both: cycles per repetition-unit of (1 MFMA + 8 v_fma) = number below x 9\n");
    run<T_MIX_F32, T_MIX_F32>("1 f32 MFMA + 8 v_fma, same wave", 144, 144, dc, ds);
    run<T_MIX_BF16, T_MIX_BF16>("1 bf16 MFMA + 8 v_fma, same wave", 144, 144, dc, ds);
    printf("-- an MFMA-only wave (A) beside a v_fma-only wave (B) on the same SIMD\n");
    run<T_MFMA_F32, T_FMA>("A f32 MFMA | B v_fma", 64, 64, dc, ds);
    run<T_MFMA_BF16, T_FMA>("A bf16 MFMA | B v_fma", 64, 64, dc, ds);
    run<T_EXP, T_FMA>("A v_exp | B v_fma", 64, 64, dc, ds);
    return 0;
}
