#include <hip/hip_runtime.h>
#include <cstdio>
// Probe: row_newbcast DPP control (0x150 + lane) through the builtin, folded into VOP2 ops, and v_cndmask_b32_dpp with a VCC lane mask.
__global__ void probe(const float *in, float *out) {
    const int lane = threadIdx.x;
    float x = in[lane], y = in[64 + lane];
    // (1) broadcast lane 5 of every row
    float b5 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + 5, 0xf, 0xf, false));
    // (2) folded multiply: y * x[lane 3 of row]
    float m3 = y * __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + 3, 0xf, 0xf, false));
    // (3) cndmask_dpp: lane k=7 of each row takes x[lane 15 of its row], the others keep y
    float sel = y;
    const unsigned long long keep = ~(0x0001000100010001ull << 7);
    asm volatile("s_mov_b64 vcc, %2\n\ts_nop 1\n\tv_cndmask_b32_dpp %0, %1, %0, vcc row_newbcast:15 row_mask:0xf bank_mask:0xf" : "+v"(sel) : "v"(x), "s"(keep) : "vcc");
    // (4) v_subrev_f32_dpp: y - x[lane 2 of row]
    float sr;
    asm volatile("s_nop 1\n\tv_subrev_f32_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "=v"(sr) : "v"(x), "v"(y));
    // (5) fmac dpp: acc += x[bcast 9] * y
    float acc = 1.0f;
    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %1, %2 row_newbcast:9 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(x), "v"(y));
    out[lane] = b5; out[64 + lane] = m3; out[128 + lane] = sel; out[192 + lane] = sr; out[256 + lane] = acc;
}
int main() {
    float h[128], o[320];
    for (int i = 0; i < 64; i++) { h[i] = 100.f + i; h[64 + i] = 0.5f * i; }
    float *d, *r; hipMalloc(&d, sizeof(h)); hipMalloc(&r, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, 64>>>(d, r);
    hipMemcpy(o, r, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; l++) {
        const int row = l & ~15;
        if (o[l] != h[row + 5]) bad |= 1;
        if (o[64 + l] != h[64 + l] * h[row + 3]) bad |= 2;
        const float want = (l & 15) == 7 ? h[row + 15] : h[64 + l];
        if (o[128 + l] != want) bad |= 4;
        if (o[192 + l] != h[64 + l] - h[row + 2]) bad |= 8;
        if (o[256 + l] != 1.0f + h[row + 9] * h[64 + l]) bad |= 16;
    }
    printf("dpp_probe: %s (mask %d)  sample: b5[20]=%g m3[20]=%g sel[23]=%g sel[24]=%g sr[40]=%g acc[50]=%g\n", bad ? "MISMATCH" : "all five forms behave as assumed", bad, o[20], o[84], o[128 + 23], o[128 + 24], o[192 + 40], o[256 + 50]);
    return bad;
}
