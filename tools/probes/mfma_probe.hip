// Probe: operand / result layout of v_mfma_f32_16x16x4_f32 on gfx950 (run on the GPU box).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
__global__ void probe(const float *A /*[16][4]*/, const float *B /*[4][16]*/, float *D /*[16][16]*/, float *raw /*[64][4]*/) {
    const int l = threadIdx.x;
    // hypothesis: lane l supplies A[i = l % 16][k = l / 16] and B[k = l / 16][j = l % 16]
    const float a = A[(l % 16) * 4 + (l / 16)];
    const float b = B[(l / 16) * 16 + (l % 16)];
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    for (int r = 0; r < 4; r++) raw[l * 4 + r] = acc[r];
    // hypothesis: lane l holds D[i = 4 * (l / 16) + r][j = l % 16]
    for (int r = 0; r < 4; r++) D[(4 * (l / 16) + r) * 16 + (l % 16)] = acc[r];
}
int main() {
    float hA[64], hB[64], hD[256], ref[256], hraw[256];
    for (int i = 0; i < 64; i++) { hA[i] = (float)((i * 7) % 13) - 6.f; hB[i] = (float)((i * 5) % 11) - 5.f; }
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { float s = 0; for (int k = 0; k < 4; k++) s += hA[i * 4 + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    float *dA, *dB, *dD, *draw;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dD, 1024); hipMalloc(&draw, 1024);
    hipMemcpy(dA, hA, 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 256, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, dD, draw);
    hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost); hipMemcpy(hraw, draw, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 256; i++) if (hD[i] != ref[i]) bad++;
    printf("layout hypothesis mismatches: %d / 256\n", bad);
    if (bad) { for (int l = 0; l < 64; l += 9) printf("lane %d: %g %g %g %g\n", l, hraw[l * 4], hraw[l * 4 + 1], hraw[l * 4 + 2], hraw[l * 4 + 3]); }
    return 0;
}
