// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the access patterns of this library: the x2 correction of
// MI355X_MICROARCH.md is calibrated for wide coalesced streams only.  Three kernels with KNOWN HBM read volumes, every byte read
// exactly once from buffers far larger than the L2 + MALL:
//   calib_stream     float4 per lane, unit stride                                  (preprocess / radix passes)
//   calib_gather64   one 64-byte record per lane at a random record index,          (compositing kernels: a splat record per list entry;
//                    48 of its 64 bytes read as three float4                         render_fwd / render_bwd stage loads)
//   calib_gather4    one 4-byte word per lane at a random word index                (gid_of[slot], slot_list: the binning's gathers)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_calib tools/probes/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -o pmc --output-format csv -- /tmp/fetch_calib      (tools/fetch_calib.sh)
// The program prints the known bytes per kernel; tools/fetch_calib.py divides them by the counter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void calib_stream(const float4 *__restrict__ src, float *__restrict__ out, size_t n4) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = src[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void calib_gather64(const float4 *__restrict__ recs, const uint32_t *__restrict__ idx, float *__restrict__ out, size_t n) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t g = idx[i];
        const float4 a = recs[g * 4], b = recs[g * 4 + 1], c = recs[g * 4 + 2];
        acc += a.x + b.y + c.z;
    }
    if (acc == 123.456f) out[0] = acc;
}
__global__ void calib_gather4(const uint32_t *__restrict__ words, const uint32_t *__restrict__ idx, float *__restrict__ out, size_t n) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc += words[idx[i]];
    if (acc == 0x12345678u) out[0] = (float)acc;
}

int main() {
    const size_t stream_bytes = 1ull << 30;                       // 1 GiB streamed once
    const size_t nrec = 1ull << 24, ngather = 1ull << 22;         // 16 M records (1 GiB), 4 M of them fetched: every one a different record
    const size_t nwords = 1ull << 28, nw = 1ull << 22;            // 1 GiB of words, 4 M fetched, every one in a different 128-byte line
    float4 *src; uint32_t *idx64, *idx4, *words; float *out;
    (void)hipMalloc(&src, stream_bytes); (void)hipMalloc(&idx64, ngather * 4); (void)hipMalloc(&idx4, nw * 4); (void)hipMalloc(&out, 4);
    words = reinterpret_cast<uint32_t *>(src);
    (void)hipMemset(src, 0, stream_bytes);
    std::vector<uint32_t> h(ngather);
    // a permutation-like spread: record i * 4 + (i * 2654435761 mod 4) of 16 M: distinct records, scattered order
    for (size_t i = 0; i < ngather; i++) h[i] = (uint32_t)(((i * 2654435761ull) % ngather) * 4 + (i & 3));
    (void)hipMemcpy(idx64, h.data(), ngather * 4, hipMemcpyHostToDevice);
    for (size_t i = 0; i < nw; i++) h[i] = (uint32_t)(((i * 2654435761ull) % nw) * 64 + (i & 31));   // one word per 256-byte stretch
    (void)hipMemcpy(idx4, h.data(), nw * 4, hipMemcpyHostToDevice);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(calib_stream, dim3(4096), dim3(256), 0, 0, src, out, stream_bytes / 16);
    hipLaunchKernelGGL(calib_gather64, dim3(4096), dim3(256), 0, 0, src, idx64, out, ngather);
    hipLaunchKernelGGL(calib_gather4, dim3(4096), dim3(256), 0, 0, words, idx4, out, nw);
    (void)hipDeviceSynchronize();
    printf("KNOWN calib_stream %zu\n", stream_bytes);
    printf("KNOWN calib_gather64 %zu %zu\n", ngather * 64 + ngather * 4, ngather * 48 + ngather * 4);   // whole records | the bytes touched (+ the index stream)
    printf("KNOWN calib_gather4 %zu %zu\n", nw * 64 + nw * 4, nw * 4 + nw * 4);                          // a 64-byte sector per word | the words (+ the index stream)
    return 0;
}
