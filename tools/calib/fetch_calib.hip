// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on the access patterns of this repo's kernels (MI355X_MICROARCH.md:
// "FETCH_SIZE reports exactly 1/2 of the bytes of a wide coalesced streaming read ... other access widths are uncalibrated:
// calibrate on a known byte count in your own access pattern").  Every kernel reads (or writes) a known number of bytes of a
// buffer far larger than the 256 MB Infinity Cache, once.  Run:
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out_f -- ./fetch_calib ; rocprofv3 --pmc WRITE_SIZE ... -- ./fetch_calib
// and compare the counter (KiB) per kernel with the bytes printed here (tools/calib/summarize.py).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// (a) streaming read, 16 bytes per lane, fully coalesced
__global__ void calib_stream16(const uint4* __restrict__ in, uint64_t n16, uint32_t* sink) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t a = 0;
    for (; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const uint4 v = in[i]; a ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (a == 0x12345u) *sink = a;
}
// (b) k_aln's pattern: a wave reads, per instruction, 16 whole 64-byte lines that lie `stride` bytes apart (4 lanes x 16 bytes per line;
// trace rows of 16 consecutive jobs), line after line along each row
__global__ void calib_lines64(const uint8_t* __restrict__ in, uint64_t row_bytes, uint64_t n_rows, uint32_t* sink) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t row0 = wave * 16;
    if (row0 + 16 > n_rows) return;
    const uint8_t* p = in + (row0 + (lane >> 2)) * row_bytes + (lane & 3) * 16;
    uint32_t a = 0;
    for (uint64_t off = 0; off + 64 <= row_bytes; off += 64) { const uint4 v = *reinterpret_cast<const uint4*>(p + off); a ^= v.x ^ v.w; }
    if (a == 0x12345u) *sink = a;
}
// (c) k_job's pattern: every lane streams its own row, 64 bytes (4 x 16) at a time
__global__ void calib_lane_rows(const uint8_t* __restrict__ in, uint64_t row_bytes, uint64_t n_rows, uint32_t* sink) {
    const uint64_t r = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_rows) return;
    const uint8_t* p = in + r * row_bytes;
    uint32_t a = 0;
    for (uint64_t off = 0; off + 64 <= row_bytes; off += 64) {
        const uint4 v0 = *reinterpret_cast<const uint4*>(p + off), v1 = *reinterpret_cast<const uint4*>(p + off + 16);
        const uint4 v2 = *reinterpret_cast<const uint4*>(p + off + 32), v3 = *reinterpret_cast<const uint4*>(p + off + 48);
        a ^= v0.x ^ v1.y ^ v2.z ^ v3.w;
    }
    if (a == 0x12345u) *sink = a;
}
// (d) k_loop's pattern: every lane gathers 16 bytes at a pseudo-random place of a large table (one request per lane and line)
__global__ void calib_gather16(const uint4* __restrict__ in, uint64_t n16, uint32_t per_lane, uint32_t* sink) {
    uint64_t x = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x9E3779B97F4A7C15ull + 12345;
    uint32_t a = 0;
    for (uint32_t k = 0; k < per_lane; k++) { x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32; const uint4 v = in[x % n16]; a ^= v.x; }
    if (a == 0x12345u) *sink = a;
}
// (e) streaming write, 16 bytes per lane; (f) k_aln's 64-byte line writes, 16 lines `stride` apart per instruction
__global__ void calib_wstream16(uint4* __restrict__ out, uint64_t n16) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i < n16; i += (uint64_t)gridDim.x * blockDim.x) out[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
__global__ void calib_wlines64(uint8_t* __restrict__ out, uint64_t row_bytes, uint64_t n_rows) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const uint64_t row0 = wave * 16;
    if (row0 + 16 > n_rows) return;
    uint8_t* p = out + (row0 + (lane >> 2)) * row_bytes + (lane & 3) * 16;
    for (uint64_t off = 0; off + 64 <= row_bytes; off += 64) *reinterpret_cast<uint4*>(p + off) = make_uint4((uint32_t)off, 1u, 2u, 3u);
}

int main() {
    const uint64_t bytes = 8ull << 30;                     // 8 GiB: far beyond the 256 MB Infinity Cache
    uint8_t* buf = nullptr; uint32_t* sink = nullptr;
    CHK(hipMalloc(&buf, bytes)); CHK(hipMalloc(&sink, 4));
    CHK(hipMemset(buf, 1, bytes));
    CHK(hipDeviceSynchronize());
    const uint64_t row = 8192, rows = bytes / row;         // 8 KB rows (a 1 kb job's trace row)
    hipLaunchKernelGGL(calib_stream16, dim3(256 * 16), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, sink);
    hipLaunchKernelGGL(calib_lines64, dim3((unsigned)(rows / 16 / 4)), dim3(256), 0, 0, buf, row, rows, sink);
    hipLaunchKernelGGL(calib_lane_rows, dim3((unsigned)(rows / 256)), dim3(256), 0, 0, buf, row, rows, sink);
    const uint32_t per_lane = 64; const uint64_t lanes = 1ull << 22;
    hipLaunchKernelGGL(calib_gather16, dim3((unsigned)(lanes / 256)), dim3(256), 0, 0, (const uint4*)buf, bytes / 16, per_lane, sink);
    hipLaunchKernelGGL(calib_wstream16, dim3(256 * 16), dim3(256), 0, 0, (uint4*)buf, bytes / 16);
    hipLaunchKernelGGL(calib_wlines64, dim3((unsigned)(rows / 16 / 4)), dim3(256), 0, 0, buf, row, rows);
    CHK(hipDeviceSynchronize());
    printf("calib_stream16 read_bytes %llu\ncalib_lines64 read_bytes %llu\ncalib_lane_rows read_bytes %llu\n", (unsigned long long)bytes,
           (unsigned long long)bytes, (unsigned long long)bytes);
    printf("calib_gather16 read_bytes_requested %llu lines64 %llu lines128 %llu\n", (unsigned long long)(lanes * per_lane * 16),
           (unsigned long long)(lanes * per_lane * 64), (unsigned long long)(lanes * per_lane * 128));
    printf("calib_wstream16 write_bytes %llu\ncalib_wlines64 write_bytes %llu\n", (unsigned long long)bytes, (unsigned long long)bytes);
    return 0;
}
