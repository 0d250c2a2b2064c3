// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the demod kernel's access pattern:
// every lane walks its own contiguous stream with 16-byte loads (uncoalesced across lanes) and
// writes 16-byte chunks to its own output stream.  Known byte counts -> correction factors.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void coalesced_read(const uint4 *in, uint32_t *out, size_t n16) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) { uint4 v = in[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
// lane-per-stream: stream s = 64 KiB region; 16-B loads, `gap` iterations of dummy work between loads
__global__ void per_lane_stream_read(const uint4 *in, uint32_t *out, uint32_t n_streams, uint32_t per_stream16) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    const uint4 *p = in + (size_t)s * per_stream16;
    uint32_t acc = 0;
    for (uint32_t i = 0; i < per_stream16; i++) { uint4 v = p[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void per_lane_stream_write(uint4 *outp, uint32_t n_streams, uint32_t per_stream16) {
    uint32_t s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_streams) return;
    uint4 *p = outp + (size_t)s * per_stream16;
    for (uint32_t i = 0; i < per_stream16; i++) p[i] = make_uint4(i, s, i ^ s, 7);
}
int main() {
    const uint32_t n_streams = 196608, per16 = 4096;          // 64 KiB per stream, 12.9 GB total
    const size_t bytes = (size_t)n_streams * per16 * 16;
    uint4 *buf; uint32_t *out; hipMalloc(&buf, bytes); hipMalloc(&out, 64); hipMemset(buf, 1, bytes);
    hipLaunchKernelGGL(coalesced_read, dim3(256 * 16), dim3(256), 0, 0, buf, out, bytes / 16);
    hipLaunchKernelGGL(per_lane_stream_read, dim3((n_streams + 255) / 256), dim3(256), 0, 0, buf, out, n_streams, per16);
    hipLaunchKernelGGL(per_lane_stream_write, dim3((n_streams + 255) / 256), dim3(256), 0, 0, buf, n_streams, per16 / 8);
    hipDeviceSynchronize();
    printf("bytes read per read-kernel = %zu ; bytes written by write-kernel = %zu\n", bytes, bytes / 8);
    return 0;
}
