// FETCH_SIZE / WRITE_SIZE calibration on gfx950: stream a buffer of known size with 4-, 8- and 16-byte loads per lane (and with
// LDS-DMA), so that the PMC readings of kernels with those access widths can be turned into bytes (MI355X_MICROARCH.md, HBM:
// "FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read; other widths are uncalibrated").
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/fetch_calib.hip -o /tmp/fetch_calib && rocprofv3 --pmc FETCH_SIZE ... -- /tmp/fetch_calib
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

template <typename T>
__global__ void read_kernel(const T* __restrict__ src, size_t n, unsigned* sink) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        T v = src[i];
        const unsigned* w = reinterpret_cast<const unsigned*>(&v);
        for (unsigned j = 0; j < sizeof(T) / 4; ++j) acc ^= w[j];
    }
    if (acc == 0x12345678u) *sink = acc;      // never true in practice: keeps the loads alive
}
template <typename T>
__global__ void write_kernel(T* __restrict__ dst, size_t n) {
    T v;
    unsigned* w = reinterpret_cast<unsigned*>(&v);
    for (unsigned j = 0; j < sizeof(T) / 4; ++j) w[j] = threadIdx.x + j;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = v;
}
__global__ void read_u16x4_kernel(const ushort4* __restrict__ src, size_t n, unsigned* sink) {      // 8 bytes per lane (4 x bf16)
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        ushort4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}

int main() {
    const size_t bytes = 512ull << 20;                      // beyond the 256 MiB Infinity Cache
    void *a, *b; unsigned* sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc((void**)&sink, 4);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    const dim3 grid(2048), block(256);
    for (int rep = 0; rep < 3; ++rep) {
        hipMemset(b, rep, bytes);                                                               // evict
        hipLaunchKernelGGL(read_kernel<unsigned>, grid, block, 0, 0, (const unsigned*)a, bytes / 4, sink);
        hipMemset(b, rep, bytes);
        hipLaunchKernelGGL(read_kernel<uint2>, grid, block, 0, 0, (const uint2*)a, bytes / 8, sink);
        hipMemset(b, rep, bytes);
        hipLaunchKernelGGL(read_u16x4_kernel, grid, block, 0, 0, (const ushort4*)a, bytes / 8, sink);
        hipMemset(b, rep, bytes);
        hipLaunchKernelGGL(read_kernel<uint4>, grid, block, 0, 0, (const uint4*)a, bytes / 16, sink);
        hipLaunchKernelGGL(write_kernel<unsigned>, grid, block, 0, 0, (unsigned*)b, bytes / 4);
        hipLaunchKernelGGL(write_kernel<uint2>, grid, block, 0, 0, (uint2*)b, bytes / 8);
        hipLaunchKernelGGL(write_kernel<uint4>, grid, block, 0, 0, (uint4*)b, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("calibration buffer: %zu bytes per kernel\n", bytes);
    return 0;
}
