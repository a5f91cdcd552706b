// Ablation of the VQ distance kernel's inner loop on gfx950: where do the cycles between f32 MFMAs go?
//   A: dependent chain of 384 v_mfma_f32_32x32x2_f32 per wave, operands in registers
//   B: A + operands re-read from LDS (ds_read_b128 per 4 MFMAs, x2)
//   C: B + one __syncthreads() per 16 MFMAs
//   D: C + 5 global 16-byte loads per thread per stage written to LDS (the real staging traffic)
// Grid 1024 x 256 threads, 40 KiB dynamic LDS -> 4 workgroups per CU, like vq_dist_tile_kernel.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>
__global__ __launch_bounds__(256, 4) void k(const float* __restrict__ src, float* out, int nst) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, i = lane & 31, h = lane >> 5;
    for (int j = tid; j < 10240; j += 256) smem[j] = src[j];
    __syncthreads();
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 a = *(const f32x4*)(smem + 4 * lane), b = *(const f32x4*)(smem + 1024 + 4 * lane);
    f32x4 st[5];
    for (int s = 0; s < nst; ++s) {
        const int buf = s & 1;
        if (MODE >= 3) {
#pragma unroll
            for (int q = 0; q < 5; ++q) st[q] = *(const f32x4*)(src + ((size_t)blockIdx.x * 5120 + (size_t)(s & 7) * 1310720 + (q * 256 + tid) * 4) % (1 << 24));
        }
        const float* erow = smem + buf * 5120 + (w * 32 + i) * 32;
        const float* zrow = smem + buf * 5120 + 4096 + i * 32;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (MODE >= 1) {
                const int slot = ((2 * g + h) ^ ((i >> 1) & 7)) << 2;
                a = *(const f32x4*)(erow + slot);
                b = *(const f32x4*)(zrow + slot);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        if (MODE >= 3) {
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int L = q * 256 + tid, r = L >> 3, c = L & 7;
                *(f32x4*)(smem + (buf ^ 1) * 5120 + r * 32 + ((c ^ ((r >> 1) & 7)) << 2)) = st[q];
            }
        }
        if (MODE >= 2) __syncthreads();
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    if (s == 12345.678f) out[blockIdx.x * 256 + tid] = s;
}

template <int MODE>
void run(const char* name, const float* src, float* out) {
    const int nst = 24;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 40960, 0, src, out, nst);
    hipEventRecord(e0);
    const int reps = 20;
    for (int it = 0; it < reps; ++it) hipLaunchKernelGGL(k<MODE>, dim3(1024), dim3(256), 40960, 0, src, out, nst);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double flop = 1024.0 * 4 * nst * 16 * (32.0 * 32 * 2 * 2);
    printf("%-40s %8.1f us  %7.1f TFLOP/s  (%.0f%% of 157.3)\n", name, us, flop / us / 1e6, flop / us / 1e6 / 157.3 * 100);
}

int main() {
    float *src, *out;
    hipMalloc(&src, (1 << 24) * 4 + 65536);
    hipMalloc(&out, 1024 * 256 * 4);
    hipMemset(src, 0x3c, (1 << 24) * 4 + 65536);
    run<0>("A: MFMA chain only", src, out);
    run<1>("B: + LDS fragment reads", src, out);
    run<2>("C: + barrier per 16 MFMA", src, out);
    run<3>("D: + global loads + LDS writes", src, out);
    run<0>("A again", src, out);
    return 0;
}
