// kvq_fp8.hip -- per-tensor fp8 (OCP e4m3fn) quantisation for the fp8 forward GEMMs of BASELINE.json configs[4] (gfx950).
//
// Extension: the reference is f32 throughout; off unless the engine is built with fp8 forward GEMMs.
//   y = x . W^T + b  (modeling_bert.py:139-352 behind models/bagon/Bagon.py:46-53)  becomes
//   y = (sat(x * sx) . sat(W * sw)^T) / (sx * sw) + b   with sx = 448 / amax|x|, sw = 448 / amax|W| (448 = largest e4m3 value),
// Weights: scales from the tensor that is quantised, in the same step (the whole flat bf16 shadow buffer in two launches per
// optimiser step, one scale per GEMM weight; a segment table maps buffer ranges to weights).  Activations, as the engine uses
// them: DELAYED scaling (kvq_fp8_quantize_delayed): one pass that quantises with the scale derived from the previous call's
// amax at that site (4x headroom, kvq_fp8_update_scales) and records this call's amax; the first call at a site runs with
// scale 1 and saturates at +-448.  kvq_fp8_quantize (amax pass + quantise pass, scale of the same tensor) is the two-launch
// form without history.  A NaN stays a NaN through the quantisation (the clamp lets it pass), so a diverged activation shows.
#include "kvq_common.h"

namespace kvq {

constexpr int Q_THREADS = 256;
// (FP8_MAX, amax8, quant8, fp8_amax_note: kvq_common.h -- the producers of csrc/kvq_nn.hip / kvq_gemm2.hip emit the same bytes)

__device__ __forceinline__ void block_amax_to(float m, float* dst) {
    __shared__ float sm[Q_THREADS / WAVE];
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = sm[0];
#pragma unroll
        for (int i = 1; i < Q_THREADS / WAVE; ++i) t = fmaxf(t, sm[i]);
        atomicMax(reinterpret_cast<unsigned*>(dst), __float_as_uint(t));          // non-negative floats order like their bit patterns
    }
}

__device__ __forceinline__ float scale_of(float amax) { return amax > 0.f ? FP8_MAX / amax : 1.0f; }

// amax accumulators are zeroed by a KERNEL, not hipMemsetAsync: a memset captured into a hipGraph is a node of another kind and
// was the one thing in the replayed fp8 step that did not keep its place in the stream's order -- with 123 segments two identically
// seeded runs parted ways after 8 - 14 steps (an amax cleared after the first maxima had landed), eager launches never did
// (tools/fp8_flake.py, profiles/r04_fp8.md)
// `step` / `period` (weights inside the training step, kvq_fp8_quantize_segments_periodic): the amax accumulators are refreshed only
// on steps that are multiples of `period` -- the step count is read on the DEVICE, so a replayed graph takes the same decision as
// an eager step
__global__ __launch_bounds__(256) void fp8_zero_kernel(float* __restrict__ p, int n, const unsigned long long* __restrict__ step = nullptr,
                                                       int period = 1) {
    if (step && period > 1 && (*step % (unsigned long long)period) != 0) return;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) p[i] = 0.f;
}

// segments: seg s covers elements [off[s], off[s] + n[s]) of src (bf16) / dst (fp8), n[s] % 8 == 0, off[s] % 8 == 0
__global__ __launch_bounds__(Q_THREADS) void fp8_seg_amax_kernel(const unsigned short* __restrict__ src, const int64_t* __restrict__ off,
                                                                  const int64_t* __restrict__ n, float* __restrict__ amax,
                                                                  const unsigned long long* __restrict__ step = nullptr, int period = 1) {
    if (step && period > 1 && (*step % (unsigned long long)period) != 0) return;      // (uniform over the grid)
    const int s = blockIdx.y;
    const int64_t chunks = n[s] >> 3;
    const uint4* p = reinterpret_cast<const uint4*>(src + off[s]);
    float m = 0.f;
    for (int64_t c = (int64_t)blockIdx.x * Q_THREADS + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * Q_THREADS) m = fmaxf(m, amax8(p[c]));
    block_amax_to(m, amax + s);
}

__global__ __launch_bounds__(Q_THREADS) void fp8_seg_quant_kernel(const unsigned short* __restrict__ src, const int64_t* __restrict__ off,
                                                                   const int64_t* __restrict__ n, const float* __restrict__ amax,
                                                                   unsigned char* __restrict__ dst, float* __restrict__ scale,
                                                                   const unsigned long long* __restrict__ step = nullptr, int period = 1) {
    // (period < 0: the conversion is gated like the amax pass -- the Adam kernel wrote this step's bytes with the scales in force)
    if (step && period < -1 && (*step % (unsigned long long)(-period)) != 0) return;
    const int s = blockIdx.y;
    const int64_t chunks = n[s] >> 3;
    const float sc = scale_of(amax[s]);
    const uint4* p = reinterpret_cast<const uint4*>(src + off[s]);
    uint2* q = reinterpret_cast<uint2*>(dst + off[s]);
    for (int64_t c = (int64_t)blockIdx.x * Q_THREADS + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * Q_THREADS) q[c] = quant8(p[c], sc);
    if (blockIdx.x == 0 && threadIdx.x == 0) scale[s] = sc;
}

// one tensor, rows of `cols` elements with row stride ld (elements): activations
__global__ __launch_bounds__(Q_THREADS) void fp8_amax_kernel(const unsigned short* __restrict__ src, int64_t rows, int cols, int64_t ld,
                                                              float* __restrict__ amax) {
    const int cpr = cols >> 3;
    const int64_t chunks = rows * cpr;
    float m = 0.f;
    for (int64_t c = (int64_t)blockIdx.x * Q_THREADS + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * Q_THREADS) {
        const int64_t r = c / cpr;
        m = fmaxf(m, amax8(*reinterpret_cast<const uint4*>(src + r * ld + (c - r * cpr) * 8)));
    }
    block_amax_to(m, amax);
}

__global__ __launch_bounds__(Q_THREADS) void fp8_quant_kernel(const unsigned short* __restrict__ src, int64_t rows, int cols, int64_t ld,
                                                               const float* __restrict__ amax, unsigned char* __restrict__ dst,
                                                               float* __restrict__ scale) {
    const int cpr = cols >> 3;
    const int64_t chunks = rows * cpr;
    const float sc = scale_of(amax[0]);
    for (int64_t c = (int64_t)blockIdx.x * Q_THREADS + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * Q_THREADS) {
        const int64_t r = c / cpr;
        *reinterpret_cast<uint2*>(dst + c * 8) = quant8(*reinterpret_cast<const uint4*>(src + r * ld + (c - r * cpr) * 8), sc);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) scale[0] = sc;
}

// Delayed scaling for activations: quantise with the scale derived from the PREVIOUS call's amax at this site and record this
// call's amax on the way -- one pass over the tensor instead of two.  A value beyond the previous range saturates;
// kvq_fp8_update_scales leaves `headroom` (4x) between the last amax and 448, which costs an fp format no precision, only
// underflow range.  state (per site) = {scale, unused x 7, partial amax of each of the Q_PARTS workgroups}: plain stores, no
// atomics (a thousand atomicMax on one word serialise at ~12 ns each: measured 28 us per call for a 4-us copy).
constexpr int Q_PARTS = 512;                    // workgroups of the quantisation pass: it writes slots 0 .. 511
constexpr int Q_STATE = 8 + FP8_PARTS;          // floats per site (the producers of round 5 spread their atomic maxima over 4096 slots)

__global__ __launch_bounds__(Q_THREADS) void fp8_quant_delayed_kernel(const unsigned short* __restrict__ src, int64_t rows, int cols, int64_t ld,
                                                                       unsigned char* __restrict__ dst, float* __restrict__ state) {
    __shared__ float sm[Q_THREADS / WAVE];
    const int cpr = cols >> 3;
    const int64_t chunks = rows * cpr;
    const float sc = state[0];
    float m = 0.f;
    if (ld == cols) {                                                       // dense: no row arithmetic
        const uint4* p = reinterpret_cast<const uint4*>(src);
        uint2* q = reinterpret_cast<uint2*>(dst);
        for (int64_t c = (int64_t)blockIdx.x * Q_THREADS + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * Q_THREADS) {
            const uint4 v = p[c];
            m = fmaxf(m, amax8(v));
            q[c] = quant8(v, sc);
        }
    } else {
        for (int64_t c = (int64_t)blockIdx.x * Q_THREADS + threadIdx.x; c < chunks; c += (int64_t)gridDim.x * Q_THREADS) {
            const int64_t r = c / cpr;
            const uint4 v = *reinterpret_cast<const uint4*>(src + r * ld + (c - r * cpr) * 8);
            m = fmaxf(m, amax8(v));
            *reinterpret_cast<uint2*>(dst + c * 8) = quant8(v, sc);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = sm[0];
#pragma unroll
        for (int i = 1; i < Q_THREADS / WAVE; ++i) t = fmaxf(t, sm[i]);
        state[8 + blockIdx.x] = t;
    }
}

__global__ __launch_bounds__(Q_THREADS) void fp8_update_scales_kernel(float* state, float headroom) {
    __shared__ float sm[Q_THREADS / WAVE];
    float* st = state + (size_t)blockIdx.x * Q_STATE;
    float m = 0.f;
    for (int i = threadIdx.x; i < FP8_PARTS; i += Q_THREADS) {
        m = fmaxf(m, st[8 + i]);
        st[8 + i] = 0.f;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, WAVE));
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = sm[0];
#pragma unroll
        for (int i = 1; i < Q_THREADS / WAVE; ++i) t = fmaxf(t, sm[i]);
        if (t > 0.f && headroom > 0.f) st[0] = FP8_MAX / (t * headroom);  // a site that was not visited keeps its scale; headroom 0: only clear
    }
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_fp8_quantize(const void* x_bf16, int64_t rows, int cols, int64_t ld, void* out_fp8, float* amax, float* scale, void* stream) {
    KVQ_REQUIRE(x_bf16 && out_fp8 && amax && scale && rows > 0 && cols > 0, "kvq_fp8_quantize: bad argument");
    KVQ_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ld >= cols && (((uintptr_t)x_bf16 | (uintptr_t)out_fp8) & 15) == 0,
                "kvq_fp8_quantize: cols, ld multiples of 8 and 16-byte aligned buffers");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(fp8_zero_kernel, dim3(1), dim3(256), 0, st, amax, 1);
    const int64_t chunks = rows * (cols / 8);
    const unsigned grid = (unsigned)((chunks + Q_THREADS * 4 - 1) / (Q_THREADS * 4) > 2048 ? 2048 : (chunks + Q_THREADS * 4 - 1) / (Q_THREADS * 4));
    hipLaunchKernelGGL(fp8_amax_kernel, dim3(grid), dim3(Q_THREADS), 0, st, (const unsigned short*)x_bf16, rows, cols, ld, amax);
    hipLaunchKernelGGL(fp8_quant_kernel, dim3(grid), dim3(Q_THREADS), 0, st, (const unsigned short*)x_bf16, rows, cols, ld, amax,
                       (unsigned char*)out_fp8, scale);
    return check_launch("fp8_quant_kernel");
}

int kvq_fp8_state_floats(void) { return Q_STATE; }

int kvq_fp8_quantize_delayed(const void* x_bf16, int64_t rows, int cols, int64_t ld, void* out_fp8, float* state, void* stream) {
    KVQ_REQUIRE(x_bf16 && out_fp8 && state && rows > 0 && cols > 0, "kvq_fp8_quantize_delayed: bad argument");
    KVQ_REQUIRE(cols % 8 == 0 && ld % 8 == 0 && ld >= cols && (((uintptr_t)x_bf16 | (uintptr_t)out_fp8) & 15) == 0,
                "kvq_fp8_quantize_delayed: cols, ld multiples of 8 and 16-byte aligned buffers");
    hipLaunchKernelGGL(fp8_quant_delayed_kernel, dim3(Q_PARTS), dim3(Q_THREADS), 0, (hipStream_t)stream, (const unsigned short*)x_bf16, rows, cols, ld,
                       (unsigned char*)out_fp8, state);
    return check_launch("fp8_quant_delayed_kernel");
}

int kvq_fp8_update_scales(float* state, int nsites, float headroom, void* stream) {
    KVQ_REQUIRE(state && nsites > 0 && (headroom >= 1.0f || headroom == 0.0f), "kvq_fp8_update_scales: headroom >= 1, or 0 to only clear the amax partials");
    hipLaunchKernelGGL(fp8_update_scales_kernel, dim3((unsigned)nsites), dim3(Q_THREADS), 0, (hipStream_t)stream, state, headroom);
    return check_launch("fp8_update_scales_kernel");
}

static int quantize_segments_impl(const void* src_bf16, const int64_t* seg_off, const int64_t* seg_n, int nseg, int64_t max_seg_n, void* dst_fp8,
                                  float* amax, float* scale, const unsigned long long* step, int period, void* stream);

int kvq_fp8_quantize_segments(const void* src_bf16, const int64_t* seg_off, const int64_t* seg_n, int nseg, int64_t max_seg_n, void* dst_fp8,
                              float* amax, float* scale, void* stream) {
    return quantize_segments_impl(src_bf16, seg_off, seg_n, nseg, max_seg_n, dst_fp8, amax, scale, nullptr, 1, stream);
}

int kvq_fp8_quantize_segments_periodic(const void* src_bf16, const int64_t* seg_off, const int64_t* seg_n, int nseg, int64_t max_seg_n,
                                       void* dst_fp8, float* amax, float* scale, const void* step_count_u64, int period, void* stream) {
    KVQ_REQUIRE(step_count_u64 && (period >= 1 || period < -1), "kvq_fp8_quantize_segments_periodic: a device step counter and |period| >= 1");
    return quantize_segments_impl(src_bf16, seg_off, seg_n, nseg, max_seg_n, dst_fp8, amax, scale, (const unsigned long long*)step_count_u64,
                                  period, stream);
}

static int quantize_segments_impl(const void* src_bf16, const int64_t* seg_off, const int64_t* seg_n, int nseg, int64_t max_seg_n, void* dst_fp8,
                                  float* amax, float* scale, const unsigned long long* step, int period, void* stream) {
    KVQ_REQUIRE(src_bf16 && seg_off && seg_n && dst_fp8 && amax && scale && nseg > 0 && max_seg_n > 0, "kvq_fp8_quantize_segments: bad argument");
    KVQ_REQUIRE((((uintptr_t)src_bf16 | (uintptr_t)dst_fp8) & 15) == 0, "kvq_fp8_quantize_segments: 16-byte aligned buffers");
    hipStream_t st = (hipStream_t)stream;
    const int gate = period < 0 ? -period : period;          // period < 0: also the conversion pass runs on refresh steps only
    hipLaunchKernelGGL(fp8_zero_kernel, dim3((unsigned)((nseg + 255) / 256)), dim3(256), 0, st, amax, nseg, step, gate);
    const int64_t chunks = max_seg_n / 8;
    unsigned gx = (unsigned)((chunks + Q_THREADS * 8 - 1) / (Q_THREADS * 8));
    gx = gx < 1 ? 1 : (gx > 256 ? 256 : gx);
    hipLaunchKernelGGL(fp8_seg_amax_kernel, dim3(gx, (unsigned)nseg), dim3(Q_THREADS), 0, st, (const unsigned short*)src_bf16, seg_off, seg_n, amax,
                       step, gate);
    hipLaunchKernelGGL(fp8_seg_quant_kernel, dim3(gx, (unsigned)nseg), dim3(Q_THREADS), 0, st, (const unsigned short*)src_bf16, seg_off, seg_n, amax,
                       (unsigned char*)dst_fp8, scale, step, period);
    return check_launch("fp8_seg_quant_kernel");
}

}  // extern "C"
