// kvq_common.h -- shared host/device helpers of libkvq.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "kvq.h"

namespace kvq {

// ---- error reporting (thread-local message, C-ABI never throws) -------------------------------------------
int fail(int code, const char* fmt, ...);
int check_launch(const char* what);
// profiling ring (kvq_prof_enable): record a start/stop event pair around one launch
bool prof_begin(hipStream_t st);
void prof_end(hipStream_t st);

#define KVQ_REQUIRE(cond, ...)                                  \
    do {                                                        \
        if (!(cond)) return ::kvq::fail(KVQ_E_INVALID, __VA_ARGS__); \
    } while (0)

static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// ---- io dtype helpers --------------------------------------------------------------------------------------
constexpr int WAVE = 64;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float(((unsigned)b) << 16); }

// round-to-nearest-even, NaN stays NaN (plain cast lowers to v_cvt_pk_bf16_f32 on gfx950)
__device__ __forceinline__ unsigned short f32_to_bf16(float x) {
    __bf16 b = (__bf16)x;
    return __builtin_bit_cast(unsigned short, b);
}

// 8 consecutive elements: HBM-bound kernels move 16 bytes per lane and instruction (8-byte accesses run at 0.54-0.70x the
// 16-byte rate on MI355X, MI355X_MICROARCH.md "sc1 / nt loads" row), i.e. 8 bf16 or 2 x 4 f32
struct f32x8 {
    f32x4 lo, hi;
};

template <int DT>
struct IO;

template <>
struct IO<KVQ_F32> {
    typedef float elem;
    static constexpr int bytes = 4;
    // 4 consecutive elements starting at element offset `off` (off % 4 == 0, 16-byte aligned rows)
    __device__ static __forceinline__ f32x4 load4(const void* base, size_t off) {
        return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
    }
    __device__ static __forceinline__ void store4(void* base, size_t off, f32x4 v) {
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off) = v;
    }
    __device__ static __forceinline__ float load1(const void* base, size_t off) {
        return reinterpret_cast<const float*>(base)[off];
    }
    __device__ static __forceinline__ void store1(void* base, size_t off, float v) {
        reinterpret_cast<float*>(base)[off] = v;
    }
    __device__ static __forceinline__ f32x8 load8(const void* base, size_t off) {
        const f32x4* p = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off);
        f32x8 v = {p[0], p[1]};
        return v;
    }
    __device__ static __forceinline__ void store8(void* base, size_t off, const f32x8& v) {
        f32x4* p = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(base) + off);
        p[0] = v.lo; p[1] = v.hi;
    }
    // value the consumer of a stored element will read back
    __device__ static __forceinline__ float round(float v) { return v; }
};

template <>
struct IO<KVQ_BF16> {
    typedef unsigned short elem;
    static constexpr int bytes = 2;
    __device__ static __forceinline__ f32x4 load4(const void* base, size_t off) {
        u16x4 r = *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(base) + off);
        f32x4 v = {bf16_to_f32(r.x), bf16_to_f32(r.y), bf16_to_f32(r.z), bf16_to_f32(r.w)};
        return v;
    }
    __device__ static __forceinline__ void store4(void* base, size_t off, f32x4 v) {
        u16x4 r = {f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
        *reinterpret_cast<u16x4*>(reinterpret_cast<unsigned short*>(base) + off) = r;
    }
    __device__ static __forceinline__ float load1(const void* base, size_t off) {
        return bf16_to_f32(reinterpret_cast<const unsigned short*>(base)[off]);
    }
    __device__ static __forceinline__ void store1(void* base, size_t off, float v) {
        reinterpret_cast<unsigned short*>(base)[off] = f32_to_bf16(v);
    }
    // off % 8 == 0 and a 16-byte aligned base
    __device__ static __forceinline__ f32x8 load8(const void* base, size_t off) {
        const uint4 r = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + off);
        f32x8 v;
        v.lo.x = __uint_as_float(r.x << 16); v.lo.y = __uint_as_float(r.x & 0xffff0000u);
        v.lo.z = __uint_as_float(r.y << 16); v.lo.w = __uint_as_float(r.y & 0xffff0000u);
        v.hi.x = __uint_as_float(r.z << 16); v.hi.y = __uint_as_float(r.z & 0xffff0000u);
        v.hi.z = __uint_as_float(r.w << 16); v.hi.w = __uint_as_float(r.w & 0xffff0000u);
        return v;
    }
    __device__ static __forceinline__ void store8(void* base, size_t off, const f32x8& v) {
        typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
        const bf2 a = {(__bf16)v.lo.x, (__bf16)v.lo.y}, b = {(__bf16)v.lo.z, (__bf16)v.lo.w};
        const bf2 c = {(__bf16)v.hi.x, (__bf16)v.hi.y}, d = {(__bf16)v.hi.z, (__bf16)v.hi.w};
        uint4 r = {__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, c), __builtin_bit_cast(unsigned, d)};
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + off) = r;
    }
    __device__ static __forceinline__ float round(float v) { return bf16_to_f32(f32_to_bf16(v)); }
};

// Philox4x32-10 (Salmon et al. 2011).  One call -> 4 x 32 random bits for counter (c0..c3), key (k0,k1).
struct U4 {
    unsigned x, y, z, w;
};
__device__ __forceinline__ U4 philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
        const unsigned n1 = (unsigned)p1;
        const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
        const unsigned n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    return {c0, c1, c2, c3};
}
// keep-mask for 4 consecutive elements starting at element index e4*4 of dropout site `site`
__device__ __forceinline__ U4 drop_bits(unsigned long long seed, unsigned site, unsigned long long e4) {
    return philox4x32((unsigned)e4, (unsigned)(e4 >> 32), site, 0x5eedu, (unsigned)seed, (unsigned)(seed >> 32));
}
__device__ __forceinline__ float keep_scale(unsigned bits, unsigned thresh, float inv_keep) {
    return bits >= thresh ? inv_keep : 0.f;   // P(drop) = thresh / 2^32
}
static inline unsigned drop_threshold(float p) {
    double t = (double)p * 4294967296.0;
    if (t < 0) t = 0;
    if (t > 4294967295.0) t = 4294967295.0;
    return (unsigned)t;
}

// device-resident addend of every dropout seed (kvq_set_seed_offset, per calling thread; csrc/kvq_nn.hip)
const unsigned long long* seed_offset_ptr();

// ---- fp8 (OCP e4m3fn) quantisation of bf16 values, shared by csrc/kvq_fp8.hip (the quantisation passes) and the kernels that emit
//      the fp8 copy of an activation while they produce it (round 5: LayerNorm forward, attention forward, the GELU epilogue)
constexpr float FP8_MAX = 448.0f;

__device__ __forceinline__ float amax8(const uint4 r) {          // (a NaN element does not enter the maximum: the scale stays usable)
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    float m = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        m = fmaxf(m, fabsf(__uint_as_float(w[u] << 16)));
        m = fmaxf(m, fabsf(__uint_as_float(w[u] & 0xffff0000u)));
    }
    return m;
}

// 8 bf16 -> 8 fp8 (two dwords), saturating: the values are clamped to +-448 before the conversion
__device__ __forceinline__ uint2 quant8(const uint4 r, float s) {
    const unsigned w[4] = {r.x, r.y, r.z, r.w};
    float f[8];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float a = __uint_as_float(w[u] << 16) * s, b = __uint_as_float(w[u] & 0xffff0000u) * s;
        // fminf / fmaxf return the non-NaN operand: clamp only what is a number, e4m3fn has a NaN encoding for the rest
        f[2 * u] = a != a ? a : fminf(fmaxf(a, -FP8_MAX), FP8_MAX);
        f[2 * u + 1] = b != b ? b : fminf(fmaxf(b, -FP8_MAX), FP8_MAX);
    }
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[4], f[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[6], f[7], hi, true);
    return make_uint2((unsigned)lo, (unsigned)hi);
}

// 4 values (already rounded to bf16) -> 4 fp8 bytes, as quant8 does it
__device__ __forceinline__ unsigned quant4(f32x4 v, float s) {
    float f[4] = {v.x * s, v.y * s, v.z * s, v.w * s};
#pragma unroll
    for (int u = 0; u < 4; ++u) f[u] = f[u] != f[u] ? f[u] : fminf(fmaxf(f[u], -FP8_MAX), FP8_MAX);
    int w = 0;
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[0], f[1], w, false);
    w = __builtin_amdgcn_cvt_pk_fp8_f32(f[2], f[3], w, true);
    return (unsigned)w;
}
__device__ __forceinline__ float amax4(f32x4 v) {                 // (NaN does not enter, as in amax8)
    return fmaxf(fmaxf(fmaxf(0.f, fabsf(v.x)), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w)));
}
constexpr int FP8_PARTS = 4096;                 // partial-amax slots of a delayed-scaling site (kvq_fp8_state_floats() = 8 + FP8_PARTS)
// this launch's contribution to a site's amax: one atomic per wave on one of the site's 4096 partial slots (non-negative floats
// order like their bit patterns; kvq_fp8_update_scales takes the maximum over the slots and clears them).  4096 slots = 128 cache
// lines: the 8192 waves of a LayerNorm launch put 2 atomics on a word and 64 on a line -- with 512 slots (16 lines) the atomics of
// a launch queued behind each other in L2 for microseconds
__device__ __forceinline__ void fp8_amax_note(float m, float* state, unsigned slot) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<unsigned*>(state + 8 + (slot & (FP8_PARTS - 1))), __float_as_uint(m));
}

// ---- wave-level reductions (64 lanes) ------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
    return v;
}
__device__ __forceinline__ float wave_sum_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
    return v;
}

// torch.argmin ordering: strictly smaller wins; NaN is smaller than any number; equal -> lower index.
__device__ __forceinline__ bool dist_less(float d, float best) { return (d < best) || ((d != d) && (best == best)); }
__device__ __forceinline__ bool dist_equal(float a, float b) { return (a == b) || ((a != a) && (b != b)); }
__device__ __forceinline__ bool cand_better(float d, int i, float bd, int bi) {
    return dist_less(d, bd) || (dist_equal(d, bd) && i < bi);
}

}  // namespace kvq
