// kvq_vq.hip -- the VectorQuantizer codebook step on gfx950 (MI355X).
//
// Replaces models/shelgon3/VectorQuantizer.py:55-93 of the reference (distance, argmin, one-hot gather,
// losses, straight-through, perplexity) and its autograd.  Numerics contract: include/kvq.h ("kvq order v1").
//
// Forward, fast path (D % 32 == 0), three launches per call (+ one re-layout of the codebook per codebook UPDATE):
//   vq_dist_packed_kernel   the distance contraction on the exact-f32 matrix core instruction v_mfma_f32_32x32x2_f32 with the
//                           codes as MFMA rows (a lane owns one token and 16 codes of a 32x32 tile: the running arg-min is
//                           lane-local); workgroup = 64 tokens x 128 codes, 4 waves, the codebook operand streamed
//                           global -> VGPR from an MFMA-fragment-ordered copy, only the token tile staged through LDS; the
//                           per-token minimum leaves as ONE 64-bit integer atomicMin on (orderable(d) << 32 | code)
//   vq_epilogue_kernel      HBM-bound: gather E[idx], z_q = z + (e - z), per-token sum of squares (f64), histogram
//   vq_finalize_kernel      one workgroup: loss and perplexity from the partials in a fixed order (bitwise reproducible; no
//                           float atomics anywhere)
//   vq_pack_codebook_kernel the fragment-ordered codebook copy: once per optimiser step through kvq_vq_pack_codebook +
//                           kvq_vq_forward_packed (the engine), or inside kvq_vq_forward for one-off calls.
// Measured alternatives that lost and were removed in round 2 (profiles/r01_summary.md): one fused kernel over all codes
// (32 tokens x K per workgroup: 100 us against 62 + 14.5 us at N = 8192, K = 512, D = 768) and the codebook staged through
// LDS (73 us).  Folding the epilogue into the distance kernel needs every code block of a token finished first, i.e. a
// workgroup per token tile over ALL codes -- the 100-us structure.
// Forward, generic path (any D): one wave per token, same chain orders, scalar fmaf.
// Backward: g_z elementwise; g_E by ordered slab reduction (per code, per token chunk, then chunks in order).
#include <limits.h>
#include <math.h>

#include "kvq_common.h"

namespace kvq {

struct FwdParams {
    const void* z;      // [G,N,D] io dtype
    const float* E;     // [G,K,D]
    void* z_q;          // [G,N,D] io dtype
    int64_t* idx;       // [G,N]
    double* tok_sumsq;  // ws [G,N]
    unsigned* counts;   // ws [G,K]
    const float* e2;    // ws [G,K]  (generic path only)
    float* dump;        // optional [N,K] distances (debug hook), G == 1
    unsigned long long* keys;  // ws [G,N]  (v2 path: packed (distance, code) minima)
    const float* epack;        // ws [G, ceil(K/32)*32, D]  codebook in MFMA-fragment order (v3 path)
    unsigned* tickets;         // ws [G, token blocks] arrivals of a token block's code blocks, then [G] finished token blocks (fused path)
    float* loss;               // [G]   (fused path: written by the workgroup that finishes last)
    float* perplexity;         // [G]
    float* counts_f;           // [G,K] or null
    float beta;
    int fused;                 // 1: the last code block of a token block runs the epilogue, the last token block the final sums
    int64_t N;
    int K, D;
};

// -------------------------------------------------------------------------------------------------------------
// per-token epilogue, executed by one full wave: gather, straight-through, squared error, histogram
//   VectorQuantizer.py:72 (z_q = E[idx]), :76-77 (squared error), :80 (z + (z_q - z))
// -------------------------------------------------------------------------------------------------------------
template <int DT, bool HIST>
__device__ __forceinline__ void token_epilogue(const FwdParams& p, int g, int64_t tok, int code, int lane) {
    const size_t zrow = ((size_t)g * p.N + (size_t)tok) * p.D;
    const float* e = p.E + ((size_t)g * p.K + (size_t)code) * p.D;
    double ss = 0.0;
    if ((p.D & 3) == 0) {
        for (int c = lane; c < (p.D >> 2); c += WAVE) {
            f32x4 zv = IO<DT>::load4(p.z, zrow + 4 * c);
            f32x4 ev = *reinterpret_cast<const f32x4*>(e + 4 * c);
            f32x4 df = ev - zv;        // fl(e - z)
            f32x4 q = zv + df;         // fl(z + fl(e - z))
            IO<DT>::store4(p.z_q, zrow + 4 * c, q);
            ss += (double)df.x * (double)df.x + (double)df.y * (double)df.y;
            ss += (double)df.z * (double)df.z + (double)df.w * (double)df.w;
        }
    } else {
        for (int j = lane; j < p.D; j += WAVE) {
            float zv = IO<DT>::load1(p.z, zrow + j);
            float df = e[j] - zv;
            IO<DT>::store1(p.z_q, zrow + j, zv + df);
            ss += (double)df * (double)df;
        }
    }
    ss = wave_sum_f64(ss);
    if (lane == 0) {
        p.tok_sumsq[(size_t)g * p.N + tok] = ss;
        p.idx[(size_t)g * p.N + tok] = (int64_t)code;
        if (HIST) atomicAdd(p.counts + (size_t)g * p.K + code, 1u);   // integer histogram: order independent, exact
    }
}

// 4 consecutive activation elements kept RAW in registers between the global load and the LDS write
template <int DT> struct ZRaw;
template <> struct ZRaw<KVQ_F32> {
    typedef f32x4 T;
    __device__ static __forceinline__ T load(const void* base, size_t off) { return *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(base) + off); }
    __device__ static __forceinline__ f32x4 cvt(T r) { return r; }
};
template <> struct ZRaw<KVQ_BF16> {
    typedef u16x4 T;
    __device__ static __forceinline__ T load(const void* base, size_t off) { return *reinterpret_cast<const u16x4*>(reinterpret_cast<const unsigned short*>(base) + off); }
    __device__ static __forceinline__ f32x4 cvt(T r) {
        f32x4 v = {bf16_to_f32(r.x), bf16_to_f32(r.y), bf16_to_f32(r.z), bf16_to_f32(r.w)};
        return v;
    }
};

// The same for U tokens of one wave at once (fused path).  The tail of the distance kernel is a latency chain, not a bandwidth
// problem -- 128 workgroups of 4 waves each move 0.4 MB -- so every load of the U rows is issued before the first is used: with
// CH > 0 (D = 256 CH, compile-time) 2 U CH independent 16-byte loads per lane in ONE round trip (CH = 0: a run-time loop over D
// with 2 U loads per pass).  D % 4 == 0.  Rows of tokens past N are loaded from row 0 and never stored.  The per-token sums of
// squares go to `ss_out` (LDS, U doubles): the workgroup adds its 64 of them in token order.
template <int DT, int U, int CH>
__device__ __forceinline__ void tokens_epilogue(const FwdParams& p, int g, int64_t tok0, const int* codes, double* ss_out, int lane) {
    size_t zrow[U];
    const float* e[U];
    double ss[U];
    bool on[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        on[u] = codes[u] >= 0;                                            // wave-uniform (codes come from LDS)
        zrow[u] = ((size_t)g * p.N + (size_t)(on[u] ? tok0 + u : 0)) * p.D;
        e[u] = p.E + ((size_t)g * p.K + (size_t)(on[u] ? codes[u] : 0)) * p.D;
        ss[u] = 0.0;
    }
    if (CH > 0) {
        constexpr int C = CH > 0 ? CH : 1;
        typename ZRaw<DT>::T zr[U][C];
        f32x4 ev[U][C];
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < C; ++j) {
                zr[u][j] = ZRaw<DT>::load(p.z, zrow[u] + 4 * (lane + WAVE * j));
                ev[u][j] = *reinterpret_cast<const f32x4*>(e[u] + 4 * (lane + WAVE * j));
            }
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const f32x4 zv = ZRaw<DT>::cvt(zr[u][j]);
                const f32x4 df = ev[u][j] - zv;        // fl(e - z)
                if (on[u]) IO<DT>::store4(p.z_q, zrow[u] + 4 * (lane + WAVE * j), zv + df);       // fl(z + fl(e - z))
                ss[u] += (double)df.x * (double)df.x + (double)df.y * (double)df.y;
                ss[u] += (double)df.z * (double)df.z + (double)df.w * (double)df.w;
            }
    } else {
        for (int c = lane; c < (p.D >> 2); c += WAVE) {
            f32x4 zv[U], ev[U];
#pragma unroll
            for (int u = 0; u < U; ++u) { zv[u] = IO<DT>::load4(p.z, zrow[u] + 4 * c); ev[u] = *reinterpret_cast<const f32x4*>(e[u] + 4 * c); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const f32x4 df = ev[u] - zv[u];
                if (on[u]) IO<DT>::store4(p.z_q, zrow[u] + 4 * c, zv[u] + df);
                ss[u] += (double)df.x * (double)df.x + (double)df.y * (double)df.y;
                ss[u] += (double)df.z * (double)df.z + (double)df.w * (double)df.w;
            }
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const double t = wave_sum_f64(ss[u]);
        if (lane == 0) {
            ss_out[u] = on[u] ? t : 0.0;
            if (on[u]) p.idx[(size_t)g * p.N + tok0 + u] = (int64_t)codes[u];
        }
    }
}

// loss and perplexity of codebook g from the per-token / per-code partials, in a fixed order, by ONE workgroup of NT threads
//   VectorQuantizer.py:76-77 and :84-85.  AGENT: the partials were written by other workgroups of THIS launch (fused path).
//   `ts`: n_part partial sums of squares (one per token: three-kernel path; one per 64-token block: fused path).
template <int NT, bool AGENT>
__device__ __forceinline__ void finalize_block(const double* __restrict__ ts, int64_t n_part, const unsigned* __restrict__ counts_u, int g,
                                               int64_t N, int K, int D, float beta, float* loss, float* perplexity,
                                               float* counts_f, double* sd, double* sf) {
    const int t = threadIdx.x;
    const unsigned* cu = counts_u + (size_t)g * K;
    double a = 0.0;
    for (int64_t n = t; n < n_part; n += NT) a += AGENT ? __hip_atomic_load(ts + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : ts[n];
    double ent = 0.0;                                // the f32 entropy terms of the reference, summed in f64 (oracle/vq_oracle.c: a
    for (int k = t; k < K; k += NT) {                // sequential f32 sum over K = 8192 terms drifts by 2e-4; torch.sum does not)
        const float cnt = (float)(AGENT ? __hip_atomic_load(cu + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cu[k]);
        if (counts_f) counts_f[(size_t)g * K + k] = cnt;
        const float pk = cnt / (float)N;                 // e_mean                       (:84)
        ent += (double)(pk * logf(pk + 1e-10f));
    }
    sd[t] = a;
    sf[t] = ent;
    __syncthreads();
    for (int s = NT / 2; s > 0; s >>= 1) {
        if (t < s) { sd[t] += sd[t + s]; sf[t] += sf[t + s]; }
        __syncthreads();
    }
    if (t == 0) {
        const float m = (float)(sd[0] / ((double)N * (double)D));
        const float bm = beta * m;
        loss[g] = m + bm;                                 // mean(.) + beta*mean(.)       (:76-77)
        perplexity[g] = expf(-(float)sf[0]);              // (:85)
    }
}

// =============================================================================================================
// fast path: distance / arg-min kernel (codes as MFMA rows) + streaming epilogue kernel
//   the per-token minimum goes to global memory as ONE 64-bit atomicMin per token on the key
//   (orderable(d) << 32 | code): minimum distance first, lowest code on ties, NaN mapped to key 0 (torch.argmin).
//   The epilogue kernel (HBM-bound) decodes the keys: gather, straight-through, squared error, histogram.
// =============================================================================================================
constexpr int T2_WAVES = 4;
constexpr int T2_THREADS = T2_WAVES * WAVE;
constexpr int T2_CN = T2_WAVES * 32;          // 128 codes per workgroup

__device__ __forceinline__ unsigned long long pack_key(float d, int code) {
    unsigned u = __float_as_uint(d);
    unsigned key = (d != d) ? 0u : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));   // monotone map, NaN lowest
    return ((unsigned long long)key << 32) | (unsigned)code;
}

// ---- codebook pre-packed in MFMA-fragment order -----------------------------------------------------------
// The A operand (32 codes x 8 contraction floats per MFMA group) belongs to ONE wave, so staging it through LDS buys
// nothing and costs the LDS write path.  A tiny kernel re-lays E (once per codebook update) as
//     Epack[code_block32][k_group8][lane = (code i, half h)][4 floats] = E[32*cb + i][8*g + 4*h .. +3]
// so that each A fragment is ONE fully coalesced 1-KiB wave load straight into VGPRs (double-buffered in registers).
// Only the shared 32-token tile still goes through LDS (8 KiB per workgroup).  Same arithmetic, same bits.
__global__ __launch_bounds__(256) void vq_pack_codebook_kernel(const float* __restrict__ E, int K, int D, float* __restrict__ Epack) {
    const int g = blockIdx.y;
    const int G8 = D / 8;
    const int64_t total = (int64_t)((K + 31) / 32) * G8 * 64;          // float4 slots
    const float* Eg = E + (size_t)g * K * D;
    float* Pg = Epack + (size_t)g * ((K + 31) / 32) * 32 * D;
    for (int64_t s = (int64_t)blockIdx.x * 256 + threadIdx.x; s < total; s += (int64_t)gridDim.x * 256) {
        const int lane = (int)(s & 63);
        const int64_t t = s >> 6;
        const int g8 = (int)(t % G8);
        const int cb = (int)(t / G8);
        int code = cb * 32 + (lane & 31);
        code = code < K ? code : K - 1;                                   // padding rows: masked at the argmin
        const f32x4 v = *reinterpret_cast<const f32x4*>(Eg + (size_t)code * D + 8 * g8 + 4 * (lane >> 5));
        *reinterpret_cast<f32x4*>(Pg + 4 * s) = v;
    }
}

constexpr int T3_KC = 32;                                    // contraction floats per z stage

// Diagnostic build only (-DKVQ_VQ_DIAG, tools/build_diag.sh; the product library has none of this): thread 0 of every workgroup
// times its prologue, its stages and its stage barriers with s_memtime and stores [workgroup][8] u64 at its end.
#ifdef KVQ_VQ_DIAG
// (stamps stored from inside the stage loop made the kernel five times slower: the loop only ACCUMULATES two scalar differences
//  -- time inside the stage barriers, time between them -- and the workgroup stores five numbers at its end)
__device__ unsigned long long* g_vq_diag = nullptr;
#define VQ_DIAG_DECL unsigned long long vq_t_entry = __builtin_amdgcn_s_memtime(), vq_t_prev = 0, vq_t_bar = 0, vq_acc_bar = 0, vq_acc_stage = 0, vq_t_pro = 0;
#define VQ_DIAG_PROLOGUE { vq_t_pro = __builtin_amdgcn_s_memtime(); vq_t_prev = vq_t_pro; }
#define VQ_DIAG_BEFORE_BARRIER { vq_t_bar = __builtin_amdgcn_s_memtime(); vq_acc_stage += vq_t_bar - vq_t_prev; }
#define VQ_DIAG_AFTER_BARRIER { vq_t_prev = __builtin_amdgcn_s_memtime(); vq_acc_bar += vq_t_prev - vq_t_bar; }
#define VQ_DIAG_END                                                                                                  \
    if (g_vq_diag != nullptr && threadIdx.x == 0) {                                                                  \
        unsigned long long* o = g_vq_diag + (size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 8;                       \
        o[0] = vq_t_entry; o[1] = vq_t_pro; o[2] = vq_acc_stage; o[3] = vq_acc_bar; o[4] = vq_t_prev;                \
        o[5] = __builtin_amdgcn_s_memtime(); o[6] = __builtin_amdgcn_s_memrealtime();                                \
    }
#else
#define VQ_DIAG_DECL
#define VQ_DIAG_PROLOGUE
#define VQ_DIAG_BEFORE_BARRIER
#define VQ_DIAG_AFTER_BARRIER
#define VQ_DIAG_END
#endif

// (Round 2 also measured the token operand loaded straight into registers in B-fragment layout -- no LDS, no barrier, waves fully
//  decoupled, same bits: 81 us against 65.5 us for this kernel at N = 8192, K = 512, D = 768.  The LDS stage is not what idles the
//  matrix pipe.  Four token tiles per wave -- 128 tokens per workgroup, one workgroup per CU, every codebook fragment used for
//  four MFMAs -- measured 70 us.)
// TT = 32-token tiles per wave (1 or 2): with 2 the wave reuses each codebook fragment for two MFMAs (two independent
// accumulator chains), halving the codebook traffic per flop; the workgroup then covers 64 tokens x 128 codes.
// NST > 0: the stage count D/32 is a compile-time constant and the stage loop is fully unrolled (no loop back-edge,
// so the compiler's s_waitcnt vmcnt counts stay exact); NST == 0: run-time loop for any D % 32 == 0.
template <int DT, bool PRIO, int TT, int NST>
__global__ __launch_bounds__(T2_THREADS, TT == 2 ? 3 : 4) void vq_dist_packed_kernel(FwdParams p) {
    constexpr int TMW = 32 * TT;                              // tokens per workgroup
    constexpr int ZSTAGE = TMW * T3_KC;                       // floats per z stage
    extern __shared__ __attribute__((aligned(16))) float smem[];
    VQ_DIAG_DECL
    const int tid = threadIdx.x;
    const int w = tid >> 6, lane = tid & 63, i = lane & 31, h = lane >> 5;
    const int g = blockIdx.z;
    const int64_t tok0 = (int64_t)blockIdx.x * TMW;
    const int code0 = blockIdx.y * T2_CN;
    const size_t zbase = (size_t)g * p.N * p.D;
    const int nst = NST > 0 ? NST : p.D / T3_KC;
    const int G8 = p.D / 8;
    const int ncb = (p.K + 31) / 32;
    int cb = code0 / 32 + w;
    const bool wave_has_codes = cb < ncb;
    cb = wave_has_codes ? cb : ncb - 1;
    const f32x4* __restrict__ Ap = reinterpret_cast<const f32x4*>(p.epack + (size_t)g * ncb * 32 * p.D) + ((size_t)cb * G8) * 64 + lane;

    // z staging: TT 16-byte chunks per thread per stage (rows of 128 B, chunk swizzle (row>>1)&7)
    typename ZRaw<DT>::T zreg[TT], zregB[TT];                  // raw 4 elements (converted only when written to LDS); B: the pair's second stage
    int zoff[TT];
    int64_t ztok[TT];
    float zmul[TT];
#pragma unroll
    for (int q = 0; q < TT; ++q) {
        const int L = q * T2_THREADS + tid;
        const int r = L >> 3, c = L & 7;
        zmul[q] = tok0 + r < p.N ? 1.0f : 0.0f;                          // rows past N: clamped load, zeroed value
        ztok[q] = tok0 + r < p.N ? tok0 + r : p.N - 1;
        zoff[q] = r * T3_KC + ((c ^ ((r >> 1) & 7)) << 2);
        zreg[q] = ZRaw<DT>::load(p.z, zbase + (size_t)ztok[q] * p.D + c * 4);
        *reinterpret_cast<f32x4*>(smem + zoff[q]) = ZRaw<DT>::cvt(zreg[q]) * zmul[q];
        if (nst > 1) {                                                   // the first PAIR of stages goes in before the first barrier
            zregB[q] = ZRaw<DT>::load(p.z, zbase + (size_t)ztok[q] * p.D + T3_KC + c * 4);
            *reinterpret_cast<f32x4*>(smem + ZSTAGE + zoff[q]) = ZRaw<DT>::cvt(zregB[q]) * zmul[q];
        }
    }
    const int zc4 = (tid & 7) * 4;
    // codebook fragments ping-pong between two NAMED register sets (a0 for even stages, a1 for odd ones): with a
    // single set + copy the compiler's wait-count merging across the loop back-edge stalls the MFMA cluster on the
    // prefetch it has just issued
    f32x4 a0[4], a1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) a0[q] = Ap[(size_t)q * 64];
    __syncthreads();
    VQ_DIAG_PROLOGUE

    f32x16 acc[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float pe = 0.f, pz[TT];
#pragma unroll
    for (int t = 0; t < TT; ++t) pz[t] = 0.f;
    const int sw = (i >> 1) & 7;

#define KVQ_PREFETCH_A(AREG, STG)                                                                             \
    { _Pragma("unroll") for (int q = 0; q < 4; ++q) AREG[q] = Ap[(size_t)(4 * (STG) + q) * 64]; }
#define KVQ_PREFETCH_Z(ZREG, STG)                                                                             \
    {                                                                                                          \
        _Pragma("unroll") for (int q = 0; q < TT; ++q)                                                         \
            ZREG[q] = ZRaw<DT>::load(p.z, zbase + (size_t)ztok[q] * p.D + (STG) * T3_KC + zc4);                \
    }
#define KVQ_COMMIT(ZREG, BUF)                                                                                  \
    {                                                                                                          \
        __builtin_amdgcn_sched_barrier(0); /* conversion + LDS write stay BEHIND the MFMA cluster */           \
        _Pragma("unroll") for (int q = 0; q < TT; ++q)                                                         \
            *reinterpret_cast<f32x4*>(smem + (BUF) * ZSTAGE + zoff[q]) = ZRaw<DT>::cvt(ZREG[q]) * zmul[q];     \
    }
#define KVQ_CLUSTER(AREG, BUF)                                                                                 \
    {                                                                                                          \
        const float* zst = smem + (BUF) * ZSTAGE;                                                              \
        if (PRIO) __builtin_amdgcn_s_setprio(1);                                                               \
        _Pragma("unroll") for (int gq = 0; gq < 4; ++gq) {                                                     \
            const f32x4 a = AREG[gq];                                                                          \
            const int slot = ((2 * gq + h) ^ sw) << 2;                                                         \
            f32x4 b[TT];                                                                                       \
            _Pragma("unroll") for (int t = 0; t < TT; ++t)                                                     \
                b[t] = *reinterpret_cast<const f32x4*>(zst + (32 * t + i) * T3_KC + slot);                     \
            _Pragma("unroll") for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b[t].x, acc[t], 0, 0, 0); \
            _Pragma("unroll") for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b[t].y, acc[t], 0, 0, 0); \
            _Pragma("unroll") for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b[t].z, acc[t], 0, 0, 0); \
            _Pragma("unroll") for (int t = 0; t < TT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b[t].w, acc[t], 0, 0, 0); \
            pe = __builtin_fmaf(a.x, a.x, pe); pe = __builtin_fmaf(a.y, a.y, pe);                             \
            pe = __builtin_fmaf(a.z, a.z, pe); pe = __builtin_fmaf(a.w, a.w, pe);                             \
            _Pragma("unroll") for (int t = 0; t < TT; ++t) {                                                   \
                pz[t] = __builtin_fmaf(b[t].x, b[t].x, pz[t]); pz[t] = __builtin_fmaf(b[t].y, b[t].y, pz[t]);  \
                pz[t] = __builtin_fmaf(b[t].z, b[t].z, pz[t]); pz[t] = __builtin_fmaf(b[t].w, b[t].w, pz[t]);  \
            }                                                                                                  \
        }                                                                                                      \
        if (PRIO) __builtin_amdgcn_s_setprio(0);                                                               \
    }

    // ONE barrier per PAIR of stages (round 3; tools/vq_stamps.py put 800 of a stage's 4900 cycles at the stage boundary: commit
    // of the next z tile, barrier, fragment reads -- with the SIMD's other wave in the same phase).  LDS holds two pairs of stage
    // buffers; the codebook fragments still ping-pong per stage (they are the wave's own registers: no barrier involved).
#define KVQ_STAGE_PAIR(ST)                                                                                     \
    {                                                                                                          \
        const int cur = (((ST) >> 1) & 1) * 2, nxt = cur ^ 2;                                                  \
        const bool more1 = (ST) + 1 < nst, more2 = (ST) + 2 < nst, more3 = (ST) + 3 < nst;                     \
        if (more2) KVQ_PREFETCH_Z(zreg, (ST) + 2)                                                              \
        if (more3) KVQ_PREFETCH_Z(zregB, (ST) + 3)                                                             \
        if (more1) KVQ_PREFETCH_A(a1, (ST) + 1)                                                                \
        KVQ_CLUSTER(a0, cur)                                                                                   \
        if (more1) {                                                                                           \
            if (more2) KVQ_PREFETCH_A(a0, (ST) + 2)                                                            \
            KVQ_CLUSTER(a1, cur + 1)                                                                           \
        }                                                                                                      \
        if (more2) KVQ_COMMIT(zreg, nxt)                                                                       \
        if (more3) KVQ_COMMIT(zregB, nxt + 1)                                                                  \
        VQ_DIAG_BEFORE_BARRIER                                                                                 \
        __syncthreads();                                                                                       \
        VQ_DIAG_AFTER_BARRIER                                                                                  \
    }
    if (NST > 0) {
#pragma unroll
        for (int sp = 0; sp < (NST + 1) / 2; ++sp) KVQ_STAGE_PAIR(2 * sp)
    } else {
        for (int st = 0; st < nst; st += 2) KVQ_STAGE_PAIR(st)
    }
#undef KVQ_STAGE_PAIR
#undef KVQ_PREFETCH_A
#undef KVQ_PREFETCH_Z
#undef KVQ_COMMIT
#undef KVQ_CLUSTER

    // ---- arg-min over the workgroup's 128 codes, per token.  In-kernel timing (tools/vq_stamps.py) put the first version of this
    // tail -- a branch per candidate for the validity test, the debug dump and the NaN / tie rule, one ds_bpermute round trip per
    // candidate for ||e||^2 -- at 14.6 k cycles of a 134 k-cycle workgroup at K = 512 (one round: nothing else uses the matrix
    // pipe meanwhile).  Now: the 32 ||e||^2 of the wave go through LDS once (four 16-byte reads per lane), every candidate becomes
    // the 64-bit key the final atomicMin orders by anyway (monotone map of the distance, NaN lowest, then the code), and minima
    // are integer selects.  Same arithmetic for the distance (fl(fl(z2 + e2) - 2 acc)), same winner: key order == cand_better
    // order (-0 is canonicalised to +0 first; the two differ in nothing else).
    if (PRIO) __builtin_amdgcn_s_setprio(3);                 // the tail ahead of the other workgroups' MFMA clusters: it frees the slot
    const float e2v = pe + __shfl_xor(pe, 32, WAVE);
    const int cbase = code0 + w * 32;
    float* e2s = smem + T2_WAVES * 2 * TMW + 32 * w;          // behind the merge buffers; 32 floats per wave (z stages are dead)
    unsigned long long* red_key = reinterpret_cast<unsigned long long*>(smem);     // [4 waves][TMW]
    if (lane < 32) e2s[i] = e2v;                              // same wave writes and reads: LDS operations of a wave stay in order in
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");    // hardware; the fence + wave barrier pin that order for the compiler as
    __builtin_amdgcn_wave_barrier();                          // well (no instruction is emitted for either)
    f32x4 e2q[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) e2q[q] = *reinterpret_cast<const f32x4*>(e2s + 8 * q + 4 * h);
    if (p.dump) {                                             // debug hook (wave-uniform): the distances as the arg-min sees them
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            const float z2 = pz[t] + __shfl_xor(pz[t], 32, WAVE);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float dd = (z2 + e2q[r >> 2][r & 3]) - 2.0f * acc[t][r];
                if (wave_has_codes && cbase + cl < p.K && tok0 + 32 * t + i < p.N) p.dump[(size_t)(tok0 + 32 * t + i) * p.K + cbase + cl] = dd;
            }
        }
    }
#pragma unroll
    for (int t = 0; t < TT; ++t) {
        const float z2 = pz[t] + __shfl_xor(pz[t], 32, WAVE);
        unsigned long long best = ~0ull;
        if (wave_has_codes && cbase + 32 <= p.K) {            // wave-uniform: every code of this wave exists (the usual case: K a multiple of 32)
            // a lane's codes ascend with r: a strict "<" on the 32-bit distance key keeps the lowest code among equals
            unsigned bk = 0xffffffffu;
            int bc = INT_MAX;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float tt = z2 + e2q[r >> 2][r & 3];
                const float dd = (tt - 2.0f * acc[t][r]) + 0.0f;                   // (+ 0: -0 -> +0, nothing else changes)
                const unsigned key = (unsigned)(pack_key(dd, 0) >> 32);
                const bool lt = key < bk;
                bk = lt ? key : bk;
                bc = lt ? cbase + (r & 3) + 8 * (r >> 2) + 4 * h : bc;
            }
            best = ((unsigned long long)bk << 32) | (unsigned)bc;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int cl = (r & 3) + 8 * (r >> 2) + 4 * h;
                const float tt = z2 + e2q[r >> 2][r & 3];
                const float dd = (tt - 2.0f * acc[t][r]) + 0.0f;
                const int code = cbase + cl;
                const unsigned long long key = (wave_has_codes && code < p.K) ? pack_key(dd, code) : ~0ull;
                best = key < best ? key : best;
            }
        }
        const unsigned long long other = __shfl_xor(best, 32, WAVE);
        best = other < best ? other : best;
        if (lane < 32) red_key[w * TMW + 32 * t + i] = best;
    }
    __syncthreads();
    if (tid < TMW && tok0 + tid < p.N) {
        unsigned long long b = red_key[tid];
#pragma unroll
        for (int ww = 1; ww < T2_WAVES; ++ww) {
            const unsigned long long v = red_key[ww * TMW + tid];
            b = v < b ? v : b;
        }
        if (b != ~0ull) atomicMin(p.keys + (size_t)g * p.N + tok0 + tid, b);
    }
    VQ_DIAG_END
    if (!p.fused) return;                                    // (kernel argument: uniform)

    // ---- fused tail (round 5; VectorQuantizer.py:65-85 in the launch that computed the distances).  A token block's minima are
    // final when the LAST of its gridDim.y code blocks has added its own: every workgroup draws an arrival ticket behind its
    // atomicMins (both agent-scope atomics; the minima were issued by wave 0 alone, whose vmcnt(0) wait below is their
    // acknowledgement), and the one that draws gridDim.y - 1 runs the token block's epilogue -- gather, straight-through, squared
    // error, histogram -- on the 64 rows of z its own XCD's L2 has just served to all the block's code blocks (workgroups
    // x, x + gridDim.x, ... land on one XCD whenever gridDim.x is a multiple of 8; a placement bonus, never relied on).  The token
    // block that finishes last of all then reduces the per-token / per-code partials to loss and perplexity in a fixed order.
    // Nothing waits on another workgroup: a workgroup that is not last leaves.
    __shared__ int s_codes[64];
    __shared__ int s_last;
    if (w == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) {
            const unsigned t = __hip_atomic_fetch_add(p.tickets + (size_t)g * gridDim.x + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = t == gridDim.y - 1;
        }
    }
    __syncthreads();
    if (!s_last) return;
    if (tid < 64) {
        int code = -1;
        if (tid < TMW && tok0 + tid < p.N)
            code = (int)(unsigned)(__hip_atomic_load(p.keys + (size_t)g * p.N + tok0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffffffffull);
        s_codes[tid] = code;
    }
    __syncthreads();
    if (tid < TMW && s_codes[tid] >= 0) {                    // one histogram atomic per distinct code of the token block
        const int c = s_codes[tid];
        unsigned cnt = 0;
        bool first = true;
        for (int j = 0; j < TMW; ++j) {
            const bool same = s_codes[j] == c;
            cnt += same ? 1u : 0u;
            first = first && !(same && j < tid);
        }
        if (first) atomicAdd(p.counts + (size_t)g * p.K + c, cnt);
    }
    constexpr int U = 4;
    constexpr int CH = (NST > 0 && (NST * T3_KC) % (4 * WAVE) == 0) ? NST * T3_KC / (4 * WAVE) : 0;        // D = 768: 3 chunks per lane
    __shared__ double s_ss[64];
    for (int t = w * U; t < TMW; t += T2_WAVES * U) tokens_epilogue<DT, U, CH>(p, g, tok0 + t, s_codes + t, s_ss + t, lane);
    __syncthreads();
    // the token block's sum of squares in token order, as ONE agent-scope (sc1, write-through) 8-byte store: what the workgroup
    // that finishes LAST reads back with agent-scope loads (MI355X_MICROARCH.md, inter-workgroup visibility: "8-B agent atomics
    // both sides"); the histogram adds above are agent-scope atomics as well
    if (tid == 0) {
        double a = 0.0;
        for (int j = 0; j < TMW; ++j) a += s_ss[j];
        __hip_atomic_store(p.tok_sumsq + (size_t)g * gridDim.x + blockIdx.x, a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // every wave: its histogram adds (wave 0: the partial sum too) are performed
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(p.tickets + (size_t)gridDim.z * gridDim.x + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = t == gridDim.x - 1;
    }
    __syncthreads();
    if (!s_last) return;
    double* sd = reinterpret_cast<double*>(smem);             // 2 x 256 doubles of the (dead) z stage buffers
    finalize_block<T2_THREADS, true>(p.tok_sumsq + (size_t)g * gridDim.x, gridDim.x, p.counts, g, p.N, p.K, p.D, p.beta, p.loss, p.perplexity,
                                     p.counts_f, sd, sd + T2_THREADS);
}

// keys := all ones, histogram := 0, arrival tickets := 0 -- ONE launch in front of the distance kernel (stream memsets are graph
// nodes of another kind; every entry point of this library launches kernels only, see include/kvq.h)
__global__ __launch_bounds__(256) void vq_init_kernel(unsigned long long* keys, int64_t n_keys, unsigned* counts, int64_t n_counts,
                                                       unsigned* tickets, int64_t n_tickets) {
    const int64_t i0 = (int64_t)blockIdx.x * 256 + threadIdx.x, step = (int64_t)gridDim.x * 256;
    for (int64_t i = i0; i < n_keys; i += step) keys[i] = ~0ull;
    for (int64_t i = i0; i < n_counts; i += step) counts[i] = 0u;
    for (int64_t i = i0; i < n_tickets; i += step) tickets[i] = 0u;
}

// epilogue of the v2 path: 64 tokens per workgroup, one wave per token at a time
constexpr int EP_TOK = 16;
constexpr int EP_THREADS = 256;

template <int DT>
__global__ __launch_bounds__(EP_THREADS) void vq_epilogue_kernel(FwdParams p) {
    __shared__ int codes[EP_TOK];
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    const int g = blockIdx.y;
    const int64_t tok0 = (int64_t)blockIdx.x * EP_TOK;
    if (tid < EP_TOK) {
        const int64_t tok = tok0 + tid;
        codes[tid] = tok < p.N ? (int)(unsigned)(p.keys[(size_t)g * p.N + tok] & 0xffffffffull) : -1;
    }
    __syncthreads();
    if (tid < EP_TOK && codes[tid] >= 0) {       // one histogram atomic per distinct code of the workgroup
        const int c = codes[tid];
        unsigned cnt = 0;
        bool first = true;
        for (int j = 0; j < EP_TOK; ++j) {
            const bool same = codes[j] == c;
            cnt += same ? 1u : 0u;
            first = first && !(same && j < tid);
        }
        if (first) atomicAdd(p.counts + (size_t)g * p.K + c, cnt);
    }
    for (int t = w; t < EP_TOK; t += EP_THREADS / WAVE) {
        const int64_t tok = tok0 + t;
        if (tok < p.N) token_epilogue<DT, false>(p, g, tok, codes[t], lane);
    }
}

// -------------------------------------------------------------------------------------------------------------
// generic path: any D.  sq(e_k) first (one thread per code), then one wave per token.
// -------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int walk8(int s) { return (s >> 1) + ((s & 1) << 2); }

__global__ void row_sq_kernel(const float* __restrict__ E, int64_t rows, int D, float* __restrict__ out) {
    const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= rows) return;
    const float* x = E + (size_t)k * D;
    float p0 = 0.f, p1 = 0.f;
    for (int j = 0; j < D; ++j) {
        const float v = x[j];
        if ((j & 7) < 4) p0 = __builtin_fmaf(v, v, p0);
        else p1 = __builtin_fmaf(v, v, p1);
    }
    out[k] = p0 + p1;
}

constexpr int GEN_WAVES = 4;

template <int DT>
__global__ __launch_bounds__(GEN_WAVES* WAVE) void vq_fwd_generic_kernel(FwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.y;
    const int64_t tok = (int64_t)blockIdx.x * GEN_WAVES + w;
    if (tok >= p.N) return;   // whole wave leaves together; no block-level barrier below
    const int D = p.D;
    const int D8 = (D + 7) & ~7;
    float* zs = smem + (size_t)w * D8;
    const size_t zrow = ((size_t)g * p.N + (size_t)tok) * D;
    for (int j = lane; j < D8; j += WAVE) zs[j] = j < D ? IO<DT>::load1(p.z, zrow + j) : 0.f;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    // sq(z): two half chains
    float p0 = 0.f, p1 = 0.f;
    for (int j = 0; j < D; ++j) {
        const float v = zs[j];
        if ((j & 7) < 4) p0 = __builtin_fmaf(v, v, p0);
        else p1 = __builtin_fmaf(v, v, p1);
    }
    const float z2 = p0 + p1;
    const float* E = p.E + (size_t)g * p.K * D;
    const float* e2 = p.e2 + (size_t)g * p.K;
    float best = INFINITY;
    int bidx = INT_MAX;
    for (int k = lane; k < p.K; k += WAVE) {
        const float* e = E + (size_t)k * D;
        float acc = 0.f;
        for (int g8 = 0; g8 < D8; g8 += 8) {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int j = g8 + walk8(s);
                const float ev = j < D ? e[j] : 0.f;
                acc = __builtin_fmaf(zs[j], ev, acc);
            }
        }
        const float t = z2 + e2[k];
        const float dd = t - 2.0f * acc;
        if (p.dump) p.dump[(size_t)tok * p.K + k] = dd;
        if (cand_better(dd, k, best, bidx)) { best = dd; bidx = k; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ob = __shfl_xor(best, m, WAVE);
        const int oi = __shfl_xor(bidx, m, WAVE);
        if (cand_better(ob, oi, best, bidx)) { best = ob; bidx = oi; }
    }
    token_epilogue<DT, true>(p, g, tok, bidx, lane);
}

// -------------------------------------------------------------------------------------------------------------
// finalize: loss and perplexity from the per-token / per-code partials, fixed reduction order
//   VectorQuantizer.py:76-77 and :84-85
// -------------------------------------------------------------------------------------------------------------
constexpr int FIN_THREADS = 1024;   // one workgroup: the per-token partials are a 64 KB latency-bound read

__global__ __launch_bounds__(FIN_THREADS) void vq_finalize_kernel(const double* __restrict__ tok_sumsq,
                                                                   const unsigned* __restrict__ counts_u,
                                                                   int64_t N, int K, int D, float beta,
                                                                   float* loss, float* perplexity, float* counts_f) {
    __shared__ double sd[FIN_THREADS];
    __shared__ double sf[FIN_THREADS];
    finalize_block<FIN_THREADS, false>(tok_sumsq + (size_t)blockIdx.x * N, N, counts_u, blockIdx.x, N, K, D, beta, loss, perplexity, counts_f, sd, sf);
}

// -------------------------------------------------------------------------------------------------------------
// backward
// -------------------------------------------------------------------------------------------------------------
struct BwdParams {
    const void* z;
    const float* E;
    const int64_t* idx;
    const void* g_zq;
    const float* g_loss;
    void* g_z;
    float* g_E;
    float* slab;   // ws [G,T,K,D]
    int* slab_cnt; // ws [G,T,K]
    int64_t N;
    int K, D, T;
    int64_t chunk;  // tokens per chunk (multiple of 64)
    float beta;
};

// g_z = g_zq - s * fl(e_idx - z),  s = g_loss * 2/(N*D)          one wave per token
template <int DT>
__global__ __launch_bounds__(256) void vq_bwd_gz_kernel(BwdParams p) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int g = blockIdx.y;
    const int64_t tok = (int64_t)blockIdx.x * 4 + w;
    if (tok >= p.N) return;
    const float gl = p.g_loss ? p.g_loss[g] : 1.0f;
    const float s = (float)((double)gl * 2.0 / ((double)p.N * (double)p.D));
    const size_t row = ((size_t)g * p.N + (size_t)tok) * p.D;
    const int code = (int)p.idx[(size_t)g * p.N + tok];
    const float* e = p.E + ((size_t)g * p.K + code) * p.D;
    if ((p.D & 3) == 0) {
        for (int c = lane; c < (p.D >> 2); c += WAVE) {
            const f32x4 zv = IO<DT>::load4(p.z, row + 4 * c);
            const f32x4 ev = *reinterpret_cast<const f32x4*>(e + 4 * c);
            f32x4 gq = {0.f, 0.f, 0.f, 0.f};
            if (p.g_zq) gq = IO<DT>::load4(p.g_zq, row + 4 * c);
            const f32x4 df = ev - zv;
            f32x4 o;
            o.x = __builtin_fmaf(-s, df.x, gq.x); o.y = __builtin_fmaf(-s, df.y, gq.y);
            o.z = __builtin_fmaf(-s, df.z, gq.z); o.w = __builtin_fmaf(-s, df.w, gq.w);
            IO<DT>::store4(p.g_z, row + 4 * c, o);
        }
    } else {
        for (int j = lane; j < p.D; j += WAVE) {
            const float df = e[j] - IO<DT>::load1(p.z, row + j);
            const float gq = p.g_zq ? IO<DT>::load1(p.g_zq, row + j) : 0.f;
            IO<DT>::store1(p.g_z, row + j, __builtin_fmaf(-s, df, gq));
        }
    }
}

// slab[t][k][:] = sum over the tokens n of chunk t with idx_n == k, in increasing n, of f(n); cnt[t][k] = how many
//   MODE 0: f = fl(e_k - z_n)   (codebook gradient)        MODE 1: f = z_n   (EMA cluster sums)
// grid (K, T, G), 256 threads.  Phase 1: the chunk's indices are scanned once (coalesced, ballot + prefix) into an
// ORDERED list of matching tokens in LDS.  Phase 2: threads own columns and walk the list with independent,
// unrolled row loads.  The visiting order is the token order: deterministic, no float atomics.
constexpr int SEG_THREADS = 256;
constexpr int SEG_MAX_CHUNK = 4096;   // tokens per chunk (LDS list capacity)

template <int DT, int MODE>
__global__ __launch_bounds__(SEG_THREADS) void vq_seg_sum_kernel(BwdParams p) {
    __shared__ int list[SEG_MAX_CHUNK];
    __shared__ int wave_cnt[SEG_THREADS / WAVE];
    __shared__ int base_s;
    const int k = blockIdx.x, t = blockIdx.y, g = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t n0 = (int64_t)t * p.chunk;
    const int64_t n1 = n0 + p.chunk < p.N ? n0 + p.chunk : p.N;
    const int64_t* idx = p.idx + (size_t)g * p.N;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (int64_t nb = n0; nb < n1; nb += SEG_THREADS) {      // uniform trip count for the whole workgroup
        const int64_t n = nb + tid;
        const bool hit = n < n1 && idx[n] == (int64_t)k;
        const unsigned long long m = __ballot(hit);
        if (lane == 0) wave_cnt[w] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int ww = 0; ww < w; ++ww) off += wave_cnt[ww];
        if (hit) list[off + __popcll(m & ((1ull << lane) - 1ull))] = (int)(n - n0);
        __syncthreads();
        if (tid == 0) base_s += wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        __syncthreads();
    }
    const int cnt = base_s;
    if (tid == 0) p.slab_cnt[((size_t)g * p.T + t) * p.K + k] = cnt;
    if (cnt == 0) return;
    const float* e = p.E ? p.E + ((size_t)g * p.K + k) * p.D : nullptr;
    float* out = p.slab + (((size_t)g * p.T + t) * p.K + k) * p.D;
    const size_t zrow0 = ((size_t)g * p.N + (size_t)n0) * p.D;
    if ((p.D & 7) == 0) {
        // Vector walk: a row is D/8 chunks of 8 elements (one 16-byte load of bf16); the workgroup is cut into RG row groups of cw
        // threads (cw = the power of two >= D/8) that take every RG-th list entry, four rows in flight per thread; the groups'
        // sums are then added in group order.  Fixed visiting order: bitwise reproducible.  With the codes of a training run
        // (a few long lists instead of many short ones) the scalar walk below was latency-bound: one 2-byte load per row and
        // thread (42-119 us per call in the step trace of round 1).
        __shared__ float red[8 * SEG_THREADS];
        const int CH = p.D >> 3;
        int cw = 8;
        while (cw < CH) cw <<= 1;
        const int RG = SEG_THREADS / cw, rg = tid / cw, c = tid - rg * cw;
        float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if (c < CH) {
            f32x8 ev;
            ev.lo = 0.f; ev.hi = 0.f;
            if (MODE == 0) ev = IO<KVQ_F32>::load8(e, (size_t)c * 8);
            const float ef[8] = {ev.lo.x, ev.lo.y, ev.lo.z, ev.lo.w, ev.hi.x, ev.hi.y, ev.hi.z, ev.hi.w};
            int i = rg;
            for (; i + 3 * RG < cnt; i += 4 * RG) {
                f32x8 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) v[u] = IO<DT>::load8(p.z, zrow0 + (size_t)list[i + u * RG] * p.D + (size_t)c * 8);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float vf[8] = {v[u].lo.x, v[u].lo.y, v[u].lo.z, v[u].lo.w, v[u].hi.x, v[u].hi.y, v[u].hi.z, v[u].hi.w};
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (MODE == 0) ? (ef[j] - vf[j]) : vf[j];
                }
            }
            for (; i < cnt; i += RG) {
                const f32x8 v = IO<DT>::load8(p.z, zrow0 + (size_t)list[i] * p.D + (size_t)c * 8);
                const float vf[8] = {v.lo.x, v.lo.y, v.lo.z, v.lo.w, v.hi.x, v.hi.y, v.hi.z, v.hi.w};
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (MODE == 0) ? (ef[j] - vf[j]) : vf[j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) red[(size_t)rg * p.D + c * 8 + j] = acc[j];
        }
        __syncthreads();
        for (int j = tid; j < p.D; j += SEG_THREADS) {
            float a = 0.f;
            for (int q = 0; q < RG; ++q) a += red[(size_t)q * p.D + j];
            out[j] = a;
        }
        return;
    }
    for (int j = tid; j < p.D; j += SEG_THREADS) {
        const float ej = (MODE == 0) ? e[j] : 0.f;
        float acc = 0.f;
        int i = 0;
        for (; i + 8 <= cnt; i += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = IO<DT>::load1(p.z, zrow0 + (size_t)list[i + u] * p.D + j);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += (MODE == 0) ? (ej - v[u]) : v[u];
        }
        for (; i < cnt; ++i) {
            const float v = IO<DT>::load1(p.z, zrow0 + (size_t)list[i] * p.D + j);
            acc += (MODE == 0) ? (ej - v) : v;
        }
        out[j] = acc;
    }
}

// g_E[k][:] = beta * s * sum_t slab[t][k][:]   (t in increasing order; empty (t,k) cells are skipped, never read)
__global__ void vq_bwd_combine_kernel(BwdParams p) {
    const int g = blockIdx.y;
    const size_t KD = (size_t)p.K * p.D;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= KD) return;
    const int k = (int)(e / p.D);
    const float gl = p.g_loss ? p.g_loss[g] : 1.0f;
    const float s = (float)((double)gl * 2.0 / ((double)p.N * (double)p.D));
    const float* sl = p.slab + (size_t)g * p.T * KD + e;
    const int* sc = p.slab_cnt + (size_t)g * p.T * p.K + k;
    float a = 0.f;
    for (int t = 0; t < p.T; ++t)
        if (sc[(size_t)t * p.K] > 0) a += sl[(size_t)t * KD];
    p.g_E[(size_t)g * KD + e] = (p.beta * s) * a;
}

// k-means centroid update from the per-chunk cluster sums of vq_seg_sum_kernel<., 1>:  E_k <- mean of the points assigned to k;
// a cluster without points keeps its centroid (scipy.cluster.vq.kmeans2, missing='warn').  Chunks are added in order.
__global__ void kmeans_combine_kernel(const float* __restrict__ slab, const int* __restrict__ slab_cnt, int T, int K, int D,
                                      float* __restrict__ E, long long* __restrict__ counts) {
    const size_t KD = (size_t)K * D;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= KD) return;
    const int k = (int)(e / D);
    double a = 0.0;
    long long n = 0;
    for (int t = 0; t < T; ++t) {
        const int c = slab_cnt[(size_t)t * K + k];
        if (c > 0) { a += (double)slab[(size_t)t * KD + e]; n += c; }
    }
    if (n > 0) E[e] = (float)(a / (double)n);
    if (counts && e % D == 0) counts[k] = n;
}

// EMA apply (extension; textbook VQ-VAE EMA, see include/kvq.h)
__global__ void vq_ema_counts_kernel(const int64_t* idx, int64_t N, int K, unsigned* cnt) {
    const int g = blockIdx.y;
    const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n < N) atomicAdd(cnt + (size_t)g * K + idx[(size_t)g * N + n], 1u);
}

__global__ __launch_bounds__(256) void vq_ema_n_kernel(const unsigned* cnt, int K, float decay, float* ema_n,
                                                        double* tot_out) {
    __shared__ double sd[256];
    const int g = blockIdx.x, t = threadIdx.x;
    double a = 0.0;
    for (int k = t; k < K; k += 256) {
        const size_t i = (size_t)g * K + k;
        const float v = (float)((double)decay * (double)ema_n[i] + (1.0 - (double)decay) * (double)cnt[i]);
        ema_n[i] = v;
        a += (double)v;
    }
    sd[t] = a;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) sd[t] += sd[t + s];
        __syncthreads();
    }
    if (t == 0) tot_out[g] = sd[0];
}

__global__ void vq_ema_apply_kernel(const float* slab, const int* slab_cnt, int T, int K, int D, float decay, float eps,
                                    const float* ema_n, const double* tot, float* ema_m, float* E) {
    const int g = blockIdx.y;
    const size_t KD = (size_t)K * D;
    const size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= KD) return;
    const int k = (int)(e / D);
    const float* sl = slab + (size_t)g * T * KD + e;
    double sum = 0.0;
    for (int t = 0; t < T; ++t)
        if (slab_cnt[((size_t)g * T + t) * K + k] > 0) sum += (double)sl[(size_t)t * KD];
    const double tt = tot[g];
    const double nk = ((double)ema_n[(size_t)g * K + k] + (double)eps) / (tt + (double)K * (double)eps) * tt;
    const size_t i = (size_t)g * KD + e;
    const float m = (float)((double)decay * (double)ema_m[i] + (1.0 - (double)decay) * sum);
    ema_m[i] = m;
    E[i] = (float)((double)m / nk);
}

__global__ void vq_one_hot_kernel(const int64_t* __restrict__ idx, int64_t N, int K, float* __restrict__ enc) {
    const size_t total = (size_t)N * K;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t n = e / K;
        enc[e] = (int64_t)(e - n * K) == idx[n] ? 1.0f : 0.0f;
    }
}

// =============================================================================================================
// host side
// =============================================================================================================
static int pick_T(int64_t N, int K, int D) {
    // chunks of >= 256 tokens, as many as keep the slab [T,K,D] under 64 MiB; a chunk never exceeds SEG_MAX_CHUNK
    int64_t T = (N + 255) / 256;
    const int64_t cap = (64ll << 20) / ((int64_t)K * D * 4);
    if (T > cap) T = cap;
    const int64_t need = (N + SEG_MAX_CHUNK - 1) / SEG_MAX_CHUNK;
    if (T < need) T = need;
    if (T < 1) T = 1;
    return (int)T;
}
static int64_t chunk_of(int64_t N, int T) { return ((N + T - 1) / T + 255) / 256 * 256; }

struct WsLayout {
    size_t counts, sumsq, e2, keys, tickets, epack, slab, slab_cnt, total;
};
static int64_t n_tickets(int64_t N, int G) { return (int64_t)G * ((N + 63) / 64) + G; }
static WsLayout ws_layout(int64_t N, int K, int D, int G) {
    WsLayout l;
    size_t off = 0;
    const int T = pick_T(N, K, D);
    l.counts = off; off = align_up(off + (size_t)G * K * sizeof(unsigned), 256);
    l.sumsq = off;  off = align_up(off + (size_t)G * N * sizeof(double), 256);
    l.e2 = off;     off = align_up(off + (size_t)G * K * sizeof(float) + 16, 256);
    l.keys = off;   off = align_up(off + (size_t)G * N * sizeof(unsigned long long), 256);
    l.tickets = off; off = align_up(off + (size_t)n_tickets(N, G) * sizeof(unsigned), 256);
    l.epack = off;  off = align_up(off + (size_t)G * ((K + 31) / 32) * 32 * D * sizeof(float), 256);
    l.slab_cnt = off; off = align_up(off + (size_t)G * T * K * sizeof(int), 256);
    l.slab = off;   off = align_up(off + (size_t)G * T * K * D * sizeof(float), 256);
    l.total = off;
    return l;
}

static bool mfma_ok(int64_t N, int K, int D) { return N > 0 && K > 0 && D > 0 && D % 32 == 0; }

static void launch_pack(const float* E, int K, int D, int G, float* epack, hipStream_t st) {
    const int64_t slots = (int64_t)((K + 31) / 32) * (D / 8) * 64;
    const unsigned pb = (unsigned)((slots + 255) / 256 > 2048 ? 2048 : (slots + 255) / 256);
    hipLaunchKernelGGL(vq_pack_codebook_kernel, dim3(pb, (unsigned)G), dim3(256), 0, st, E, K, D, epack);
}

// distances + arg-min + epilogue (everything of the forward except the final scalar reduction); `packed`: p.epack already
// holds the fragment-ordered codebook (kvq_vq_pack_codebook)
// 1 (default): the distance kernel's last-arriving workgroups also run the epilogue and the final sums (one launch behind the
// init kernel); 0: distance kernel + vq_epilogue_kernel + vq_finalize_kernel (rounds 1 - 4; kept as the A/B arm and the checker
// of the fused tail -- kvq_vq_set_variant).  Per calling thread, like kvq_attn_set_variant.
static thread_local int g_vq_fused = 1;

static void launch_init(const FwdParams& p, int G, bool use_mfma, hipStream_t st) {
    const int64_t nk = use_mfma ? (int64_t)G * p.N : 0, nc = (int64_t)G * p.K, nt = use_mfma ? n_tickets(p.N, G) : 0;
    const int64_t most = nk > nc ? nk : nc;
    const unsigned blocks = (unsigned)((most + 255) / 256 > 1024 ? 1024 : (most + 255) / 256);
    hipLaunchKernelGGL(vq_init_kernel, dim3(blocks ? blocks : 1), dim3(256), 0, st, p.keys, nk, p.counts, nc, p.tickets, nt);
}

// everything of the forward; `packed`: p.epack already holds the fragment-ordered codebook (kvq_vq_pack_codebook)
template <int DT>
static int launch_forward(FwdParams p, int G, bool use_mfma, bool packed, hipStream_t st) {
    launch_init(p, G, use_mfma, st);
    int rc = check_launch("vq_init_kernel");
    if (rc) return rc;
    if (use_mfma) {
        if (!packed) launch_pack(p.E, p.K, p.D, G, const_cast<float*>(p.epack), st);
        const bool prof = prof_begin(st);
        dim3 grid((unsigned)((p.N + 63) / 64), (unsigned)((p.K + T2_CN - 1) / T2_CN), (unsigned)G);
        const size_t lds = 4 * 64 * T3_KC * sizeof(float);      // two pairs of z stage buffers
        if (p.D == 768) hipLaunchKernelGGL((vq_dist_packed_kernel<DT, true, 2, 24>), grid, dim3(T2_THREADS), lds, st, p);
        else hipLaunchKernelGGL((vq_dist_packed_kernel<DT, true, 2, 0>), grid, dim3(T2_THREADS), lds, st, p);
        if (prof) prof_end(st);
        rc = check_launch("vq_dist_packed_kernel");
        if (rc || p.fused) return rc;
        dim3 egrid((unsigned)((p.N + EP_TOK - 1) / EP_TOK), (unsigned)G);
        hipLaunchKernelGGL(vq_epilogue_kernel<DT>, egrid, dim3(EP_THREADS), 0, st, p);
        rc = check_launch("vq_epilogue_kernel");
    } else {
        const int64_t rows = (int64_t)G * p.K;
        hipLaunchKernelGGL(row_sq_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, p.E, rows, p.D,
                           const_cast<float*>(p.e2));
        const size_t lds = (size_t)GEN_WAVES * ((p.D + 7) & ~7) * sizeof(float);
        if (lds > 64 * 1024) return fail(KVQ_E_INVALID, "generic path: D=%d too large (needs D %% 32 == 0 above 4096)", p.D);
        dim3 grid((unsigned)((p.N + GEN_WAVES - 1) / GEN_WAVES), (unsigned)G);
        hipLaunchKernelGGL(vq_fwd_generic_kernel<DT>, grid, dim3(GEN_WAVES * WAVE), lds, st, p);
        rc = check_launch("vq_fwd_generic_kernel");
    }
    if (rc || !p.loss) return rc;                                // (the distance dump of the debug hook stops here)
    hipLaunchKernelGGL(vq_finalize_kernel, dim3((unsigned)G), dim3(FIN_THREADS), 0, st, p.tok_sumsq, p.counts, p.N, p.K, p.D,
                       p.beta, p.loss, p.perplexity, p.counts_f);
    return check_launch("vq_finalize_kernel");
}

}  // namespace kvq

using namespace kvq;

extern "C" {

#ifdef KVQ_VQ_DIAG
int kvq_vq_diag_set_buffer(void* buf) {       // diagnostic library only: [workgroups of the next launch][8] u64, or null
    return hipMemcpyToSymbol(HIP_SYMBOL(kvq::g_vq_diag), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

size_t kvq_vq_workspace_bytes(int64_t N, int K, int D, int G) {
    if (N <= 0 || K <= 0 || D <= 0 || G <= 0) return 0;
    return ws_layout(N, K, D, G).total;
}

int kvq_vq_uses_mfma(int64_t N, int K, int D) { return mfma_ok(N, K, D) ? 1 : 0; }

size_t kvq_vq_packed_bytes(int K, int D, int G) {
    return (size_t)G * ((K + 31) / 32) * 32 * D * sizeof(float);
}

int kvq_vq_pack_codebook(const float* E, int K, int D, int G, float* packed, void* stream) {
    KVQ_REQUIRE(E && packed && K > 0 && D > 0 && G > 0, "kvq_vq_pack_codebook: bad argument");
    KVQ_REQUIRE(D % 32 == 0, "kvq_vq_pack_codebook: the fragment-ordered copy exists for D %% 32 == 0 only (got %d)", D);
    KVQ_REQUIRE((((uintptr_t)E | (uintptr_t)packed) & 15) == 0, "kvq_vq_pack_codebook: E and packed must be 16-byte aligned");
    launch_pack(E, K, D, G, packed, (hipStream_t)stream);
    return check_launch("vq_pack_codebook_kernel");
}

static int vq_forward_impl(const void* z, const float* E, const float* packed, int64_t N, int K, int D, int G, int io_dtype, float beta,
                           void* z_q, int64_t* idx, float* loss, float* perplexity, float* counts, void* ws,
                           size_t ws_bytes, void* stream) {
    KVQ_REQUIRE(z && E && z_q && idx && loss && perplexity, "kvq_vq_forward: null pointer argument");
    KVQ_REQUIRE(N > 0 && K > 0 && D > 0 && G > 0, "kvq_vq_forward: N=%lld K=%d D=%d G=%d must be positive", (long long)N, K, D, G);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "kvq_vq_forward: unsupported io dtype %d", io_dtype);
    KVQ_REQUIRE(N < (1ll << 36), "kvq_vq_forward: N too large");
    const WsLayout l = ws_layout(N, K, D, G);
    if (!ws || ws_bytes < l.total) return fail(KVQ_E_WORKSPACE, "kvq_vq_forward: workspace %zu < %zu bytes", ws_bytes, l.total);
    KVQ_REQUIRE(((uintptr_t)ws & 255) == 0, "kvq_vq_forward: workspace must be 256-byte aligned");
    KVQ_REQUIRE(((uintptr_t)z & 15) == 0 && ((uintptr_t)E & 15) == 0 && ((uintptr_t)z_q & 15) == 0,
                "kvq_vq_forward: z, E, z_q must be 16-byte aligned");
    const bool fast = mfma_ok(N, K, D);
    KVQ_REQUIRE(!packed || (fast && ((uintptr_t)packed & 15) == 0), "kvq_vq_forward_packed: needs D %% 32 == 0 and a 16-byte aligned copy");
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    FwdParams p;
    p.z = z; p.E = E; p.z_q = z_q; p.idx = idx;
    p.tok_sumsq = (double*)(w + l.sumsq);
    p.counts = (unsigned*)(w + l.counts);
    p.e2 = (const float*)(w + l.e2);
    p.keys = (unsigned long long*)(w + l.keys);
    p.tickets = (unsigned*)(w + l.tickets);
    p.epack = packed ? packed : (const float*)(w + l.epack);
    p.dump = nullptr;
    p.loss = loss; p.perplexity = perplexity; p.counts_f = counts; p.beta = beta;
    p.fused = fast && g_vq_fused ? 1 : 0;
    p.N = N; p.K = K; p.D = D;
    return io_dtype == KVQ_F32 ? launch_forward<KVQ_F32>(p, G, fast, packed != nullptr, st)
                               : launch_forward<KVQ_BF16>(p, G, fast, packed != nullptr, st);
}

int kvq_vq_set_variant(int fused) {
    KVQ_REQUIRE(fused == 0 || fused == 1, "kvq_vq_set_variant: 0 (three kernels) or 1 (fused tail), got %d", fused);
    g_vq_fused = fused;
    return KVQ_OK;
}

int kvq_vq_forward(const void* z, const float* E, int64_t N, int K, int D, int G, int io_dtype, float beta,
                   void* z_q, int64_t* idx, float* loss, float* perplexity, float* counts, void* ws,
                   size_t ws_bytes, void* stream) {
    return vq_forward_impl(z, E, nullptr, N, K, D, G, io_dtype, beta, z_q, idx, loss, perplexity, counts, ws, ws_bytes, stream);
}

int kvq_vq_forward_packed(const void* z, const float* E, const float* packed, int64_t N, int K, int D, int G, int io_dtype,
                          float beta, void* z_q, int64_t* idx, float* loss, float* perplexity, float* counts, void* ws,
                          size_t ws_bytes, void* stream) {
    KVQ_REQUIRE(packed, "kvq_vq_forward_packed: null packed codebook");
    return vq_forward_impl(z, E, packed, N, K, D, G, io_dtype, beta, z_q, idx, loss, perplexity, counts, ws, ws_bytes, stream);
}

int kvq_vq_debug_distances(const void* z, const float* E, int64_t N, int K, int D, int io_dtype, int use_mfma,
                           float* d, void* stream) {
    KVQ_REQUIRE(z && E && d && N > 0 && K > 0 && D > 0, "kvq_vq_debug_distances: bad argument");
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    if (use_mfma && !mfma_ok(N, K, D)) return fail(KVQ_E_INVALID, "shape N=%lld K=%d D=%d has no MFMA path", (long long)N, K, D);
    // self-contained scratch (test hook only: allocation here is fine, this is never on the product path)
    const WsLayout l = ws_layout(N, K, D, 1);
    char* w = nullptr;
    void *zq = nullptr, *ix = nullptr;
    const size_t esz = io_dtype == KVQ_F32 ? 4 : 2;
    if (hipMalloc((void**)&w, l.total) != hipSuccess || hipMalloc(&zq, (size_t)N * D * esz) != hipSuccess ||
        hipMalloc(&ix, (size_t)N * 8) != hipSuccess)
        return fail(KVQ_E_LAUNCH, "debug scratch allocation failed");
    hipStream_t st = (hipStream_t)stream;
    FwdParams p;
    p.z = z; p.E = E; p.z_q = zq; p.idx = (int64_t*)ix;
    p.tok_sumsq = (double*)(w + l.sumsq); p.counts = (unsigned*)(w + l.counts); p.e2 = (const float*)(w + l.e2);
    p.keys = (unsigned long long*)(w + l.keys);
    p.tickets = (unsigned*)(w + l.tickets);
    p.epack = (const float*)(w + l.epack);
    p.loss = p.perplexity = p.counts_f = nullptr; p.beta = 0.f; p.fused = 0;
    p.dump = d; p.N = N; p.K = K; p.D = D;
    int rc = io_dtype == KVQ_F32 ? launch_forward<KVQ_F32>(p, 1, use_mfma != 0, false, st)
                                 : launch_forward<KVQ_BF16>(p, 1, use_mfma != 0, false, st);
    (void)hipStreamSynchronize(st);
    (void)hipFree(w); (void)hipFree(zq); (void)hipFree(ix);
    return rc;
}

int kvq_vq_backward(const void* z, const float* E, const int64_t* idx, const void* g_zq, const float* g_loss,
                    int64_t N, int K, int D, int G, int io_dtype, float beta, void* g_z, float* g_E, void* ws,
                    size_t ws_bytes, void* stream) {
    KVQ_REQUIRE(z && E && idx, "kvq_vq_backward: null pointer argument");
    KVQ_REQUIRE(N > 0 && K > 0 && D > 0 && G > 0, "kvq_vq_backward: sizes must be positive");
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "kvq_vq_backward: unsupported io dtype %d", io_dtype);
    const WsLayout l = ws_layout(N, K, D, G);
    if (g_E && (!ws || ws_bytes < l.total)) return fail(KVQ_E_WORKSPACE, "kvq_vq_backward: workspace %zu < %zu bytes", ws_bytes, l.total);
    hipStream_t st = (hipStream_t)stream;
    BwdParams p;
    p.z = z; p.E = E; p.idx = idx; p.g_zq = g_zq; p.g_loss = g_loss; p.g_z = g_z; p.g_E = g_E;
    p.slab = ws ? (float*)((char*)ws + l.slab) : nullptr;
    p.N = N; p.K = K; p.D = D; p.T = pick_T(N, K, D); p.chunk = chunk_of(N, p.T); p.beta = beta;
    p.slab_cnt = ws ? (int*)((char*)ws + l.slab_cnt) : nullptr;
    if (g_z) {
        dim3 grid((unsigned)((N + 3) / 4), (unsigned)G);
        if (io_dtype == KVQ_F32) hipLaunchKernelGGL(vq_bwd_gz_kernel<KVQ_F32>, grid, dim3(256), 0, st, p);
        else hipLaunchKernelGGL(vq_bwd_gz_kernel<KVQ_BF16>, grid, dim3(256), 0, st, p);
        int rc = check_launch("vq_bwd_gz_kernel");
        if (rc) return rc;
    }
    if (g_E) {
        dim3 grid((unsigned)K, (unsigned)p.T, (unsigned)G);
        if (io_dtype == KVQ_F32) hipLaunchKernelGGL((vq_seg_sum_kernel<KVQ_F32, 0>), grid, dim3(SEG_THREADS), 0, st, p);
        else hipLaunchKernelGGL((vq_seg_sum_kernel<KVQ_BF16, 0>), grid, dim3(SEG_THREADS), 0, st, p);
        int rc = check_launch("vq_seg_sum_kernel");
        if (rc) return rc;
        const size_t KD = (size_t)K * D;
        hipLaunchKernelGGL(vq_bwd_combine_kernel, dim3((unsigned)((KD + 255) / 256), (unsigned)G), dim3(256), 0, st, p);
        rc = check_launch("vq_bwd_combine_kernel");
        if (rc) return rc;
    }
    return KVQ_OK;
}

int kvq_vq_ema_update(const void* z, const int64_t* idx, int64_t N, int K, int D, int G, int io_dtype, float decay,
                      float eps, float* ema_n, float* ema_m, float* E, void* ws, size_t ws_bytes, void* stream) {
    KVQ_REQUIRE(z && idx && ema_n && ema_m && E, "kvq_vq_ema_update: null pointer argument");
    KVQ_REQUIRE(N > 0 && K > 0 && D > 0 && G > 0, "kvq_vq_ema_update: sizes must be positive");
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "kvq_vq_ema_update: unsupported io dtype %d", io_dtype);
    const WsLayout l = ws_layout(N, K, D, G);
    if (!ws || ws_bytes < l.total) return fail(KVQ_E_WORKSPACE, "kvq_vq_ema_update: workspace %zu < %zu bytes", ws_bytes, l.total);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    unsigned* cnt = (unsigned*)(w + l.counts);
    double* tot = (double*)(w + l.sumsq);
    BwdParams p;
    p.z = z; p.E = nullptr; p.idx = idx; p.g_zq = nullptr; p.g_loss = nullptr; p.g_z = nullptr; p.g_E = nullptr;
    p.slab = (float*)(w + l.slab);
    p.N = N; p.K = K; p.D = D; p.T = pick_T(N, K, D); p.chunk = chunk_of(N, p.T); p.beta = 0.f;
    p.slab_cnt = (int*)(w + l.slab_cnt);
    hipLaunchKernelGGL(vq_init_kernel, dim3((unsigned)(((int64_t)G * K + 255) / 256)), dim3(256), 0, st, (unsigned long long*)nullptr,
                       (int64_t)0, cnt, (int64_t)G * K, (unsigned*)nullptr, (int64_t)0);
    hipLaunchKernelGGL(vq_ema_counts_kernel, dim3((unsigned)((N + 255) / 256), (unsigned)G), dim3(256), 0, st, idx, N, K, cnt);
    hipLaunchKernelGGL(vq_ema_n_kernel, dim3((unsigned)G), dim3(256), 0, st, cnt, K, decay, ema_n, tot);
    dim3 grid((unsigned)K, (unsigned)p.T, (unsigned)G);
    if (io_dtype == KVQ_F32) hipLaunchKernelGGL((vq_seg_sum_kernel<KVQ_F32, 1>), grid, dim3(SEG_THREADS), 0, st, p);
    else hipLaunchKernelGGL((vq_seg_sum_kernel<KVQ_BF16, 1>), grid, dim3(SEG_THREADS), 0, st, p);
    const size_t KD = (size_t)K * D;
    hipLaunchKernelGGL(vq_ema_apply_kernel, dim3((unsigned)((KD + 255) / 256), (unsigned)G), dim3(256), 0, st, p.slab, p.slab_cnt, p.T, K, D,
                       decay, eps, ema_n, tot, ema_m, E);
    return check_launch("vq_ema_update");
}

int kvq_kmeans_update(const void* z, const int64_t* idx, int64_t N, int K, int D, int io_dtype, float* E, int64_t* counts,
                      void* ws, size_t ws_bytes, void* stream) {
    KVQ_REQUIRE(z && idx && E, "kvq_kmeans_update: null pointer argument");
    KVQ_REQUIRE(N > 0 && K > 0 && D > 0, "kvq_kmeans_update: sizes must be positive");
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "kvq_kmeans_update: unsupported io dtype %d", io_dtype);
    const WsLayout l = ws_layout(N, K, D, 1);
    if (!ws || ws_bytes < l.total) return fail(KVQ_E_WORKSPACE, "kvq_kmeans_update: workspace %zu < %zu bytes", ws_bytes, l.total);
    hipStream_t st = (hipStream_t)stream;
    char* w = (char*)ws;
    BwdParams p;
    p.z = z; p.E = nullptr; p.idx = idx; p.g_zq = nullptr; p.g_loss = nullptr; p.g_z = nullptr; p.g_E = nullptr;
    p.slab = (float*)(w + l.slab);
    p.N = N; p.K = K; p.D = D; p.T = pick_T(N, K, D); p.chunk = chunk_of(N, p.T); p.beta = 0.f;
    p.slab_cnt = (int*)(w + l.slab_cnt);
    dim3 grid((unsigned)K, (unsigned)p.T, 1u);
    if (io_dtype == KVQ_F32) hipLaunchKernelGGL((vq_seg_sum_kernel<KVQ_F32, 1>), grid, dim3(SEG_THREADS), 0, st, p);
    else hipLaunchKernelGGL((vq_seg_sum_kernel<KVQ_BF16, 1>), grid, dim3(SEG_THREADS), 0, st, p);
    int rc = check_launch("vq_seg_sum_kernel");
    if (rc) return rc;
    const size_t KD = (size_t)K * D;
    hipLaunchKernelGGL(kmeans_combine_kernel, dim3((unsigned)((KD + 255) / 256)), dim3(256), 0, st, p.slab, p.slab_cnt, p.T, K, D, E,
                       (long long*)counts);
    return check_launch("kmeans_combine_kernel");
}

int kvq_vq_one_hot(const int64_t* idx, int64_t N, int K, float* enc, void* stream) {
    KVQ_REQUIRE(idx && enc && N > 0 && K > 0, "kvq_vq_one_hot: bad argument");
    const size_t total = (size_t)N * K;
    unsigned blocks = (unsigned)((total + 255) / 256);
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(vq_one_hot_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, idx, N, K, enc);
    return check_launch("vq_one_hot_kernel");
}

}  // extern "C"
