// kvq_gemm2.hip -- the bf16 MFMA GEMM family of the training step (gfx950), all three operand layouts, grouped launches.
//
//   C[M,N] (bf16) (+)= op(A) . op(B)  (+ bias[N]),  f32 accumulation on v_mfma_f32_16x16x32_bf16
//
//   layout NT  A[M][K]  B[N][K]   forward projections  y = x . W^T + b          (modeling_bert.py:139-352, Bagon.py:46-53)
//          NN  A[M][K]  B[K][N]   input gradients      gx = gy . W              (autograd of the same nn.Linear)
//          TN  A[K][M]  B[K][N]   weight gradients     gW = gy^T . x            (contraction over the 8192 tokens)
//
// One structure for all of them:
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave-instruction, no VGPR hop) into a ring of NS k-tiles
//     (64 deep).  Every operand tile is a stack of 128-byte LDS rows whatever its layout in memory:
//       k-major operand ([rows][K] in memory)  image [row][64 k]   read by ds_read_b128,   16-byte chunks XORed with row & 7
//       m-major operand ([K][rows] in memory)  images [64 k][64 m] read by ds_read_b64_tr_b16 (the hardware transpose read:
//                                              the MFMA wants 8 consecutive k per lane, memory has them 2*ld bytes apart),
//                                              32-byte segments XORed with ((k>>1)&1) | ((k>>3)&1)<<1
//     both swizzles are applied on the DMA's per-lane SOURCE address (its LDS destination is lane-linear) and undone by the
//     reads; both are bank-conflict free for their read instruction (derivation in profiles/NOTES_r01-r03_design_and_experiments.md §2.3).
//   * one raw s_barrier per k-tile; the DMA of tile kt+NS is issued right behind the barrier that frees its slot and is waited
//     for NS-1 tiles later with a COUNTED s_waitcnt vmcnt (never 0 inside the loop): NS-1 tiles stay in flight per CU.
//   * every wave software-pipelines itself: the fragments of k-step s+1 are read from LDS while the MFMAs of k-step s run
//     (two named fragment sets, static indexing), so LDS latency never sits in front of the matrix pipe; with 8 waves per
//     workgroup (two per SIMD) the partner's MFMAs also cover the DMA issue slots.
//   * epilogue through LDS: bf16 tile -> whole row segments, bias / accumulate in f32, 16-byte stores.
//   * tiles are numbered so that each XCD (private L2) owns a contiguous band of tiles.
//   * a launch can cover several problems of one layout ("grouped"): the weight gradients of one transformer layer are ONE
//     launch of ~250 tiles of 128 x 256 -- one tile per CU over the whole 8192-token contraction, no split-K, no partial slabs.
#include <limits.h>
#include <stdlib.h>

#include <atomic>
#include <type_traits>

#include "kvq_common.h"

namespace kvq {
namespace g2 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef __attribute__((address_space(3))) s16x4* l4ptr_t;

constexpr int BK = 64;                 // k-tile depth
constexpr int ROWB = 128;              // bytes per LDS row
constexpr int MAX_PROBLEMS = 16;

struct Problem {
    const unsigned short* A;
    const unsigned short* B;
    unsigned short* C;
    const unsigned short* bias;        // [N] bf16 or null
    int M, N, K;
    int lda, ldb, ldc;
    int tiles_m, tiles_n;
    int tile0;                         // first tile id of this problem in the launch
    int band;                          // > 0: tile rows per band, row-fastest inside a band; < 0: -band tile COLUMNS per band (see locate_tile)
    unsigned mg_per_band, mg_full, mg_rem;   // division by multiplication (host-computed, exact for tile counts below 2^16): by the
                                       // tiles of a band, by the width of a full band, by the width of the last, narrower band
    int accumulate;                    // C += result
    unsigned short* C2;                // EPI_GELU: second output gelu(C)
    const unsigned short* H;           // EPI_DGELU: pre-activation h [M, N] (row stride ldc); C = (A.B) * gelu'(h)
    float* part;                       // EPI_DGELU: [tiles_m][N] f32 column sums of C over each tile's rows (bias-gradient partials)
                                       // EPI_CE: [M][tiles_n][4] softmax statistics of each row over each tile's columns < vlimit
    int vlimit;                        // EPI_CE: columns >= vlimit (vocabulary padding) do not take part
    const float* scaleA;               // fp8 kernel: per-tensor quantisation scales of the two operands (device scalars)
    const float* scaleB;
    // EPI_DROPRES: C = dropout(A.B + bias) + R, R = H ([M, N], row stride ldc) -- the input of the LayerNorm behind a BertSelfOutput /
    // BertOutput dense (modeling_bert.py:282-296, 339-352), with the masks of csrc/kvq_nn.hip's drln kernels (same Philox stream)
    unsigned long long seed;
    const unsigned long long* seed_off;
    unsigned site, thresh;
    float inv_keep;
    // EPI_GELU, optional (round 5): the fp8 (e4m3) copy of C2 = gelu(C) for the fp8 GEMM that reads it next (BertOutput.dense), row
    // stride ld8 bytes, quantised with the scale in st8[0]; this launch's amax goes to st8's partial slots (kvq_fp8_quantize_delayed)
    unsigned char* C8;
    float* st8;
    int ld8;
};

constexpr int EPI_NONE = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_CE = 3, EPI_DROPRES = 4;

// A/B switch of the one-tile-per-workgroup kernels' prologue (round 5; -DKVQ_G2_EARLY=0 builds the round-4 form)
#ifndef KVQ_G2_EARLY
#define KVQ_G2_EARLY 1
#endif
constexpr bool G2_EARLY_START = KVQ_G2_EARLY != 0;

// Diagnostic build only (-DKVQ_G2_DIAG, tools/build_diag.sh -> lib/libkvq_diag.so; the product library has none of this): one
// wave per workgroup stamps s_memtime at the phase boundaries of its tile into a buffer nothing else reads (guide, "In-kernel stamps").
#ifdef KVQ_G2_DIAG
__device__ unsigned long long* g_diag = nullptr;
__device__ __forceinline__ void g2_stamp(int slot, unsigned long long v, int row = -1) {
    // wave-uniform condition (scalar branch), one lane stores
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) == 0 && g_diag != nullptr) {
        unsigned vb = row >= 0 ? row : blockIdx.x;   // through an opaque VGPR: keeps the compiler from moving the kernel's own scalar
        asm volatile("" : "+v"(vb));       // blockIdx chain (tile origin -> DMA bases, "s" asm operands) to the vector unit
        if ((threadIdx.x & 63) == 0) __builtin_nontemporal_store(v, g_diag + (size_t)vb * 16 + slot);
    }
}
#define G2_STAMP(slot) g2_stamp((slot), __builtin_amdgcn_s_memtime())
#define G2_STAMP_VAL(slot, v) g2_stamp((slot), (unsigned long long)(v))
#define G2_STAMP_ROW(row, slot) g2_stamp((slot), __builtin_amdgcn_s_memtime(), (row))
#define G2_STAMP_ROW_VAL(row, slot, v) g2_stamp((slot), (unsigned long long)(v), (row))
#else
#define G2_STAMP(slot) do { } while (0)
#define G2_STAMP_VAL(slot, v) do { } while (0)
#define G2_STAMP_ROW(row, slot) do { } while (0)
#define G2_STAMP_ROW_VAL(row, slot, v) do { } while (0)
#endif

// GELU / GELU' of the fused epilogues, two elements at a time (v_pk_* on the f32 pairs a bf16 dword unpacks to).
//   Phi(x) = 1/2 (1 + erf(x / sqrt2)),  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below bf16 output rounding):
//   1 - erf(|y|) = t (a1 + t (a2 + t (a3 + t (a4 + t a5)))) exp(-y^2),  t = 1 / (1 + p |y|)
// With y = x / sqrt2 the exponential is exp(-x^2 / 2) -- the SAME one the density phi(x) of GELU' needs: one v_exp_f32 and one
// v_rcp_f32 per element for either function (the transcendental pipe runs at quarter rate; everything else is full-rate FMAs).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ void phi_pair(f32x2 x, f32x2& cdf, f32x2& e) {
    const f32x2 ax = __builtin_elementwise_abs(x);
    const f32x2 d = fma2(ax, f32x2{0.3275911f * 0.70710678118654752f, 0.3275911f * 0.70710678118654752f}, f32x2{1.0f, 1.0f});
    const f32x2 t = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
    const f32x2 xx = x * x * -0.72134752044448170f;                       // -x^2 / 2 * log2(e)
    e = f32x2{__builtin_amdgcn_exp2f(xx.x), __builtin_amdgcn_exp2f(xx.y)};
    // half the tail polynomial: 0.5 * (a1 .. a5)
    f32x2 q = fma2(t, f32x2{0.5f * 1.061405429f, 0.5f * 1.061405429f}, f32x2{0.5f * -1.453152027f, 0.5f * -1.453152027f});
    q = fma2(q, t, f32x2{0.5f * 1.421413741f, 0.5f * 1.421413741f});
    q = fma2(q, t, f32x2{0.5f * -0.284496736f, 0.5f * -0.284496736f});
    q = fma2(q, t, f32x2{0.5f * 0.254829592f, 0.5f * 0.254829592f});
    const f32x2 tail = q * t * e;                                          // 1 - Phi(|x|)
    cdf.x = x.x >= 0.f ? 1.0f - tail.x : tail.x;
    cdf.y = x.y >= 0.f ? 1.0f - tail.y : tail.y;
}
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
    f32x2 cdf, e;
    phi_pair(x, cdf, e);
    return x * cdf;
}
__device__ __forceinline__ f32x2 dgelu2(f32x2 x) {                         // Phi(x) + x phi(x)
    f32x2 cdf, e;
    phi_pair(x, cdf, e);
    return fma2(x * 0.39894228040143268f, e, cdf);
}
__device__ __forceinline__ f32x2 unpack_bf16x2(unsigned w) { return f32x2{__uint_as_float(w << 16), __uint_as_float(w & 0xffff0000u)}; }

struct Params {
    Problem p[MAX_PROBLEMS];
    int nprob;
    int ntiles;
};

template <int BM_, int BN_, int WM_, int WN_, bool AK_, bool BKM_, int NS_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, NS = NS_;
    static constexpr bool AK = AK_, BKM = BKM_;             // operand is k-major in memory
    static constexpr int WAVES = WM * WN, THREADS = 64 * WAVES;
    static constexpr int MINW = WAVES >= 8 ? WAVES / 4 : 2;  // waves per SIMD the register budget must allow: four-wave workgroups run in pairs
    static constexpr int TM = BM / WM, TN = BN / WN;        // wave tile
    static constexpr int FA = TM / 16, FB = TN / 16;        // fragments per k-step
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int PIECES = (BM + BN) / 8;            // 1-KiB DMA pieces per k-tile
    static constexpr int PPW = PIECES / WAVES;              // per wave
    static constexpr int CLD = BN * 2 + 16;                 // epilogue row stride (bytes)
    static constexpr int CPRP = 32;                         // epilogue: threads per tile row (>= BN / 8 chunks, power of two)
    static constexpr int RPP = THREADS / CPRP;              // epilogue: tile rows per pass
    static constexpr int EPI_BYTES = BM * CLD + RPP * BN * 4;   // bf16 tile + f32 column-sum scratch (EPI_DGELU)
    static constexpr int LDS = (NS * STAGE > EPI_BYTES) ? NS * STAGE : EPI_BYTES;
    static_assert(BN / 8 <= CPRP, "epilogue mapping");
    static_assert(PIECES % WAVES == 0, "DMA pieces must divide evenly over the waves");
    static_assert(TM % 16 == 0 && TN % 16 == 0 && BM % 64 == 0 && BN % 64 == 0, "tile shape");
    static_assert(LDS <= 160 * 1024, "LDS budget");
    static_assert(AK_ || TM % 64 == 0 || 64 % TM == 0, "m-major A: a wave tile must not straddle 64-column images unevenly");
};

// swizzle key of row k of a [64 k][64 m] image (32-byte segments)
__device__ __forceinline__ int trkey(int k) { return ((k >> 1) & 1) | (((k >> 3) & 1) << 1); }

// ---- staging: one k-tile of both operands into a ring slot ---------------------------------------------------------------
// A wave owns PPW of the tile's 1-KiB pieces (piece = q * WAVES + w; the A pieces come first, and BM/8 is a multiple of WAVES,
// so "is this an A piece" depends on q only).  Per piece the lane's byte offset from the operand's tile base never changes
// along k: it is computed once (one VGPR per piece); per k-tile only the two 64-bit scalar bases advance.  The DMA itself is
// issued from inline asm: hipcc orders every LDS read behind a builtin LDS-DMA with s_waitcnt vmcnt(0), which would drain the
// ring at every k-step; from asm the loads are invisible to its bookkeeping and are waited for by the counted waits below.
template <class C>
struct Stager {
    unsigned off[C::PPW];          // per-lane byte offset of this wave's piece q inside the operand tile
    unsigned long long baseA, baseB;   // scalar: tile base of the current k-tile (bytes)
    unsigned long long stepA, stepB;   // scalar: bytes per k-tile
    unsigned lds0;                 // scalar: LDS byte address of this wave's piece 0 in slot 0

    __device__ __forceinline__ void init(const Problem& pr, char* smem, int m0, int n0, int w, int lane) {
        const int r8 = lane >> 3, c8 = lane & 7;
#pragma unroll
        for (int q = 0; q < C::PPW; ++q) {
            const int piece = q * C::WAVES + w;
            const bool isA = q < (C::BM / 8) / C::WAVES;
            const int pp = isA ? piece : piece - C::BM / 8;
            const bool kmajor = isA ? C::AK : C::BKM;
            const int ld = isA ? pr.lda : pr.ldb;
            const int lim = isA ? pr.M : pr.N, o0 = isA ? m0 : n0;
            if (kmajor) {
                const int row = pp * 8 + r8;
                const int sc = c8 ^ (row & 7);
                int gr = o0 + row;
                gr = gr < lim ? gr : lim - 1;                               // clamped, never branched
                off[q] = (unsigned)(gr - o0) * (unsigned)ld * 2u + (unsigned)sc * 16u;
            } else {
                const int img = pp >> 3;
                const int krow = (pp & 7) * 8 + r8;
                const int sc = c8 ^ (trkey(krow) << 1);
                int gc = o0 + img * 64 + sc * 8;
                gc = gc + 8 <= lim ? gc : lim - 8;
                off[q] = (unsigned)krow * (unsigned)ld * 2u + (unsigned)(gc - o0) * 2u;
            }
        }
        baseA = (unsigned long long)(C::AK ? pr.A + (size_t)m0 * pr.lda : pr.A + m0);
        baseB = (unsigned long long)(C::BKM ? pr.B + (size_t)n0 * pr.ldb : pr.B + n0);
        stepA = C::AK ? (unsigned long long)BK * 2 : (unsigned long long)BK * 2 * pr.lda;
        stepB = C::BKM ? (unsigned long long)BK * 2 : (unsigned long long)BK * 2 * pr.ldb;
        lds0 = (unsigned)(size_t)(lptr_t)smem + (unsigned)w * 1024u;
    }

    // persistent tile loop: point the stager at another tile of the same problem (the ring, and with it lds0, stays)
    __device__ __forceinline__ void retarget(const Problem& pr, char* smem, int m0, int n0, int w, int lane) { init(pr, smem, m0, n0, w, lane); }

    // one piece of the next k-tile (the one the bases point at) into ring slot `slot`
    template <int Q>
    __device__ __forceinline__ void piece(int slot) {
        constexpr bool isA = Q < (C::BM / 8) / C::WAVES;
        const unsigned dst = lds0 + (unsigned)slot * (unsigned)C::STAGE + (unsigned)Q * (unsigned)(C::WAVES * 1024);
        const unsigned long long base = isA ? baseA : baseB;
        unsigned keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 0\n\t"
            "global_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(off[Q]), "s"(base), "s"(dst)
            : "memory");
    }
    template <int Q0, int Q1>
    __device__ __forceinline__ void piece_rt(int j, int slot) {           // j is a compile-time constant after unrolling
        if constexpr (Q0 < Q1) {
            if (j == 0) piece<Q0>(slot);
            else piece_rt<Q0 + 1, Q1>(j - 1, slot);
        }
    }
    __device__ __forceinline__ void advance() {
        baseA += stepA;
        baseB += stepB;
    }
    template <int Q0, int Q1>
    __device__ __forceinline__ void pieces(int slot) {
        if constexpr (Q0 < Q1) {
            piece<Q0>(slot);
            pieces<Q0 + 1, Q1>(slot);
        }
    }
    // the whole next k-tile at once (prologue)
    __device__ __forceinline__ void issue(int slot) {
        pieces<0, C::PPW>(slot);
        advance();
    }
};

// ---- fragment reads of one k-step (ks = 0, 1) ----------------------------------------------------------------------
template <class C>
struct Frags {
    bf16x8 a[C::FA];
    bf16x8 b[C::FB];
};

template <bool KMAJOR>
__device__ __forceinline__ bf16x8 read_frag(const char* op, int r0, int ks, int lane) {
    if (KMAJOR) {
        const int row = r0 + (lane & 15);
        const int chunk = ks * 4 + (lane >> 4);
        return *reinterpret_cast<const bf16x8*>(op + row * ROWB + ((chunk ^ (row & 7)) << 4));
    } else {
        const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
        const int img = r0 >> 6, seg = (r0 & 63) >> 4;
        const int key = ((q >> 1) & 1) | ((g & 1) << 1);                     // trkey(32ks + 8g + 4h + q)
        const char* base = op + img * 8192 + (ks * 32 + g * 8 + q) * ROWB + ((seg ^ key) << 5) + p * 8;
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((l4ptr_t)(base));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((l4ptr_t)(base + 4 * ROWB));
        typedef short s16x8 __attribute__((ext_vector_type(8)));
        const s16x8 v = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <class C>
__device__ __forceinline__ void read_frags(Frags<C>& f, const char* stage, int wm, int wn, int ks, int lane) {
#pragma unroll
    for (int mi = 0; mi < C::FA; ++mi) f.a[mi] = read_frag<C::AK>(stage, wm * C::TM + mi * 16, ks, lane);
#pragma unroll
    for (int ni = 0; ni < C::FB; ++ni) f.b[ni] = read_frag<C::BKM>(stage + C::A_BYTES, wn * C::TN + ni * 16, ks, lane);
}

// One MFMA cluster (a k-step of the wave tile) that also (a) reads the NEXT k-step's fragments from LDS and (b) issues the DMA
// pieces [Q0, Q1) of the pending ring refill, both interleaved with the MFMAs:
//   * a DMA piece occupies the CU's address path for ~16-20 cycles: 8 waves issuing theirs back to back behind the barrier stall
//     each other for hundreds of cycles with the matrix pipe idle; one piece every few MFMAs never queues;
//   * an LDS read issued between two MFMAs costs the matrix pipe nothing (MI355X_MICROARCH.md: <= 3 cycles per gap for two reads),
//     a block of 16-20 reads in front of the cluster costs their issue time with the pipe empty.
// The cluster is cut into NP + 1 sub-blocks by the pieces (pinned with sched_barrier); the reads go into the first sub-blocks so
// that they have landed when the cluster ends; inside a sub-block the scheduler is asked for an MFMA / read alternation.
template <class C, int Q0, int Q1>
__device__ __forceinline__ void mma(f32x4 (&acc)[C::FA][C::FB], const Frags<C>& f, Frags<C>& fn, const char* nstage, int nks,
                                    int wm, int wn, int lane, Stager<C>& sg, int slot, bool pending) {
    constexpr int NM = C::FA * C::FB, NP = Q1 - Q0, NSUB = NP + 1, NF = C::FA + C::FB;
    constexpr int RSUB = NSUB > 1 ? NSUB - 1 : 1;                       // sub-blocks that carry reads
    constexpr int RPF = (C::AK && C::BKM) ? 1 : 2;                      // upper bound of LDS instructions per fragment
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb) {
        const int m0 = sb * NM / NSUB, m1 = (sb + 1) * NM / NSUB;
        const int t0 = sb < RSUB ? sb * NF / RSUB : NF, t1 = sb < RSUB ? (sb + 1) * NF / RSUB : NF;
#pragma unroll
        for (int t = t0; t < t1; ++t) {
            if (t < C::FA) fn.a[t] = read_frag<C::AK>(nstage, wm * C::TM + t * 16, nks, lane);
            else fn.b[t - C::FA] = read_frag<C::BKM>(nstage + C::A_BYTES, wn * C::TN + (t - C::FA) * 16, nks, lane);
        }
#pragma unroll
        for (int i = m0; i < m1; ++i) {
            const int mi = i / C::FB, ni = i % C::FB;
            acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[ni], f.a[mi], acc[mi][ni], 0, 0, 0);
        }
#pragma unroll
        for (int i = m0; i < m1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                   // one MFMA
            if (i - m0 < t1 - t0) __builtin_amdgcn_sched_group_barrier(0x100, RPF, 0);           // one fragment's LDS reads
        }
        if (sb + 1 < NSUB) {
            __builtin_amdgcn_sched_barrier(0);
            if (pending) sg.template piece_rt<Q0, Q1>(sb, slot);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- which tile does this workgroup own ------------------------------------------------------------------------------------
struct TileId {
    int pi, tm, tn, m0, n0;
};
// n / d for n, d < 2^16 with mg = 2^32 / d + 1: scalar multiply-high instead of the ~30-instruction v_rcp sequence hipcc emits
// for an integer division (four of them sat between kernel entry and the first DMA of every workgroup)
__device__ __forceinline__ int div_mg(int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; }     // mg == 0: d == 1
// (tm, tn) of local tile lt: bands of `band` tile rows (row-fastest inside) or of -band tile columns (column-fastest inside)
__device__ __forceinline__ void band_walk(const Problem& pr, int lt, int& tm, int& tn) {
    const bool colb = pr.band < 0;
    const int w = colb ? -pr.band : pr.band;              // band width: tile columns (colb) or tile rows
    const int across = colb ? pr.tiles_m : pr.tiles_n;    // tiles swept per unit of width
    const int along = colb ? pr.tiles_n : pr.tiles_m;     // what the bands cut
    const int per_band = w * across;
    const int bnd = div_mg(lt, pr.mg_per_band), inb = lt - bnd * per_band;
    const int o0 = bnd * w;
    const bool last = along - o0 < w;                     // the last band may be narrower
    const int wb = last ? along - o0 : w;
    const int outer = div_mg(inb, last ? pr.mg_rem : pr.mg_full);
    const int inner = o0 + inb - outer * wb;
    tm = colb ? outer : inner;
    tn = colb ? inner : outer;
}
template <class C, bool SINGLE = false>
__device__ __forceinline__ TileId locate_tile(const Params& P, int id) {
    // XCD-aware numbering (bijective form), then problem lookup
    {
        const int nt = P.ntiles, q = nt >> 3, r = nt & 7, x = id & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    // One problem (every launch but the grouped weight gradients): its fields sit at fixed kernel-argument offsets and are
    // requested in one batch at kernel entry.  Several: the tile0 look-up and then the chosen problem's fields are two more
    // dependent trips to the scalar cache.
    int pi = 0;
    if constexpr (!SINGLE) {
#pragma unroll
        for (int i = 1; i < MAX_PROBLEMS; ++i)
            if (i < P.nprob && id >= P.p[i].tile0) pi = i;
    }
    const Problem& pr = P.p[pi];
    // Tile order inside the problem: bands of `band` tile rows, row-fastest inside a band: the tiles an XCD runs together share
    // `band` A tile rows (resident in its 4 MiB L2) and each B tile is fetched once per band instead of once per tile row
    // (fabric traffic per shape: profiles/r02_gemm_pmc.md).  band = 1: column-fastest.
    const int lt = id - pr.tile0;
    TileId t;
    t.pi = pi;
    // Tile order inside the problem.  band > 0: bands of `band` tile rows, row-fastest inside (band = 1: column-fastest).
    // band < 0 (wide outputs): -band tile COLUMNS per band, column-fastest inside, every tile row swept per band: the band's B
    // tiles (<= 3 MiB) stay in the XCD's L2 while the A row tiles stream past once per band; with row bands every XCD re-reads
    // ALL of B once per band of rows (LM head: 16 x 47 MB from the fabric)
    band_walk(pr, lt, t.tm, t.tn);
    t.m0 = t.tm * C::BM;
    t.n0 = t.tn * C::BN;
    return t;
}

// the bias values this lane will add in the epilogue: requested at kernel start, FB dependent round trips to L2 later otherwise
template <class C>
__device__ __forceinline__ void load_bias(const Problem& pr, int n0, int wn, int lane, u16x4 (&biasv)[C::FB]) {
    // No branch around a load: hipcc would wait vmcnt(0) behind each of them (three dependent round trips to L2 at the start of
    // every workgroup: 1.5 of the 2.7 us the stamps showed between kernel entry and the first DMA).  Without a bias the loads
    // read the first elements of B (always there, N of them at least) and the values are masked to zero.
    const int fk0 = lane >> 4;
    const unsigned short* bp = pr.bias ? pr.bias : pr.B;
    const unsigned short keep = pr.bias ? 0xffffu : 0u;
#pragma unroll
    for (int ni = 0; ni < C::FB; ++ni) {
        int nb = n0 + wn * C::TN + ni * 16 + 4 * fk0;
        nb = nb + 4 <= pr.N ? nb : pr.N - 4;
        const u16x4 t = *reinterpret_cast<const u16x4*>(bp + nb);
        biasv[ni] = u16x4{(unsigned short)(t.x & keep), (unsigned short)(t.y & keep), (unsigned short)(t.z & keep), (unsigned short)(t.w & keep)};
    }
}

// ---- epilogue: acc * mul -> (bias) -> bf16 tile in LDS -> whole row segments -> (+C | * gelu'(H)) -> global
template <class C, int EPI>
__device__ __forceinline__ void epilogue(const Problem& pr, char* smem, f32x4 (&acc)[C::FA][C::FB], const u16x4 (&biasv)[C::FB],
                                         int tid, int lane, int wm, int wn, int m0, int n0, int tm, float mul) {
    // (Storing straight from the accumulators -- 8 bytes per lane, 16 rows x 32 bytes per wave-instruction, no LDS -- was measured and
    //  lost: 49 vs 44 us on the FFN1 shape, 599 vs 370 us on the LM head; partial-line writes cost more than the LDS round trip.)
    // A thread owns ONE 16-byte column chunk of the tile and NIT rows (r = rr + it * RPP).  Whatever the row segments need from
    // memory (the old C of an accumulate, the pre-activation H) is requested for all NIT rows up front, before the transposition
    // through LDS: one round trip instead of NIT dependent ones.
    constexpr int CPR = C::BN / 8;                                         // 16-byte chunks per tile row
    constexpr int NIT = C::BM / C::RPP;
    const int c16 = tid % C::CPRP, rr = tid / C::CPRP;
    const int n = n0 + c16 * 8;
    const bool colok = c16 < CPR && n < pr.N;                              // N % 8 == 0: a chunk is inside or outside as a whole
    const bool accum = EPI == EPI_NONE && pr.accumulate != 0;
    const unsigned short* auxp = (EPI == EPI_DGELU || EPI == EPI_DROPRES) ? pr.H : pr.C;
    uint4 aux[NIT];
    unsigned long long dseed = 0;
    if constexpr (EPI == EPI_DROPRES) dseed = pr.seed + (pr.seed_off ? *pr.seed_off : 0ull);      // (scalar load, waited for in the store loop)
    if (EPI == EPI_DGELU || EPI == EPI_DROPRES || accum) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            int m = m0 + rr + it * C::RPP;
            m = m < pr.M ? m : pr.M - 1;                                   // clamped; out-of-range rows are never stored
            aux[it] = colok ? *reinterpret_cast<const uint4*>(auxp + (size_t)m * pr.ldc + n) : make_uint4(0, 0, 0, 0);
        }
    }
    // lane holds, for (mi, ni): row m = wm*TM + mi*16 + (lane & 15), columns n = wn*TN + ni*16 + 4*(lane >> 4) + 0..3
    const int frow = lane & 15, fk = lane >> 4;
    // EPI_CE: running softmax statistics of this lane's part of each of its FA rows, over the values AS STORED (bf16): maximum,
    // sum of exp(x - maximum), first arg-max.  Columns are visited in increasing order, so an equal value never replaces an earlier one.
    float cm[EPI == EPI_CE ? C::FA : 1], csum[EPI == EPI_CE ? C::FA : 1];
    int cbi[EPI == EPI_CE ? C::FA : 1];
    if constexpr (EPI == EPI_CE) {
#pragma unroll
        for (int mi = 0; mi < C::FA; ++mi) { cm[mi] = -INFINITY; csum[mi] = 0.f; cbi[mi] = INT_MAX; }
    }
#pragma unroll
    for (int ni = 0; ni < C::FB; ++ni) {
        const int nl = wn * C::TN + ni * 16 + 4 * fk;
        const u16x4 t = biasv[ni];
        const f32x4 bv = {bf16_to_f32(t.x), bf16_to_f32(t.y), bf16_to_f32(t.z), bf16_to_f32(t.w)};
#pragma unroll
        for (int mi = 0; mi < C::FA; ++mi) {
            const int ml = wm * C::TM + mi * 16 + frow;
            const f32x4 v = acc[mi][ni] * mul + bv;
            const u16x4 o = {f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
            *reinterpret_cast<u16x4*>(smem + ml * C::CLD + nl * 2) = o;
            if constexpr (EPI == EPI_CE) {
                const int col = n0 + nl;
                const float NEG = -INFINITY;
                const float x0 = col + 0 < pr.vlimit ? bf16_to_f32(o.x) : NEG, x1 = col + 1 < pr.vlimit ? bf16_to_f32(o.y) : NEG;
                const float x2 = col + 2 < pr.vlimit ? bf16_to_f32(o.z) : NEG, x3 = col + 3 < pr.vlimit ? bf16_to_f32(o.w) : NEG;
                const float vm = fmaxf(fmaxf(x0, x1), fmaxf(x2, x3));
                if (vm > cm[mi]) {                                         // (strictly greater: the first maximum keeps the arg-max)
                    csum[mi] *= __expf(cm[mi] - vm);                       // exp(-inf) = 0 the first time
                    cm[mi] = vm;
                    cbi[mi] = col + (x0 == vm ? 0 : x1 == vm ? 1 : x2 == vm ? 2 : 3);
                }
                if (cm[mi] > NEG) csum[mi] += (__expf(x0 - cm[mi]) + __expf(x1 - cm[mi])) + (__expf(x2 - cm[mi]) + __expf(x3 - cm[mi]));
                // (a variant in base 2 with the padding test and the maximum update behind workgroup- / wave-uniform branches
                //  measured SLOWER in the step: 18.78 against 18.55 ms)
            }
        }
    }
    if constexpr (EPI == EPI_CE) {
        // the four lanes fk = 0..3 of a row hold interleaved 4-column groups: merge (the arg-max by value, then by lower index),
        // then one entry per (row, wave column) into the scratch behind the bf16 tile; merged over the WN wave columns below
        float4* cest = reinterpret_cast<float4*>(smem + C::BM * C::CLD);
#pragma unroll
        for (int mi = 0; mi < C::FA; ++mi) {
            float m = cm[mi], sum = csum[mi];
            int bi = cbi[mi];
#pragma unroll
            for (int mk = 16; mk <= 32; mk <<= 1) {
                const float om = __shfl_xor(m, mk, 64), os = __shfl_xor(sum, mk, 64);
                const int obi = __shfl_xor(bi, mk, 64);
                const float nm = fmaxf(m, om);
                const float sa = m > -INFINITY ? sum * __expf(m - nm) : 0.f, sb = om > -INFINITY ? os * __expf(om - nm) : 0.f;
                bi = (om > m || (om == m && obi < bi)) ? obi : bi;
                m = nm; sum = sa + sb;
            }
            if (fk == 0) cest[(wm * C::TM + mi * 16 + frow) * C::WN + wn] = make_float4(m, sum, __int_as_float(bi), 0.f);
        }
    }
    __syncthreads();
    G2_STAMP(6);
    if constexpr (EPI == EPI_CE) {
        const float4* cest = reinterpret_cast<const float4*>(smem + C::BM * C::CLD);
        if (tid < C::BM && m0 + tid < pr.M) {
            float m = -INFINITY, sum = 0.f;
            int bi = INT_MAX;
#pragma unroll
            for (int q = 0; q < C::WN; ++q) {                              // wave columns in increasing column order
                const float4 e = cest[tid * C::WN + q];
                const int ebi = __float_as_int(e.z);
                const float nm = fmaxf(m, e.x);
                const float sa = m > -INFINITY ? sum * __expf(m - nm) : 0.f, sb = e.x > -INFINITY ? e.y * __expf(e.x - nm) : 0.f;
                bi = (e.x > m || (e.x == m && ebi < bi)) ? ebi : bi;
                m = nm; sum = sa + sb;
            }
            const int tn_ = n0 / C::BN;
            reinterpret_cast<float4*>(pr.part)[(size_t)(m0 + tid) * pr.tiles_n + tn_] = make_float4(m, sum, __int_as_float(bi), 0.f);
        }
    }
    float cs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const bool emit8 = EPI == EPI_GELU && pr.C8 != nullptr;               // (uniform)
    const float s8 = emit8 ? pr.st8[0] : 1.0f;
    float am8 = 0.f;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
        const int r = rr + it * C::RPP;
        const int m = m0 + r;
        if (colok && m < pr.M) {
            uint4 v = *reinterpret_cast<const uint4*>(smem + r * C::CLD + c16 * 16);
            unsigned* vn = reinterpret_cast<unsigned*>(&v);
            const size_t off = (size_t)m * pr.ldc + n;
            if (accum) {
                const unsigned* vo = reinterpret_cast<const unsigned*>(&aux[it]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = __uint_as_float(vn[u] << 16) + __uint_as_float(vo[u] << 16);
                    const float hi = __uint_as_float(vn[u] & 0xffff0000u) + __uint_as_float(vo[u] & 0xffff0000u);
                    vn[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
            }
            if constexpr (EPI == EPI_DROPRES) {
                // what drln_fwd_kernel does with the dense output y (here: the bf16 values of the LDS tile) and the residual, in
                // its order: y * keep-scale, + residual, rounded to the io dtype.  One Philox call covers 4 consecutive elements
                // of a row, counted row * N/4 + chunk.
                const unsigned* rv = reinterpret_cast<const unsigned*>(&aux[it]);
                float ks[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
                if (pr.thresh != 0) {                                      // (uniform: no dropout, no Philox rounds -- they are NOT free here)
                    const unsigned long long e4 = (unsigned long long)m * (unsigned)(pr.N >> 2) + (unsigned)(n >> 2);
                    const U4 k0 = drop_bits(dseed, pr.site, e4), k1 = drop_bits(dseed, pr.site, e4 + 1);
                    ks[0] = keep_scale(k0.x, pr.thresh, pr.inv_keep); ks[1] = keep_scale(k0.y, pr.thresh, pr.inv_keep);
                    ks[2] = keep_scale(k0.z, pr.thresh, pr.inv_keep); ks[3] = keep_scale(k0.w, pr.thresh, pr.inv_keep);
                    ks[4] = keep_scale(k1.x, pr.thresh, pr.inv_keep); ks[5] = keep_scale(k1.y, pr.thresh, pr.inv_keep);
                    ks[6] = keep_scale(k1.z, pr.thresh, pr.inv_keep); ks[7] = keep_scale(k1.w, pr.thresh, pr.inv_keep);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = __uint_as_float(vn[u] << 16) * ks[2 * u] + __uint_as_float(rv[u] << 16);
                    const float hi = __uint_as_float(vn[u] & 0xffff0000u) * ks[2 * u + 1] + __uint_as_float(rv[u] & 0xffff0000u);
                    vn[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
            }
            if (EPI == EPI_DGELU) {                                        // C = (A.B) * gelu'(h); column sums of what is stored
                const unsigned* hv = reinterpret_cast<const unsigned*>(&aux[it]);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x2 gv2 = unpack_bf16x2(vn[u]) * dgelu2(unpack_bf16x2(hv[u]));
                    const unsigned short lo = f32_to_bf16(gv2.x);
                    const unsigned short hi = f32_to_bf16(gv2.y);
                    cs[2 * u] += bf16_to_f32(lo);
                    cs[2 * u + 1] += bf16_to_f32(hi);
                    vn[u] = (unsigned)lo | ((unsigned)hi << 16);
                }
            }
            {   // streaming store: the tile is not read again by this kernel, and written through now it does not wait in L2 for
                // the end-of-kernel write-back (cold-operand probe: FFN1 forward 52.7 -> 49.5 us, 768 x 768 16.9 -> 15.2 us)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 vv = {v.x, v.y, v.z, v.w};
                __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(pr.C + off));
            }
            if (EPI == EPI_GELU) {                                         // second output a = gelu(h), h = the bf16 value just stored
                uint4 g;
                unsigned* gv = reinterpret_cast<unsigned*>(&g);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const f32x2 a2 = gelu2(unpack_bf16x2(vn[u]));
                    gv[u] = (unsigned)f32_to_bf16(a2.x) | ((unsigned)f32_to_bf16(a2.y) << 16);
                }
                {
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    const u32x4 gg = {g.x, g.y, g.z, g.w};
                    __builtin_nontemporal_store(gg, reinterpret_cast<u32x4*>(pr.C2 + off));
                }
                if (emit8) {
                    am8 = fmaxf(am8, amax8(g));
                    *reinterpret_cast<uint2*>(pr.C8 + (size_t)m * pr.ld8 + n) = quant8(g, s8);
                }
            }
        }
    }
    if (emit8) fp8_amax_note(am8, pr.st8, blockIdx.x);
    if (EPI == EPI_DGELU) {
        // column sums over the tile's rows: RPP partial rows through LDS (behind the bf16 tile), then one thread per column
        float* scratch = reinterpret_cast<float*>(smem + C::BM * C::CLD);
        if (c16 < CPR) {
#pragma unroll
            for (int j = 0; j < 8; ++j) scratch[rr * C::BN + c16 * 8 + j] = cs[j];
        }
        __syncthreads();
        if (tid < C::BN && n0 + tid < pr.N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < C::RPP; ++q) t += scratch[q * C::BN + tid];
            pr.part[(size_t)tm * pr.N + n0 + tid] = t;
        }
    }
}

// ---- epilogue without LDS (the persistent kernel's, epilogue_regs below): the accumulators go to memory from the registers they
// are in.  A lane of the 16x16x32 accumulator holds 4 consecutive columns (8 bytes of bf16) of one row; stored like that a
// wave-instruction writes 16 rows x 32 bytes and the memory system sees quarter lines (measured in round 2: LM head 599 vs
// 370 us).  One v_permlane16_swap per dword exchanges, between the lane pairs (fk, fk ^ 1), the quads of two neighbouring
// 16-column blocks: afterwards an even-fk lane holds 8 consecutive columns of block ni0 and its odd partner 8 consecutive
// columns of block ni0 + 1 -- one 16-byte store per lane, 16 rows x 64 contiguous bytes per wave-instruction, the other half of
// each 128-byte line by the wave's next instruction.  No LDS, no barrier; in the one-tile-per-workgroup kernel it ties with the
// LDS-transposed epilogue on every shape (measured at round 3 with a probe build since removed: 768^2 14.2 vs 14.3 us, FFN1 44.4 vs 44.4, LM
// head 362.8 vs 366.1) -- what it buys is that a persistent tile loop keeps its ring in flight across it.

// ---- the k loop of one tile.  On entry the ring holds k-tiles 0 .. min(NS, nkt) - 1 of the tile (issued, not yet waited for)
//      and `sg` points at k-tile NS; on exit every wave has read its last fragments (the ring is free).
template <class C>
__device__ __forceinline__ void mainloop(f32x4 (&acc)[C::FA][C::FB], Stager<C>& sg, char* smem, int nkt, int wm, int wn, int lane) {
    // EARLY (rings of three slots and more): the prologue issued k-tiles 0 .. NS-2 only; the last slot's first fill is issued
    // inside the first MFMA cluster (see gemm2_tile)
    constexpr bool EARLY = G2_EARLY_START && C::NS >= 3;
    constexpr int NPRO = EARLY ? C::NS - 1 : C::NS;                        // k-tiles the prologue issued
    if (nkt >= NPRO) wait_vm<(NPRO - 1) * C::PPW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    G2_STAMP(4);

    // ---- main loop: per k-tile
    //   MFMA(f0) || read f1 || 2nd half of the pending refill  |  f1 ready, tile kt+1 landed, barrier  |
    //   MFMA(f1) || read next f0 || 1st half of the refill of the slot just freed
#ifdef KVQ_G2_DIAG
    unsigned long long diag_vm = 0, diag_bar = 0;      // (diagnostic build) cycles of wave 0 in the k loop's vmcnt waits / barriers
#endif
    Frags<C> f0, f1;
    read_frags<C>(f0, smem, wm, wn, 0, lane);
    // pieces issued inside the first cluster after the barrier; a two-slot ring needs the whole refill there (the tile is
    // waited for at the very next barrier), a deeper ring spreads it over both clusters
    constexpr int PH = C::NS == 2 ? C::PPW : C::PPW / 2;
    int slot = 0, pslot = 0;
    bool pending = false;
    // One k-tile.  STEADY = a refill is pending on entry, another k-tile follows and another refill starts: true for every
    // k-tile but the first and the last NS.  With these three as compile-time constants the clusters carry no branch around
    // the DMA pieces (as run-time flags each piece sat behind an s_cbranch plus the v_cndmask / v_cmp pair hipcc builds for a
    // uniform bool: a basic-block cut every four or five MFMAs)
    auto ktile = [&](int kt, auto steady, auto first) {
        constexpr bool STEADY = decltype(steady)::value;
        constexpr bool FIRST = decltype(first)::value;                     // EARLY only: k-tile 0 also issues the WHOLE fill of slot NS-1
        const char* st = smem + slot * C::STAGE;
        __builtin_amdgcn_s_waitcnt(0xC07F);                                // lgkmcnt(0): f0 (its reads ended half a cluster ago)
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FIRST) {
            pending = C::NS - 1 < nkt;
            mma<C, 0, C::PPW>(acc, f0, f1, st, 1, wm, wn, lane, sg, C::NS - 1, pending);
        } else {
            mma<C, PH, C::PPW>(acc, f0, f1, st, 1, wm, wn, lane, sg, pslot, STEADY ? true : pending);
        }
        if (STEADY || pending) sg.advance();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                                // f1 arrived: this wave is done with `slot`
        pending = false;
        const bool more = STEADY || kt + 1 < nkt;
        const int nslot = slot + 1 == C::NS ? 0 : slot + 1;
        if (more) {
#ifdef KVQ_G2_DIAG
            const unsigned long long t_pre = __builtin_amdgcn_s_memtime();
            if (STEADY || kt + C::NS <= nkt) wait_vm<(C::NS - 2) * C::PPW>();
            else wait_vm<0>();
            const unsigned long long t_mid = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            const unsigned long long t_post = __builtin_amdgcn_s_memtime();
            diag_vm += t_mid - t_pre;                                      // waiting for this wave's DMA pieces of tile kt+1
            diag_bar += t_post - t_mid;                                    // waiting for the other waves
#else
            if (STEADY || kt + C::NS <= nkt) wait_vm<(C::NS - 2) * C::PPW>();   // tiles kt+2 .. kt+NS-1 may still be in flight
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();                                  // tile kt+1 landed for everybody; `slot` is free
#endif
            pending = STEADY || kt + C::NS < nkt;
            pslot = slot;
        }
        __builtin_amdgcn_sched_barrier(0);
        // (after the last k-tile this still reads a ring slot -- stale bytes, inside the ring, never used)
        mma<C, 0, PH>(acc, f1, f0, smem + nslot * C::STAGE, 0, wm, wn, lane, sg, pslot, STEADY ? true : pending);
        slot = nslot;
    };
    int kt = 0;
    if (nkt > 0) ktile(kt++, std::false_type{}, std::integral_constant<bool, EARLY>{});
    for (; kt + C::NS < nkt; ++kt) ktile(kt, std::true_type{}, std::false_type{});
    for (; kt < nkt; ++kt) ktile(kt, std::false_type{}, std::false_type{});
    __builtin_amdgcn_s_barrier();      // every wave has read its last fragments: the ring is free (it becomes the epilogue tile)
#ifdef KVQ_G2_DIAG
    G2_STAMP_VAL(12, diag_vm);
    G2_STAMP_VAL(13, diag_bar);
#endif
}
// (Round 3, measured and removed: an L2 prefetch for the weight-gradient layout.  The in-kernel timers above showed wave 0 of the
//  256 x 256 TN kernel a quarter of its k loop in the vmcnt wait for its own DMA pieces -- two ring slots mean a k-tile is asked
//  for one k-tile, ~1.4 us, ahead.  Every thread then touched one 128-byte line of the k-tile four further on with a 4-byte load
//  (one more entry in the vmcnt queue per k-tile, the counted waits adjusted): the vmcnt share fell to 0.6 %, the barrier share
//  rose from 13 % to 38 %, the k-tile took 2899 cycles instead of 2856 and the grouped launches 226 / 250 us instead of 215 / 231.
//  Wave 0's wait was slack, not the loop's pace.)

// one tile of problem `pr`: ring prologue, k loop, epilogue
template <class C, int EPI>
__device__ __forceinline__ void gemm2_tile(const Problem& pr, const TileId& ti, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / C::WN, wn = w % C::WN;
    const int m0 = ti.m0, n0 = ti.n0;
    const int nkt = pr.K / BK;

    f32x4 acc[C::FA][C::FB];
#pragma unroll
    for (int mi = 0; mi < C::FA; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::FB; ++ni) acc[mi][ni] = 0.f;
    u16x4 biasv[C::FB];
    load_bias<C>(pr, n0, wn, lane, biasv);

    // ---- prologue: fill the ring
    Stager<C> sg;
    sg.init(pr, smem, m0, n0, w, lane);
    G2_STAMP(10);
    // Rings of three slots: only NS - 1 k-tiles here.  The stamps put 0.5 us per 40-KB k-tile between "issued" and "issued" (the
    // CU's address path takes a DMA piece every ~30 cycles whoever asks), and the k loop could start 0.3 us after the FIRST k-tile
    // was issued: the last slot's fill waits for nothing in front of the loop, so it goes into the first MFMA cluster instead.
    constexpr int NPRO = (G2_EARLY_START && C::NS >= 3) ? C::NS - 1 : C::NS;
#pragma unroll
    for (int s = 0; s < NPRO; ++s) {
        if (s < nkt) sg.issue(s);
        if (s == 0) G2_STAMP(11);
    }
    G2_STAMP(3);
    mainloop<C>(acc, sg, smem, nkt, wm, wn, lane);
    G2_STAMP(5);
    epilogue<C, EPI>(pr, smem, acc, biasv, tid, lane, wm, wn, m0, n0, ti.tm, 1.0f);
#ifdef KVQ_G2_DIAG
    G2_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    G2_STAMP(8);
    G2_STAMP_VAL(9, __builtin_amdgcn_s_memrealtime());
#endif
}

// several problems in one grid (the grouped weight gradients): the problem of a tile is looked up in the argument struct
template <class C, int EPI>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm2_kernel(Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    G2_STAMP(0);
    G2_STAMP_VAL(1, __builtin_amdgcn_s_memrealtime());
    G2_STAMP_VAL(2, ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));
    const TileId ti = locate_tile<C, false>(P, blockIdx.x);
    gemm2_tile<C, EPI>(P.p[ti.pi], ti, smem);
}

// ONE problem (every launch of the step but the grouped weight gradients).  What the workgroup needs before its first DMA --
// the tile walk's constants, the operand and bias pointers, the leading dimensions, K, M, N -- are the FIRST 14 DWORDS of the kernel
// arguments, as scalars (four pairs of 16-bit numbers packed; a problem with a number above 65535 takes gemm2_kernel): with -amdgpu-kernarg-preload-count they arrive in SGPRs with the
// wave, no s_load and no wait (14 is what the hardware preloads; a by-value struct cannot be preloaded at all).  Everything else
// (C, bias, the epilogue's pointers) is read from `rest` when used.
template <class C, int EPI>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm2s_kernel(int ntiles_band, int tiles_mn, unsigned mg_per_band, unsigned mg_full,
                                                                          unsigned mg_rem, int K, const unsigned short* A, const unsigned short* B,
                                                                          const unsigned short* bias, int ld_ab, int mn, Problem rest) {
    const int ntiles = ntiles_band & 0xffff, band = ntiles_band >> 16;                  // (band is signed: arithmetic shift)
    const int tiles_m = tiles_mn & 0xffff, tiles_n = (int)((unsigned)tiles_mn >> 16);
    const int lda = ld_ab & 0xffff, ldb = (int)((unsigned)ld_ab >> 16), M = mn & 0xffff, N = (int)((unsigned)mn >> 16);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    G2_STAMP(0);
    G2_STAMP_VAL(1, __builtin_amdgcn_s_memrealtime());
    G2_STAMP_VAL(2, ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));
    Problem pr = rest;
    pr.band = band; pr.tiles_m = tiles_m; pr.tiles_n = tiles_n; pr.mg_per_band = mg_per_band; pr.mg_full = mg_full; pr.mg_rem = mg_rem;
    pr.K = K; pr.A = A; pr.B = B; pr.bias = bias; pr.lda = lda; pr.ldb = ldb; pr.M = M; pr.N = N; pr.tile0 = 0;
    TileId ti;
    {
        int id = blockIdx.x;
        const int q = ntiles >> 3, r = ntiles & 7, x = id & 7;               // XCD-aware numbering (see locate_tile)
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
        ti.pi = 0;
        band_walk(pr, id, ti.tm, ti.tn);
        ti.m0 = ti.tm * C::BM;
        ti.n0 = ti.tn * C::BN;
    }
    gemm2_tile<C, EPI>(pr, ti, smem);
}

// (A persistent form -- gridDim.x = CUs workgroups walking the tiles, the next tile's ring issued before this tile's epilogue, the
//  epilogue transposed through two LDS pass regions behind the ring -- was built and measured in round 2 and removed again:
//  256x192 tiles 41.7 vs 43.6 us on the FFN1 shape and 228 vs 236 us on the cross-K/V shape, but 394 vs 389 us on the LM head, and
//  256x256 tiles (16-row passes, 16 barriers per tile) 442 vs 357 us.  vmcnt being one in-order queue for loads and stores, the next
//  tile's first waits also wait for the previous tile's stores, which is most of what the overlap was meant to hide.)

// (Weight gradients -- TN, both operands streamed from HBM, each byte used by 3-12 tiles -- run at 1.0 us per 64-deep k-tile on
//  128 x 256 and 1.7 us on 256 x 256 with every CU busy (1.17 us with 9 tiles on 9 CUs).  Two attempts to buy that back with a
//  deeper ring were built, measured and removed in round 2: a 192 x 192 tile with three slots (129 us per encoder layer against
//  125 for 128 x 256) and 32-deep k-tiles for m-major operands -- five slots for 256 x 256, six for 128 x 256, one barrier per
//  32 -- which left the two-layer launch at 219 us (221 before) and cost 128 x 256 11 %.  Neither the prefetch distance nor the
//  bytes per k-tile sets that time; tools/gemm2_probe_wgrad.py is the probe.)

// ---- persistent form: one workgroup per CU walks its tiles, the ring NEVER drains -------------------------------------------------
// Stamps of the one-tile-per-workgroup kernel (tools/gemm2_stamps.py, profiles/r03_gemm_stamps.md): a 256 x 256 tile of the
// LM head lives 24 us, of which the 12 k-tiles take 14-16; the rest is start-up (2.9 us until the ring is issued, 1 us until
// its first tile has landed), the epilogue (1.6 us accumulators -> LDS, 2.1 us LDS -> stores, 0.9 us until they are done) and
// 0.6 us until the CU's next workgroup starts -- 9 us per tile in which the CU's L2 -> LDS path, the thing that bounds the
// k loop (~60 GB/s per CU), moves nothing.  Here the k-tiles of a workgroup's tiles form ONE stream: the refill issued while
// the last k-tiles of tile t are multiplied already belongs to tile t + 1, and the epilogue goes from the registers to memory
// (epilogue_regs: no LDS, no barrier), so the only gap in the stream is the epilogue's own issue time.
//   * the DMA loads and the epilogue's stores share the wave's in-order vmcnt queue: every counted wait of the loop stays
//     correct (a static count can only wait for MORE than it needs when stores sit between the loads), and the first refill
//     issued behind the stores is not waited for until NS - 1 k-tiles later
//   * tiles of one workgroup: b, b + G, b + 2G ... (G = gridDim.x, a multiple of 8): all on one XCD's residue, and at every
//     step the 32 CUs of an XCD work on 32 consecutive tiles of that XCD's band (locate_tile)
//   * no per-lane clamping anywhere: an edge tile's ORIGIN is pulled back inside the matrix (n0 = N - BN: it overlaps its
//     neighbour and both write the same values to the shared columns), so the stager's per-lane offsets are the same for every
//     tile and a tile switch is two scalar base addresses; nothing the epilogue needs (bias) is held in registers across the
//     k loop: the bias segment of a tile is one more DMA piece (wave 0) into a two-deep LDS buffer behind the ring
template <class C>
__device__ __forceinline__ void tile_origin(const Params& P, int id, int& m0, int& n0) {
    const Problem& pr = P.p[0];
    const int nt = P.ntiles, q = nt >> 3, r = nt & 7, x = id & 7;
    id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);                 // XCD-aware numbering (see locate_tile)
    int tm, tn;
    band_walk(pr, id, tm, tn);
    m0 = tm * C::BM < pr.M - C::BM ? tm * C::BM : pr.M - C::BM;
    n0 = tn * C::BN < pr.N - C::BN ? tn * C::BN : pr.N - C::BN;
}

constexpr int BIAS_SLOT = 1024;                      // bytes of one bias buffer behind the ring (one DMA piece)

// the bias segment [n0, n0 + BN) of a tile -> LDS bias buffer `par` (wave 0 only; lanes past the segment re-read its last chunk)
template <class C>
__device__ __forceinline__ void bias_dma(const Problem& pr, char* smem, int n0, int par, int lane) {
    const int c = lane < C::BN / 8 ? lane : C::BN / 8 - 1;
    const unsigned long long src = (unsigned long long)(pr.bias + n0 + c * 8);
    const unsigned dst = (unsigned)(size_t)(lptr_t)smem + (unsigned)(C::NS * C::STAGE + par * BIAS_SLOT);
    unsigned keep;
    asm volatile(
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %1, off\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(src), "s"(dst)
        : "memory");
}

// registers -> memory ("epilogue without LDS" above), bias from the LDS buffer, no masks (tile origins are inside the matrix)
template <class C, int EPI>
__device__ __forceinline__ void epilogue_regs(const Problem& pr, const char* biasl, f32x4 (&acc)[C::FA][C::FB], int lane, int wm, int wn,
                                              int m0, int n0) {
    static_assert(C::FB % 2 == 0, "column blocks are stored in pairs");
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int frow = lane & 15, fk = lane >> 4;
    const size_t coff = (size_t)(m0 + wm * C::TM + frow) * pr.ldc + n0 + wn * C::TN + (fk & 1) * 16 + 8 * (fk >> 1);
    unsigned short* crow = pr.C + coff;
    unsigned short* crow2 = EPI == EPI_GELU ? pr.C2 + coff : nullptr;
#pragma unroll
    for (int np = 0; np < C::FB / 2; ++np) {
        f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
        if (biasl) {
            const u16x4 t0 = *reinterpret_cast<const u16x4*>(biasl + (wn * C::TN + (2 * np) * 16 + 4 * fk) * 2);
            const u16x4 t1 = *reinterpret_cast<const u16x4*>(biasl + (wn * C::TN + (2 * np + 1) * 16 + 4 * fk) * 2);
            b0 = f32x4{bf16_to_f32(t0.x), bf16_to_f32(t0.y), bf16_to_f32(t0.z), bf16_to_f32(t0.w)};
            b1 = f32x4{bf16_to_f32(t1.x), bf16_to_f32(t1.y), bf16_to_f32(t1.z), bf16_to_f32(t1.w)};
        }
#pragma unroll
        for (int mi = 0; mi < C::FA; ++mi) {
            const f32x4 v0 = acc[mi][2 * np] + b0, v1 = acc[mi][2 * np + 1] + b1;
            unsigned x0 = (unsigned)f32_to_bf16(v0.x) | ((unsigned)f32_to_bf16(v0.y) << 16), x1 = (unsigned)f32_to_bf16(v0.z) | ((unsigned)f32_to_bf16(v0.w) << 16);
            unsigned y0 = (unsigned)f32_to_bf16(v1.x) | ((unsigned)f32_to_bf16(v1.y) << 16), y1 = (unsigned)f32_to_bf16(v1.z) | ((unsigned)f32_to_bf16(v1.w) << 16);
            const size_t eoff = (size_t)(mi * 16) * pr.ldc + np * 32;
            {
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const u32x4 vv = {s0.x, s1.x, s0.y, s1.y};
                __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(crow + eoff));
            }
            if constexpr (EPI == EPI_GELU) {                               // second output: gelu of the values as stored (bf16)
                const f32x2 g0 = gelu2(unpack_bf16x2(x0)), g1 = gelu2(unpack_bf16x2(x1));
                const f32x2 g2 = gelu2(unpack_bf16x2(y0)), g3 = gelu2(unpack_bf16x2(y1));
                x0 = (unsigned)f32_to_bf16(g0.x) | ((unsigned)f32_to_bf16(g0.y) << 16); x1 = (unsigned)f32_to_bf16(g1.x) | ((unsigned)f32_to_bf16(g1.y) << 16);
                y0 = (unsigned)f32_to_bf16(g2.x) | ((unsigned)f32_to_bf16(g2.y) << 16); y1 = (unsigned)f32_to_bf16(g3.x) | ((unsigned)f32_to_bf16(g3.y) << 16);
                const u32x2 s0 = __builtin_amdgcn_permlane16_swap(x0, y0, false, false);
                const u32x2 s1 = __builtin_amdgcn_permlane16_swap(x1, y1, false, false);
                const u32x4 vv = {s0.x, s1.x, s0.y, s1.y};
                __builtin_nontemporal_store(vv, reinterpret_cast<u32x4*>(crow2 + eoff));
            }
        }
    }
}

template <class C, int EPI>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm3_kernel(Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / C::WN, wn = w % C::WN;
    const Problem& pr = P.p[0];
    const int nkt = pr.K / BK;                       // >= NS (host check)
    const int G = gridDim.x, ntiles = P.ntiles;
    int tc = blockIdx.x;                             // the tile being multiplied
    int ts = tc, sk = 0, spar = 0;                   // the stager's cursor: tile, k-tile to issue next, parity of its tile count
    const int total = ((ntiles - 1 - (int)blockIdx.x) / G + 1) * nkt;      // k-tiles of this workgroup's stream
    int g = 0;                                       // k-tiles of the stream consumed so far
    const bool has_bias = pr.bias != nullptr;

    int m0, n0;
    tile_origin<C>(P, tc, m0, n0);
    Stager<C> sg;
    {
        Problem un = pr;                             // the per-lane offsets of an interior tile serve every tile (no clamping)
        un.M = INT_MAX; un.N = INT_MAX;
        sg.init(un, smem, 0, 0, w, lane);
    }
    auto aim = [&](int am0, int an0) {               // the two scalar bases of a tile's k-tile 0
        sg.baseA = (unsigned long long)(C::AK ? pr.A + (size_t)am0 * pr.lda : pr.A + am0);
        sg.baseB = (unsigned long long)(C::BKM ? pr.B + (size_t)an0 * pr.ldb : pr.B + an0);
    };
    aim(m0, n0);
    if (has_bias && w == 0) bias_dma<C>(pr, smem, n0, 0, lane);
    auto next_ktile = [&]() {                        // after a whole k-tile has been issued
        sg.advance();
        if (++sk == nkt) {
            sk = 0;
            ts += G;
            spar ^= 1;
            if (ts < ntiles) {
                int sm0, sn0;
                tile_origin<C>(P, ts, sm0, sn0);
                aim(sm0, sn0);
                if (has_bias && w == 0) bias_dma<C>(pr, smem, sn0, spar, lane);
            }
        }
    };
#pragma unroll
    for (int s = 0; s < C::NS; ++s) {
        sg.template pieces<0, C::PPW>(s);
        next_ktile();
    }
    f32x4 acc[C::FA][C::FB];
#pragma unroll
    for (int mi = 0; mi < C::FA; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::FB; ++ni) acc[mi][ni] = 0.f;

    wait_vm<(C::NS - 1) * C::PPW>();
    __builtin_amdgcn_s_barrier();
    Frags<C> f0, f1;
    read_frags<C>(f0, smem, wm, wn, 0, lane);
    constexpr int PH = C::NS == 2 ? C::PPW : C::PPW / 2;
    constexpr int NST = C::FA * (C::FB / 2) * (EPI == EPI_GELU ? 2 : 1);     // stores of one epilogue, per wave
    int slot = 0, pslot = 0, cpar = 0;
    bool pending = false;
    for (;;) {
        G2_STAMP_ROW(tc, 0);
        G2_STAMP_ROW_VAL(tc, 1, __builtin_amdgcn_s_memrealtime());
        G2_STAMP_ROW_VAL(tc, 2, ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32) | (unsigned)__builtin_amdgcn_s_getreg((31 << 11) | 4));
        for (int kt = 0; kt < nkt; ++kt, ++g) {
            if (kt == 1) G2_STAMP_ROW(tc, 3);
            if (kt == 2) G2_STAMP_ROW(tc, 4);
            const char* st = smem + slot * C::STAGE;
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_sched_barrier(0);
            mma<C, PH, C::PPW>(acc, f0, f1, st, 1, wm, wn, lane, sg, pslot, pending);
            if (pending) next_ktile();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            pending = false;
            const int nslot = slot + 1 == C::NS ? 0 : slot + 1;
            if (g + 1 < total) {                                           // the stream goes on (in this tile or the next)
                // first k-tile behind a tile boundary: the previous epilogue's NST stores are YOUNGER than every refill issued
                // so far -- they may stay in flight, the loop must not wait for them here (extra operations in the queue, such as
                // wave 0's bias piece, only make a counted wait more conservative: completion is in issue order)
                if (g + C::NS > total) wait_vm<0>();
                else if (kt == 0 && g > 0) wait_vm<(C::NS - 2) * C::PPW + NST>();
                else wait_vm<(C::NS - 2) * C::PPW>();
                __builtin_amdgcn_s_barrier();
                pending = ts < ntiles;
                pslot = slot;
            }
            __builtin_amdgcn_sched_barrier(0);
            mma<C, 0, PH>(acc, f1, f0, smem + nslot * C::STAGE, 0, wm, wn, lane, sg, pslot, pending);
            slot = nslot;
        }
        // ---- tile finished: the rest of the pending refill first (the stream must not wait for the stores), then the stores
        if constexpr (PH < C::PPW) {
            if (pending) {
                sg.template pieces<PH, C::PPW>(pslot);
                next_ktile();
                pending = false;
            }
        }
        G2_STAMP_ROW(tc, 5);
        epilogue_regs<C, EPI>(pr, has_bias ? smem + C::NS * C::STAGE + cpar * BIAS_SLOT : nullptr, acc, lane, wm, wn, m0, n0);
        G2_STAMP_ROW(tc, 7);
        G2_STAMP_ROW_VAL(tc, 9, __builtin_amdgcn_s_memrealtime());
        tc += G;
        if (tc >= ntiles) break;
        cpar ^= 1;
        tile_origin<C>(P, tc, m0, n0);
#pragma unroll
        for (int mi = 0; mi < C::FA; ++mi)
#pragma unroll
            for (int ni = 0; ni < C::FB; ++ni) acc[mi][ni] = 0.f;
    }
}

// ---- fp8 (OCP e4m3) operands, NT layout: forward projections of BASELINE.json configs[4] --------------------------------------
//   C[M,N] (bf16) = (A8[M,K] . B8[N,K]^T) / (sA * sB) + bias,   A8 = sat(x * sA), B8 = sat(W * sB)  (per-tensor scales, on device)
// Same ring, same LDS images and the same two conflict-free ds_read_b128 per fragment as the bf16 kernel -- a 128-byte LDS row
// now holds 128 contraction elements -- but ONE block-scaled v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales) per k-tile
// and fragment pair: 2x the contraction per staged byte and 2x the bf16 matrix rate.  Which k a byte of the 32-byte operand
// belongs to does not matter as long as A and B agree (both are read the same way): the instruction sums over all 128.
// One MFMA cluster per k-tile; the next tile's fragments are read into a second register set inside it.
typedef int i32x8 __attribute__((ext_vector_type(8)));

template <class C>
struct Frags8 {
    bf16x8 a[2][C::FA];
    bf16x8 b[2][C::FB];
};

template <class C>
__device__ __forceinline__ i32x8 join(bf16x8 lo, bf16x8 hi) {
    typedef int i32x4 __attribute__((ext_vector_type(4)));
    const i32x4 l = __builtin_bit_cast(i32x4, lo), h = __builtin_bit_cast(i32x4, hi);
    const i32x8 v = {l.x, l.y, l.z, l.w, h.x, h.y, h.z, h.w};
    return v;
}

template <class C>
__device__ __forceinline__ void cluster8(f32x4 (&acc)[C::FA][C::FB], const Frags8<C>& f, Frags8<C>& fn, const char* nstage,
                                         int wm, int wn, int lane, Stager<C>& sg, int slot, bool pending) {
    constexpr int NM = C::FA * C::FB, NP = C::PPW, NSUB = NP + 1, NF = 2 * (C::FA + C::FB), RSUB = NSUB - 1;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int sb = 0; sb < NSUB; ++sb) {
        const int m0 = sb * NM / NSUB, m1 = (sb + 1) * NM / NSUB;
        const int t0 = sb < RSUB ? sb * NF / RSUB : NF, t1 = sb < RSUB ? (sb + 1) * NF / RSUB : NF;
#pragma unroll
        for (int t = t0; t < t1; ++t) {                                   // the next k-tile's fragments: a (both halves), then b
            const int ks = t & 1, fi = t >> 1;
            if (fi < C::FA) fn.a[ks][fi] = read_frag<true>(nstage, wm * C::TM + fi * 16, ks, lane);
            else fn.b[ks][fi - C::FA] = read_frag<true>(nstage + C::A_BYTES, wn * C::TN + (fi - C::FA) * 16, ks, lane);
        }
#pragma unroll
        for (int i = m0; i < m1; ++i) {
            const int mi = i / C::FB, ni = i % C::FB;
            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(join<C>(f.b[0][ni], f.b[1][ni]), join<C>(f.a[0][mi], f.a[1][mi]),
                                                                            acc[mi][ni], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
#pragma unroll
        for (int i = m0; i < m1; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (i - m0 < t1 - t0) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        if (sb + 1 < NSUB) {
            __builtin_amdgcn_sched_barrier(0);
            if (pending) sg.template piece_rt<0, C::PPW>(sb, slot);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

template <class C, int EPI = EPI_NONE>
__global__ __launch_bounds__(C::THREADS, C::MINW) void gemm2_f8_kernel(Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / C::WN, wn = w % C::WN;
    const TileId ti = locate_tile<C, true>(P, blockIdx.x);
    const Problem& pr = P.p[0];
    const int m0 = ti.m0, n0 = ti.n0;
    const int nkt = pr.K / BK;                 // K counted in bf16-sized units (= 2 fp8 elements): 64 units = 128 bytes per row

    f32x4 acc[C::FA][C::FB];
#pragma unroll
    for (int mi = 0; mi < C::FA; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::FB; ++ni) acc[mi][ni] = 0.f;
    u16x4 biasv[C::FB];
    load_bias<C>(pr, n0, wn, lane, biasv);
    const float mul = 1.0f / (pr.scaleA[0] * pr.scaleB[0]);

    Stager<C> sg;
    sg.init(pr, smem, m0, n0, w, lane);
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
        if (s < nkt) sg.issue(s);
    if (nkt >= C::NS) wait_vm<(C::NS - 1) * C::PPW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    Frags8<C> f0, f1;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int mi = 0; mi < C::FA; ++mi) f0.a[ks][mi] = read_frag<true>(smem, wm * C::TM + mi * 16, ks, lane);
#pragma unroll
        for (int ni = 0; ni < C::FB; ++ni) f0.b[ks][ni] = read_frag<true>(smem + C::A_BYTES, wn * C::TN + ni * 16, ks, lane);
    }
    int slot = 0;
    // per k-tile: [this tile's fragments landed; tile kt+1 landed for everybody; slot kt free] -> one cluster: MFMAs of tile kt ||
    // fragment reads of tile kt+1 || the DMA refill of slot kt.  Two named fragment sets: the loop body is written out twice.
    for (int kt = 0; kt < nkt; kt += 2) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            if (kt + half < nkt) {
                const int k = kt + half;
                __builtin_amdgcn_s_waitcnt(0xC07F);                            // lgkmcnt(0): the fragments of tile k
                bool pending = false;
                const int nslot = slot + 1 == C::NS ? 0 : slot + 1;
                if (k + 1 < nkt) {
                    if (k + C::NS <= nkt) wait_vm<(C::NS - 2) * C::PPW>();
                    else wait_vm<0>();
                    __builtin_amdgcn_s_barrier();
                    pending = k + C::NS < nkt;
                }
                __builtin_amdgcn_sched_barrier(0);
                if (half == 0) cluster8<C>(acc, f0, f1, smem + nslot * C::STAGE, wm, wn, lane, sg, slot, pending);
                else cluster8<C>(acc, f1, f0, smem + nslot * C::STAGE, wm, wn, lane, sg, slot, pending);
                if (pending) sg.advance();
                __builtin_amdgcn_sched_barrier(0);
                slot = nslot;
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    epilogue<C, EPI>(pr, smem, acc, biasv, tid, lane, wm, wn, m0, n0, ti.tm, mul);
}


// (Round 5, measured and removed: a STAGGERED form of the 256 x 256 tile -- the guide's "256^2 eight-phase" idiom: four phases per k-tile
//  (one quadrant of the wave tile = 16 MFMAs each), a phase = fragment reads + two DMA pieces | barrier | MFMAs | barrier, the two wave
//  rows one barrier apart so that one is in its MFMA block while the other reads.  Bit-identical to the kernels above and 2 - 13 %
//  SLOWER everywhere: grouped weight gradients of two decoder layers 252.9 against 223.9 us, [18432, 768] x 8192 206 against 195, LM
//  head 366 against 340, 8192^3 1512 against 1535 TF (profiles/r05_probe_staggered.txt).  Eight barriers per k-tile cost more than
//  the one barrier's bubble; with two waves per SIMD the intra-wave overlap of reads and MFMAs above is the better use of the pipe.
//  On the guide's own reference problems the kernels above hold 1357 / 1377 TF at 4096^3 and 1535 TF at 8192^3 (its template: 1320 - 1340
//  and ~1470; hipBLASLt: 1459 and 1621).)

// ---- tile configurations ---------------------------------------------------------------------------------------------
//                 BM   BN  WM WN  A k-major  B k-major  ring
template <bool AK, bool BKM> using Cfg128x256 = Cfg<128, 256, 2, 4, AK, BKM, 3>;     // 8 waves, 144 KiB: one tile per CU (wgrad)
template <bool AK, bool BKM> using Cfg256x192 = Cfg<256, 192, 4, 2, AK, BKM, 2>;     // 8 waves, 112 KiB
template <bool AK, bool BKM> using Cfg128x192 = Cfg<128, 192, 2, 4, AK, BKM, 3>;     // 8 waves, 120 KiB: 256 tiles at N = 768
template <bool AK, bool BKM> using Cfg256x256 = Cfg<256, 256, 2, 4, AK, BKM, 2>;     // 8 waves, 128 KiB
template <bool AK, bool BKM> using Cfg128x192p = Cfg<128, 192, 4, 2, AK, BKM, 3>;    // persistent form: an even number of 16-column blocks per wave
template <bool AK, bool BKM> using Cfg64x128 = Cfg<64, 128, 1, 4, AK, BKM, 3>;       // 4 waves, 72 KiB: two workgroups per CU, small outputs
template <bool AK, bool BKM> using Cfg128x192h = Cfg<128, 192, 2, 2, AK, BKM, 2>;    // 4 waves, 80 KiB: TWO workgroups per CU, each one wave
                                                                                     // per SIMD; one's start-up / epilogue under the other's k loop

template <class C, int EPI = EPI_NONE>
static int launch_cfg(const Params& P, hipStream_t st) {
    static std::atomic<bool> attr_done{false};      // per instantiation; idempotent, so a race only repeats the call
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<C, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e == hipSuccess) e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2s_kernel<C, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "hipFuncSetAttribute(gemm2): %s", hipGetErrorString(e));
        attr_done = true;
    }
    const Problem& q = P.p[0];
    const bool packs = P.nprob == 1 && P.ntiles < 65536 && q.M < 65536 && q.N < 65536 && q.lda < 65536 && q.ldb < 65536;
    if (!packs) {
        hipLaunchKernelGGL((gemm2_kernel<C, EPI>), dim3((unsigned)P.ntiles), dim3(C::THREADS), C::LDS, st, P);
    } else {
        hipLaunchKernelGGL((gemm2s_kernel<C, EPI>), dim3((unsigned)P.ntiles), dim3(C::THREADS), C::LDS, st, (int)((unsigned)P.ntiles | ((unsigned)q.band << 16)),
                           (int)((unsigned)q.tiles_m | ((unsigned)q.tiles_n << 16)), q.mg_per_band, q.mg_full, q.mg_rem, q.K, q.A, q.B, q.bias,
                           (int)((unsigned)q.lda | ((unsigned)q.ldb << 16)), (int)((unsigned)q.M | ((unsigned)q.N << 16)), q);
    }
    return check_launch("gemm2_kernel");
}

static int persistent_grid() {
    static int cus = 0;
    if (!cus) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n / 8 * 8 > 0 ? n / 8 * 8 : 8;        // a multiple of the 8 XCDs: a workgroup's tiles then stay on one XCD's residue
    }
    return cus;
}

template <class C, int EPI>
static int launch_persistent(const Params& P, hipStream_t st) {
    if (P.nprob != 1 || P.p[0].accumulate) return fail(KVQ_E_INVALID, "kvq_gemm (persistent): one problem, no accumulate");
    if (P.p[0].K / BK < C::NS) return fail(KVQ_E_INVALID, "kvq_gemm (persistent): K=%d below %d k-tiles", P.p[0].K, C::NS);
    if (P.p[0].M < C::BM || P.p[0].N < C::BN) return fail(KVQ_E_INVALID, "kvq_gemm (persistent): the matrix must hold one whole tile");
    constexpr int LDSP = C::NS * C::STAGE + 2 * BIAS_SLOT;
    static_assert(LDSP <= 160 * 1024, "LDS budget (persistent)");
    static std::atomic<bool> attr_done{false};
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm3_kernel<C, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDSP);
        if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "hipFuncSetAttribute(gemm3): %s", hipGetErrorString(e));
        attr_done = true;
    }
    const int grid = P.ntiles < persistent_grid() ? P.ntiles : persistent_grid();
    hipLaunchKernelGGL((gemm3_kernel<C, EPI>), dim3((unsigned)grid), dim3(C::THREADS), LDSP, st, P);
    return check_launch("gemm3_kernel");
}

template <template <bool, bool> class T>
static int launch_layout(int layout, const Params& P, hipStream_t st) {
    switch (layout) {
        case KVQ_GEMM_NT: return launch_cfg<T<true, true>>(P, st);
        case KVQ_GEMM_NN: return launch_cfg<T<true, false>>(P, st);
        case KVQ_GEMM_TN: return launch_cfg<T<false, false>>(P, st);
    }
    return fail(KVQ_E_INVALID, "kvq_gemm_bf16: unknown layout %d", layout);
}

static void tile_of(int tile, int& bm, int& bn) {
    switch (tile) {
        case KVQ_GEMM_TILE_128x256: bm = 128; bn = 256; break;
        case KVQ_GEMM_TILE_256x192: bm = 256; bn = 192; break;
        case KVQ_GEMM_TILE_256x256: bm = 256; bn = 256; break;
        case KVQ_GEMM_TILE_64x128: bm = 64; bn = 128; break;
        case KVQ_GEMM_TILE_128x192H: bm = 128; bn = 192; break;
        default: bm = 128; bn = 192; break;
    }
}

}  // namespace g2
}  // namespace kvq

using namespace kvq;

static int build_params(const kvq_gemm_problem* probs, int nprob, int layout, int tile, g2::Params& P, const char* who) {
    KVQ_REQUIRE(probs && nprob >= 1 && nprob <= g2::MAX_PROBLEMS, "%s: 1..%d problems per launch", who, g2::MAX_PROBLEMS);
    KVQ_REQUIRE(layout == KVQ_GEMM_NT || layout == KVQ_GEMM_NN || layout == KVQ_GEMM_TN, "%s: unknown layout %d", who, layout);
    const bool persistent = (tile & KVQ_GEMM_PERSISTENT) != 0;
    tile &= ~KVQ_GEMM_PERSISTENT;
    KVQ_REQUIRE(tile >= KVQ_GEMM_TILE_128x192 && tile <= KVQ_GEMM_TILE_128x192H, "%s: unknown tile %d", who, tile);
    KVQ_REQUIRE(!(persistent && (tile == KVQ_GEMM_TILE_64x128 || tile == KVQ_GEMM_TILE_128x192H)), "%s: the four-wave tiles have no persistent form", who);
    int bm, bn;
    g2::tile_of(tile, bm, bn);
    (void)persistent;
    P.nprob = nprob;
    int t0 = 0;
    for (int i = 0; i < nprob; ++i) {
        const kvq_gemm_problem& q = probs[i];
        KVQ_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0, "%s: problem %d: bad argument", who, i);
        KVQ_REQUIRE(q.K % g2::BK == 0, "%s: problem %d: K=%d must be a multiple of %d", who, i, q.K, g2::BK);
        KVQ_REQUIRE(q.M % 8 == 0 && q.N % 8 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 8 == 0 && q.M >= 8 && q.N >= 8,
                    "%s: problem %d: M, N, lda, ldb, ldc must be multiples of 8", who, i);
        KVQ_REQUIRE((((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) & 15) == 0 && (!q.bias || ((uintptr_t)q.bias & 7) == 0),
                    "%s: problem %d: operands must be 16-byte aligned", who, i);
        // leading dimensions must cover the rows the kernel reads
        const int a_cols = layout == KVQ_GEMM_TN ? q.M : q.K, b_cols = layout == KVQ_GEMM_NT ? q.K : q.N;
        KVQ_REQUIRE(q.lda >= a_cols && q.ldb >= b_cols && q.ldc >= q.N, "%s: problem %d: leading dimension too small", who, i);
        g2::Problem& d = P.p[i];
        d.A = (const unsigned short*)q.A; d.B = (const unsigned short*)q.B; d.C = (unsigned short*)q.C; d.bias = (const unsigned short*)q.bias;
        d.M = q.M; d.N = q.N; d.K = q.K; d.lda = q.lda; d.ldb = q.ldb; d.ldc = q.ldc;
        d.tiles_m = (q.M + bm - 1) / bm; d.tiles_n = (q.N + bn - 1) / bn;
        d.tile0 = t0; d.accumulate = q.accumulate;
        // Tile order.  A few tile columns: column-fastest.  Wide outputs (>= 8 tile columns): COLUMN bands of 6 tile columns, every
        // tile row swept per band -- the band's B tiles (6 x 256 x 768 x 2 B = 2.4 MB) stay in the XCD's L2 and the A row tiles
        // stream past once per band; with bands of two tile ROWS (round 2) every XCD pulled all of B through the fabric once per
        // band.  tools/gemm2_probe_band.py on MI355X: LM head 362.6 -> 354.7 us (one tile per workgroup), 430 -> 362 (persistent);
        // cross-K/V persistent 208 -> 203; 4, 8 or 9 columns and taller row bands are all slower.
        d.band = d.tiles_n >= 8 ? -6 : 1;
        if (getenv("KVQ_GEMM_BAND") && atoi(getenv("KVQ_GEMM_BAND")) != 0) d.band = atoi(getenv("KVQ_GEMM_BAND"));
        if (d.band > d.tiles_m) d.band = d.tiles_m;
        if (-d.band > d.tiles_n) d.band = -d.tiles_n;
        {
            const int w = d.band < 0 ? -d.band : d.band, across = d.band < 0 ? d.tiles_m : d.tiles_n, along = d.band < 0 ? d.tiles_n : d.tiles_m;
            KVQ_REQUIRE(d.tiles_m * d.tiles_n < 65536 && w * across < 65536, "%s: problem %d: more than 65535 tiles", who, i);
            auto mg = [](int dv) { return dv > 1 ? (unsigned)(0x100000000ull / (unsigned)dv + 1) : 0u; };      // 0 = divide by one
            d.mg_per_band = mg(w * across);
            d.mg_full = mg(w);
            d.mg_rem = mg(along % w ? along % w : w);
        }
        d.C2 = nullptr; d.H = nullptr; d.part = nullptr; d.vlimit = 0; d.scaleA = nullptr; d.scaleB = nullptr;
        d.seed = 0; d.seed_off = nullptr; d.site = 0; d.thresh = 0; d.inv_keep = 1.0f;
        d.C8 = nullptr; d.st8 = nullptr; d.ld8 = 0;
        t0 += d.tiles_m * d.tiles_n;
    }
    for (int i = nprob; i < g2::MAX_PROBLEMS; ++i) P.p[i] = P.p[0];
    P.ntiles = t0;
    return KVQ_OK;
}

extern "C" {

#ifdef KVQ_G2_DIAG
int kvq_diag_set_buffer(void* buf) {          // diagnostic library only: [tiles of the next launches][16] u64, or null
    return hipMemcpyToSymbol(HIP_SYMBOL(g2::g_diag), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

int kvq_gemm_grouped_bf16(const kvq_gemm_problem* probs, int nprob, int layout, int tile, void* stream) {
    g2::Params P;
    if (int rc = build_params(probs, nprob, layout, tile, P, "kvq_gemm_grouped_bf16")) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (tile & KVQ_GEMM_PERSISTENT) {
        KVQ_REQUIRE(layout == KVQ_GEMM_NT, "kvq_gemm (persistent): layout NT only");
        switch (tile & ~KVQ_GEMM_PERSISTENT) {
            case KVQ_GEMM_TILE_128x256: return g2::launch_persistent<g2::Cfg128x256<true, true>, g2::EPI_NONE>(P, st);
            case KVQ_GEMM_TILE_256x192: return g2::launch_persistent<g2::Cfg256x192<true, true>, g2::EPI_NONE>(P, st);
            case KVQ_GEMM_TILE_256x256: return g2::launch_persistent<g2::Cfg256x256<true, true>, g2::EPI_NONE>(P, st);
            default: return g2::launch_persistent<g2::Cfg128x192p<true, true>, g2::EPI_NONE>(P, st);
        }
    }
    switch (tile) {
        case KVQ_GEMM_TILE_128x256: return g2::launch_layout<g2::Cfg128x256>(layout, P, st);
        case KVQ_GEMM_TILE_256x192: return g2::launch_layout<g2::Cfg256x192>(layout, P, st);
        case KVQ_GEMM_TILE_256x256: return g2::launch_layout<g2::Cfg256x256>(layout, P, st);
        case KVQ_GEMM_TILE_64x128: return g2::launch_layout<g2::Cfg64x128>(layout, P, st);
        case KVQ_GEMM_TILE_128x192H: return g2::launch_layout<g2::Cfg128x192h>(layout, P, st);
        default: return g2::launch_layout<g2::Cfg128x192>(layout, P, st);
    }
}

int kvq_gemm_bf16(const void* A, const void* B, const void* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                  int layout, int tile, int accumulate, void* stream) {
    kvq_gemm_problem q;
    q.A = A; q.B = B; q.C = C; q.bias = bias; q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.accumulate = accumulate;
    return kvq_gemm_grouped_bf16(&q, 1, layout, tile, stream);
}

int kvq_gemm_bf16_gelu(const void* A, const void* B, const void* bias, void* Hout, void* Aout, int M, int N, int K, int lda, int ldb,
                       int ldc, int tile, void* stream) {
    KVQ_REQUIRE(Aout && ((uintptr_t)Aout & 15) == 0, "kvq_gemm_bf16_gelu: null / misaligned second output");
    const int base = tile & ~KVQ_GEMM_PERSISTENT;
    KVQ_REQUIRE(base == KVQ_GEMM_TILE_256x192 || base == KVQ_GEMM_TILE_128x256, "kvq_gemm_bf16_gelu: tile must be 256x192 or 128x256");
    kvq_gemm_problem q;
    q.A = A; q.B = B; q.C = Hout; q.bias = bias; q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.accumulate = 0;
    g2::Params P;
    if (int rc = build_params(&q, 1, KVQ_GEMM_NT, tile, P, "kvq_gemm_bf16_gelu")) return rc;
    for (int i = 0; i < g2::MAX_PROBLEMS; ++i) P.p[i].C2 = (unsigned short*)Aout;
    hipStream_t st = (hipStream_t)stream;
    if (tile & KVQ_GEMM_PERSISTENT) {
        if (base == KVQ_GEMM_TILE_256x192) return g2::launch_persistent<g2::Cfg256x192<true, true>, g2::EPI_GELU>(P, st);
        return g2::launch_persistent<g2::Cfg128x256<true, true>, g2::EPI_GELU>(P, st);
    }
    if (tile == KVQ_GEMM_TILE_256x192) return g2::launch_cfg<g2::Cfg256x192<true, true>, g2::EPI_GELU>(P, st);
    return g2::launch_cfg<g2::Cfg128x256<true, true>, g2::EPI_GELU>(P, st);
}

int kvq_gemm_bf16_dropres(const void* A, const void* B, const void* bias, const void* R, void* C, int M, int N, int K, int lda, int ldb,
                          int ldc, int tile, float p_drop, uint64_t seed, uint32_t site, void* stream) {
    KVQ_REQUIRE(R && ((uintptr_t)R & 15) == 0, "kvq_gemm_bf16_dropres: null / misaligned residual");
    KVQ_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "kvq_gemm_bf16_dropres: p_drop out of range");
    KVQ_REQUIRE(tile == KVQ_GEMM_TILE_128x192 || tile == KVQ_GEMM_TILE_128x256 || tile == KVQ_GEMM_TILE_64x128,
                "kvq_gemm_bf16_dropres: tile must be 128x192, 128x256 or 64x128");
    kvq_gemm_problem q;
    q.A = A; q.B = B; q.C = C; q.bias = bias; q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.accumulate = 0;
    g2::Params P;
    if (int rc = build_params(&q, 1, KVQ_GEMM_NT, tile, P, "kvq_gemm_bf16_dropres")) return rc;
    for (int i = 0; i < g2::MAX_PROBLEMS; ++i) {
        g2::Problem& d = P.p[i];
        d.H = (const unsigned short*)R; d.seed = seed; d.seed_off = seed_offset_ptr(); d.site = site; d.thresh = drop_threshold(p_drop);
        d.inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    }
    hipStream_t st = (hipStream_t)stream;
    if (tile == KVQ_GEMM_TILE_128x256) return g2::launch_cfg<g2::Cfg128x256<true, true>, g2::EPI_DROPRES>(P, st);
    if (tile == KVQ_GEMM_TILE_64x128) return g2::launch_cfg<g2::Cfg64x128<true, true>, g2::EPI_DROPRES>(P, st);
    return g2::launch_cfg<g2::Cfg128x192<true, true>, g2::EPI_DROPRES>(P, st);
}

size_t kvq_gemm_ce_stats_bytes(int M, int N) { return (size_t)(M > 0 ? M : 0) * (size_t)((N + 255) / 256) * 16; }

int kvq_gemm_bf16_ce(const void* A, const void* B, const void* bias, void* Cout, int M, int N, int K, int lda, int ldb, int ldc, int V,
                     float* stats, size_t stats_bytes, void* stream) {
    KVQ_REQUIRE(stats && V > 0 && V <= N, "kvq_gemm_bf16_ce: stats buffer and 0 < V <= N required");
    KVQ_REQUIRE(stats_bytes >= kvq_gemm_ce_stats_bytes(M, N), "kvq_gemm_bf16_ce: stats buffer %zu < %zu bytes", stats_bytes, kvq_gemm_ce_stats_bytes(M, N));
    KVQ_REQUIRE(((uintptr_t)stats) % 16 == 0, "kvq_gemm_bf16_ce: 16-byte aligned stats buffer required");
    kvq_gemm_problem q;
    q.A = A; q.B = B; q.C = Cout; q.bias = bias; q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.accumulate = 0;
    g2::Params P;
    if (int rc = build_params(&q, 1, KVQ_GEMM_NT, KVQ_GEMM_TILE_256x256, P, "kvq_gemm_bf16_ce")) return rc;
    for (int i = 0; i < g2::MAX_PROBLEMS; ++i) { P.p[i].part = stats; P.p[i].vlimit = V; }
    static_assert(g2::Cfg256x256<true, true>::RPP * g2::Cfg256x256<true, true>::BN * 4 >= g2::Cfg256x256<true, true>::BM * g2::Cfg256x256<true, true>::WN * 16,
                  "statistics scratch behind the epilogue tile");
    return g2::launch_cfg<g2::Cfg256x256<true, true>, g2::EPI_CE>(P, (hipStream_t)stream);
}

int kvq_gemm_fp8_nt(const void* A8, const void* B8, const float* scale_a, const float* scale_b, const void* bias, void* C, int M, int N,
                    int K, int lda, int ldb, int ldc, void* stream) {
    KVQ_REQUIRE(scale_a && scale_b, "kvq_gemm_fp8_nt: null scale pointer");
    KVQ_REQUIRE(K > 0 && K % 128 == 0 && lda % 16 == 0 && ldb % 16 == 0, "kvq_gemm_fp8_nt: K %% 128 == 0 and lda, ldb %% 16 == 0 (fp8 elements)");
    kvq_gemm_problem q;                         // the ring moves bytes: an fp8 row of K elements is a bf16 row of K / 2
    q.A = A8; q.B = B8; q.C = C; q.bias = bias; q.M = M; q.N = N; q.K = K / 2; q.lda = lda / 2; q.ldb = ldb / 2; q.ldc = ldc; q.accumulate = 0;
    g2::Params P;
    // Tile (round 5): 128 x 256 unless that leaves CUs without a tile and 128 x 192 does not -- an [8192, 768] output is 192 tiles of
    // the first and 256 of the second (KVQ_FP8_TILE=0: always 128 x 256, the rounds 2 - 4 behaviour; A/B switch)
    static const bool narrow_ok = !(getenv("KVQ_FP8_TILE") && atoi(getenv("KVQ_FP8_TILE")) == 0);
    const int t256 = ((M + 127) / 128) * ((N + 255) / 256), t192 = ((M + 127) / 128) * ((N + 191) / 192);
    const bool narrow = narrow_ok && t256 < g2::persistent_grid() && t192 > t256;
    if (int rc = build_params(&q, 1, KVQ_GEMM_NT, narrow ? KVQ_GEMM_TILE_128x192 : KVQ_GEMM_TILE_128x256, P, "kvq_gemm_fp8_nt")) return rc;
    for (int i = 0; i < g2::MAX_PROBLEMS; ++i) { P.p[i].scaleA = scale_a; P.p[i].scaleB = scale_b; }
    hipStream_t st = (hipStream_t)stream;
    if (narrow) {
        typedef g2::Cfg128x192<true, true> Cn;
        static std::atomic<bool> attr_n{false};
        if (!attr_n) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&g2::gemm2_f8_kernel<Cn>), hipFuncAttributeMaxDynamicSharedMemorySize, Cn::LDS);
            if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "hipFuncSetAttribute(gemm2_f8 128x192): %s", hipGetErrorString(e));
            attr_n = true;
        }
        hipLaunchKernelGGL((g2::gemm2_f8_kernel<Cn>), dim3((unsigned)P.ntiles), dim3(Cn::THREADS), Cn::LDS, st, P);
        return check_launch("gemm2_f8_kernel<128x192>");
    }
    typedef g2::Cfg128x256<true, true> Cf;
    static std::atomic<bool> attr_done{false};
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&g2::gemm2_f8_kernel<Cf>), hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS);
        if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "hipFuncSetAttribute(gemm2_f8): %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL((g2::gemm2_f8_kernel<Cf>), dim3((unsigned)P.ntiles), dim3(Cf::THREADS), Cf::LDS, st, P);
    return check_launch("gemm2_f8_kernel");
}

int kvq_gemm_fp8_nt_gelu(const void* A8, const void* B8, const float* scale_a, const float* scale_b, const void* bias, void* Hout, void* Aout,
                         void* Aout_fp8, int ld8, float* fp8_state, int M, int N, int K, int lda, int ldb, int ldc, void* stream) {
    KVQ_REQUIRE(scale_a && scale_b, "kvq_gemm_fp8_nt_gelu: null scale pointer");
    KVQ_REQUIRE(K > 0 && K % 128 == 0 && lda % 16 == 0 && ldb % 16 == 0, "kvq_gemm_fp8_nt_gelu: K %% 128 == 0 and lda, ldb %% 16 == 0 (fp8 elements)");
    KVQ_REQUIRE(Aout && ((uintptr_t)Aout & 15) == 0, "kvq_gemm_fp8_nt_gelu: null / misaligned second output");
    KVQ_REQUIRE(!Aout_fp8 || (fp8_state && ld8 >= N && ld8 % 8 == 0 && ((uintptr_t)Aout_fp8 & 7) == 0),
                "kvq_gemm_fp8_nt_gelu: the fp8 copy needs a state, 8-byte alignment and a row stride >= N in multiples of 8");
    kvq_gemm_problem q;
    q.A = A8; q.B = B8; q.C = Hout; q.bias = bias; q.M = M; q.N = N; q.K = K / 2; q.lda = lda / 2; q.ldb = ldb / 2; q.ldc = ldc; q.accumulate = 0;
    g2::Params P;
    if (int rc = build_params(&q, 1, KVQ_GEMM_NT, KVQ_GEMM_TILE_128x256, P, "kvq_gemm_fp8_nt_gelu")) return rc;
    for (int i = 0; i < g2::MAX_PROBLEMS; ++i) {
        g2::Problem& d = P.p[i];
        d.scaleA = scale_a; d.scaleB = scale_b; d.C2 = (unsigned short*)Aout; d.C8 = (unsigned char*)Aout_fp8; d.st8 = fp8_state; d.ld8 = ld8;
    }
    typedef g2::Cfg128x256<true, true> Cf;
    static std::atomic<bool> attr_done{false};
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&g2::gemm2_f8_kernel<Cf, g2::EPI_GELU>), hipFuncAttributeMaxDynamicSharedMemorySize, Cf::LDS);
        if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "hipFuncSetAttribute(gemm2_f8 gelu): %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL((g2::gemm2_f8_kernel<Cf, g2::EPI_GELU>), dim3((unsigned)P.ntiles), dim3(Cf::THREADS), Cf::LDS, (hipStream_t)stream, P);
    return check_launch("gemm2_f8_kernel<gelu>");
}

int64_t kvq_gemm_dgelu_partial_rows(int64_t M, int tile) {
    int bm, bn;
    g2::tile_of(tile, bm, bn);
    return (M + bm - 1) / bm;
}

int kvq_gemm_bf16_dgelu(const void* A, const void* B, const void* H, void* C, float* part, size_t part_bytes, int M, int N, int K,
                        int lda, int ldb, int ldc, int tile, void* stream) {
    KVQ_REQUIRE(H && part && ((uintptr_t)H & 15) == 0, "kvq_gemm_bf16_dgelu: null / misaligned pre-activation or partial buffer");
    KVQ_REQUIRE(tile == KVQ_GEMM_TILE_256x192 || tile == KVQ_GEMM_TILE_128x256, "kvq_gemm_bf16_dgelu: tile must be 256x192 or 128x256");
    KVQ_REQUIRE(part_bytes >= (size_t)kvq_gemm_dgelu_partial_rows(M, tile) * N * sizeof(float), "kvq_gemm_bf16_dgelu: partial buffer too small");
    kvq_gemm_problem q;
    q.A = A; q.B = B; q.C = C; q.bias = nullptr; q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.accumulate = 0;
    g2::Params P;
    if (int rc = build_params(&q, 1, KVQ_GEMM_NN, tile, P, "kvq_gemm_bf16_dgelu")) return rc;
    for (int i = 0; i < g2::MAX_PROBLEMS; ++i) { P.p[i].H = (const unsigned short*)H; P.p[i].part = part; }
    hipStream_t st = (hipStream_t)stream;
    if (tile == KVQ_GEMM_TILE_256x192) return g2::launch_cfg<g2::Cfg256x192<true, false>, g2::EPI_DGELU>(P, st);
    return g2::launch_cfg<g2::Cfg128x256<true, false>, g2::EPI_DGELU>(P, st);
}

}  // extern "C"
