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
//     reads; both are bank-conflict free for their read instruction (derivation in DESIGN.md §2.3).
//   * one raw s_barrier per k-tile; the DMA of tile kt+NS is issued right behind the barrier that frees its slot and is waited
//     for NS-1 tiles later with a COUNTED s_waitcnt vmcnt (never 0 inside the loop): NS-1 tiles stay in flight per CU.
//   * every wave software-pipelines itself: the fragments of k-step s+1 are read from LDS while the MFMAs of k-step s run
//     (two named fragment sets, static indexing), so LDS latency never sits in front of the matrix pipe; with 8 waves per
//     workgroup (two per SIMD) the partner's MFMAs also cover the DMA issue slots.
//   * epilogue through LDS: bf16 tile -> whole row segments, bias / accumulate in f32, 16-byte stores.
//   * tiles are numbered so that each XCD (private L2) owns a contiguous band of tiles.
//   * a launch can cover several problems of one layout ("grouped"): the weight gradients of one transformer layer are ONE
//     launch of ~250 tiles of 128 x 256 -- one tile per CU over the whole 8192-token contraction, no split-K, no partial slabs.
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
constexpr int MAX_PROBLEMS = 8;

struct Problem {
    const unsigned short* A;
    const unsigned short* B;
    unsigned short* C;
    const unsigned short* bias;        // [N] bf16 or null
    int M, N, K;
    int lda, ldb, ldc;
    int tiles_m, tiles_n;
    int tile0;                         // first tile id of this problem in the launch
    int accumulate;                    // C += result
};

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
    static constexpr int TM = BM / WM, TN = BN / WN;        // wave tile
    static constexpr int FA = TM / 16, FB = TN / 16;        // fragments per k-step
    static constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB, STAGE = A_BYTES + B_BYTES;
    static constexpr int PIECES = (BM + BN) / 8;            // 1-KiB DMA pieces per k-tile
    static constexpr int PPW = PIECES / WAVES;              // per wave
    static constexpr int CLD = BN * 2 + 16;                 // epilogue row stride (bytes)
    static constexpr int LDS = (NS * STAGE > BM * CLD) ? NS * STAGE : BM * CLD;
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

// One MFMA cluster (a k-step of the wave tile) with the DMA pieces [Q0, Q1) of the pending ring refill issued in between, evenly
// spaced: a piece occupies the CU's address path for ~16-20 cycles, so 8 waves issuing their pieces back to back behind the
// barrier stall each other for hundreds of cycles with the matrix pipe idle; one piece every few MFMAs never queues.
template <class C, int Q0, int Q1>
__device__ __forceinline__ void mma(f32x4 (&acc)[C::FA][C::FB], const Frags<C>& f, Stager<C>& sg, int slot, bool pending) {
    constexpr int NM = C::FA * C::FB, NP = Q1 - Q0;
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < NM; ++i) {
        const int mi = i / C::FB, ni = i % C::FB;
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.b[ni], f.a[mi], acc[mi][ni], 0, 0, 0);
        if constexpr (NP > 0) {
#pragma unroll
            for (int j = 0; j < NP; ++j)
                if (i == (j + 1) * NM / (NP + 1) - 1) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (pending) sg.template piece_rt<Q0, Q1>(j, slot);
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
    }
    __builtin_amdgcn_s_setprio(0);
}

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <class C>
__global__ __launch_bounds__(C::THREADS, C::WAVES / 4) void gemm2_kernel(Params P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w / C::WN, wn = w % C::WN;

    // ---- tile of this workgroup: XCD-aware numbering (bijective form), then problem lookup
    int id = blockIdx.x;
    {
        const int nt = P.ntiles, q = nt >> 3, r = nt & 7, x = id & 7;
        id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (id >> 3);
    }
    int pi = 0;
#pragma unroll
    for (int i = 1; i < MAX_PROBLEMS; ++i)
        if (i < P.nprob && id >= P.p[i].tile0) pi = i;
    const Problem& pr = P.p[pi];
    const int lt = id - pr.tile0;
    const int tm = lt / pr.tiles_n, tn = lt - tm * pr.tiles_n;
    const int m0 = tm * C::BM, n0 = tn * C::BN;
    const int nkt = pr.K / BK;

    f32x4 acc[C::FA][C::FB];
#pragma unroll
    for (int mi = 0; mi < C::FA; ++mi)
#pragma unroll
        for (int ni = 0; ni < C::FB; ++ni) acc[mi][ni] = 0.f;

    // ---- prologue: fill the ring
    Stager<C> sg;
    sg.init(pr, smem, m0, n0, w, lane);
#pragma unroll
    for (int s = 0; s < C::NS; ++s)
        if (s < nkt) sg.issue(s);
    if (nkt >= C::NS) wait_vm<(C::NS - 1) * C::PPW>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();

    // ---- main loop: per k-tile  [f0 ready] read f1 | MFMA(f0) + 2nd half of the refill | f1 ready, tile kt+1 landed, barrier,
    //                             read next f0 | MFMA(f1) + 1st half of the refill of the slot just freed
    Frags<C> f0, f1;
    read_frags<C>(f0, smem, wm, wn, 0, lane);
    constexpr int PH = C::PPW / 2;                                         // pieces issued inside the first cluster after the barrier
    int slot = 0, pslot = 0;
    bool pending = false;
    for (int kt = 0; kt < nkt; ++kt) {
        const char* st = smem + slot * C::STAGE;
        __builtin_amdgcn_s_waitcnt(0xC07F);                                // lgkmcnt(0): f0 (issued a whole MFMA cluster ago)
        read_frags<C>(f1, st, wm, wn, 1, lane);
        __builtin_amdgcn_sched_barrier(0);
        mma<C, PH, C::PPW>(acc, f0, sg, pslot, pending);
        if (pending) sg.advance();
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(0xC07F);                                // f1 arrived: this wave is done with `slot`
        pending = false;
        if (kt + 1 < nkt) {
            if (kt + C::NS <= nkt) wait_vm<(C::NS - 2) * C::PPW>();         // tiles kt+2 .. kt+NS-1 may still be in flight
            else wait_vm<0>();
            __builtin_amdgcn_s_barrier();                                  // tile kt+1 landed for everybody; `slot` is free
            pending = kt + C::NS < nkt;
            pslot = slot;
            const int nslot = slot + 1 == C::NS ? 0 : slot + 1;
            read_frags<C>(f0, smem + nslot * C::STAGE, wm, wn, 0, lane);
            slot = nslot;
        }
        __builtin_amdgcn_sched_barrier(0);
        mma<C, 0, PH>(acc, f1, sg, pslot, pending);
    }
    __builtin_amdgcn_s_barrier();      // every wave has read its last fragments: the ring becomes the epilogue tile

    // ---- epilogue: acc -> (bias) -> bf16 tile in LDS -> whole row segments (+C) -> global
    // lane holds, for (mi, ni): row m = wm*TM + mi*16 + (lane & 15), columns n = wn*TN + ni*16 + 4*(lane >> 4) + 0..3
    const int frow = lane & 15, fk = lane >> 4;
#pragma unroll
    for (int ni = 0; ni < C::FB; ++ni) {
        const int nl = wn * C::TN + ni * 16 + 4 * fk;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (pr.bias) {
            int nb = n0 + nl;
            nb = nb + 4 <= pr.N ? nb : pr.N - 4;
            const u16x4 t = *reinterpret_cast<const u16x4*>(pr.bias + nb);
            bv.x = bf16_to_f32(t.x); bv.y = bf16_to_f32(t.y); bv.z = bf16_to_f32(t.z); bv.w = bf16_to_f32(t.w);
        }
#pragma unroll
        for (int mi = 0; mi < C::FA; ++mi) {
            const int ml = wm * C::TM + mi * 16 + frow;
            const f32x4 v = acc[mi][ni] + bv;
            const u16x4 o = {f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
            *reinterpret_cast<u16x4*>(smem + ml * C::CLD + nl * 2) = o;
        }
    }
    __syncthreads();
    constexpr int CPR = C::BN / 8;                                         // 16-byte chunks per tile row
    constexpr int CHUNKS = C::BM * CPR;
#pragma unroll 4
    for (int cid = tid; cid < CHUNKS; cid += C::THREADS) {
        const int r = cid / CPR, c16 = cid - r * CPR;
        const int m = m0 + r, n = n0 + c16 * 8;
        if (m < pr.M && n < pr.N) {                                        // N % 8 == 0: a chunk is inside or outside as a whole
            uint4 v = *reinterpret_cast<const uint4*>(smem + r * C::CLD + c16 * 16);
            const size_t off = (size_t)m * pr.ldc + n;
            if (pr.accumulate) {
                unsigned* vn = reinterpret_cast<unsigned*>(&v);
                const uint4 old = *reinterpret_cast<const uint4*>(pr.C + off);
                const unsigned* vo = reinterpret_cast<const unsigned*>(&old);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = __uint_as_float(vn[u] << 16) + __uint_as_float(vo[u] << 16);
                    const float hi = __uint_as_float(vn[u] & 0xffff0000u) + __uint_as_float(vo[u] & 0xffff0000u);
                    vn[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
            }
            *reinterpret_cast<uint4*>(pr.C + off) = v;
        }
    }
}

// ---- tile configurations ---------------------------------------------------------------------------------------------
//                 BM   BN  WM WN  A k-major  B k-major  ring
template <bool AK, bool BKM> using Cfg128x256 = Cfg<128, 256, 2, 4, AK, BKM, 3>;     // 8 waves, 144 KiB: one tile per CU (wgrad)
template <bool AK, bool BKM> using Cfg256x192 = Cfg<256, 192, 4, 2, AK, BKM, 2>;     // 8 waves, 112 KiB
template <bool AK, bool BKM> using Cfg128x192 = Cfg<128, 192, 2, 2, AK, BKM, 3>;     // 4 waves, 120 KiB: 256 tiles at N = 768
template <bool AK, bool BKM> using Cfg256x256 = Cfg<256, 256, 2, 4, AK, BKM, 2>;     // 8 waves, 128 KiB

template <class C>
static int launch_cfg(const Params& P, hipStream_t st) {
    static bool attr_done = false;                 // per instantiation; idempotent, so a race only repeats the call
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm2_kernel<C>), hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS);
        if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "hipFuncSetAttribute(gemm2): %s", hipGetErrorString(e));
        attr_done = true;
    }
    hipLaunchKernelGGL(gemm2_kernel<C>, dim3((unsigned)P.ntiles), dim3(C::THREADS), C::LDS, st, P);
    return check_launch("gemm2_kernel");
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
        default: bm = 128; bn = 192; break;
    }
}

}  // namespace g2
}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_gemm_grouped_bf16(const kvq_gemm_problem* probs, int nprob, int layout, int tile, void* stream) {
    KVQ_REQUIRE(probs && nprob >= 1 && nprob <= g2::MAX_PROBLEMS, "kvq_gemm_grouped_bf16: 1..%d problems per launch", g2::MAX_PROBLEMS);
    KVQ_REQUIRE(layout == KVQ_GEMM_NT || layout == KVQ_GEMM_NN || layout == KVQ_GEMM_TN, "kvq_gemm_grouped_bf16: unknown layout %d", layout);
    KVQ_REQUIRE(tile >= KVQ_GEMM_TILE_128x192 && tile <= KVQ_GEMM_TILE_256x256, "kvq_gemm_grouped_bf16: unknown tile %d", tile);
    int bm, bn;
    g2::tile_of(tile, bm, bn);
    g2::Params P;
    P.nprob = nprob;
    int t0 = 0;
    for (int i = 0; i < nprob; ++i) {
        const kvq_gemm_problem& q = probs[i];
        KVQ_REQUIRE(q.A && q.B && q.C && q.M > 0 && q.N > 0 && q.K > 0, "kvq_gemm_grouped_bf16: problem %d: bad argument", i);
        KVQ_REQUIRE(q.K % g2::BK == 0, "kvq_gemm_grouped_bf16: problem %d: K=%d must be a multiple of %d", i, q.K, g2::BK);
        KVQ_REQUIRE(q.M % 8 == 0 && q.N % 8 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc % 8 == 0 && q.M >= 8 && q.N >= 8,
                    "kvq_gemm_grouped_bf16: problem %d: M, N, lda, ldb, ldc must be multiples of 8", i);
        KVQ_REQUIRE((((uintptr_t)q.A | (uintptr_t)q.B | (uintptr_t)q.C) & 15) == 0 && (!q.bias || ((uintptr_t)q.bias & 7) == 0),
                    "kvq_gemm_grouped_bf16: problem %d: operands must be 16-byte aligned", i);
        // leading dimensions must cover the rows the kernel reads
        const int a_cols = layout == KVQ_GEMM_TN ? q.M : q.K, b_cols = layout == KVQ_GEMM_NT ? q.K : q.N;
        KVQ_REQUIRE(q.lda >= a_cols && q.ldb >= b_cols && q.ldc >= q.N, "kvq_gemm_grouped_bf16: problem %d: leading dimension too small", i);
        g2::Problem& d = P.p[i];
        d.A = (const unsigned short*)q.A; d.B = (const unsigned short*)q.B; d.C = (unsigned short*)q.C; d.bias = (const unsigned short*)q.bias;
        d.M = q.M; d.N = q.N; d.K = q.K; d.lda = q.lda; d.ldb = q.ldb; d.ldc = q.ldc;
        d.tiles_m = (q.M + bm - 1) / bm; d.tiles_n = (q.N + bn - 1) / bn;
        d.tile0 = t0; d.accumulate = q.accumulate;
        t0 += d.tiles_m * d.tiles_n;
    }
    for (int i = nprob; i < g2::MAX_PROBLEMS; ++i) P.p[i] = P.p[0];
    P.ntiles = t0;
    hipStream_t st = (hipStream_t)stream;
    switch (tile) {
        case KVQ_GEMM_TILE_128x256: return g2::launch_layout<g2::Cfg128x256>(layout, P, st);
        case KVQ_GEMM_TILE_256x192: return g2::launch_layout<g2::Cfg256x192>(layout, P, st);
        case KVQ_GEMM_TILE_256x256: return g2::launch_layout<g2::Cfg256x256>(layout, P, st);
        default: return g2::launch_layout<g2::Cfg128x192>(layout, P, st);
    }
}

int kvq_gemm_bf16(const void* A, const void* B, const void* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                  int layout, int tile, int accumulate, void* stream) {
    kvq_gemm_problem q;
    q.A = A; q.B = B; q.C = C; q.bias = bias; q.M = M; q.N = N; q.K = K; q.lda = lda; q.ldb = ldb; q.ldc = ldc; q.accumulate = accumulate;
    return kvq_gemm_grouped_bf16(&q, 1, layout, tile, stream);
}

}  // extern "C"
