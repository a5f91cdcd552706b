// kvq_gemm.hip -- bf16 MFMA GEMM for the BERT projections of the training step (gfx950).
//
//   C[M,N] (bf16) = A[M,K] (bf16, row-major) . B[N,K]^T (bf16, row-major)  [+ bias[N]]  [+ C]      "NT": both operands k-contiguous
//
// This is the shape of every forward projection of HuggingFace's BertLayer as used by models/bagon/Bagon.py:46-53
// (x[tokens, in] . W[out, in]^T + b: modeling_bert.py:139-352) and, with a transposed shadow copy of the weight, of every
// input-gradient GEMM of its backward.
//
// Structure (one workgroup = one 128 x 128 output tile, 4 waves as 2 x 2, each wave 64 x 64 = 4 x 4 MFMA tiles):
//   * v_mfma_f32_16x16x32_bf16, f32 accumulation; operands swapped (the MFMA's rows are the N index) so that a lane ends up
//     with 4 CONSECUTIVE columns of one output row;
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B per lane, no VGPR hop), 64-deep k-tiles, two LDS stages
//     (64 KiB: two workgroups per CU); LDS rows are 128 B, 16-byte chunks XOR-swizzled by (row & 7) through the SOURCE
//     address (the DMA destination is lane-linear), undone on the ds_read_b128 side: conflict-free fragment reads;
//   * epilogue through LDS: the tile is re-read as whole 256-byte row segments, bias / accumulate applied in f32,
//     stored with coalesced 16-byte stores;
//   * tiles are numbered so that the 8 XCDs each own a contiguous band of tiles (neighbouring tiles share their A rows
//     in that XCD's L2).
#include "kvq_common.h"

namespace kvq {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

constexpr int GM = 128, GN = 128, GK = 64;
constexpr int G_THREADS = 256;
constexpr int G_ROWB = GK * 2;                        // bytes per LDS row (128)
constexpr int G_TILE_B = GM * G_ROWB;                 // bytes per operand tile (16 KiB)
constexpr int G_STAGE_B = 2 * G_TILE_B;               // A + B
constexpr int G_LDS_B = 2 * G_STAGE_B;                // 64 KiB
constexpr int G_CLD = GN * 2 + 16;                    // epilogue tile row stride in bytes (272: conflict-free row reads)
static_assert(GM * G_CLD <= G_LDS_B, "epilogue tile must fit in the staging buffers");

struct GemmParams {
    const unsigned short* A;
    const unsigned short* B;
    const unsigned short* bias;   // [N] bf16 or null
    unsigned short* C;
    int M, N, K;
    int lda, ldb, ldc;
    int accumulate;
    int tiles_m, tiles_n, ntiles;
    int group_m;                  // > 0: tiles are walked in groups of `group_m` tile-rows per tile-column (big N: keeps A in L2)
    int epi;                      // EPI_*
    unsigned short* C2;           // EPI_GELU: second output a = gelu(C)
    const unsigned short* H;      // EPI_DGELU: pre-activation h; C = acc * gelu'(h)
};

constexpr int EPI_NONE = 0, EPI_GELU = 1, EPI_DGELU = 2;

// erf to ~1.2e-7 absolute (Abramowitz & Stegun 7.1.26): plenty under bf16 output rounding, ~12 VALU ops
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float r = 1.0f - poly * __expf(-ax * ax);
    return copysignf(r, x);
}
__device__ __forceinline__ float gelu_fast(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752f)); }
__device__ __forceinline__ float dgelu_fast(float x) {
    const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752f));
    return cdf + x * 0.39894228040143268f * __expf(-0.5f * x * x);
}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// one k-tile of A and B into LDS stage `stage`: 8 LDS-DMA instructions per wave (4 for A, 4 for B)
__device__ __forceinline__ void g_stage(const GemmParams& p, char* smem, int stage, int m0, int n0, int kt, int w, int lane) {
    const int rsub = lane >> 3, cdst = lane & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (w * 4 + q) * 8 + rsub;                         // tile row 0..127
        const int c = cdst ^ (r & 7);                                  // source chunk (swizzle on the source side)
        int ra = m0 + r; ra = ra < p.M ? ra : p.M - 1;
        int rb = n0 + r; rb = rb < p.N ? rb : p.N - 1;
        const unsigned short* ga = p.A + (size_t)ra * p.lda + kt * GK + c * 8;
        const unsigned short* gb = p.B + (size_t)rb * p.ldb + kt * GK + c * 8;
        char* la = smem + stage * G_STAGE_B + (w * 4 + q) * 8 * G_ROWB;                 // wave-uniform 1-KiB block
        char* lb = la + G_TILE_B;
        __builtin_amdgcn_global_load_lds((gptr_t)ga, (lptr_t)la, 16, 0, 0);
        __builtin_amdgcn_global_load_lds((gptr_t)gb, (lptr_t)lb, 16, 0, 0);
    }
}

__global__ __launch_bounds__(G_THREADS, 2) void gemm_nt_bf16_kernel(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    // XCD-aware tile numbering (bijective only when ntiles % 8 == 0; otherwise plain order)
    int id = blockIdx.x;
    if ((p.ntiles & 7) == 0) id = (id & 7) * (p.ntiles >> 3) + (id >> 3);
    int tm, tn;
    if (p.group_m > 0) {
        const int per_group = p.group_m * p.tiles_n;
        const int grp = id / per_group, in = id - grp * per_group;
        const int gm0 = grp * p.group_m;
        const int gsz = p.tiles_m - gm0 < p.group_m ? p.tiles_m - gm0 : p.group_m;
        tm = gm0 + in % gsz;
        tn = in / gsz;
    } else {
        tm = id / p.tiles_n;
        tn = id - tm * p.tiles_n;
    }
    const int m0 = tm * GM, n0 = tn * GN;
    const int nkt = p.K / GK;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = 0.f;

    g_stage(p, smem, 0, m0, n0, 0, w, lane);
    __syncthreads();

    const int frow = lane & 15, fk = lane >> 4;
    for (int kt = 0; kt < nkt; ++kt) {
        const int st = kt & 1;
        if (kt + 1 < nkt) g_stage(p, smem, st ^ 1, m0, n0, kt + 1, w, lane);
        const char* At = smem + st * G_STAGE_B;
        const char* Bt = At + G_TILE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int r = wm * 64 + mi * 16 + frow;
                a[mi] = *reinterpret_cast<const bf16x8*>(At + r * G_ROWB + (((ks * 4 + fk) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int r = wn * 64 + ni * 16 + frow;
                b[ni] = *reinterpret_cast<const bf16x8*>(Bt + r * G_ROWB + (((ks * 4 + fk) ^ (r & 7)) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        __syncthreads();     // LDS-DMA of the next stage landed (vmcnt(0) is part of the barrier) and this stage is free again
    }

    // ---- epilogue: acc -> (bias) -> bf16 tile in LDS -> coalesced rows (+C) -> global
    // lane holds, for tile (mi, ni): output row m = wm*64 + mi*16 + (lane & 15), columns n = wn*64 + ni*16 + 4*(lane >> 4) + 0..3
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int nl = wn * 64 + ni * 16 + 4 * fk;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            int nb = n0 + nl; nb = nb + 3 < p.N ? nb : (p.N >= 4 ? p.N - 4 : 0);
            const u16x4 t = *reinterpret_cast<const u16x4*>(p.bias + nb);
            bv.x = bf16_to_f32(t.x); bv.y = bf16_to_f32(t.y); bv.z = bf16_to_f32(t.z); bv.w = bf16_to_f32(t.w);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int ml = wm * 64 + mi * 16 + frow;
            const f32x4 v = acc[mi][ni] + bv;
            u16x4 o = {f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
            *reinterpret_cast<u16x4*>(smem + ml * G_CLD + nl * 2) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int cid = q * G_THREADS + tid;
        const int r = cid >> 4, c16 = cid & 15;
        const int m = m0 + r, n = n0 + c16 * 8;
        if (m < p.M && n < p.N) {                                     // N % 8 == 0: a chunk is inside or outside as a whole
            uint4 v = *reinterpret_cast<const uint4*>(smem + r * G_CLD + c16 * 16);
            const size_t off = (size_t)m * p.ldc + n;
            unsigned* vn = reinterpret_cast<unsigned*>(&v);
            if (p.accumulate) {
                const uint4 old = *reinterpret_cast<const uint4*>(p.C + off);
                const unsigned* vo = reinterpret_cast<const unsigned*>(&old);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = __uint_as_float(vn[u] << 16) + __uint_as_float(vo[u] << 16);
                    const float hi = __uint_as_float(vn[u] & 0xffff0000u) + __uint_as_float(vo[u] & 0xffff0000u);
                    vn[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
            }
            if (p.epi == EPI_DGELU) {                                  // C = (A.B^T) * gelu'(h)
                const uint4 hh = *reinterpret_cast<const uint4*>(p.H + off);
                const unsigned* hv = reinterpret_cast<const unsigned*>(&hh);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = __uint_as_float(vn[u] << 16) * dgelu_fast(__uint_as_float(hv[u] << 16));
                    const float hi = __uint_as_float(vn[u] & 0xffff0000u) * dgelu_fast(__uint_as_float(hv[u] & 0xffff0000u));
                    vn[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
            }
            *reinterpret_cast<uint4*>(p.C + off) = v;
            if (p.epi == EPI_GELU) {                                   // second output: a = gelu(h), h = the bf16 value just stored
                uint4 g;
                unsigned* gv = reinterpret_cast<unsigned*>(&g);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = gelu_fast(__uint_as_float(vn[u] << 16));
                    const float hi = gelu_fast(__uint_as_float(vn[u] & 0xffff0000u));
                    gv[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
                *reinterpret_cast<uint4*>(p.C2 + off) = g;
            }
        }
    }
}

// 3-stage variant: two k-tiles of LDS-DMA stay in flight across the barrier (counted s_waitcnt vmcnt, raw s_barrier), which is
// what covers the L2 latency of the 12..48-tile contractions of this model; 96 KiB of LDS, one workgroup per CU.
__global__ __launch_bounds__(G_THREADS, 1) void gemm_nt_bf16_kernel3(GemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = w >> 1, wn = w & 1;
    // XCD-aware tile numbering (bijective only when ntiles % 8 == 0; otherwise plain order)
    int id = blockIdx.x;
    if ((p.ntiles & 7) == 0) id = (id & 7) * (p.ntiles >> 3) + (id >> 3);
    int tm, tn;
    if (p.group_m > 0) {
        const int per_group = p.group_m * p.tiles_n;
        const int grp = id / per_group, in = id - grp * per_group;
        const int gm0 = grp * p.group_m;
        const int gsz = p.tiles_m - gm0 < p.group_m ? p.tiles_m - gm0 : p.group_m;
        tm = gm0 + in % gsz;
        tn = in / gsz;
    } else {
        tm = id / p.tiles_n;
        tn = id - tm * p.tiles_n;
    }
    const int m0 = tm * GM, n0 = tn * GN;
    const int nkt = p.K / GK;

    f32x4 acc[4][4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = 0.f;

    g_stage(p, smem, 0, m0, n0, 0, w, lane);
    if (nkt > 1) g_stage(p, smem, 1, m0, n0, 1, w, lane);

    const int frow = lane & 15, fk = lane >> 4;
    int st = 0;
    for (int kt = 0; kt < nkt; ++kt) {
        // stage kt has landed when at most the 8 LDS-DMA instructions of stage kt+1 are still outstanding
        if (kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        // the buffer of stage kt-1 is free now (every wave passed the barrier after reading it): refill it with stage kt+2
        if (kt + 2 < nkt) g_stage(p, smem, st == 0 ? 2 : st - 1, m0, n0, kt + 2, w, lane);
        const char* At = smem + st * G_STAGE_B;
        const char* Bt = At + G_TILE_B;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const int r = wm * 64 + mi * 16 + frow;
                a[mi] = *reinterpret_cast<const bf16x8*>(At + r * G_ROWB + (((ks * 4 + fk) ^ (r & 7)) << 4));
            }
#pragma unroll
            for (int ni = 0; ni < 4; ++ni) {
                const int r = wn * 64 + ni * 16 + frow;
                b[ni] = *reinterpret_cast<const bf16x8*>(Bt + r * G_ROWB + (((ks * 4 + fk) ^ (r & 7)) << 4));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[ni], a[mi], acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        st = st == 2 ? 0 : st + 1;
    }
    __syncthreads();

    // ---- epilogue: acc -> (bias) -> bf16 tile in LDS -> coalesced rows (+C) -> global
    // lane holds, for tile (mi, ni): output row m = wm*64 + mi*16 + (lane & 15), columns n = wn*64 + ni*16 + 4*(lane >> 4) + 0..3
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) {
        const int nl = wn * 64 + ni * 16 + 4 * fk;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            int nb = n0 + nl; nb = nb + 3 < p.N ? nb : (p.N >= 4 ? p.N - 4 : 0);
            const u16x4 t = *reinterpret_cast<const u16x4*>(p.bias + nb);
            bv.x = bf16_to_f32(t.x); bv.y = bf16_to_f32(t.y); bv.z = bf16_to_f32(t.z); bv.w = bf16_to_f32(t.w);
        }
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int ml = wm * 64 + mi * 16 + frow;
            const f32x4 v = acc[mi][ni] + bv;
            u16x4 o = {f32_to_bf16(v.x), f32_to_bf16(v.y), f32_to_bf16(v.z), f32_to_bf16(v.w)};
            *reinterpret_cast<u16x4*>(smem + ml * G_CLD + nl * 2) = o;
        }
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int cid = q * G_THREADS + tid;
        const int r = cid >> 4, c16 = cid & 15;
        const int m = m0 + r, n = n0 + c16 * 8;
        if (m < p.M && n < p.N) {                                     // N % 8 == 0: a chunk is inside or outside as a whole
            uint4 v = *reinterpret_cast<const uint4*>(smem + r * G_CLD + c16 * 16);
            const size_t off = (size_t)m * p.ldc + n;
            unsigned* vn = reinterpret_cast<unsigned*>(&v);
            if (p.accumulate) {
                const uint4 old = *reinterpret_cast<const uint4*>(p.C + off);
                const unsigned* vo = reinterpret_cast<const unsigned*>(&old);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = __uint_as_float(vn[u] << 16) + __uint_as_float(vo[u] << 16);
                    const float hi = __uint_as_float(vn[u] & 0xffff0000u) + __uint_as_float(vo[u] & 0xffff0000u);
                    vn[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
            }
            if (p.epi == EPI_DGELU) {                                  // C = (A.B^T) * gelu'(h)
                const uint4 hh = *reinterpret_cast<const uint4*>(p.H + off);
                const unsigned* hv = reinterpret_cast<const unsigned*>(&hh);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = __uint_as_float(vn[u] << 16) * dgelu_fast(__uint_as_float(hv[u] << 16));
                    const float hi = __uint_as_float(vn[u] & 0xffff0000u) * dgelu_fast(__uint_as_float(hv[u] & 0xffff0000u));
                    vn[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
            }
            *reinterpret_cast<uint4*>(p.C + off) = v;
            if (p.epi == EPI_GELU) {                                   // second output: a = gelu(h), h = the bf16 value just stored
                uint4 g;
                unsigned* gv = reinterpret_cast<unsigned*>(&g);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float lo = gelu_fast(__uint_as_float(vn[u] << 16));
                    const float hi = gelu_fast(__uint_as_float(vn[u] & 0xffff0000u));
                    gv[u] = (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
                }
                *reinterpret_cast<uint4*>(p.C2 + off) = g;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Batched bf16 transposes: dst_i[C][R] = src_i[R][C]^T for up to KVQ_TRANSPOSE_MAX matrices of one shape in one launch.
// The input-gradient GEMM g . W of a projection is an "NN" product; with W^T kept beside W (refreshed once per optimiser
// step by this kernel, ~57 MB for the 48 768x768 weights of bert-base) it becomes the NT shape the kernel above is fast at.
// 64x64 tiles through LDS, rows padded by one dword pair: conflict-free both ways.
// ---------------------------------------------------------------------------------------------------------------
struct TransposeBatch {
    const unsigned short* src[KVQ_TRANSPOSE_MAX];
    unsigned short* dst[KVQ_TRANSPOSE_MAX];
    int R, C, n;
};
__global__ __launch_bounds__(256) void transpose_batch_kernel(TransposeBatch tb) {
    __shared__ unsigned short tile[64][66];
    const unsigned short* src = tb.src[blockIdx.z];
    unsigned short* dst = tb.dst[blockIdx.z];
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int r = r0 + ty + 4 * i, c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < tb.R && c < tb.C) ? src[(size_t)r * tb.C + c] : (unsigned short)0;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i, r = r0 + tx;
        if (c < tb.C && r < tb.R) dst[(size_t)c * tb.R + r] = tile[tx][ty + 4 * i];
    }
}

}  // namespace kvq

using namespace kvq;

extern "C" {

static int g_gemm_stages = 2;

int kvq_gemm_set_stages(int stages) {
    if (stages != 2 && stages != 3) return fail(KVQ_E_INVALID, "kvq_gemm_set_stages: 2 or 3");
    g_gemm_stages = stages;
    return KVQ_OK;
}

static int gemm_nt_launch(const void* A, const void* B, const void* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                          int accumulate, int epi, void* C2, const void* H, void* stream) {
    KVQ_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "kvq_gemm_nt_bf16: bad argument");
    KVQ_REQUIRE(K % GK == 0, "kvq_gemm_nt_bf16: K=%d must be a multiple of %d", K, GK);
    KVQ_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "kvq_gemm_nt_bf16: N, lda, ldb, ldc must be multiples of 8");
    KVQ_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)C2 | (uintptr_t)H) & 15) == 0 && (!bias || ((uintptr_t)bias & 7) == 0),
                "kvq_gemm_nt_bf16: operands must be 16-byte aligned");
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, G_LDS_B);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_nt_bf16_kernel3),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, 3 * G_STAGE_B);
        if (e != hipSuccess) return fail(KVQ_E_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
        attr_done = true;
    }
    GemmParams p;
    p.A = (const unsigned short*)A; p.B = (const unsigned short*)B; p.bias = (const unsigned short*)bias; p.C = (unsigned short*)C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc; p.accumulate = accumulate;
    p.tiles_m = (M + GM - 1) / GM;
    p.tiles_n = (N + GN - 1) / GN;
    p.ntiles = p.tiles_m * p.tiles_n;
    p.group_m = p.tiles_n > 32 ? 8 : 0;       // wide outputs (LM head): 8 tile-rows per tile-column keep the A band in L2
    p.epi = epi; p.C2 = (unsigned short*)C2; p.H = (const unsigned short*)H;
    if (g_gemm_stages == 3)
        hipLaunchKernelGGL(gemm_nt_bf16_kernel3, dim3((unsigned)p.ntiles), dim3(G_THREADS), 3 * G_STAGE_B, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(gemm_nt_bf16_kernel, dim3((unsigned)p.ntiles), dim3(G_THREADS), G_LDS_B, (hipStream_t)stream, p);
    return check_launch("gemm_nt_bf16_kernel");
}

int kvq_gemm_nt_bf16(const void* A, const void* B, const void* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                     int accumulate, void* stream) {
    return gemm_nt_launch(A, B, bias, C, M, N, K, lda, ldb, ldc, accumulate, EPI_NONE, nullptr, nullptr, stream);
}

int kvq_gemm_nt_bf16_gelu(const void* A, const void* B, const void* bias, void* Hout, void* Aout, int M, int N, int K, int lda,
                          int ldb, int ldc, void* stream) {
    KVQ_REQUIRE(Aout, "kvq_gemm_nt_bf16_gelu: null output");
    return gemm_nt_launch(A, B, bias, Hout, M, N, K, lda, ldb, ldc, 0, EPI_GELU, Aout, nullptr, stream);
}

int kvq_gemm_nt_bf16_dgelu(const void* A, const void* B, const void* H, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                           void* stream) {
    KVQ_REQUIRE(H, "kvq_gemm_nt_bf16_dgelu: null pre-activation");
    return gemm_nt_launch(A, B, nullptr, C, M, N, K, lda, ldb, ldc, 0, EPI_DGELU, nullptr, H, stream);
}

int kvq_transpose_batch_bf16(const void* const* src, void* const* dst, int n, int R, int C, void* stream) {
    KVQ_REQUIRE(src && dst && n >= 1 && R > 0 && C > 0, "kvq_transpose_batch_bf16: bad argument");
    hipStream_t st = (hipStream_t)stream;
    for (int first = 0; first < n; first += KVQ_TRANSPOSE_MAX) {
        TransposeBatch tb;
        tb.R = R; tb.C = C; tb.n = n - first < KVQ_TRANSPOSE_MAX ? n - first : KVQ_TRANSPOSE_MAX;
        for (int i = 0; i < tb.n; ++i) {
            KVQ_REQUIRE(src[first + i] && dst[first + i] && src[first + i] != dst[first + i], "kvq_transpose_batch_bf16: matrix %d: null or in-place", first + i);
            tb.src[i] = (const unsigned short*)src[first + i];
            tb.dst[i] = (unsigned short*)dst[first + i];
        }
        hipLaunchKernelGGL(transpose_batch_kernel, dim3((unsigned)((C + 63) / 64), (unsigned)((R + 63) / 64), (unsigned)tb.n), dim3(256), 0, st, tb);
    }
    return check_launch("transpose_batch_kernel");
}

}  // extern "C"
