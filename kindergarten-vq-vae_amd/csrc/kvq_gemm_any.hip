// kvq_gemm_any.hip -- the any-shape member of the bf16 GEMM family (gfx950).
//
//   C[M,N] (bf16) (+)= op(A) . op(B) (+ bias[N]),  f32 accumulation in ascending k
//
// csrc/kvq_gemm2.hip (LDS-DMA + MFMA) wants K in multiples of 64, M / N / leading dimensions in multiples of 8 and 16-byte aligned
// operands -- true for every product of a BERT-shaped step whose token count is a multiple of 64.  The training step must not
// fall back to a vendor library for the rest (6 x 12 = 72 tokens of a unit test, 9 codes of the reference's Gumbel analysis,
// an odd vocabulary): this kernel takes ANY M, N, K, leading dimension and 2-byte alignment.  It serves launch-latency-sized
// problems; nothing in it is tuned beyond coalesced loads and an LDS tile (64 x 64 outputs per workgroup, 4 x 4 per thread,
// 16 deep), and a problem that meets the MFMA kernel's requirements never comes here (kvq/nnops.py::gemm routes).
// Same operand layouts as kvq_gemm_bf16 (NT forward projection / NN input gradient / TN weight gradient;
// modeling_bert.py:139-352 + autograd), same C ABI conventions.
#include "kvq_common.h"

namespace kvq {

constexpr int GA_T = 64, GA_K = 16;

// element (r, k) of op(X): row-major [rows][K] when kmajor, [K][rows] otherwise
__device__ __forceinline__ float ga_load(const unsigned short* X, bool kmajor, int ld, int r, int k, int rows, int K) {
    if (r >= rows || k >= K) return 0.f;
    return bf16_to_f32(kmajor ? X[(size_t)r * ld + k] : X[(size_t)k * ld + r]);
}

__global__ __launch_bounds__(256) void gemm_any_kernel(const unsigned short* __restrict__ A, const unsigned short* __restrict__ B,
                                                        const unsigned short* __restrict__ bias, unsigned short* C, int M, int N, int K,
                                                        int lda, int ldb, int ldc, int a_kmajor, int b_kmajor, int accumulate) {
    __shared__ float As[GA_K][GA_T + 1], Bs[GA_K][GA_T + 1];
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    const int m0 = blockIdx.y * GA_T, n0 = blockIdx.x * GA_T;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += GA_K) {
        // 64 x 16 elements per operand, four per thread; consecutive threads walk the operand's contiguous direction
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = tid + i * 256;
            {
                const int r = a_kmajor ? e / GA_K : e % GA_T, k = a_kmajor ? e % GA_K : e / GA_T;
                As[k][r] = ga_load(A, a_kmajor, lda, m0 + r, k0 + k, M, K);
            }
            {
                const int r = b_kmajor ? e / GA_K : e % GA_T, k = b_kmajor ? e % GA_K : e / GA_T;
                Bs[k][r] = ga_load(B, b_kmajor, ldb, n0 + r, k0 + k, N, K);
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < GA_K; ++k) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = As[k][ty * 4 + i]; b[i] = Bs[k][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ty * 4 + i;
        if (m >= M) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + tx * 4 + j;
            if (n >= N) continue;
            float v = acc[i][j] + (bias ? bf16_to_f32(bias[n]) : 0.f);
            unsigned short* c = C + (size_t)m * ldc + n;
            if (accumulate) v = bf16_to_f32(f32_to_bf16(v)) + bf16_to_f32(*c);      // as kvq_gemm_bf16: the rounded product joins C
            *c = f32_to_bf16(v);
        }
    }
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_gemm_any_bf16(const void* A, const void* B, const void* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                      int layout, int accumulate, void* stream) {
    KVQ_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0, "kvq_gemm_any_bf16: bad argument");
    KVQ_REQUIRE(layout == KVQ_GEMM_NT || layout == KVQ_GEMM_NN || layout == KVQ_GEMM_TN, "kvq_gemm_any_bf16: unknown layout %d", layout);
    KVQ_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C | (uintptr_t)bias) & 1) == 0, "kvq_gemm_any_bf16: operands must be 2-byte aligned");
    const int a_kmajor = layout != KVQ_GEMM_TN, b_kmajor = layout == KVQ_GEMM_NT;
    KVQ_REQUIRE(lda >= (a_kmajor ? K : M) && ldb >= (b_kmajor ? K : N) && ldc >= N, "kvq_gemm_any_bf16: leading dimension too small");
    const long long gx = ((long long)N + GA_T - 1) / GA_T, gy = ((long long)M + GA_T - 1) / GA_T;
    KVQ_REQUIRE(gy <= 65535, "kvq_gemm_any_bf16: M=%d too large for this kernel", M);
    hipLaunchKernelGGL(gemm_any_kernel, dim3((unsigned)gx, (unsigned)gy), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned short*)A, (const unsigned short*)B, (const unsigned short*)bias, (unsigned short*)C, M, N, K, lda, ldb, ldc,
                       a_kmajor, b_kmajor, accumulate);
    return check_launch("gemm_any_kernel");
}

}  // extern "C"
