// kvq_census.hip -- the consumer of the quantiser's indices: joint (word, code) counts on gfx950.
//
// Replaces the Python walk of analyses/unsupervised_vq_disentanglement/unsupervised_vq_disentanglement.py:166-200, which for
// every sentence, every word and every token of the word does
//     vq_words_distrib[code].append(word); seen_v_is.add(code)                    (:181-183, every token of every word)
//     words_of_interest_vq_distrib[word].append(v_is[0])                          (:192-199, the word's FIRST token)
// and afterwards only uses counts and sets of those lists (:208-235).  Both lists are projections of one table
//     counts[g][w][k] = number of tokens of word w whose factor-g code is k       (all tokens | first tokens only)
// so the device keeps the two tables and the host derives the three result files from them.
//
// Integer work, HBM-bound: 4 + 8 G bytes read per token, nothing written but the tables.  The tables of the reference's
// configuration (K = 9 codes, a few hundred distinct words) are tiny, so plain global atomics would serialise on a handful of
// cache lines: each workgroup counts into a private copy in LDS (ds_add_u32) and adds only its non-zero cells to the global
// tables at the end.  Tables beyond the LDS budget (K = 8192) are sparse per workgroup and take the global atomics directly.
#include "kvq_common.h"

namespace kvq {

constexpr int CENSUS_THREADS = 256;
constexpr int CENSUS_LDS_CELLS = 16384;          // 64 KiB: both tables of G * W * K <= 8192 cells
constexpr int CENSUS_TOKENS_PER_WG = 1024;       // a workgroup's span of tokens (4 per lane): amortises the LDS flush

// slot_first[n] = -1 (no word: padding, or a position past the sentence's words) | (slot << 1) | first-token flag
template <bool LDS>
__global__ __launch_bounds__(CENSUS_THREADS) void census_kernel(const int32_t* __restrict__ slot_first, const int64_t* __restrict__ idx,
                                                                int64_t N, int G, int K, int W, uint32_t* __restrict__ all,
                                                                uint32_t* __restrict__ first, uint32_t* __restrict__ n_bad) {
    extern __shared__ uint32_t cells[];          // [2][G][W][K] when LDS
    const int64_t table = (int64_t)G * W * K;
    if (LDS) {
        for (int c = threadIdx.x; c < 2 * table; c += CENSUS_THREADS) cells[c] = 0u;
        __syncthreads();
    }
    const int64_t n0 = (int64_t)blockIdx.x * CENSUS_TOKENS_PER_WG;
    const int64_t n1 = n0 + CENSUS_TOKENS_PER_WG < N ? n0 + CENSUS_TOKENS_PER_WG : N;
    unsigned bad = 0;
    for (int64_t n = n0 + threadIdx.x; n < n1; n += CENSUS_THREADS) {      // consecutive lanes, consecutive tokens
        const int32_t sf = slot_first[n];
        if (sf < 0) continue;
        const int w = sf >> 1;
        if (w >= W) { ++bad; continue; }
        for (int g = 0; g < G; ++g) {
            const int64_t k = idx[n * G + g];
            if (k < 0 || k >= K) { ++bad; continue; }
            const int64_t cell = ((int64_t)g * W + w) * K + k;
            if (LDS) {
                atomicAdd(&cells[cell], 1u);
                if (sf & 1) atomicAdd(&cells[table + cell], 1u);
            } else {
                atomicAdd(&all[cell], 1u);
                if (sf & 1) atomicAdd(&first[cell], 1u);
            }
        }
    }
    if (bad && n_bad) atomicAdd(n_bad, bad);
    if (LDS) {
        __syncthreads();
        for (int c = threadIdx.x; c < table; c += CENSUS_THREADS) {
            const uint32_t a = cells[c], f = cells[table + c];
            if (a) atomicAdd(&all[c], a);
            if (f) atomicAdd(&first[c], f);
        }
    }
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_code_census(const int32_t* slot_first, const int64_t* idx, int64_t N, int G, int K, int W, uint32_t* counts_all,
                    uint32_t* counts_first, uint32_t* n_bad, void* stream) {
    KVQ_REQUIRE(N >= 0 && G >= 1 && K >= 1 && W >= 1, "kvq_code_census: N >= 0, G, K, W >= 1 required (N=%lld G=%d K=%d W=%d)", (long long)N, G, K, W);
    KVQ_REQUIRE((int64_t)G * W * K < (int64_t)1 << 31, "kvq_code_census: G * W * K = %lld cells do not fit 31 bits", (long long)G * W * K);
    if (N == 0) return KVQ_OK;
    KVQ_REQUIRE(slot_first && idx && counts_all && counts_first, "kvq_code_census: null pointer argument");
    hipStream_t st = (hipStream_t)stream;
    const unsigned grid = (unsigned)((N + CENSUS_TOKENS_PER_WG - 1) / CENSUS_TOKENS_PER_WG);
    const int64_t table = (int64_t)G * W * K;
    if (2 * table <= CENSUS_LDS_CELLS)
        hipLaunchKernelGGL(census_kernel<true>, dim3(grid), dim3(CENSUS_THREADS), (size_t)(2 * table) * sizeof(uint32_t), st, slot_first, idx, N, G, K, W,
                           counts_all, counts_first, n_bad);
    else
        hipLaunchKernelGGL(census_kernel<false>, dim3(grid), dim3(CENSUS_THREADS), 0, st, slot_first, idx, N, G, K, W, counts_all, counts_first, n_bad);
    return check_launch("census_kernel");
}

}  // extern "C"
