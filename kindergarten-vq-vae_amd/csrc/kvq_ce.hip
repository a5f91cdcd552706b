// kvq_ce.hip -- reconstruction loss of step() on gfx950.
//
// Replaces models/shelgon3/Trainer.py:94-101:
//     kl_div(log_softmax(logits.reshape(-1,V)), one_hot(ids,V).reshape(-1,V).float(), "batchmean")
//     recon_ids = argmax(softmax(logits)) ; seq_acc(recon_ids, ids)          (common/metrics.py:18-30)
// With an exactly one-hot target the KL sum is  logsumexp(x_n) - x_n[id_n]  per row, so the [N,V] one-hot
// (1 GB at N=8192) and the [N,V] log-softmax are never materialised: one pass over the logits per direction.
// HBM-bound: forward reads N*V elements once; backward reads and writes them once (in place allowed).
// One workgroup per row; rows are only 4-byte aligned (V = 30522), so each row is walked as
// scalar head | 16-byte vectors | scalar tail.
#include <math.h>

#include "kvq_common.h"

namespace kvq {

constexpr int CE_THREADS = 256;

template <int DT>
struct RowWalk {
    static constexpr int VEC = 16 / IO<DT>::bytes;
    const char* base;  // row start
    int V, head, nvec, tail0;
    __device__ RowWalk(const void* p, int64_t n, int V_, int64_t ld) : V(V_) {
        base = reinterpret_cast<const char*>(p) + (size_t)n * ld * IO<DT>::bytes;
        const int mis = (int)((uintptr_t)base & 15);
        head = mis ? (16 - mis) / IO<DT>::bytes : 0;
        head = head < V ? head : V;
        nvec = (V - head) / VEC;
        tail0 = head + nvec * VEC;
    }
};

template <int DT>
__device__ __forceinline__ void load_vec(const char* p, float (&v)[RowWalk<DT>::VEC]);
template <>
__device__ __forceinline__ void load_vec<KVQ_F32>(const char* p, float (&v)[4]) {
    const f32x4 r = *reinterpret_cast<const f32x4*>(p);
    v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}
template <>
__device__ __forceinline__ void load_vec<KVQ_BF16>(const char* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}

struct MaxSum {
    float m, s;   // running max, sum of exp(x - m)
    float bv;     // best value for argmax
    int bi;       // its (lowest) index
};

__device__ __forceinline__ void ms_push(MaxSum& a, float x, int i) {
    if (x > a.m) { a.s = a.s * __expf(a.m - x) + 1.0f; a.m = x; }
    else a.s += __expf(x - a.m);
    if (x > a.bv || (x == a.bv && i < a.bi)) { a.bv = x; a.bi = i; }
}
// a whole 16-byte vector at once: one rescale at most, VEC exponentials, the arg-max only looked at when the vector's
// maximum beats the running one (indices grow along a thread's walk, so an equal value never replaces an earlier one)
template <int VEC>
__device__ __forceinline__ void ms_push_vec(MaxSum& a, const float (&v)[VEC], int i0) {
    float vm = v[0];
#pragma unroll
    for (int u = 1; u < VEC; ++u) vm = fmaxf(vm, v[u]);
    if (vm > a.m) { a.s *= __expf(a.m - vm); a.m = vm; }
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < VEC; ++u) s += __expf(v[u] - a.m);
    a.s += s;
    if (vm > a.bv) {
        int iu = VEC - 1;
#pragma unroll
        for (int u = VEC - 2; u >= 0; --u) iu = v[u] == vm ? u : iu;
        a.bv = vm; a.bi = i0 + iu;
    }
}
__device__ __forceinline__ void ms_merge(MaxSum& a, float m, float s, float bv, int bi) {
    const float M = fmaxf(a.m, m);
    const float sa = (a.m == -INFINITY) ? 0.f : a.s * __expf(a.m - M);
    const float sb = (m == -INFINITY) ? 0.f : s * __expf(m - M);
    a.m = M; a.s = sa + sb;
    if (bv > a.bv || (bv == a.bv && bi < a.bi)) { a.bv = bv; a.bi = bi; }
}

template <int DT>
__global__ __launch_bounds__(CE_THREADS) void ce_fwd_kernel(const void* __restrict__ logits, const int64_t* __restrict__ target,
                                                             int64_t N, int V, int64_t ld, float* __restrict__ row_loss,
                                                             float* __restrict__ row_lse, int64_t* __restrict__ pred) {
    __shared__ float sm[CE_THREADS / WAVE], ss[CE_THREADS / WAVE], sv[CE_THREADS / WAVE];
    __shared__ int si[CE_THREADS / WAVE];
    const int64_t n = blockIdx.x;
    const int t = threadIdx.x;
    RowWalk<DT> rw(logits, n, V, ld);
    constexpr int VEC = RowWalk<DT>::VEC;
    MaxSum a = {-INFINITY, 0.f, -INFINITY, INT_MAX};
    if (t < rw.head) ms_push(a, IO<DT>::load1(rw.base, t), t);
    for (int q = t; q < rw.nvec; q += CE_THREADS) {
        float v[VEC];
        load_vec<DT>(rw.base + ((size_t)rw.head + (size_t)q * VEC) * IO<DT>::bytes, v);
        ms_push_vec<VEC>(a, v, rw.head + q * VEC);
    }
    for (int j = rw.tail0 + t; j < V; j += CE_THREADS) ms_push(a, IO<DT>::load1(rw.base, j), j);
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1)
        ms_merge(a, __shfl_xor(a.m, mk, WAVE), __shfl_xor(a.s, mk, WAVE), __shfl_xor(a.bv, mk, WAVE), __shfl_xor(a.bi, mk, WAVE));
    const int w = t >> 6;
    if ((t & 63) == 0) { sm[w] = a.m; ss[w] = a.s; sv[w] = a.bv; si[w] = a.bi; }
    __syncthreads();
    if (t == 0) {
        for (int ww = 1; ww < CE_THREADS / WAVE; ++ww) ms_merge(a, sm[ww], ss[ww], sv[ww], si[ww]);
        const float lse = a.m + logf(a.s);
        const int64_t tg = target[n];
        const float xt = IO<DT>::load1(rw.base, (size_t)tg);
        row_lse[n] = lse;
        row_loss[n] = lse - xt;
        pred[n] = (int64_t)a.bi;
    }
}

// loss = mean(row_loss), acc = mean(pred == target): single workgroup, fixed order, f64 accumulate
__global__ __launch_bounds__(256) void ce_finalize_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ pred,
                                                           const int64_t* __restrict__ target, int64_t N, float* loss, float* acc) {
    __shared__ double sd[256];
    __shared__ unsigned sh[256];
    const int t = threadIdx.x;
    double a = 0.0;
    unsigned hits = 0;
    for (int64_t n = t; n < N; n += 256) {
        a += (double)row_loss[n];
        hits += pred[n] == target[n];
    }
    sd[t] = a; sh[t] = hits;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (t < s) { sd[t] += sd[t + s]; sh[t] += sh[t + s]; }
        __syncthreads();
    }
    if (t == 0) {
        if (loss) *loss = (float)(sd[0] / (double)N);
        if (acc) *acc = (float)((double)sh[0] / (double)N);
    }
}

// seq_acc's second result (common/metrics.py:32-36): per_sentence[b] = mean_s(pred[b, s] == target[b, s]).  One wave per sentence.
__global__ __launch_bounds__(256) void seq_acc_kernel(const int64_t* __restrict__ pred, const int64_t* __restrict__ target, int64_t B,
                                                       int S, float* __restrict__ per_sentence) {
    const int64_t b = (int64_t)blockIdx.x * (256 / WAVE) + (threadIdx.x >> 6);
    if (b >= B) return;
    const int lane = threadIdx.x & 63;
    int hits = 0;
    for (int s = lane; s < S; s += WAVE) hits += pred[b * S + s] == target[b * S + s];
#pragma unroll
    for (int mk = 32; mk >= 1; mk >>= 1) hits += __shfl_xor(hits, mk, WAVE);
    if (lane == 0) per_sentence[b] = (float)hits / (float)S;
}

// The same row results from per-tile statistics left by the LM-head GEMM's epilogue (kvq_gemm_bf16_ce): stats [N][tiles][4] =
// (max, sum of exp(x - max), first arg-max as int bits, -) of row n over the columns of tile t below V.  One thread per row merges
// the tiles in column order (so the first maximum wins) and reads ONE logit, the target's: the [N, V] logits are not read again.
template <int DT>
__global__ __launch_bounds__(256) void ce_from_stats_kernel(const void* __restrict__ logits, const int64_t* __restrict__ target,
                                                             int64_t N, int64_t ld, const float4* __restrict__ stats, int tiles,
                                                             float* __restrict__ row_loss, float* __restrict__ row_lse,
                                                             int64_t* __restrict__ pred) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    float m = -INFINITY, s = 0.f;
    int bi = INT_MAX;
    const float4* row = stats + (size_t)n * tiles;
    for (int t = 0; t < tiles; ++t) {
        const float4 e = row[t];
        const int ebi = __float_as_int(e.z);
        const float nm = fmaxf(m, e.x);
        const float sa = m > -INFINITY ? s * __expf(m - nm) : 0.f, sb = e.x > -INFINITY ? e.y * __expf(e.x - nm) : 0.f;
        bi = (e.x > m || (e.x == m && ebi < bi)) ? ebi : bi;
        m = nm; s = sa + sb;
    }
    const float lse = m + logf(s);
    const float xt = IO<DT>::load1(logits, (size_t)n * ld + (size_t)target[n]);
    row_lse[n] = lse;
    row_loss[n] = lse - xt;
    pred[n] = (int64_t)bi;
}

template <int DT>
__global__ __launch_bounds__(CE_THREADS) void ce_bwd_kernel(const void* logits, const int64_t* __restrict__ target,
                                                             const float* __restrict__ row_lse, const float* __restrict__ g_loss,
                                                             int64_t N, int V, int64_t ld, void* g_logits) {
    const int64_t n = blockIdx.x;
    const int t = threadIdx.x;
    RowWalk<DT> rw(logits, n, V, ld);
    constexpr int VEC = RowWalk<DT>::VEC;
    char* out = reinterpret_cast<char*>(g_logits) + (size_t)n * ld * IO<DT>::bytes;
    for (int j = V + t; j < ld; j += CE_THREADS) IO<DT>::store1(out, j, 0.f);   // padding columns carry no gradient
    const float c = (g_loss ? *g_loss : 1.0f) / (float)N;
    const float lse = row_lse[n];
    const int tg = (int)target[n];
    if (t < rw.head) {
        const float x = IO<DT>::load1(rw.base, t);
        IO<DT>::store1(out, t, c * (__expf(x - lse) - (t == tg ? 1.f : 0.f)));
    }
    for (int q = t; q < rw.nvec; q += CE_THREADS) {
        float v[VEC];
        const size_t off = (size_t)rw.head + (size_t)q * VEC;
        load_vec<DT>(rw.base + off * IO<DT>::bytes, v);
#pragma unroll
        for (int u = 0; u < VEC; ++u) v[u] = c * (__expf(v[u] - lse) - ((int)off + u == tg ? 1.f : 0.f));
        if (DT == KVQ_F32) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            *reinterpret_cast<f32x4*>(out + off * 4) = o;
        } else {
            uint4 o;
            o.x = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
            o.y = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
            o.z = (unsigned)f32_to_bf16(v[4 % VEC]) | ((unsigned)f32_to_bf16(v[5 % VEC]) << 16);
            o.w = (unsigned)f32_to_bf16(v[6 % VEC]) | ((unsigned)f32_to_bf16(v[7 % VEC]) << 16);
            *reinterpret_cast<uint4*>(out + off * 2) = o;
        }
    }
    for (int j = rw.tail0 + t; j < V; j += CE_THREADS) {
        const float x = IO<DT>::load1(rw.base, j);
        IO<DT>::store1(out, j, c * (__expf(x - lse) - (j == tg ? 1.f : 0.f)));
    }
}

// Backward tiled over (column slab, row block) -- every element only needs its row's lse -- which lets a thread keep 8 columns
// and accumulate their sums over the block's rows: the LM-head bias gradient (colsum of g_logits) comes out as partial rows
// instead of a second 500 MB pass.  Needs 16-byte aligned rows (ld % 8 == 0 elements, aligned base).
constexpr int CEB_ROWS = 32;
template <int DT>
__global__ __launch_bounds__(256) void ce_bwd_tiled_kernel(const void* logits, const int64_t* __restrict__ target,
                                                            const float* __restrict__ row_lse, const float* __restrict__ g_loss,
                                                            int64_t N, int V, int64_t ld, void* g_logits, float* __restrict__ part) {
    const int64_t c0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (c0 >= ld) return;
    const int64_t r0 = (int64_t)blockIdx.y * CEB_ROWS;
    const int64_t r1 = r0 + CEB_ROWS < N ? r0 + CEB_ROWS : N;
    const float c = (g_loss ? *g_loss : 1.0f) / (float)N;
    f32x4 acc_lo = {0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
    for (int64_t r = r0; r < r1; r += 4) {
        f32x8 x[4];
        float lse[4];
        int tg[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t rr = r + u < r1 ? r + u : r1 - 1;               // clamped, never branched around
            x[u] = IO<DT>::load8(logits, (size_t)rr * ld + c0);
            lse[u] = row_lse[rr];
            tg[u] = (int)target[rr];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (r + u >= r1) break;
            f32x8 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int cl = (int)c0 + e, ch = (int)c0 + 4 + e;
                o.lo[e] = cl < V ? c * (__expf(x[u].lo[e] - lse[u]) - (cl == tg[u] ? 1.f : 0.f)) : 0.f;   // padding columns: no gradient
                o.hi[e] = ch < V ? c * (__expf(x[u].hi[e] - lse[u]) - (ch == tg[u] ? 1.f : 0.f)) : 0.f;
            }
            IO<DT>::store8(g_logits, (size_t)(r + u) * ld + c0, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc_lo[e] += IO<DT>::round(o.lo[e]); acc_hi[e] += IO<DT>::round(o.hi[e]); }
        }
    }
    if (part) {
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * ld + c0) = acc_lo;
        *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * ld + c0 + 4) = acc_hi;
    }
}

}  // namespace kvq

using namespace kvq;

extern "C" {

int kvq_ce_forward(const void* logits, const int64_t* target, int64_t N, int V, int64_t ld, int io_dtype, float* row_loss,
                   float* row_lse, int64_t* pred, float* loss, float* acc, void* stream) {
    KVQ_REQUIRE(logits && target && row_loss && row_lse && pred, "kvq_ce_forward: null pointer argument");
    KVQ_REQUIRE(N > 0 && V > 0 && N < (1ll << 31) && ld >= V, "kvq_ce_forward: N=%lld V=%d ld=%lld out of range", (long long)N, V, (long long)ld);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "kvq_ce_forward: unsupported io dtype %d", io_dtype);
    KVQ_REQUIRE(((uintptr_t)logits & (io_dtype == KVQ_F32 ? 3 : 1)) == 0, "kvq_ce_forward: misaligned logits");
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == KVQ_F32)
        hipLaunchKernelGGL(ce_fwd_kernel<KVQ_F32>, dim3((unsigned)N), dim3(CE_THREADS), 0, st, logits, target, N, V, ld, row_loss, row_lse, pred);
    else
        hipLaunchKernelGGL(ce_fwd_kernel<KVQ_BF16>, dim3((unsigned)N), dim3(CE_THREADS), 0, st, logits, target, N, V, ld, row_loss, row_lse, pred);
    int rc = check_launch("ce_fwd_kernel");
    if (rc) return rc;
    if (loss || acc) {
        hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, row_loss, pred, target, N, loss, acc);
        rc = check_launch("ce_finalize_kernel");
    }
    return rc;
}

int kvq_seq_acc(const int64_t* pred, const int64_t* target, int64_t B, int S, float* per_sentence, void* stream) {
    KVQ_REQUIRE(pred && target && per_sentence, "kvq_seq_acc: null pointer argument");
    KVQ_REQUIRE(B > 0 && S > 0 && B < (1ll << 31), "kvq_seq_acc: B=%lld S=%d out of range", (long long)B, S);
    hipLaunchKernelGGL(seq_acc_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pred, target, B, S, per_sentence);
    return check_launch("seq_acc_kernel");
}

int kvq_ce_forward_stats(const void* logits, const int64_t* target, int64_t N, int64_t ld, int io_dtype, const float* stats, int tiles,
                         float* row_loss, float* row_lse, int64_t* pred, float* loss, float* acc, void* stream) {
    KVQ_REQUIRE(logits && target && stats && row_loss && row_lse && pred, "kvq_ce_forward_stats: null pointer argument");
    KVQ_REQUIRE(N > 0 && N < (1ll << 31) && tiles > 0 && ld > 0, "kvq_ce_forward_stats: N=%lld tiles=%d ld=%lld out of range", (long long)N, tiles, (long long)ld);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "kvq_ce_forward_stats: unsupported io dtype %d", io_dtype);
    KVQ_REQUIRE(((uintptr_t)stats) % 16 == 0, "kvq_ce_forward_stats: 16-byte aligned statistics required");
    hipStream_t st = (hipStream_t)stream;
    const unsigned blocks = (unsigned)((N + 255) / 256);
    const float4* s4 = reinterpret_cast<const float4*>(stats);
    if (io_dtype == KVQ_F32)
        hipLaunchKernelGGL(ce_from_stats_kernel<KVQ_F32>, dim3(blocks), dim3(256), 0, st, logits, target, N, ld, s4, tiles, row_loss, row_lse, pred);
    else
        hipLaunchKernelGGL(ce_from_stats_kernel<KVQ_BF16>, dim3(blocks), dim3(256), 0, st, logits, target, N, ld, s4, tiles, row_loss, row_lse, pred);
    int rc = check_launch("ce_from_stats_kernel");
    if (rc) return rc;
    if (loss || acc) {
        hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(256), 0, st, row_loss, pred, target, N, loss, acc);
        rc = check_launch("ce_finalize_kernel");
    }
    return rc;
}

int kvq_ce_backward(const void* logits, const int64_t* target, const float* row_lse, const float* g_loss, int64_t N,
                    int V, int64_t ld, int io_dtype, void* g_logits, void* stream) {
    KVQ_REQUIRE(logits && target && row_lse && g_logits, "kvq_ce_backward: null pointer argument");
    KVQ_REQUIRE(N > 0 && V > 0 && N < (1ll << 31) && ld >= V, "kvq_ce_backward: N=%lld V=%d ld=%lld out of range", (long long)N, V, (long long)ld);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "kvq_ce_backward: unsupported io dtype %d", io_dtype);
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == KVQ_F32)
        hipLaunchKernelGGL(ce_bwd_kernel<KVQ_F32>, dim3((unsigned)N), dim3(CE_THREADS), 0, st, logits, target, row_lse, g_loss, N, V, ld, g_logits);
    else
        hipLaunchKernelGGL(ce_bwd_kernel<KVQ_BF16>, dim3((unsigned)N), dim3(CE_THREADS), 0, st, logits, target, row_lse, g_loss, N, V, ld, g_logits);
    return check_launch("ce_bwd_kernel");
}

int64_t kvq_ce_bwd_partial_rows(int64_t N) { return (N + CEB_ROWS - 1) / CEB_ROWS; }

int kvq_ce_backward_bias(const void* logits, const int64_t* target, const float* row_lse, const float* g_loss, int64_t N,
                         int V, int64_t ld, int io_dtype, void* g_logits, float* bias_part, size_t part_bytes, void* stream) {
    KVQ_REQUIRE(logits && target && row_lse && g_logits && bias_part, "kvq_ce_backward_bias: null pointer argument");
    KVQ_REQUIRE(N > 0 && V > 0 && N < (1ll << 31) && ld >= V, "kvq_ce_backward_bias: N=%lld V=%d ld=%lld out of range", (long long)N, V, (long long)ld);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "kvq_ce_backward_bias: unsupported io dtype %d", io_dtype);
    KVQ_REQUIRE(ld % 8 == 0 && (((uintptr_t)logits | (uintptr_t)g_logits | (uintptr_t)bias_part) & 15) == 0,
                "kvq_ce_backward_bias: rows must be 16-byte aligned (ld %% 8 == 0)");
    const int64_t P = (N + CEB_ROWS - 1) / CEB_ROWS;
    if (part_bytes < (size_t)P * ld * sizeof(float)) return fail(KVQ_E_WORKSPACE, "kvq_ce_backward_bias: partial buffer %zu < %zu", part_bytes, (size_t)P * ld * sizeof(float));
    KVQ_REQUIRE(P <= 65535, "kvq_ce_backward_bias: N too large");
    dim3 grid((unsigned)((ld + 2047) / 2048), (unsigned)P);
    hipStream_t st = (hipStream_t)stream;
    if (io_dtype == KVQ_F32)
        hipLaunchKernelGGL(ce_bwd_tiled_kernel<KVQ_F32>, grid, dim3(256), 0, st, logits, target, row_lse, g_loss, N, V, ld, g_logits, bias_part);
    else
        hipLaunchKernelGGL(ce_bwd_tiled_kernel<KVQ_BF16>, grid, dim3(256), 0, st, logits, target, row_lse, g_loss, N, V, ld, g_logits, bias_part);
    return check_launch("ce_bwd_tiled_kernel");
}

}  // extern "C"
