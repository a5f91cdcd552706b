// kvq_nn.hip -- the memory-bound pieces of the BERT encoder/decoder blocks of the training step, fused for gfx950.
//
// The reference runs these as separate ATen ops inside HuggingFace's BertLayer (modeling_bert.py:139-352): bias add,
// dropout, residual add, LayerNorm, GELU, softmax attention, and torch.optim.Adam.  Here each block boundary is one
// kernel pass over the activations (bf16 storage, f32 arithmetic):
//   kvq_dropout_residual_ln_{fwd,bwd}   BertSelfOutput / BertOutput (:282-293, :340-352): LN(dropout(y) + residual)
//   kvq_gelu_{fwd,bwd}                   BertIntermediate activation (:325-337), erf form
//   kvq_colsum                           bias gradients (sum over tokens)
//   kvq_attn_{fwd,bwd}                   BertSelfAttention / BertCrossAttention core (:111-204) for sentences of <= 32 tokens:
//                                        one wave per (sentence, head), scores never leave registers / LDS
//   kvq_adam_step                        torch.optim.Adam semantics (main.py:91) on flat buffers + bf16 shadow weights
// Dropout masks are never stored: both directions regenerate them from a counter-based generator (Philox4x32-10)
// keyed by (seed, site) and indexed by the element position.
#include <math.h>

#include <stdlib.h>

#include "kvq_common.h"
#include <type_traits>

namespace kvq {

// (Philox4x32-10, drop_bits, keep_scale, drop_threshold: kvq_common.h -- the GEMM epilogue of csrc/kvq_gemm2.hip draws the same masks)

// ---------------------------------------------------------------------------------------------------------------
// LN(dropout(y) + residual): one wave per row, rows of H <= 4096 (H % 4 == 0); bf16 or f32 io
// ---------------------------------------------------------------------------------------------------------------
constexpr int LN_MAX_PER_LANE = 16;   // H <= 64 * 4 * 16 = 4096

// RES = false (round 5): no residual operand at all -- the LayerNorm-only pass behind kvq_gemm_bf16_dropres, whose GEMM epilogue
// already added the residual: one load stream instead of two.
template <int DT, int PER, bool RES = true>
__global__ __launch_bounds__(256) void drln_fwd_kernel(const void* __restrict__ y, const void* __restrict__ resid,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        int64_t N, int H, float eps, float p_drop, unsigned thresh,
                                                        unsigned long long seed, const unsigned long long* __restrict__ seed_off,
                                                        unsigned site, void* __restrict__ out,
                                                        void* __restrict__ pre, float* __restrict__ mean_out,
                                                        float* __restrict__ rstd_out, unsigned char* __restrict__ out8 = nullptr,
                                                        float* __restrict__ st8 = nullptr) {
    // out8 / st8 (round 5, bf16 io only): also the fp8 (e4m3) copy of `out` for the fp8 GEMM that reads it next -- the bytes
    // kvq_fp8_quantize_delayed(out) would write (this site's scale of the previous step, this call's amax noted in its state)
    // EVERY load of the row is requested before anything is done with one of them, and no load sits behind a branch: chunk
    // indices past the row are clamped (and masked out of the sums), a missing residual reads y again and is weighted 0.
    // As first written ("if (c < nchunk) { load y; ...; if (resid) load resid; ... }" per chunk) hipcc put s_waitcnt vmcnt(0)
    // behind each load: six dependent trips to memory per row, then three more for gamma / beta -- 13.3 us for the 50 MB of a
    // [8192, 768] call, one round of waves doing nothing but waiting (ISA audit of round 3, profiles/NOTES_r01-r03_design_and_experiments.md section 2.5).
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int nchunk = H >> 2;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    const void* rp = resid ? resid : y;
    const float rw = resid ? 1.0f : 0.0f;
    constexpr bool EARLY_GB = PER <= 4;        // (wide rows: gamma / beta stay L2-resident small loads of the last pass)
    f32x4 v[PER], r[RES ? PER : 1], g[EARLY_GB ? PER : 1], b[EARLY_GB ? PER : 1];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int c = lane + WAVE * t < nchunk ? lane + WAVE * t : nchunk - 1;
        const size_t off = (size_t)row * H + 4 * c;
        v[t] = IO<DT>::load4(y, off);
        if constexpr (RES) r[t] = IO<DT>::load4(rp, off);
    }
    if constexpr (EARLY_GB) {
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int c = lane + WAVE * t < nchunk ? lane + WAVE * t : nchunk - 1;
            g[t] = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
            b[t] = *reinterpret_cast<const f32x4*>(beta + 4 * c);
        }
    }
    if (seed_off) seed += *seed_off;          // device-resident step counter (a captured graph replays with fresh masks): the
                                              // dependent scalar load waits while the row is on its way
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int c = lane + WAVE * t;
        f32x4 a = v[t];
        if (p_drop > 0.f) {                   // (uniform; arithmetic only: the Philox rounds run while the loads are in flight)
            const U4 kb = drop_bits(seed, site, (unsigned long long)row * nchunk + c);
            a.x *= keep_scale(kb.x, thresh, inv_keep); a.y *= keep_scale(kb.y, thresh, inv_keep);
            a.z *= keep_scale(kb.z, thresh, inv_keep); a.w *= keep_scale(kb.w, thresh, inv_keep);
        }
        if constexpr (RES) a += r[t] * rw;
        // LayerNorm sees the STORED pre-activation (bf16-rounded when io is bf16): backward re-reads exactly that
        a.x = IO<DT>::round(a.x); a.y = IO<DT>::round(a.y); a.z = IO<DT>::round(a.z); a.w = IO<DT>::round(a.w);
        if (c < nchunk) {
            if (pre) IO<DT>::store4(pre, (size_t)row * H + 4 * c, a);
            sum += (a.x + a.y) + (a.z + a.w);
        }
        v[t] = a;
    }
    const float mean = wave_sum_f32(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        if (lane + WAVE * t < nchunk) {
            const f32x4 d = v[t] - mean;
            sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
    const float var = wave_sum_f32(sq) / (float)H;
    const float rstd = rsqrtf(var + eps);
    const float s8 = (DT == KVQ_BF16 && out8) ? st8[0] : 1.0f;
    float am8 = 0.f;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int c = lane + WAVE * t;
        if (c < nchunk) {
            f32x4 gg, bb;
            if constexpr (EARLY_GB) { gg = g[t]; bb = b[t]; }
            else { gg = *reinterpret_cast<const f32x4*>(gamma + 4 * c); bb = *reinterpret_cast<const f32x4*>(beta + 4 * c); }
            const f32x4 o = (v[t] - mean) * rstd * gg + bb;
            IO<DT>::store4(out, (size_t)row * H + 4 * c, o);
            if (DT == KVQ_BF16 && out8) {                     // (uniform)
                const f32x4 ob = {IO<DT>::round(o.x), IO<DT>::round(o.y), IO<DT>::round(o.z), IO<DT>::round(o.w)};
                am8 = fmaxf(am8, amax4(ob));
                *reinterpret_cast<unsigned*>(out8 + (size_t)row * H + 4 * c) = quant4(ob, s8);
            }
        }
    }
    if (DT == KVQ_BF16 && out8) fp8_amax_note(am8, st8, (unsigned)row);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// BertEmbeddings (modeling_bert.py:53-110) in one pass: out = dropout(LayerNorm(word[ids[n]] + (pos[n % S] + type[0]))).
// Replaces F.embedding + the position/type add + its repeat over the batch + the LayerNorm kernel + the dropout kernel (five
// launches, three [N, H] round trips).  Rounding as those kernels rounded: pos + type to the io dtype, the sum to the io dtype
// (= `pre`, what backward re-reads), the LayerNorm output to the io dtype BEFORE the keep scale.  One wave per row.
template <int DT, int PER>
__global__ __launch_bounds__(256) void embed_ln_fwd_kernel(const int64_t* __restrict__ ids, const void* __restrict__ word,
                                                            const void* __restrict__ pos, const void* __restrict__ type_row,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            int64_t N, int S, int H, int64_t V, float eps, float p_drop, unsigned thresh,
                                                            unsigned long long seed, const unsigned long long* __restrict__ seed_off,
                                                            unsigned site, void* __restrict__ out, void* __restrict__ pre,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    if (seed_off) seed += *seed_off;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const int nchunk = H >> 2;
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);                 // (torch raises on an id outside the table; the kernel must not fault)
    const int ps = (int)(row % S);
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    f32x4 v[PER];
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int c = lane + WAVE * t;
        f32x4 a = {0.f, 0.f, 0.f, 0.f};
        if (c < nchunk) {
            f32x4 pt = IO<DT>::load4(pos, (size_t)ps * H + 4 * c) + IO<DT>::load4(type_row, 4 * c);
            pt.x = IO<DT>::round(pt.x); pt.y = IO<DT>::round(pt.y); pt.z = IO<DT>::round(pt.z); pt.w = IO<DT>::round(pt.w);
            a = IO<DT>::load4(word, (size_t)id * H + 4 * c) + pt;
            a.x = IO<DT>::round(a.x); a.y = IO<DT>::round(a.y); a.z = IO<DT>::round(a.z); a.w = IO<DT>::round(a.w);
            if (pre) IO<DT>::store4(pre, (size_t)row * H + 4 * c, a);
            sum += (a.x + a.y) + (a.z + a.w);
        }
        v[t] = a;
    }
    const float mean = wave_sum_f32(sum) / (float)H;
    float sq = 0.f;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        if (lane + WAVE * t < nchunk) {
            const f32x4 d = v[t] - mean;
            sq += (d.x * d.x + d.y * d.y) + (d.z * d.z + d.w * d.w);
        }
    }
    const float var = wave_sum_f32(sq) / (float)H;
    const float rstd = rsqrtf(var + eps);
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int c = lane + WAVE * t;
        if (c < nchunk) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
            const f32x4 b = *reinterpret_cast<const f32x4*>(beta + 4 * c);
            f32x4 o = (v[t] - mean) * rstd * g + b;
            if (p_drop > 0.f) {
                const U4 kb = drop_bits(seed, site, (unsigned long long)row * nchunk + c);
                o.x = IO<DT>::round(o.x) * keep_scale(kb.x, thresh, inv_keep); o.y = IO<DT>::round(o.y) * keep_scale(kb.y, thresh, inv_keep);
                o.z = IO<DT>::round(o.z) * keep_scale(kb.z, thresh, inv_keep); o.w = IO<DT>::round(o.w) * keep_scale(kb.w, thresh, inv_keep);
            }
            IO<DT>::store4(out, (size_t)row * H + 4 * c, o);
        }
    }
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
}

// backward: g_pre = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat));  g_resid = g_pre;
//           g_y = g_pre * dropout_mask/(1-p);  dgamma/dbeta partials per workgroup (summed by colsum_final_kernel)
constexpr int LNB_ROWS = 16;   // rows per workgroup (4 waves x 4 rows, all in flight at once): 512 workgroups and 512 partial rows at N = 8192

template <int DT, int PER>
__global__ __launch_bounds__(256) void drln_bwd_kernel(const void* __restrict__ g_out, const void* __restrict__ pre,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        const float* __restrict__ gamma, int64_t N, int H, float p_drop,
                                                        unsigned thresh, unsigned long long seed,
                                                        const unsigned long long* __restrict__ seed_off, unsigned site,
                                                        void* __restrict__ g_y, void* __restrict__ g_resid,
                                                        float* __restrict__ part_dgamma, int want_dbias, int drop_on_out) {
    // drop_on_out: the block is dropout(LayerNorm(.)) (BertEmbeddings) instead of LayerNorm(dropout(y) + resid): the mask of
    // (seed, site) then applies to the incoming gradient g_out (rounded to the io dtype, as the separate kernel stored it)
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [4 waves][3][H]
    if (seed_off) seed += *seed_off;
    constexpr int RW = LNB_ROWS / 4;                              // rows per wave
    constexpr int RF = PER <= 4 ? RW : 1;                         // ... of which RF are in flight at once (register budget)
    typedef typename std::conditional<(PER > 8), unsigned long long, unsigned>::type KeepBits;   // 4 bits per chunk
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nchunk = H >> 2;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    f32x4 dg[PER], db[PER], dy[PER], gm[PER];     // dy: column sums of g_y = bias gradient of the dense layer in front
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        dg[t] = 0.f; db[t] = 0.f; dy[t] = 0.f;
        const int c = lane + WAVE * t < nchunk ? lane + WAVE * t : nchunk - 1;
        gm[t] = *reinterpret_cast<const f32x4*>(gamma + 4 * c);
    }
#pragma unroll
    for (int rb = 0; rb < RW; rb += RF) {
        const int64_t row0 = (int64_t)blockIdx.x * LNB_ROWS + w * RW + rb;
        if (row0 >= N) break;                                     // wave-uniform
        // 1. every load of these rows goes out first (rows past N are clamped and weighted 0: no branch around a load)
        f32x4 go[RF][PER], x[RF][PER];
        float mu[RF], rs[RF], live[RF];
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            const int64_t row = row0 + r < N ? row0 + r : N - 1;
            live[r] = row0 + r < N ? 1.0f : 0.0f;
            mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int c = lane + WAVE * t < nchunk ? lane + WAVE * t : nchunk - 1;
                const size_t off = (size_t)row * H + 4 * c;
                go[r][t] = IO<DT>::load4(g_out, off);
                x[r][t] = IO<DT>::load4(pre, off);
            }
        }
        // 2. the dropout keep bits do not depend on the loads: Philox runs while they are in flight
        KeepBits keepbits[RF];
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            keepbits[r] = ~(KeepBits)0;
            if ((g_y || drop_on_out) && p_drop > 0.f) {
                KeepBits kb = 0;
#pragma unroll
                for (int t = 0; t < PER; ++t) {
                    const int c = lane + WAVE * t;
                    const U4 b = drop_bits(seed, site, (unsigned long long)(row0 + r) * nchunk + c);
                    kb |= (KeepBits)((b.x >= thresh ? 1u : 0u) | (b.y >= thresh ? 2u : 0u) | (b.z >= thresh ? 4u : 0u) | (b.w >= thresh ? 8u : 0u)) << (4 * t);
                }
                keepbits[r] = kb;
            }
        }
        // 3. row statistics of all rows in flight, then their outputs
        float s1[RF], s2[RF];
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            float a1 = 0.f, a2 = 0.f;
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const float m = (lane + WAVE * t < nchunk) ? live[r] : 0.0f;
                f32x4 g = go[r][t] * m;
                if (drop_on_out) {
                    const unsigned kb = (unsigned)(keepbits[r] >> (4 * t));
                    g.x = IO<DT>::round(g.x * ((kb & 1u) ? inv_keep : 0.f)); g.y = IO<DT>::round(g.y * ((kb & 2u) ? inv_keep : 0.f));
                    g.z = IO<DT>::round(g.z * ((kb & 4u) ? inv_keep : 0.f)); g.w = IO<DT>::round(g.w * ((kb & 8u) ? inv_keep : 0.f));
                }
                const f32x4 xh = (x[r][t] - mu[r]) * rs[r];
                const f32x4 tt = g * gm[t];
                dg[t] += g * xh;
                db[t] += g;
                a1 += (tt.x + tt.y) + (tt.z + tt.w);
                a2 += (tt.x * xh.x + tt.y * xh.y) + (tt.z * xh.z + tt.w * xh.w);
                go[r][t] = tt; x[r][t] = xh;
            }
            s1[r] = a1; s2[r] = a2;
        }
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            s1[r] = wave_sum_f32(s1[r]) / (float)H;
            s2[r] = wave_sum_f32(s2[r]) / (float)H;
        }
#pragma unroll
        for (int r = 0; r < RF; ++r) {
            if (row0 + r >= N) break;                   // wave-uniform; only stores below
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int c = lane + WAVE * t;
                if (c < nchunk) {
                    const size_t off = (size_t)(row0 + r) * H + 4 * c;
                    f32x4 gp = (go[r][t] - s1[r] - x[r][t] * s2[r]) * rs[r];
                    if (g_resid) IO<DT>::store4(g_resid, off, gp);
                    if (g_y) {
                        if (!drop_on_out) {
                            const unsigned kb = (unsigned)(keepbits[r] >> (4 * t));
                            gp.x *= (kb & 1u) ? inv_keep : 0.f; gp.y *= (kb & 2u) ? inv_keep : 0.f;
                            gp.z *= (kb & 4u) ? inv_keep : 0.f; gp.w *= (kb & 8u) ? inv_keep : 0.f;
                        }
                        IO<DT>::store4(g_y, off, gp);
                        // what the consumer of g_y reads back is the STORED (possibly bf16-rounded) value
                        dy[t].x += IO<DT>::round(gp.x); dy[t].y += IO<DT>::round(gp.y);
                        dy[t].z += IO<DT>::round(gp.z); dy[t].w += IO<DT>::round(gp.w);
                    }
                }
            }
        }
    }
    float* l_dy = lds + (size_t)w * 3 * H;
    float* l_dg = l_dy + H;
    float* l_db = l_dg + H;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int c = lane + WAVE * t;
        if (c < nchunk) {
            *reinterpret_cast<f32x4*>(l_dy + 4 * c) = dy[t];
            *reinterpret_cast<f32x4*>(l_dg + 4 * c) = dg[t];
            *reinterpret_cast<f32x4*>(l_db + 4 * c) = db[t];
        }
    }
    __syncthreads();
    // partial row layout: [dbias_prev(H) | dgamma(H) | dbeta(H)]
    for (int j = threadIdx.x; j < 3 * H; j += 256) {
        if (j < H && !want_dbias) continue;
        float a = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) a += lds[(size_t)ww * 3 * H + j];
        part_dgamma[(size_t)blockIdx.x * 3 * H + j] = a;
    }
}

__device__ __forceinline__ float half_wave_sum_f32(float v) {
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) v += __shfl_xor(v, m, WAVE);
    return v;
}


// bf16 io, H % 8 == 0, H <= 8 * 32 * PER: HALF a wave per row, 16 bytes (8 elements) per lane and access instead of 8; LNB_ROWS rows
// per workgroup = 2 per half wave, both in flight.  Same partial-row layout and the same dropout stream (one Philox call per 4
// elements, indexed row * H/4 + chunk) as drln_bwd_kernel.  In the step: 16.7 -> 14.3 us per [8192, 768] call.  (The forward
// kernel rebuilt the same way measured 13.3 against 13.4 us and was dropped again: it is not the access width that holds it.)
template <int PER>
__global__ __launch_bounds__(256) void drln_bwd16_kernel(const void* __restrict__ g_out, const void* __restrict__ pre,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, int64_t N, int H, float p_drop,
                                                          unsigned thresh, unsigned long long seed,
                                                          const unsigned long long* __restrict__ seed_off, unsigned site,
                                                          void* __restrict__ g_y, void* __restrict__ g_resid,
                                                          float* __restrict__ part_dgamma, int want_dbias, int drop_on_out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // [4 waves][3][H]
    if (seed_off) seed += *seed_off;
    constexpr int RF = LNB_ROWS / 8;                              // rows per half wave
    const int hl = threadIdx.x & 31, hw = threadIdx.x >> 5, w = threadIdx.x >> 6;
    const int nchunk8 = H >> 3, nchunk4 = H >> 2;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    f32x8 dg[PER], db[PER], dy[PER], gm[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        dg[t].lo = 0.f; dg[t].hi = 0.f; db[t].lo = 0.f; db[t].hi = 0.f; dy[t].lo = 0.f; dy[t].hi = 0.f;
        const int c = hl + 32 * t < nchunk8 ? hl + 32 * t : nchunk8 - 1;
        const f32x4* gp = reinterpret_cast<const f32x4*>(gamma + 8 * c);
        gm[t].lo = gp[0]; gm[t].hi = gp[1];
    }
    const int64_t row0 = (int64_t)blockIdx.x * LNB_ROWS + hw * RF;
    // 1. every load of these rows goes out first (rows past N are clamped and weighted 0: no branch around a load)
    f32x8 go[RF][PER], x[RF][PER];
    float mu[RF], rs[RF], live[RF];
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        const int64_t row = row0 + r < N ? row0 + r : N - 1;
        live[r] = row0 + r < N ? 1.0f : 0.0f;
        mu[r] = mean[row]; rs[r] = rstd[row];
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const int c = hl + 32 * t < nchunk8 ? hl + 32 * t : nchunk8 - 1;
            const size_t off = (size_t)row * H + 8 * c;
            go[r][t] = IO<KVQ_BF16>::load8(g_out, off);
            x[r][t] = IO<KVQ_BF16>::load8(pre, off);
        }
    }
    // 2. the dropout keep bits do not depend on the loads: Philox runs while they are in flight (8 bits per chunk)
    unsigned keepbits[RF];
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        keepbits[r] = ~0u;
        if ((g_y || drop_on_out) && p_drop > 0.f) {
            unsigned kb = 0;
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int c = hl + 32 * t;
                const U4 b0 = drop_bits(seed, site, (unsigned long long)(row0 + r) * nchunk4 + 2 * c);
                const U4 b1 = drop_bits(seed, site, (unsigned long long)(row0 + r) * nchunk4 + 2 * c + 1);
                kb |= ((b0.x >= thresh ? 1u : 0u) | (b0.y >= thresh ? 2u : 0u) | (b0.z >= thresh ? 4u : 0u) | (b0.w >= thresh ? 8u : 0u) |
                       (b1.x >= thresh ? 16u : 0u) | (b1.y >= thresh ? 32u : 0u) | (b1.z >= thresh ? 64u : 0u) | (b1.w >= thresh ? 128u : 0u)) << (8 * t);
            }
            keepbits[r] = kb;
        }
    }
    // 3. row statistics, then the outputs
    float s1[RF], s2[RF];
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        float a1 = 0.f, a2 = 0.f;
#pragma unroll
        for (int t = 0; t < PER; ++t) {
            const float m = (hl + 32 * t < nchunk8) ? live[r] : 0.0f;
            f32x4 g0 = go[r][t].lo * m, g1 = go[r][t].hi * m;
            if (drop_on_out) {                       // see drln_bwd_kernel
                const unsigned kb = keepbits[r] >> (8 * t);
                g0.x = IO<KVQ_BF16>::round(g0.x * ((kb & 1u) ? inv_keep : 0.f)); g0.y = IO<KVQ_BF16>::round(g0.y * ((kb & 2u) ? inv_keep : 0.f));
                g0.z = IO<KVQ_BF16>::round(g0.z * ((kb & 4u) ? inv_keep : 0.f)); g0.w = IO<KVQ_BF16>::round(g0.w * ((kb & 8u) ? inv_keep : 0.f));
                g1.x = IO<KVQ_BF16>::round(g1.x * ((kb & 16u) ? inv_keep : 0.f)); g1.y = IO<KVQ_BF16>::round(g1.y * ((kb & 32u) ? inv_keep : 0.f));
                g1.z = IO<KVQ_BF16>::round(g1.z * ((kb & 64u) ? inv_keep : 0.f)); g1.w = IO<KVQ_BF16>::round(g1.w * ((kb & 128u) ? inv_keep : 0.f));
            }
            const f32x4 xh0 = (x[r][t].lo - mu[r]) * rs[r], xh1 = (x[r][t].hi - mu[r]) * rs[r];
            const f32x4 t0 = g0 * gm[t].lo, t1 = g1 * gm[t].hi;
            dg[t].lo += g0 * xh0; dg[t].hi += g1 * xh1;
            db[t].lo += g0; db[t].hi += g1;
            a1 += ((t0.x + t0.y) + (t0.z + t0.w)) + ((t1.x + t1.y) + (t1.z + t1.w));
            a2 += ((t0.x * xh0.x + t0.y * xh0.y) + (t0.z * xh0.z + t0.w * xh0.w)) + ((t1.x * xh1.x + t1.y * xh1.y) + (t1.z * xh1.z + t1.w * xh1.w));
            go[r][t].lo = t0; go[r][t].hi = t1; x[r][t].lo = xh0; x[r][t].hi = xh1;
        }
        s1[r] = a1; s2[r] = a2;
    }
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        s1[r] = half_wave_sum_f32(s1[r]) / (float)H;
        s2[r] = half_wave_sum_f32(s2[r]) / (float)H;
    }
#pragma unroll
    for (int r = 0; r < RF; ++r) {
        if (row0 + r < N) {                          // uniform over the half wave; only stores below
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int c = hl + 32 * t;
                if (c < nchunk8) {
                    const size_t off = (size_t)(row0 + r) * H + 8 * c;
                    f32x8 gp;
                    gp.lo = (go[r][t].lo - s1[r] - x[r][t].lo * s2[r]) * rs[r];
                    gp.hi = (go[r][t].hi - s1[r] - x[r][t].hi * s2[r]) * rs[r];
                    if (g_resid) IO<KVQ_BF16>::store8(g_resid, off, gp);
                    if (g_y) {
                        if (!drop_on_out) {
                            const unsigned kb = keepbits[r] >> (8 * t);
                            gp.lo.x *= (kb & 1u) ? inv_keep : 0.f; gp.lo.y *= (kb & 2u) ? inv_keep : 0.f;
                            gp.lo.z *= (kb & 4u) ? inv_keep : 0.f; gp.lo.w *= (kb & 8u) ? inv_keep : 0.f;
                            gp.hi.x *= (kb & 16u) ? inv_keep : 0.f; gp.hi.y *= (kb & 32u) ? inv_keep : 0.f;
                            gp.hi.z *= (kb & 64u) ? inv_keep : 0.f; gp.hi.w *= (kb & 128u) ? inv_keep : 0.f;
                        }
                        IO<KVQ_BF16>::store8(g_y, off, gp);
                        // what the consumer of g_y reads back is the STORED (bf16-rounded) value
                        dy[t].lo.x += IO<KVQ_BF16>::round(gp.lo.x); dy[t].lo.y += IO<KVQ_BF16>::round(gp.lo.y);
                        dy[t].lo.z += IO<KVQ_BF16>::round(gp.lo.z); dy[t].lo.w += IO<KVQ_BF16>::round(gp.lo.w);
                        dy[t].hi.x += IO<KVQ_BF16>::round(gp.hi.x); dy[t].hi.y += IO<KVQ_BF16>::round(gp.hi.y);
                        dy[t].hi.z += IO<KVQ_BF16>::round(gp.hi.z); dy[t].hi.w += IO<KVQ_BF16>::round(gp.hi.w);
                    }
                }
            }
        }
    }
    // the two halves of a wave hold the same columns: add them, then the 4 waves through LDS as in drln_bwd_kernel
    float* l_dy = lds + (size_t)w * 3 * H;
    float* l_dg = l_dy + H;
    float* l_db = l_dg + H;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        f32x8 a = dy[t], b = dg[t], c3 = db[t];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a.lo[j] += __shfl_xor(a.lo[j], 32, WAVE); a.hi[j] += __shfl_xor(a.hi[j], 32, WAVE);
            b.lo[j] += __shfl_xor(b.lo[j], 32, WAVE); b.hi[j] += __shfl_xor(b.hi[j], 32, WAVE);
            c3.lo[j] += __shfl_xor(c3.lo[j], 32, WAVE); c3.hi[j] += __shfl_xor(c3.hi[j], 32, WAVE);
        }
        const int c = hl + 32 * t;
        if ((threadIdx.x & 32) == 0 && c < nchunk8) {
            f32x4* py = reinterpret_cast<f32x4*>(l_dy + 8 * c); py[0] = a.lo; py[1] = a.hi;
            f32x4* pg = reinterpret_cast<f32x4*>(l_dg + 8 * c); pg[0] = b.lo; pg[1] = b.hi;
            f32x4* pb = reinterpret_cast<f32x4*>(l_db + 8 * c); pb[0] = c3.lo; pb[1] = c3.hi;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < 3 * H; j += 256) {
        if (j < H && !want_dbias) continue;
        float a = 0.f;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) a += lds[(size_t)ww * 3 * H + j];
        part_dgamma[(size_t)blockIdx.x * 3 * H + j] = a;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// column sums: out[c] = sum_n x[n,c]  (bias gradients; also the second stage of the LN partials).
// Two kernels: row-block partials (coalesced, each thread owns 4 or 8 columns), then a fixed-order final sum.
// ---------------------------------------------------------------------------------------------------------------
constexpr int CS_ROWS = 128;   // rows per partial block

// grid (ceil(C/256), ceil(N/CS_ROWS)), 256 threads: each WAVE streams whole 64x4-column row segments (512 B of bf16,
// 1 KiB of f32 per row: fully coalesced), waves interleave rows, then the 4 waves are summed through LDS.
template <int DT_IN>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const void* __restrict__ x, int64_t N, int64_t C,
                                                              int64_t ld, float* __restrict__ part, int rows_per_block, int64_t ldpart) {
    __shared__ f32x4 red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t c0 = ((int64_t)blockIdx.x * 64 + lane) * 4;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
    const int64_t r1 = r0 + rows_per_block < N ? r0 + rows_per_block : N;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    if (c0 < C) {
        const bool vec = (c0 + 4 <= C) && (ld % 4 == 0) &&
                         ((((uintptr_t)x) + (size_t)c0 * IO<DT_IN>::bytes) % (4 * IO<DT_IN>::bytes) == 0);
        if (vec) {
            f32x4 a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f}, a3 = {0.f, 0.f, 0.f, 0.f};
            int64_t r = r0 + w;
            for (; r + 12 < r1; r += 16) {       // 4 independent loads in flight per lane
                a += IO<DT_IN>::load4(x, (size_t)r * ld + c0);
                a1 += IO<DT_IN>::load4(x, (size_t)(r + 4) * ld + c0);
                a2 += IO<DT_IN>::load4(x, (size_t)(r + 8) * ld + c0);
                a3 += IO<DT_IN>::load4(x, (size_t)(r + 12) * ld + c0);
            }
            for (; r < r1; r += 4) a += IO<DT_IN>::load4(x, (size_t)r * ld + c0);
            a = (a + a1) + (a2 + a3);
        } else {
            for (int64_t r = r0 + w; r < r1; r += 4)
                for (int u = 0; u < 4 && c0 + u < C; ++u) a[u] += IO<DT_IN>::load1(x, (size_t)r * ld + c0 + u);
        }
    }
    red[w][lane] = a;
    __syncthreads();
    if (w == 0 && c0 < C) {
        const f32x4 t = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
        for (int u = 0; u < 4 && c0 + u < C; ++u) part[(size_t)blockIdx.y * ldpart + c0 + u] = t[u];
    }
}


// out[i] = sum_s part[s*n + i]  (f32 accumulate over S split-K slabs of a weight-gradient GEMM), 8 elements per thread
template <int DT>
__global__ __launch_bounds__(256) void sum_slabs_kernel(const void* __restrict__ part, int S, int64_t n4, void* __restrict__ out) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 a = IO<DT>::load4(part, 4 * i);
        for (int s2 = 1; s2 < S; ++s2) a += IO<DT>::load4(part, (size_t)s2 * n4 * 4 + 4 * i);
        IO<DT>::store4(out, 4 * i, a);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Batched reductions: up to KVQ_REDUCE_MAX_ITEMS independent  dst[c] = scale * sum_{p < count} src[p*ld + c] (+ dst[c])
// in ONE launch.  A backward layer produces ~8-14 such sums (split-K slabs of the weight-gradient GEMMs, the per-workgroup
// partials of the LayerNorm and bias gradients); each one alone is a 5-8 us launch-latency-bound kernel.
//   count <= 32 ("slabs", long rows):  1024 threads x 8 columns per workgroup, slabs summed in order
//   count  > 32 ("tree", short rows):  64 columns per workgroup, 16 row phases, fixed summation order
// ---------------------------------------------------------------------------------------------------------------
struct ReduceBatch {
    kvq_reduce_item it[KVQ_REDUCE_MAX_ITEMS];
    int first_block[KVQ_REDUCE_MAX_ITEMS + 1];
    int n;
};
constexpr int RB_SLAB_COLS = 8192;   // columns per workgroup in slab mode
__host__ __device__ __forceinline__ bool reduce_is_slab(const kvq_reduce_item& d) {   // needs 8-element (16-byte) vector access
    return d.count <= 32 && d.cols % 8 == 0 && d.ld % 8 == 0 && ((uintptr_t)d.src & 15) == 0 && ((uintptr_t)d.dst & 15) == 0;
}

template <int DT_SRC, int DT_DST>
__device__ __forceinline__ void reduce_slab_block(const kvq_reduce_item& d, int blk) {
    // 8 columns (16 bytes of bf16) per thread; four slabs in flight; the grouping is fixed, so the sum stays deterministic
    const int64_t c = (int64_t)blk * RB_SLAB_COLS + 8 * threadIdx.x;
    if (c >= d.cols) return;
    f32x8 a = IO<DT_SRC>::load8(d.src, c);
    f32x8 a1 = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, a2 = a1, a3 = a1;
    int64_t p = 1;
    for (; p + 4 <= d.count; p += 4) {
        const f32x8 t0 = IO<DT_SRC>::load8(d.src, (size_t)p * d.ld + c), t1 = IO<DT_SRC>::load8(d.src, (size_t)(p + 1) * d.ld + c);
        const f32x8 t2 = IO<DT_SRC>::load8(d.src, (size_t)(p + 2) * d.ld + c), t3 = IO<DT_SRC>::load8(d.src, (size_t)(p + 3) * d.ld + c);
        a.lo += t0.lo; a.hi += t0.hi; a1.lo += t1.lo; a1.hi += t1.hi;
        a2.lo += t2.lo; a2.hi += t2.hi; a3.lo += t3.lo; a3.hi += t3.hi;
    }
    for (; p < d.count; ++p) {
        const f32x8 t = IO<DT_SRC>::load8(d.src, (size_t)p * d.ld + c);
        a.lo += t.lo; a.hi += t.hi;
    }
    a.lo = ((a.lo + a1.lo) + (a2.lo + a3.lo)) * d.scale;
    a.hi = ((a.hi + a1.hi) + (a2.hi + a3.hi)) * d.scale;
    if (d.accumulate) {
        const f32x8 o = IO<DT_DST>::load8(d.dst, c);
        a.lo += o.lo; a.hi += o.hi;
    }
    IO<DT_DST>::store8(d.dst, c, a);
}
template <int DT_SRC, int DT_DST>
__device__ __forceinline__ void reduce_tree_block(const kvq_reduce_item& d, int blk, float (*red)[64]) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const bool vec = d.cols % 4 == 0 && d.ld % 4 == 0 && ((uintptr_t)d.src % (4 * IO<DT_SRC>::bytes)) == 0;   // block-uniform
    if (vec) {
        // 64 columns x 64 row phases: a lane owns 4 columns (one 16-byte access), a wave covers 4 rows x 64 columns
        const int cg = lane & 15, ph = threadIdx.x >> 4;
        const int64_t c = (int64_t)blk * 64 + 4 * cg;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
        if (c < d.cols) {
            int64_t p = ph;
            for (; p + 64 < d.count; p += 128) {
                a0 += IO<DT_SRC>::load4(d.src, (size_t)p * d.ld + c);
                a1 += IO<DT_SRC>::load4(d.src, (size_t)(p + 64) * d.ld + c);
            }
            for (; p < d.count; p += 64) a0 += IO<DT_SRC>::load4(d.src, (size_t)p * d.ld + c);
        }
        a0 += a1;
#pragma unroll
        for (int e = 0; e < 4; ++e) {                       // the wave's 4 phases of one column group sit 16 lanes apart
            a0[e] += __shfl_xor(a0[e], 16, WAVE);
            a0[e] += __shfl_xor(a0[e], 32, WAVE);
        }
        if (lane < 16) *reinterpret_cast<f32x4*>(&red[w][4 * cg]) = a0;
    } else {
        const int cl = lane, ph = w;
        const int64_t c = (int64_t)blk * 64 + cl;
        float a0 = 0.f, a1 = 0.f;
        if (c < d.cols) {
            int64_t p = ph;
            for (; p + 16 < d.count; p += 32) {
                a0 += IO<DT_SRC>::load1(d.src, (size_t)p * d.ld + c);
                a1 += IO<DT_SRC>::load1(d.src, (size_t)(p + 16) * d.ld + c);
            }
            for (; p < d.count; p += 16) a0 += IO<DT_SRC>::load1(d.src, (size_t)p * d.ld + c);
        }
        red[ph][cl] = a0 + a1;
    }
    __syncthreads();
    const int64_t c = (int64_t)blk * 64 + lane;
    if (w == 0 && c < d.cols) {
        float a = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) a += red[q][lane];
        a *= d.scale;
        if (d.accumulate) a += IO<DT_DST>::load1(d.dst, c);
        IO<DT_DST>::store1(d.dst, c, a);
    }
}

__global__ __launch_bounds__(1024) void reduce_batch_kernel(ReduceBatch rb) {
    __shared__ float red[16][64];
    // which item: first_block[] is increasing, so the index is a COUNT -- every comparison independent of the others, all
    // table entries requested in one batch (a `while` over the table is a chain of up to 32 dependent scalar loads at the start
    // of every workgroup)
    int i = 0;
#pragma unroll
    for (int k = 1; k < KVQ_REDUCE_MAX_ITEMS; ++k) i += (k < rb.n && (int)blockIdx.x >= rb.first_block[k]) ? 1 : 0;
    const kvq_reduce_item& d = rb.it[i];
    const int blk = (int)blockIdx.x - rb.first_block[i];
    const int combo = (d.src_dtype == KVQ_F32 ? 0 : 2) | (d.dst_dtype == KVQ_F32 ? 0 : 1);
    if (reduce_is_slab(d)) {
        switch (combo) {
            case 0: reduce_slab_block<KVQ_F32, KVQ_F32>(d, blk); break;
            case 1: reduce_slab_block<KVQ_F32, KVQ_BF16>(d, blk); break;
            case 2: reduce_slab_block<KVQ_BF16, KVQ_F32>(d, blk); break;
            default: reduce_slab_block<KVQ_BF16, KVQ_BF16>(d, blk); break;
        }
    } else {
        switch (combo) {
            case 0: reduce_tree_block<KVQ_F32, KVQ_F32>(d, blk, red); break;
            case 1: reduce_tree_block<KVQ_F32, KVQ_BF16>(d, blk, red); break;
            case 2: reduce_tree_block<KVQ_BF16, KVQ_F32>(d, blk, red); break;
            default: reduce_tree_block<KVQ_BF16, KVQ_BF16>(d, blk, red); break;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// GELU (erf form), elementwise.  fwd: a = gelu(h);  bwd: g_h = g_a * gelu'(h)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad_f(float x) {
    const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752f));
    const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
    return cdf + x * pdf;
}
// bf16 io: erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, far below the bf16 rounding of the result) -- libm's erff
// makes these kernels VALU-bound (~40 instructions per element) instead of HBM-bound.  exp(-x^2/2) is shared between the
// erf tail and the Gaussian density of the derivative.
__device__ __forceinline__ void gelu_parts_fast(float x, float& cdf, float& e) {
    const float z = fabsf(x) * 0.70710678118654752f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);      // v_rcp_f32 (1 ulp); __frcp_rn expands to a 12-instruction IEEE division
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    e = __expf(-z * z);                                   // = exp(-x^2 / 2)
    const float half_tail = 0.5f * poly * e;              // 0.5 * erfc(|x| / sqrt 2)
    cdf = x >= 0.f ? 1.0f - half_tail : half_tail;
}
template <int DT> __device__ __forceinline__ float gelu_v(float x) {
    if (DT == KVQ_F32) return gelu_f(x);
    float cdf, e;
    gelu_parts_fast(x, cdf, e);
    return x * cdf;
}
template <int DT> __device__ __forceinline__ float gelu_grad_v(float x) {
    if (DT == KVQ_F32) return gelu_grad_f(x);
    float cdf, e;
    gelu_parts_fast(x, cdf, e);
    return cdf + x * 0.39894228040143268f * e;
}

template <int DT, bool BWD>
__device__ __forceinline__ f32x4 gelu_apply4(const f32x4& x, const f32x4& g) {
    f32x4 o;
    if (BWD) {
        o.x = g.x * gelu_grad_v<DT>(x.x); o.y = g.y * gelu_grad_v<DT>(x.y);
        o.z = g.z * gelu_grad_v<DT>(x.z); o.w = g.w * gelu_grad_v<DT>(x.w);
    } else {
        o.x = gelu_v<DT>(x.x); o.y = gelu_v<DT>(x.y); o.z = gelu_v<DT>(x.z); o.w = gelu_v<DT>(x.w);
    }
    return o;
}

// n8 chunks of 8 elements (16 bytes of bf16 per lane and access), grid-stride
template <int DT, bool BWD>
__global__ __launch_bounds__(256) void gelu_kernel(const void* __restrict__ h, const void* __restrict__ g_a,
                                                    void* __restrict__ out, int64_t n8) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const f32x8 x = IO<DT>::load8(h, 8 * i);
        f32x8 g = x;
        if (BWD) g = IO<DT>::load8(g_a, 8 * i);
        f32x8 o;
        o.lo = gelu_apply4<DT, BWD>(x.lo, g.lo);
        o.hi = gelu_apply4<DT, BWD>(x.hi, g.hi);
        IO<DT>::store8(out, 8 * i, o);
    }
}
// tail of fewer than 8 elements (n % 8 != 0 callers): chunks of 4
template <int DT, bool BWD>
__global__ __launch_bounds__(64) void gelu_tail_kernel(const void* __restrict__ h, const void* __restrict__ g_a,
                                                        void* __restrict__ out, int64_t first4, int64_t n4) {
    const int64_t i = first4 + threadIdx.x;
    if (i < n4) {
        const f32x4 x = IO<DT>::load4(h, 4 * i);
        const f32x4 g = BWD ? IO<DT>::load4(g_a, 4 * i) : x;
        IO<DT>::store4(out, 4 * i, gelu_apply4<DT, BWD>(x, g));
    }
}

// GELU backward that also leaves the column sums of g_h (the bias gradient of the dense layer in front of the GELU) as partial
// rows: grid (ceil(C/1024), ceil(N/GB_ROWS)) x 128 threads, each thread owns 8 columns and walks GB_ROWS rows, 4 rows in flight.
constexpr int GB_ROWS = 8;
template <int DT>
__global__ __launch_bounds__(128) void gelu_bwd_bias_kernel(const void* __restrict__ h, const void* __restrict__ g_a,
                                                             void* __restrict__ g_h, int64_t N, int64_t C, float* __restrict__ part) {
    const int64_t c = ((int64_t)blockIdx.x * 128 + threadIdx.x) * 8;       // 8 columns (16 bytes of bf16) per thread
    if (c >= C) return;
    const int64_t r0 = (int64_t)blockIdx.y * GB_ROWS;
    const int64_t r1 = r0 + GB_ROWS < N ? r0 + GB_ROWS : N;
    f32x4 acc_lo = {0.f, 0.f, 0.f, 0.f}, acc_hi = acc_lo;
    int64_t r = r0;
    for (; r + 4 <= r1; r += 4) {
        f32x8 x[4], g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u] = IO<DT>::load8(h, (size_t)(r + u) * C + c); g[u] = IO<DT>::load8(g_a, (size_t)(r + u) * C + c); }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            f32x8 o;
            o.lo = gelu_apply4<DT, true>(x[u].lo, g[u].lo);
            o.hi = gelu_apply4<DT, true>(x[u].hi, g[u].hi);
            IO<DT>::store8(g_h, (size_t)(r + u) * C + c, o);
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc_lo[e] += IO<DT>::round(o.lo[e]); acc_hi[e] += IO<DT>::round(o.hi[e]); }   // sum what the weight-gradient GEMM will read
        }
    }
    for (; r < r1; ++r) {
        const f32x8 x = IO<DT>::load8(h, (size_t)r * C + c), g = IO<DT>::load8(g_a, (size_t)r * C + c);
        f32x8 o;
        o.lo = gelu_apply4<DT, true>(x.lo, g.lo);
        o.hi = gelu_apply4<DT, true>(x.hi, g.hi);
        IO<DT>::store8(g_h, (size_t)r * C + c, o);
#pragma unroll
        for (int e = 0; e < 4; ++e) { acc_lo[e] += IO<DT>::round(o.lo[e]); acc_hi[e] += IO<DT>::round(o.hi[e]); }
    }
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * C + c) = acc_lo;
    *reinterpret_cast<f32x4*>(part + (size_t)blockIdx.y * C + c + 4) = acc_hi;
}

// ---------------------------------------------------------------------------------------------------------------
// Adam (torch.optim.Adam, coupled L2 weight decay, optional amsgrad) on flat buffers:
//   p32 (master, f32), g (bf16 or f32), m, v (f32) [, vmax], shadow (bf16 copy of p32 for the next forward)
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void adam_update4(f32x4& pv, f32x4 gv, f32x4& mv, f32x4& vv, float* vmax4, float lr, float b1, float b2,
                                             float eps, float wd, float bc1, float bc2_sqrt) {
    gv += pv * wd;
    mv = mv * b1 + gv * (1.0f - b1);
    vv = vv * b2 + (gv * gv) * (1.0f - b2);
    f32x4 den_src = vv;
    if (vmax4) {
        f32x4 vm = *reinterpret_cast<f32x4*>(vmax4);
        vm.x = fmaxf(vm.x, vv.x); vm.y = fmaxf(vm.y, vv.y); vm.z = fmaxf(vm.z, vv.z); vm.w = fmaxf(vm.w, vv.w);
        *reinterpret_cast<f32x4*>(vmax4) = vm;
        den_src = vm;
    }
    f32x4 den;
    den.x = sqrtf(den_src.x) / bc2_sqrt + eps; den.y = sqrtf(den_src.y) / bc2_sqrt + eps;
    den.z = sqrtf(den_src.z) / bc2_sqrt + eps; den.w = sqrtf(den_src.w) / bc2_sqrt + eps;
    const float step = lr / bc1;
    pv.x -= step * (mv.x / den.x); pv.y -= step * (mv.y / den.y);
    pv.z -= step * (mv.z / den.z); pv.w -= step * (mv.w / den.w);
}

// Optional (round 5, fp8 forward GEMMs): the fp8 (e4m3) mirror of the GEMM weights written by the update itself, from the bf16 value
// it has just rounded for the shadow -- instead of a conversion pass over the whole shadow buffer after every step.  The mirror is
// indexed like the flat parameter buffer (one byte per element); `gseg[global element >> 11]` names the quantisation segment (= GEMM
// weight, one scale each) that covers a 2048-element span: >= 0 the segment, -1 none, -2 more than one thing (the thread then walks
// the segment table).  Segments start and end at multiples of 16 elements, so a thread's 8 elements never straddle one.
struct AdamFp8 {
    unsigned char* w8;            // null: no mirror
    const int* gseg;
    const float* scale;           // [nseg] current scales (kvq_fp8_quantize_segments*)
    const int64_t* seg_off;       // [nseg] first element of each segment
    const int64_t* seg_n;         // [nseg]
    int nseg;
    int64_t e_base;               // global element index of p[0]
};

// 8 parameters per thread and pass: every array moves in 16-byte accesses (the bf16 gradient and shadow included), two
// independent 4-element updates in flight.  n8 = n / 8 chunks; a tail of 4 (n % 8 == 4) is handled by the last thread.
template <int DT_G>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const void* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, float* __restrict__ vmax,
                                                    unsigned short* __restrict__ shadow, int64_t n4, float lr, float b1,
                                                    float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float grad_scale, const float* __restrict__ hyper, int n_tail, AdamFp8 f8) {
    if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; }     // device-resident step state (kvq_step_state_advance)
    const int64_t n8 = n4 >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (int64_t)gridDim.x * 256) {
        const int64_t e = 8 * i;
        f32x4 p0 = *reinterpret_cast<f32x4*>(p + e), p1 = *reinterpret_cast<f32x4*>(p + e + 4);
        f32x4 m0 = *reinterpret_cast<f32x4*>(m + e), m1 = *reinterpret_cast<f32x4*>(m + e + 4);
        f32x4 v0 = *reinterpret_cast<f32x4*>(v + e), v1 = *reinterpret_cast<f32x4*>(v + e + 4);
        const f32x8 gv = IO<DT_G>::load8(g, e);
        adam_update4(p0, gv.lo * grad_scale, m0, v0, vmax ? vmax + e : nullptr, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        adam_update4(p1, gv.hi * grad_scale, m1, v1, vmax ? vmax + e + 4 : nullptr, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        *reinterpret_cast<f32x4*>(p + e) = p0; *reinterpret_cast<f32x4*>(p + e + 4) = p1;
        *reinterpret_cast<f32x4*>(m + e) = m0; *reinterpret_cast<f32x4*>(m + e + 4) = m1;
        *reinterpret_cast<f32x4*>(v + e) = v0; *reinterpret_cast<f32x4*>(v + e + 4) = v1;
        if (shadow) { f32x8 o = {p0, p1}; IO<KVQ_BF16>::store8(shadow, e, o); }
        if (f8.w8) {                                                        // (uniform)
            const int64_t ge = f8.e_base + e;
            int sg = f8.gseg[ge >> 11];
            if (sg == -2) {
                sg = -1;
                for (int q = 0; q < f8.nseg; ++q)
                    if (ge >= f8.seg_off[q] && ge < f8.seg_off[q] + f8.seg_n[q]) sg = q;
            }
            if (sg >= 0) {
                const uint4 r = {(unsigned)f32_to_bf16(p0.x) | ((unsigned)f32_to_bf16(p0.y) << 16), (unsigned)f32_to_bf16(p0.z) | ((unsigned)f32_to_bf16(p0.w) << 16),
                                 (unsigned)f32_to_bf16(p1.x) | ((unsigned)f32_to_bf16(p1.y) << 16), (unsigned)f32_to_bf16(p1.z) | ((unsigned)f32_to_bf16(p1.w) << 16)};
                *reinterpret_cast<uint2*>(f8.w8 + ge) = quant8(r, f8.scale[sg]);
            }
        }
    }
    if ((n4 & 1) && blockIdx.x == 0 && threadIdx.x == 0) {                  // the odd 4-element chunk
        const int64_t e = 4 * (n4 - 1);
        f32x4 pv = *reinterpret_cast<f32x4*>(p + e), mv = *reinterpret_cast<f32x4*>(m + e), vv = *reinterpret_cast<f32x4*>(v + e);
        adam_update4(pv, IO<DT_G>::load4(g, e) * grad_scale, mv, vv, vmax ? vmax + e : nullptr, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
        *reinterpret_cast<f32x4*>(p + e) = pv; *reinterpret_cast<f32x4*>(m + e) = mv; *reinterpret_cast<f32x4*>(v + e) = vv;
        if (shadow) IO<KVQ_BF16>::store4(shadow, e, pv);
    }
    if (n_tail && blockIdx.x == 0 && threadIdx.x == 1) {                    // 1 .. 3 last elements (9-code Gumbel bias, ...)
        for (int64_t e = 4 * n4; e < 4 * n4 + n_tail; ++e) {
            f32x4 pv = {p[e], 0.f, 0.f, 0.f}, mv = {m[e], 0.f, 0.f, 0.f}, vv = {v[e], 0.f, 0.f, 0.f};
            f32x4 vm4 = {vmax ? vmax[e] : 0.f, 0.f, 0.f, 0.f};                // (16-byte aligned: adam_update4 reads it as one vector)
            const f32x4 gv = {IO<DT_G>::load1(g, e) * grad_scale, 0.f, 0.f, 0.f};
            adam_update4(pv, gv, mv, vv, vmax ? reinterpret_cast<float*>(&vm4) : nullptr, lr, b1, b2, eps, wd, bc1, bc2_sqrt);
            p[e] = pv.x; m[e] = mv.x; v[e] = vv.x;
            if (vmax) vmax[e] = vm4.x;
            if (shadow) IO<KVQ_BF16>::store1(shadow, e, pv.x);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Device-resident step state: what changes from one training step to the next (step count -> dropout seed offset, learning
// rate after the MultiStepLR milestones, Adam bias corrections) lives in 24 bytes of HBM and is advanced by a one-thread kernel,
// so that a whole step captured in a hipGraph replays without any host-side argument patching.
// ---------------------------------------------------------------------------------------------------------------
struct StepState {
    unsigned long long step;   // optimiser steps completed; dropout kernels add it to their seed
    float lr, bc1, bc2s, pad;  // for the step being applied: lr after milestones, 1 - beta1^t, sqrt(1 - beta2^t)
};
struct Milestones {
    long long at[8];
    int n;
};
// phase 1: lr / bias corrections of the step about to be applied; phase 2: step += 1; 3: both (kvq_step_state_advance)
__global__ void step_state_advance_kernel(StepState* st, float lr0, float gamma, Milestones ms, float beta1, float beta2, int phase) {
    const unsigned long long t = st->step + 1;             // the step now being applied (1-based)
    if (phase & 1) {
        int k = 0;
        for (int i = 0; i < ms.n; ++i) k += ((long long)(t - 1) >= ms.at[i]) ? 1 : 0;   // MultiStepLR ticked once per finished step
        double lr = lr0;
        for (int i = 0; i < k; ++i) lr *= (double)gamma;
        st->lr = (float)lr;
        st->bc1 = (float)(1.0 - pow((double)beta1, (double)t));
        st->bc2s = (float)sqrt(1.0 - pow((double)beta2, (double)t));
    }
    if (phase & 2) st->step = t;
}

// out = x * keep / (1-p) with the Philox mask of (seed, site): the dropout behind the embedding LayerNorm (modeling_bert.py:58,
// BertEmbeddings) and, applied to the incoming gradient, its backward
template <int DT>
__global__ __launch_bounds__(256) void dropout_kernel(const void* __restrict__ x, void* __restrict__ out, int64_t n4, float p_drop,
                                                       unsigned thresh, unsigned long long seed,
                                                       const unsigned long long* __restrict__ seed_off, unsigned site) {
    if (seed_off) seed += *seed_off;
    const float inv_keep = p_drop > 0.f ? 1.0f / (1.0f - p_drop) : 1.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
        f32x4 a = IO<DT>::load4(x, 4 * i);
        if (p_drop > 0.f) {
            const U4 b = drop_bits(seed, site, (unsigned long long)i);
            a.x *= keep_scale(b.x, thresh, inv_keep); a.y *= keep_scale(b.y, thresh, inv_keep);
            a.z *= keep_scale(b.z, thresh, inv_keep); a.w *= keep_scale(b.w, thresh, inv_keep);
        }
        IO<DT>::store4(out, 4 * i, a);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Word-embedding gradient (autograd of the gather in BertEmbeddings, modeling_bert.py:53-58): gW[id] (= | +=) sum over the
// tokens n with ids[n] == id of g[n], tokens visited in increasing n.  Input is the token order sorted by id (stable), so
// every id is one run of consecutive positions.  Deterministic: no float atomics (torch's index_add_ is an atomic scatter
// whose hot row -- the pad token, ~60 % of all positions -- serialises: 157 us per call at N = 8192).
//   pass 1  one wave per block of EG_R sorted positions walks its runs; a run that lies inside the block is final and is
//           written straight to gW; a run cut by a block boundary leaves an f32 partial row (head / tail slot of the block)
//   pass 2  one wave per block that holds the FIRST piece of a cut run adds the following blocks' head pieces in order
// ---------------------------------------------------------------------------------------------------------------
constexpr int EG_R = 8;    // sorted positions per wave: one batch of row loads, 1024 waves at N = 8192

template <int DT_G, int DT_W, int PER>
__device__ __forceinline__ void eg_emit(const f32x4 (&acc)[PER], void* gW, long long id, int H, int lane, int accumulate) {
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int c = lane + WAVE * t;
        if (4 * c < H) {
            f32x4 a = acc[t];
            const size_t off = (size_t)id * H + 4 * c;
            if (accumulate) a += IO<DT_W>::load4(gW, off);
            IO<DT_W>::store4(gW, off, a);
        }
    }
}

// edge layout: edge[(2 b + slot) * H ..], slot 0 = head piece (continues a run of the previous block), 1 = tail piece that
// STARTS a cut run; meta[b] bit 0: has tail piece, bit 1: the head piece covers the whole block and the run goes on
template <int DT_G, int DT_W, int PER>
__global__ __launch_bounds__(64) void embed_grad_runs_kernel(const void* __restrict__ g, const int64_t* __restrict__ perm,
                                                              const int64_t* __restrict__ sid, int64_t N, int H, int64_t V,
                                                              void* __restrict__ gW, int accumulate, float* __restrict__ edge,
                                                              int* __restrict__ meta) {
    const int lane = threadIdx.x;
    const int64_t b = blockIdx.x, s0 = b * EG_R;
    const int64_t s1 = s0 + EG_R < N ? s0 + EG_R : N;
    const int n = (int)(s1 - s0);
    if (sid[s1 - 1] < 0 || sid[s0] >= V) {                   // ascending ids: the whole block is ignored ids (e.g. masked pad tokens)
        if (lane == 0) meta[b] = 0;
        return;
    }
    // lane l keeps position s0 + l's id and token (lanes >= n mirror the last one)
    const int64_t my = s0 + (lane < n ? lane : n - 1);
    const long long my_id = sid[my], my_tok = perm[my];
    const bool head_cont = s0 > 0 && sid[s0 - 1] == sid[s0];
    const bool tail_cont = s1 < N && sid[s1] == sid[s1 - 1];
    f32x4 acc[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) acc[t] = 0.f;
    int m = 0;
    int run_start = 0;
    long long cur = __shfl(my_id, 0, WAVE);
    constexpr int EG_FLY = EG_R;                             // every row of the block in flight at once
    for (int i0 = 0; i0 < n; i0 += EG_FLY) {
        f32x4 v[EG_FLY][PER];
        long long ids4[EG_FLY];
#pragma unroll
        for (int u = 0; u < EG_FLY; ++u) {
            const int i = i0 + u < n ? i0 + u : n - 1;
            ids4[u] = __shfl(my_id, i, WAVE);
            const long long tok = __shfl(my_tok, i, WAVE);
#pragma unroll
            for (int t = 0; t < PER; ++t) {
                const int c = lane + WAVE * t;
                v[u][t] = IO<DT_G>::load4(g, (size_t)tok * H + 4 * (4 * c < H ? c : 0));
            }
        }
#pragma unroll
        for (int u = 0; u < EG_FLY; ++u) {
            const int i = i0 + u;
            if (i >= n) break;                               // wave-uniform
            if (ids4[u] != cur) {                            // wave-uniform: the run [run_start, i) of id `cur` is complete
                if (run_start == 0 && head_cont) {
#pragma unroll
                    for (int t = 0; t < PER; ++t) { const int c = lane + WAVE * t; if (4 * c < H) *reinterpret_cast<f32x4*>(edge + (size_t)(2 * b) * H + 4 * c) = acc[t]; }
                } else if (cur >= 0 && cur < V) {
                    eg_emit<DT_G, DT_W, PER>(acc, gW, cur, H, lane, accumulate);
                }
#pragma unroll
                for (int t = 0; t < PER; ++t) acc[t] = 0.f;
                cur = ids4[u];
                run_start = i;
            }
#pragma unroll
            for (int t = 0; t < PER; ++t) acc[t] += v[u][t];
        }
    }
    // last run [run_start, n)
    const bool first_piece_of_cut = tail_cont && !(run_start == 0 && head_cont);
    if (run_start == 0 && head_cont) {
#pragma unroll
        for (int t = 0; t < PER; ++t) { const int c = lane + WAVE * t; if (4 * c < H) *reinterpret_cast<f32x4*>(edge + (size_t)(2 * b) * H + 4 * c) = acc[t]; }
        if (tail_cont) m |= 2;                               // the whole block is the middle of a run
    } else if (first_piece_of_cut) {
#pragma unroll
        for (int t = 0; t < PER; ++t) { const int c = lane + WAVE * t; if (4 * c < H) *reinterpret_cast<f32x4*>(edge + (size_t)(2 * b + 1) * H + 4 * c) = acc[t]; }
        m |= 1;
    } else if (cur >= 0 && cur < V) {
        eg_emit<DT_G, DT_W, PER>(acc, gW, cur, H, lane, accumulate);
    }
    if (lane == 0) meta[b] = m;
}

// pass 2: 16 waves per cut run.  The run's pieces are the tail piece of block b and the head pieces of blocks b+1 .. b+np;
// wave w adds pieces w, w+16, ... in order, the 16 partial rows are then added in wave order: a fixed tree, deterministic.
template <int DT_W, int PER>
__global__ __launch_bounds__(1024) void embed_grad_edges_kernel(const int64_t* __restrict__ sid, int64_t N, int H, int64_t V,
                                                                void* __restrict__ gW, int accumulate, const float* __restrict__ edge,
                                                                const int* __restrict__ meta, int64_t nblocks) {
    __shared__ __attribute__((aligned(16))) float red[16][PER * 256];
    __shared__ int np_s;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t b = blockIdx.x;
    if (!(meta[b] & 1)) return;                              // no cut run starts in this block (workgroup-uniform)
    if (w == 0) {                                            // how many following blocks carry a head piece of this run
        int np = 0;
        for (int64_t k = b + 1;; k += 64) {
            const int64_t bb = k + lane;
            const bool last = bb >= nblocks || !(meta[bb] & 2);      // this head piece ends the run (or there is none)
            const unsigned long long m = __ballot(last);
            if (m) { np += __ffsll((long long)m); break; }
            np += 64;
        }
        if (lane == 0) np_s = np;
    }
    __syncthreads();
    const int np = np_s;
    f32x4 acc[PER];
#pragma unroll
    for (int t = 0; t < PER; ++t) acc[t] = 0.f;
    if (w == 0) {
#pragma unroll
        for (int t = 0; t < PER; ++t) { const int c = lane + WAVE * t; if (4 * c < H) acc[t] = *reinterpret_cast<const f32x4*>(edge + (size_t)(2 * b + 1) * H + 4 * c); }
    }
    for (int j0 = w; j0 < np; j0 += 128) {                   // pieces j0, j0+16, ..., j0+112: eight loads in flight
        f32x4 v[8][PER];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int j = j0 + 16 * u < np ? j0 + 16 * u : j0;
#pragma unroll
            for (int t = 0; t < PER; ++t) { const int c = lane + WAVE * t; v[u][t] = *reinterpret_cast<const f32x4*>(edge + (size_t)(2 * (b + 1 + j)) * H + 4 * (4 * c < H ? c : 0)); }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (j0 + 16 * u < np) {
#pragma unroll
                for (int t = 0; t < PER; ++t) acc[t] += v[u][t];
            }
        }
    }
#pragma unroll
    for (int t = 0; t < PER; ++t) *reinterpret_cast<f32x4*>(&red[w][4 * (lane + WAVE * t)]) = acc[t];
    __syncthreads();
    if (w != 0) return;
    const int64_t s1 = (b + 1) * EG_R < N ? (b + 1) * EG_R : N;
    const long long id = sid[s1 - 1];
    if (id < 0 || id >= V) return;
#pragma unroll
    for (int t = 0; t < PER; ++t) {
        const int c = lane + WAVE * t;
        if (4 * c < H) {
            f32x4 a = *reinterpret_cast<const f32x4*>(&red[0][4 * c]);
#pragma unroll
            for (int q = 1; q < 16; ++q) a += *reinterpret_cast<const f32x4*>(&red[q][4 * c]);
            const size_t off = (size_t)id * H + 4 * c;
            if (accumulate) a += IO<DT_W>::load4(gW, off);
            IO<DT_W>::store4(gW, off, a);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Gumbel-softmax quantiser rows (models/shelgon3/GumbelQuantizer.py:56-76; torch.nn.functional.gumbel_softmax):
//   t = (logits + g) / tau,  g = -log(Exp(1) sample);  y_soft = softmax(t);  ind = argmax(y_soft) (first maximum);
//   y = y_soft, or with `hard`  fl(fl(one_hot(ind) - y_soft) + y_soft)  (the straight-through value torch returns);
//   kl_row = sum_k q_k log(q_k K + 1e-10),  q = softmax(logits).
// One wave per token row, K <= 64 * PER.  g comes from `noise` [N,K] f32 when given (parity tests), else from Philox
// (seed, site, element index / 4; two uniforms -> g = -log(-log u)).  All arithmetic f32.
// ---------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m, WAVE));
    return v;
}
__device__ __forceinline__ float gumbel_from_bits(unsigned b) {
    const float u = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
    return -__logf(-__logf(u));
}

template <int DT, int PER>
__global__ __launch_bounds__(256) void gumbel_fwd_kernel(const void* __restrict__ logits, const float* __restrict__ noise,
                                                          int64_t N, int K, float inv_tau, int hard, unsigned long long seed,
                                                          const unsigned long long* __restrict__ seed_off, unsigned site,
                                                          void* __restrict__ y_out, float* __restrict__ y_soft,
                                                          int64_t* __restrict__ ind, float* __restrict__ kl_row) {
    if (seed_off) seed += *seed_off;
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    float l[PER], t[PER];
    float tmax = -INFINITY, lmax = -INFINITY;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int k = lane + WAVE * u;
        l[u] = -INFINITY; t[u] = -INFINITY;
        if (k < K) {
            const size_t e = (size_t)row * K + k;
            l[u] = IO<DT>::load1(logits, e);
            float g;
            if (noise) g = noise[e];
            else {
                const U4 b = drop_bits(seed, site, e >> 2);
                const unsigned w4[4] = {b.x, b.y, b.z, b.w};
                g = gumbel_from_bits(w4[e & 3]);
            }
            t[u] = (l[u] + g) * inv_tau;
            tmax = fmaxf(tmax, t[u]); lmax = fmaxf(lmax, l[u]);
        }
    }
    tmax = wave_max_f32(tmax); lmax = wave_max_f32(lmax);
    float ts = 0.f, ls = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const bool in = lane + WAVE * u < K;
        t[u] = in ? __expf(t[u] - tmax) : 0.f;
        l[u] = in ? __expf(l[u] - lmax) : 0.f;
        ts += t[u]; ls += l[u];
    }
    ts = wave_sum_f32(ts); ls = wave_sum_f32(ls);
    const float tinv = 1.0f / ts, linv = 1.0f / ls;
    // arg-max of y_soft: largest value, lowest index on ties
    float best = -1.f; int bi = 0x7fffffff;
    float kl = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int k = lane + WAVE * u;
        if (k < K) {
            t[u] *= tinv;
            if (t[u] > best) { best = t[u]; bi = k; }
            const float q = l[u] * linv;
            kl += q * __logf(q * (float)K + 1e-10f);
        }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float ob = __shfl_xor(best, m, WAVE);
        const int oi = __shfl_xor(bi, m, WAVE);
        if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
    }
    kl = wave_sum_f32(kl);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int k = lane + WAVE * u;
        if (k < K) {
            const size_t e = (size_t)row * K + k;
            if (y_soft) y_soft[e] = t[u];
            float y = t[u];
            if (hard) { const float h = k == bi ? 1.0f : 0.0f; y = (h - t[u]) + t[u]; }
            IO<DT>::store1(y_out, e, y);
        }
    }
    if (lane == 0) { ind[row] = bi; kl_row[row] = kl; }
}

// g_logits = (1/tau) y (g_y - <y, g_y>)  +  c q (L - <q, L>),   L_k = log(q_k K + eps) + q_k K / (q_k K + eps),
// y = y_soft (the straight-through estimator passes the gradient of the soft sample), c = g_diff * kld_scale / N
template <int DT, int PER>
__global__ __launch_bounds__(256) void gumbel_bwd_kernel(const void* __restrict__ logits, const float* __restrict__ y_soft,
                                                          const void* __restrict__ g_y, const float* __restrict__ g_diff,
                                                          int64_t N, int K, float inv_tau, float kld_scale,
                                                          void* __restrict__ g_logits) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float c = (g_diff ? g_diff[0] : 1.0f) * kld_scale / (float)N;
    float y[PER], gy[PER], l[PER];
    float lmax = -INFINITY, dot = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int k = lane + WAVE * u;
        y[u] = 0.f; gy[u] = 0.f; l[u] = -INFINITY;
        if (k < K) {
            const size_t e = (size_t)row * K + k;
            y[u] = y_soft[e];
            gy[u] = g_y ? IO<DT>::load1(g_y, e) : 0.f;
            l[u] = IO<DT>::load1(logits, e);
            lmax = fmaxf(lmax, l[u]);
            dot += y[u] * gy[u];
        }
    }
    lmax = wave_max_f32(lmax);
    dot = wave_sum_f32(dot);
    float ls = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) { l[u] = lane + WAVE * u < K ? __expf(l[u] - lmax) : 0.f; ls += l[u]; }
    ls = wave_sum_f32(ls);
    const float linv = 1.0f / ls;
    float L[PER], qL = 0.f;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        l[u] *= linv;                                                    // q
        const float qk = l[u] * (float)K;
        L[u] = __logf(qk + 1e-10f) + qk / (qk + 1e-10f);
        qL += l[u] * L[u];
    }
    qL = wave_sum_f32(qL);
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int k = lane + WAVE * u;
        if (k < K) IO<DT>::store1(g_logits, (size_t)row * K + k, inv_tau * y[u] * (gy[u] - dot) + c * l[u] * (L[u] - qL));
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Attention for short sentences: S_q, S_k <= 32, head dim 64.  One wave per (sentence, head).
//   lane = (query i = lane & 31, half h = lane >> 5).  QK^T: the lane scores its query against keys 16h..16h+15
//   (K rows are LDS broadcasts); softmax row = 16 in-lane values + one exchange with lane^32; PV: the lane produces
//   output dims 32h..32h+31 of its query over all keys.  Backward recomputes P, then switches to lane = key for the
//   two reductions over queries (dK, dV) through 8 KiB of LDS.  All arithmetic f32, io bf16 or f32.
//   q: [B*Sq, ldq] rows, head hd at columns hd*64; k, v likewise with ldk; mask [B, Sk] (1 = keep) or NULL.
// ---------------------------------------------------------------------------------------------------------------
constexpr int AT_S = 32;
constexpr int AT_D = 64;

struct AttnParams {
    const void *q, *k, *v;
    void* out;            // fwd: ctx [B*Sq, ldo]
    float* lse;           // [B, nh, Sq] log-sum-exp of the scaled, masked scores
    const int64_t* mask;  // [B, Sk] or null
    // backward
    const void* g_out;
    void *g_q, *g_k, *g_v;
    float *pb_q, *pb_k, *pb_v;   // optional per-batch column sums of g_q / g_k / g_v: [B][ldp_q] / [B][ldp_kv] f32 (bias-gradient partials)
    int ldp_q, ldp_kv;
    int B, nh, Sq, Sk;
    int ldq, ldk, ldv, ldo;   // row strides in elements
    int causal;
    float scale, p_drop;
    unsigned thresh;
    unsigned long long seed;
    const unsigned long long* seed_off;   // optional device-resident addend of the seed (see kvq_set_seed_offset)
    unsigned site;
    unsigned char* out8;                  // forward, optional (round 5): fp8 (e4m3) copy of `out`, row stride ld8 bytes, and the
    float* st8;                           // delayed-scaling state of the fp8 GEMM that reads it (kvq_fp8_quantize_delayed's contract)
    int ld8;
};

// LDS tiles hold the io dtype (bf16 tiles halve the footprint -> twice the resident waves); arithmetic is f32.
template <int DT> struct Lds;
template <> struct Lds<KVQ_F32> { typedef float T; };
template <> struct Lds<KVQ_BF16> { typedef unsigned short T; };

constexpr int AT_QLD = AT_D + 8;   // padded row stride (elements) of tiles that lanes read row-per-lane (conflict-free)
constexpr int AT_PLD = AT_S + 1;   // padded row stride of the f32 32x32 probability tiles

// 8 consecutive tile elements -> f32
__device__ __forceinline__ void ld8(const float* p, float (&v)[8]) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
    v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void ld8(const unsigned short* p, float (&v)[8]) {
    const uint4 r = *reinterpret_cast<const uint4*>(p);
    v[0] = __uint_as_float(r.x << 16); v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16); v[3] = __uint_as_float(r.y & 0xffff0000u);
    v[4] = __uint_as_float(r.z << 16); v[5] = __uint_as_float(r.z & 0xffff0000u);
    v[6] = __uint_as_float(r.w << 16); v[7] = __uint_as_float(r.w & 0xffff0000u);
}

// lane (row i, half h) copies its 32 elements of one row of a [rows, ld] matrix (head slice) into an LDS tile row.
// `row_off` must address a VALID row (callers clamp the row index); rows that do not exist are zeroed by a select, never by
// a branch around the loads: hipcc waits vmcnt(0) behind every conditional load, which would serialise the tile fill.
template <int DT>
__device__ __forceinline__ void stage_row(const void* base, size_t row_off, bool valid, typename Lds<DT>::T* tile_row) {
    if (DT == KVQ_F32) {
        f32x4 t[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) t[c] = IO<KVQ_F32>::load4(base, row_off + 4 * c);
        const float m = valid ? 1.0f : 0.0f;
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(tile_row) + 4 * c) = t[c] * m;
    } else {
        uint4 t[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) t[c] = *reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + row_off + 8 * c);
        const unsigned m = valid ? 0xffffffffu : 0u;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            uint4 v = {t[c].x & m, t[c].y & m, t[c].z & m, t[c].w & m};
            *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(tile_row) + 8 * c) = v;
        }
    }
}

// Which key a lane's score slot jj (0..15) belongs to.  LAY 0: lane (query i, half h) owns keys 16h .. 16h+15 (the LDS-tile
// kernels).  LAY 1: the C/D layout of v_mfma_f32_32x32x16_bf16 with the query on the lane: slot jj <-> key 8(jj>>2) + 4h + (jj&3).
template <int LAY>
__device__ __forceinline__ int attn_key(int h, int jj) { return LAY ? 8 * (jj >> 2) + 4 * h + (jj & 3) : 16 * h + jj; }

// dropout keep-scales of the lane's 16 prob elements (b, head, i, j): element index ((bh*32 + i)*32 + j), 4 per Philox call
// (both layouts own keys in aligned groups of four, so every flavour of the kernels draws the same mask)
// key-padding mask of sentence b as one wave-wide bit mask (lane l < 32 loads mask[b][min(l, Sk-1)] unconditionally, ballot
// collects) and the effective dropout seed: two dependent global loads the MFMA kernels issue at their very top, long before
// the scores exist
__device__ __forceinline__ unsigned long long attn_key_mask(const AttnParams& p, int b) {
    unsigned long long kmask = ~0ull;
    if (p.mask) {                                                   // wave-uniform branch
        const int l = threadIdx.x & 31;
        const int64_t mv = p.mask[(size_t)b * p.Sk + (l < p.Sk ? l : p.Sk - 1)];
        kmask = __ballot(mv != 0);
    }
    return kmask;
}
__device__ __forceinline__ unsigned long long attn_seed(const AttnParams& p) { return p.seed + (p.seed_off ? *p.seed_off : 0ull); }
// The MFMA kernels split the mask in two so that no load sits behind a branch (hipcc waits vmcnt(0) at the join of one, i.e. for
// every row load issued before it): the element load goes out FIRST and unconditionally -- without a mask it reads the head of q,
// always there -- and the ballot happens after the dropout bits have been drawn, with the row loads still in flight.
__device__ __forceinline__ int64_t attn_mask_element(const AttnParams& p, int b) {
    const int l = threadIdx.x & 31;
    const int64_t* mp = p.mask ? p.mask + ((size_t)b * p.Sk + (l < p.Sk ? l : p.Sk - 1)) : reinterpret_cast<const int64_t*>(p.q);
    return *mp;
}
__device__ __forceinline__ unsigned long long attn_mask_ballot(const AttnParams& p, int64_t mv) {
    const unsigned long long kb = __ballot(mv != 0);
    return p.mask ? kb : ~0ull;
}

template <int LAY = 0>
__device__ __forceinline__ void attn_keep16(const AttnParams& p, int bh, int i, int h, float (&keep)[16], const unsigned long long* seed_pre = nullptr) {
    const float inv_keep = 1.0f / (1.0f - p.p_drop);
    const unsigned long long seed = seed_pre ? *seed_pre : attn_seed(p);
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const unsigned long long e4 = (((unsigned long long)bh * AT_S + i) * AT_S + attn_key<LAY>(h, 4 * c)) >> 2;
        const U4 r = drop_bits(seed, p.site, e4);
        keep[4 * c] = keep_scale(r.x, p.thresh, inv_keep); keep[4 * c + 1] = keep_scale(r.y, p.thresh, inv_keep);
        keep[4 * c + 2] = keep_scale(r.z, p.thresh, inv_keep); keep[4 * c + 3] = keep_scale(r.w, p.thresh, inv_keep);
    }
}

// acc[jj] += A[i][:] . B[16h + jj][:]   (A row-per-lane tile with stride AT_QLD, B broadcast tile with stride ldb)
template <typename T>
__device__ __forceinline__ void rows_dot16(const T* Arow, const T* B, int ldb, int h, float (&acc)[16]) {
#pragma unroll 2
    for (int d = 0; d < AT_D; d += 8) {
        float a[8];
        ld8(Arow + d, a);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            float bb[8];
            ld8(B + (16 * h + jj) * ldb + d, bb);
#pragma unroll
            for (int u = 0; u < 8; ++u) acc[jj] = __builtin_fmaf(a[u], bb[u], acc[jj]);
        }
    }
}

// o[0..31] += sum_j coef[j] * B[j][32h .. 32h+31]   with coef[j] = C[j * cs] (per-lane f32 LDS scalar)
template <typename T>
__device__ __forceinline__ void weighted_rows32(const float* C, int cs, const T* B, int ldb, int h, float (&o)[32]) {
#pragma unroll 2
    for (int j = 0; j < AT_S; ++j) {
        const float cj = C[j * cs];
        const T* br = B + j * ldb + 32 * h;
#pragma unroll
        for (int d = 0; d < 32; d += 8) {
            float bb[8];
            ld8(br + d, bb);
#pragma unroll
            for (int u = 0; u < 8; ++u) o[d + u] = __builtin_fmaf(cj, bb[u], o[d + u]);
        }
    }
}

// scaled + masked scores -> probabilities (before dropout) of query i against the lane's 16 keys, and the row's lse
template <int LAY = 0>
__device__ __forceinline__ void scores_to_probs(const AttnParams& p, int b, int i, int h, bool qvalid, float (&s)[16], float& lse_out,
                                                const unsigned long long* kmask_pre = nullptr) {
    const unsigned long long kmask = kmask_pre ? *kmask_pre : attn_key_mask(p, b);
    float mx = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = attn_key<LAY>(h, jj);
        bool ok = j < p.Sk && qvalid && ((kmask >> j) & 1ull);
        if (p.causal) ok = ok && j <= i;
        s[jj] = ok ? s[jj] * p.scale : -INFINITY;
        mx = fmaxf(mx, s[jj]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, WAVE));
    const float mref = mx == -INFINITY ? 0.f : mx;   // fully masked row -> all probabilities 0
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        s[jj] = __expf(s[jj] - mref);
        sum += s[jj];
    }
    sum += __shfl_xor(sum, 32, WAVE);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) s[jj] *= inv;
    lse_out = mref + __logf(fmaxf(sum, 1e-37f));
}

template <int DT>
__device__ __forceinline__ void store32(void* base, size_t off, const float (&o)[32]) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        f32x4 t = {o[4 * c], o[4 * c + 1], o[4 * c + 2], o[4 * c + 3]};
        IO<DT>::store4(base, off + 4 * c, t);
    }
}

template <int DT>
__global__ __launch_bounds__(64) void attn_fwd_kernel(AttnParams p) {
    typedef typename Lds<DT>::T T;
    __shared__ __attribute__((aligned(16))) T Ks[AT_S * AT_D];
    __shared__ __attribute__((aligned(16))) T Vs[AT_S * AT_D];
    __shared__ __attribute__((aligned(16))) T Qs[AT_S * AT_QLD];
    __shared__ float Ps[AT_S * AT_PLD];
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
    const bool kvalid = i < p.Sk, qvalid = i < p.Sq;
    const int ik = kvalid ? i : p.Sk - 1, iq = qvalid ? i : p.Sq - 1;      // clamped rows for the loads
    stage_row<DT>(p.k, ((size_t)b * p.Sk + ik) * p.ldk + hd * AT_D + 32 * h, kvalid, Ks + i * AT_D + 32 * h);
    stage_row<DT>(p.v, ((size_t)b * p.Sk + ik) * p.ldv + hd * AT_D + 32 * h, kvalid, Vs + i * AT_D + 32 * h);
    stage_row<DT>(p.q, ((size_t)b * p.Sq + iq) * p.ldq + hd * AT_D + 32 * h, qvalid, Qs + i * AT_QLD + 32 * h);
    __syncthreads();
    float s[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) s[jj] = 0.f;
    rows_dot16<T>(Qs + i * AT_QLD, Ks, AT_D, h, s);
    float lse;
    scores_to_probs(p, b, i, h, qvalid, s, lse);
    if (p.p_drop > 0.f) {
        float keep[16];
        attn_keep16(p, bh, i, h, keep);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) s[jj] *= keep[jj];
    }
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) Ps[i * AT_PLD + 16 * h + jj] = s[jj];
    __syncthreads();
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    weighted_rows32<T>(Ps + i * AT_PLD, 1, Vs, AT_D, h, o);      // out[i][32h + d] = sum_j P[i][j] V[j][32h + d]
    if (qvalid) {
        store32<DT>(p.out, ((size_t)b * p.Sq + i) * p.ldo + hd * AT_D + 32 * h, o);
        if (h == 0 && p.lse) p.lse[((size_t)b * p.nh + hd) * p.Sq + i] = lse;
    }
}

template <int DT>
__global__ __launch_bounds__(64) void attn_bwd_kernel(AttnParams p) {
    typedef typename Lds<DT>::T T;
    __shared__ __attribute__((aligned(16))) T Ks[AT_S * AT_D];
    __shared__ __attribute__((aligned(16))) T Vs[AT_S * AT_D];
    __shared__ __attribute__((aligned(16))) T Qs[AT_S * AT_QLD];
    __shared__ __attribute__((aligned(16))) T Gs[AT_S * AT_QLD];
    __shared__ float Ps[AT_S * AT_PLD];     // dropped probabilities  P~[i][j]
    __shared__ float Ds[AT_S * AT_PLD];     // dS[i][j] (already times scale)
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
    const bool kvalid = i < p.Sk, qvalid = i < p.Sq;
    const int ik = kvalid ? i : p.Sk - 1, iq = qvalid ? i : p.Sq - 1;      // clamped rows for the loads
    stage_row<DT>(p.k, ((size_t)b * p.Sk + ik) * p.ldk + hd * AT_D + 32 * h, kvalid, Ks + i * AT_D + 32 * h);
    stage_row<DT>(p.v, ((size_t)b * p.Sk + ik) * p.ldv + hd * AT_D + 32 * h, kvalid, Vs + i * AT_D + 32 * h);
    stage_row<DT>(p.q, ((size_t)b * p.Sq + iq) * p.ldq + hd * AT_D + 32 * h, qvalid, Qs + i * AT_QLD + 32 * h);
    stage_row<DT>(p.g_out, ((size_t)b * p.Sq + iq) * p.ldo + hd * AT_D + 32 * h, qvalid, Gs + i * AT_QLD + 32 * h);
    __syncthreads();
    float s[16], dp[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { s[jj] = 0.f; dp[jj] = 0.f; }
    rows_dot16<T>(Qs + i * AT_QLD, Ks, AT_D, h, s);
    float lse;
    scores_to_probs(p, b, i, h, qvalid, s, lse);
    rows_dot16<T>(Gs + i * AT_QLD, Vs, AT_D, h, dp);            // dP~[i][j] = dO[i] . V[j]
    float keep[16];
    if (p.p_drop > 0.f) attn_keep16(p, bh, i, h, keep);
    else {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) keep[jj] = 1.0f;
    }
    // dP = dP~ * keep;  delta_i = sum_j P[i][j] dP[i][j];  dS = P * (dP - delta) * scale
    float delta = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        dp[jj] *= keep[jj];
        delta += s[jj] * dp[jj];
    }
    delta += __shfl_xor(delta, 32, WAVE);
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        Ps[i * AT_PLD + 16 * h + jj] = s[jj] * keep[jj];
        Ds[i * AT_PLD + 16 * h + jj] = s[jj] * (dp[jj] - delta) * p.scale;
    }
    __syncthreads();
    {
        float o[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
        weighted_rows32<T>(Ds + i * AT_PLD, 1, Ks, AT_D, h, o);   // dQ[i][32h+d] = sum_j dS[i][j] K[j][32h+d]
        if (qvalid) store32<DT>(p.g_q, ((size_t)b * p.Sq + i) * p.ldq + hd * AT_D + 32 * h, o);
    }
    {   // lane = key j (= i): dK[j] = sum_q dS[q][j] Q[q];  dV[j] = sum_q P~[q][j] dO[q]
        float o[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
        weighted_rows32<T>(Ds + i, AT_PLD, Qs, AT_QLD, h, o);
        if (kvalid) store32<DT>(p.g_k, ((size_t)b * p.Sk + i) * p.ldk + hd * AT_D + 32 * h, o);
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
        weighted_rows32<T>(Ps + i, AT_PLD, Gs, AT_QLD, h, o);
        if (kvalid) store32<DT>(p.g_v, ((size_t)b * p.Sk + i) * p.ldv + hd * AT_D + 32 * h, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 fast path of the attention kernels: same decomposition, but every inner product runs on v_dot2c_f32_bf16
// (two bf16 MACs per instruction, no conversion), which needs operands PAIRED along the contraction index:
//   * row-major tiles [row][64] pair along d            -> Q.K^T and dO.V^T      (dot16)
//   * "pair-interleaved" tiles Xt[row/2][64] hold (X[2r][d], X[2r+1][d]) in one dword -> P.V, dS.K, dS^T.Q, P^T.dO
//     (the contraction runs over rows); built at staging time with one neighbour-lane exchange.
// Probabilities / dS enter the row-contractions rounded to bf16 (as in every bf16 flash-attention kernel).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), c, false);
}
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    bf16x2_t v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}

constexpr int AB_LD = 36;   // dwords per row of the per-lane-row tiles (32 + 4 pad: conflict-free b128 row reads)

// lane (row i, half h) loads its 32 bf16 (16 dwords) of row `row_off`; rows that do not exist come back as zeros
__device__ __forceinline__ void load_half_row(const void* base, size_t row_off, bool valid, unsigned (&d)[16]) {
    const uint4* p = reinterpret_cast<const uint4*>(reinterpret_cast<const unsigned short*>(base) + row_off);
    const unsigned m = valid ? 0xffffffffu : 0u;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const uint4 t = p[c];
        d[4 * c] = t.x & m; d[4 * c + 1] = t.y & m; d[4 * c + 2] = t.z & m; d[4 * c + 3] = t.w & m;
    }
}
__device__ __forceinline__ void store_row_major(unsigned* tile, int ld, int i, int h, const unsigned (&d)[16]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint4 t = {d[4 * c], d[4 * c + 1], d[4 * c + 2], d[4 * c + 3]};
        *reinterpret_cast<uint4*>(tile + i * ld + 16 * h + 4 * c) = t;
    }
}
// Xt[i/2][d] = (X[i & ~1][d] | X[i | 1][d] << 16): even lanes emit d = 32h .. 32h+15, odd lanes d = 32h+16 .. 32h+31
__device__ __forceinline__ void store_pair_interleaved(unsigned* tile_t, int i, int h, const unsigned (&d)[16]) {
    const int odd = i & 1;
    unsigned out[16];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const unsigned mine = odd ? d[8 + c] : d[c];            // the 16 elements of my d-range, my row
        const unsigned give = odd ? d[c] : d[8 + c];            // the partner's d-range, my row
        const unsigned got = __shfl_xor(give, 1, WAVE);         // my d-range, partner's row
        const unsigned ev = odd ? got : mine, od = odd ? mine : got;   // even row / odd row words (2 elements each)
        out[2 * c] = (ev & 0xffffu) | (od << 16);
        out[2 * c + 1] = (ev >> 16) | (od & 0xffff0000u);
    }
    unsigned* dst = tile_t + (i >> 1) * 64 + 32 * h + 16 * odd;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint4 t = {out[4 * c], out[4 * c + 1], out[4 * c + 2], out[4 * c + 3]};
        *reinterpret_cast<uint4*>(dst + 4 * c) = t;
    }
}

// acc[jj] += A[i][0..63] . B[16h + jj][0..63]: A per-lane rows (stride AB_LD dwords), B broadcast rows (32 dwords)
__device__ __forceinline__ void dot16_bf16(const unsigned* Arow, const unsigned* B, int h, float (&acc)[16]) {
#pragma unroll 2
    for (int c = 0; c < 8; ++c) {
        const uint4 a = *reinterpret_cast<const uint4*>(Arow + 4 * c);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const uint4 b = *reinterpret_cast<const uint4*>(B + (16 * h + jj) * 32 + 4 * c);
            acc[jj] = dot2(a.x, b.x, acc[jj]); acc[jj] = dot2(a.y, b.y, acc[jj]);
            acc[jj] = dot2(a.z, b.z, acc[jj]); acc[jj] = dot2(a.w, b.w, acc[jj]);
        }
    }
}
// o[d] += sum_r coef(r) * X[r][32h + d], contraction over the 32 rows r, coefficients as 16 packed pairs cp[rp]
__device__ __forceinline__ void rows32_bf16(const unsigned (&cp)[16], const unsigned* Xt, int h, float (&o)[32]) {
#pragma unroll 2
    for (int rp = 0; rp < 16; ++rp) {
        const unsigned c = cp[rp];
        const unsigned* xr = Xt + rp * 64 + 32 * h;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const uint4 x = *reinterpret_cast<const uint4*>(xr + 4 * q);
            o[4 * q] = dot2(c, x.x, o[4 * q]); o[4 * q + 1] = dot2(c, x.y, o[4 * q + 1]);
            o[4 * q + 2] = dot2(c, x.z, o[4 * q + 2]); o[4 * q + 3] = dot2(c, x.w, o[4 * q + 3]);
        }
    }
}
// this lane's 16 values v[jj] (columns 16h + jj) -> the row's 16 packed pairs over all 32 columns (partner half via shuffle)
__device__ __forceinline__ void row_pairs(const float (&v)[16], int h, unsigned (&cp)[16]) {
    unsigned mine[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) mine[q] = pack_bf16(v[2 * q], v[2 * q + 1]);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const unsigned oth = __shfl_xor(mine[q], 32, WAVE);
        cp[q] = h ? oth : mine[q];
        cp[8 + q] = h ? mine[q] : oth;
    }
}
__device__ __forceinline__ void store32_bf16(void* base, size_t off, const float (&o)[32]) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        uint4 t = {pack_bf16(o[8 * c], o[8 * c + 1]), pack_bf16(o[8 * c + 2], o[8 * c + 3]),
                   pack_bf16(o[8 * c + 4], o[8 * c + 5]), pack_bf16(o[8 * c + 6], o[8 * c + 7])};
        *reinterpret_cast<uint4*>(reinterpret_cast<unsigned short*>(base) + off + 8 * c) = t;
    }
}

__global__ __launch_bounds__(64) void attn_fwd_bf16_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned Ks[AT_S * 32];      // row-major, pairs along d
    __shared__ __attribute__((aligned(16))) unsigned Vt[16 * 64];        // pair-interleaved over keys
    __shared__ __attribute__((aligned(16))) unsigned Qs[AT_S * AB_LD];   // per-lane rows
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
    const bool kvalid = i < p.Sk, qvalid = i < p.Sq;
    const int ik = kvalid ? i : p.Sk - 1, iq = qvalid ? i : p.Sq - 1;
    unsigned kd[16], vd[16], qd[16];
    load_half_row(p.k, ((size_t)b * p.Sk + ik) * p.ldk + hd * AT_D + 32 * h, kvalid, kd);
    load_half_row(p.v, ((size_t)b * p.Sk + ik) * p.ldv + hd * AT_D + 32 * h, kvalid, vd);
    load_half_row(p.q, ((size_t)b * p.Sq + iq) * p.ldq + hd * AT_D + 32 * h, qvalid, qd);
    store_row_major(Ks, 32, i, h, kd);
    store_pair_interleaved(Vt, i, h, vd);
    store_row_major(Qs, AB_LD, i, h, qd);
    __syncthreads();
    float s[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) s[jj] = 0.f;
    dot16_bf16(Qs + i * AB_LD, Ks, h, s);
    float lse;
    scores_to_probs(p, b, i, h, qvalid, s, lse);
    if (p.p_drop > 0.f) {
        float keep[16];
        attn_keep16(p, bh, i, h, keep);
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) s[jj] *= keep[jj];
    }
    unsigned cp[16];
    row_pairs(s, h, cp);
    float o[32];
#pragma unroll
    for (int d = 0; d < 32; ++d) o[d] = 0.f;
    rows32_bf16(cp, Vt, h, o);
    if (qvalid) {
        store32_bf16(p.out, ((size_t)b * p.Sq + i) * p.ldo + hd * AT_D + 32 * h, o);
        if (h == 0 && p.lse) p.lse[((size_t)b * p.nh + hd) * p.Sq + i] = lse;
    }
}

__global__ __launch_bounds__(64) void attn_bwd_bf16_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned Ks[AT_S * 32], Vs[AT_S * 32];          // row-major broadcast tiles
    __shared__ __attribute__((aligned(16))) unsigned Kt[16 * 64], Qt[16 * 64], Gt[16 * 64];  // pair-interleaved over rows
    __shared__ __attribute__((aligned(16))) unsigned U[2 * AT_S * AB_LD];  // phase 1: Q | dO per-lane rows; phase 2: P~ | dS as f32 [32][33]
    unsigned* Qs = U;
    unsigned* Gs = U + AT_S * AB_LD;
    float* Ps = reinterpret_cast<float*>(U);
    float* Ds = reinterpret_cast<float*>(U) + AT_S * AT_PLD;
    static_assert(2 * AT_S * AT_PLD <= 2 * AT_S * AB_LD, "P/dS tiles must fit in the Q/dO staging area");
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
    const bool kvalid = i < p.Sk, qvalid = i < p.Sq;
    const int ik = kvalid ? i : p.Sk - 1, iq = qvalid ? i : p.Sq - 1;
    {
        unsigned t[16];
        load_half_row(p.k, ((size_t)b * p.Sk + ik) * p.ldk + hd * AT_D + 32 * h, kvalid, t);
        store_row_major(Ks, 32, i, h, t);
        store_pair_interleaved(Kt, i, h, t);
        load_half_row(p.v, ((size_t)b * p.Sk + ik) * p.ldv + hd * AT_D + 32 * h, kvalid, t);
        store_row_major(Vs, 32, i, h, t);
        load_half_row(p.q, ((size_t)b * p.Sq + iq) * p.ldq + hd * AT_D + 32 * h, qvalid, t);
        store_row_major(Qs, AB_LD, i, h, t);
        store_pair_interleaved(Qt, i, h, t);
        load_half_row(p.g_out, ((size_t)b * p.Sq + iq) * p.ldo + hd * AT_D + 32 * h, qvalid, t);
        store_row_major(Gs, AB_LD, i, h, t);
        store_pair_interleaved(Gt, i, h, t);
    }
    __syncthreads();
    float s[16], dp[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { s[jj] = 0.f; dp[jj] = 0.f; }
    dot16_bf16(Qs + i * AB_LD, Ks, h, s);
    dot16_bf16(Gs + i * AB_LD, Vs, h, dp);                 // dP~[i][j] = dO[i] . V[j]
    float lse;
    scores_to_probs(p, b, i, h, qvalid, s, lse);
    float keep[16];
    if (p.p_drop > 0.f) attn_keep16(p, bh, i, h, keep);
    else {
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) keep[jj] = 1.0f;
    }
    float delta = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        dp[jj] *= keep[jj];
        delta += s[jj] * dp[jj];
    }
    delta += __shfl_xor(delta, 32, WAVE);
    float ds[16];
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        ds[jj] = s[jj] * (dp[jj] - delta) * p.scale;
        s[jj] *= keep[jj];                                  // P~
    }
    __syncthreads();                                        // every lane is done reading Q | dO rows: reuse as P~ | dS
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        Ps[i * AT_PLD + 16 * h + jj] = s[jj];
        Ds[i * AT_PLD + 16 * h + jj] = ds[jj];
    }
    {   // dQ[i][32h+d] = sum_j dS[i][j] K[j][32h+d]
        unsigned cp[16];
        row_pairs(ds, h, cp);
        float o[32];
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
        rows32_bf16(cp, Kt, h, o);
        if (qvalid) store32_bf16(p.g_q, ((size_t)b * p.Sq + i) * p.ldq + hd * AT_D + 32 * h, o);
    }
    __syncthreads();
    {   // lane = key j (= i): dK[j] = sum_q dS[q][j] Q[q];  dV[j] = sum_q P~[q][j] dO[q]   (contraction over queries)
        unsigned cp[16];
        float o[32];
#pragma unroll
        for (int qp = 0; qp < 16; ++qp) cp[qp] = pack_bf16(Ds[(2 * qp) * AT_PLD + i], Ds[(2 * qp + 1) * AT_PLD + i]);
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
        rows32_bf16(cp, Qt, h, o);
        if (kvalid) store32_bf16(p.g_k, ((size_t)b * p.Sk + i) * p.ldk + hd * AT_D + 32 * h, o);
#pragma unroll
        for (int qp = 0; qp < 16; ++qp) cp[qp] = pack_bf16(Ps[(2 * qp) * AT_PLD + i], Ps[(2 * qp + 1) * AT_PLD + i]);
#pragma unroll
        for (int d = 0; d < 32; ++d) o[d] = 0.f;
        rows32_bf16(cp, Gt, h, o);
        if (kvalid) store32_bf16(p.g_v, ((size_t)b * p.Sk + i) * p.ldv + hd * AT_D + 32 * h, o);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 MFMA path (default): one wave per (batch, head), every product on v_mfma_f32_32x32x16_bf16.
//   S^T = K.Q^T and dP~^T = V.dO^T contract over d: both operands are 16-byte row chunks straight from global memory.
//   The results sit with the QUERY on the lane and 16 keys in the registers (LAY 1), so softmax / dropout / delta are
//   lane-local (+ one cross-half exchange) and P~ / dS feed the products that contract over keys (O^T = V^T.P~^T, dQ^T = K^T.dS^T)
//   directly as the B operand, in the permuted k order  element t of half h <-> key 16s + 8(t>>2) + 4h + (t&3).
//   Their A operands (V^T, K^T: k-strided) come from pair-interleaved LDS tiles Xt[row/2][d] = (X[2r][d] | X[2r+1][d] << 16).
//   dK^T = Q^T.dS and dV^T = dO^T.P~ contract over queries (the lane index of dS): dS / P~ go through a packed bf16
//   [key][query] LDS image once and come back as natural-order B fragments (ds_read_b128).
// ---------------------------------------------------------------------------------------------------------------
typedef __bf16 abf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int AM_LDT = 72;   // dwords per row of the pair-interleaved tiles (64 + 8: conflict-free 16-byte staging writes)
constexpr int AM_LDX = 20;   // dwords per row of the packed [key][query] images (16 + 4)

__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, const f32x16& c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(abf16x8, a), __builtin_bit_cast(abf16x8, b), c, 0, 0, 0);
}
// lane (row r, half h): the four chunks X[r][16s + 8h .. 16s + 8h + 7], s = 0..3 = the A/B fragments of the d-contractions
__device__ __forceinline__ void load_row_chunks(const void* base, size_t row_off, int h, bool valid, uint4 (&f)[4]) {
    const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + row_off + 8 * h;
    const unsigned m = valid ? 0xffffffffu : 0u;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const uint4 t = *reinterpret_cast<const uint4*>(p + 16 * s);
        f[s].x = t.x & m; f[s].y = t.y & m; f[s].z = t.z & m; f[s].w = t.w & m;
    }
}
// The same row chunks, fetched in WHOLE cache lines: wave-instruction j reads rows 8j .. 8j + 7 of the (sentence, head) operand,
// eight lanes per row, 16 bytes per lane -- eight full 128-byte lines per instruction.  load_row_chunks() asks for 32 bytes of each
// of 32 lines per instruction and comes back to every line four times: the same bytes from L2 (counted: profiles/r04_attn_bwd.md), but
// 3.8 x the vector-cache accesses -- 29.9 -> 28.6 us for the backward kernel, and with the stores below 24.7.  rows_to_chunks() then moves
// the data to the lanes the MFMA operands want it on (lane (r, h): chunks h, 2 + h, 4 + h, 6 + h of row r) through a 32 x 144-byte
// LDS image -- the 4608 bytes of the kernel's pair-interleaved tile, before that tile is used for anything else.
__device__ __forceinline__ void load_rows_coalesced(const void* base, size_t row0_off, int ld, int S, int lane, uint4 (&g)[4]) {
    const unsigned short* p = reinterpret_cast<const unsigned short*>(base) + row0_off + 8 * (lane & 7);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * j + (lane >> 3);
        const unsigned m = row < S ? 0xffffffffu : 0u;
        const uint4 t = *reinterpret_cast<const uint4*>(p + (size_t)(row < S ? row : S - 1) * ld);      // clamped, never branched around
        g[j].x = t.x & m; g[j].y = t.y & m; g[j].z = t.z & m; g[j].w = t.w & m;
    }
}
__device__ __forceinline__ void rows_to_chunks(unsigned* T, int r, int h, int lane, const uint4 (&g)[4], uint4 (&f)[4]) {
    char* t = reinterpret_cast<char*>(T);
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(t + (8 * j + (lane >> 3)) * 144 + (lane & 7) * 16) = g[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // one wave per workgroup: its LDS operations stay in order; the fence
    __builtin_amdgcn_wave_barrier();                           // and the wave barrier pin that order for the compiler (no instruction)
#pragma unroll
    for (int s = 0; s < 4; ++s) f[s] = *reinterpret_cast<const uint4*>(t + r * 144 + (2 * s + h) * 16);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// the same chunks -> pair-interleaved tile: even lanes emit the first four d of each chunk, odd lanes the last four
__device__ __forceinline__ void stage_pairs_from_chunks(unsigned* Xt, int r, int h, const uint4 (&f)[4]) {
    const int odd = r & 1;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned k0 = odd ? f[s].z : f[s].x, k1 = odd ? f[s].w : f[s].y;     // my row, my d-range
        const unsigned g0 = odd ? f[s].x : f[s].z, g1 = odd ? f[s].y : f[s].w;     // my row, the partner's d-range
        const unsigned r0 = __shfl_xor(g0, 1, WAVE), r1 = __shfl_xor(g1, 1, WAVE); // partner's row, my d-range
        const unsigned e0 = odd ? r0 : k0, e1 = odd ? r1 : k1, o0 = odd ? k0 : r0, o1 = odd ? k1 : r1;
        uint4 w = {(e0 & 0xffffu) | (o0 << 16), (e0 >> 16) | (o0 & 0xffff0000u), (e1 & 0xffffu) | (o1 << 16), (e1 >> 16) | (o1 & 0xffff0000u)};
        *reinterpret_cast<uint4*>(Xt + (r >> 1) * AM_LDT + 16 * s + 8 * h + 4 * odd) = w;
    }
}
// A fragment X^T[d = dcol][k] of k-step st from a pair-interleaved tile.  PERM 1: the permuted k order of an accumulator
// used as B operand; PERM 0: natural order k = 16 st + 8h + t.
template <int PERM>
__device__ __forceinline__ uint4 frag_from_pairs(const unsigned* Xt, int dcol, int h, int st) {
    uint4 a;
    if (PERM) {
        const unsigned* b = Xt + (8 * st + 2 * h) * AM_LDT + dcol;
        a.x = b[0]; a.y = b[AM_LDT]; a.z = b[4 * AM_LDT]; a.w = b[5 * AM_LDT];
    } else {
        const unsigned* b = Xt + (8 * st + 4 * h) * AM_LDT + dcol;
        a.x = b[0]; a.y = b[AM_LDT]; a.z = b[2 * AM_LDT]; a.w = b[3 * AM_LDT];
    }
    return a;
}
// accumulator-layout values x[16] (row r on the lane) -> the two B fragments of the key contraction
__device__ __forceinline__ void acc_to_frags(const float (&x)[16], uint4 (&f)[2]) {
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        f[st].x = pack_bf16(x[8 * st], x[8 * st + 1]); f[st].y = pack_bf16(x[8 * st + 2], x[8 * st + 3]);
        f[st].z = pack_bf16(x[8 * st + 4], x[8 * st + 5]); f[st].w = pack_bf16(x[8 * st + 6], x[8 * st + 7]);
    }
}
// accumulator-layout values x (query r on the lane, key 8g + 4h + t in slot 4g + t) -> packed image T[key][query pair]
__device__ __forceinline__ void stage_transposed(unsigned* T, int r, int h, const float (&x)[16]) {
    const int odd = r & 1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const unsigned m01 = pack_bf16(x[4 * g], x[4 * g + 1]), m23 = pack_bf16(x[4 * g + 2], x[4 * g + 3]);
        const unsigned got = __shfl_xor(odd ? m01 : m23, 1, WAVE);
        const unsigned mine = odd ? m23 : m01;
        const unsigned ev = odd ? got : mine, od = odd ? mine : got;     // the even / odd query's values of my two keys
        const int j = 8 * g + 4 * h + 2 * odd;
        T[j * AM_LDX + (r >> 1)] = (ev & 0xffffu) | (od << 16);
        T[(j + 1) * AM_LDX + (r >> 1)] = (ev >> 16) | (od & 0xffff0000u);
    }
}
// C tile of a [d][row] product (row r on the lane, d = d_base + 8g + 4h + t in register 4g + t).  As it stands a lane holds four
// separate 8-byte pieces of its row; the lane pair (r, h = 0 / 1) holds neighbouring pieces.  One v_permlane32_swap per dword
// (lanes 32..63 of the first operand <-> lanes 0..31 of the second) trades piece 2j + 1 of the lower lane for piece 2j of the
// upper one: the lower lane then owns d_base + 16j .. + 7, the upper lane d_base + 16j + 8 .. + 15 -- two 16-byte stores per
// lane instead of four 8-byte ones (guide T21: a row-per-lane epilogue is store-ISSUE bound).  Round 2 measured the stores at
// 10.6 of the backward kernel's 31 us for 37.7 MB; plain stores (the streaming policy, which bypasses L2's write combining,
// took the kernel to 80 us).
__device__ __forceinline__ void store_ct(void* base, size_t row_off, int h, int d_base, const f32x16& c) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    unsigned short* p = reinterpret_cast<unsigned short*>(base) + row_off + d_base + 8 * h;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const unsigned xa = pack_bf16(c[8 * j], c[8 * j + 1]), xb = pack_bf16(c[8 * j + 2], c[8 * j + 3]);          // piece 2j
        const unsigned ya = pack_bf16(c[8 * j + 4], c[8 * j + 5]), yb = pack_bf16(c[8 * j + 6], c[8 * j + 7]);      // piece 2j + 1
        const u32x2 sa = __builtin_amdgcn_permlane32_swap(xa, ya, false, false);
        const u32x2 sb = __builtin_amdgcn_permlane32_swap(xb, yb, false, false);
        uint4 w = {sa.x, sb.x, sa.y, sb.y};
        *reinterpret_cast<uint4*>(p + 16 * j) = w;
    }
}
// The C tiles of both halves of d (dt = 0, 1) as WHOLE-LINE stores: after store_ct()'s permlane swap lane (r, h) owns the 16-byte
// chunks h, 2 + h, 4 + h, 6 + h of row r -- the distribution rows_to_chunks() produces for the loads.  They go through a 32 x 144-byte
// LDS row image and leave as eight full 128-byte lines per wave-instruction (eight lanes per row) instead of 32 bytes of each of 32
// lines.  T must be free (4608 bytes): every earlier read of it complete (one wave per workgroup: LDS operations stay in order).
__device__ __forceinline__ void store_rows_coalesced(unsigned* T, void* base, size_t row0_off, int ld, int S, int r, int h, int lane,
                                                     const f32x16& c0, const f32x16& c1, unsigned char* base8 = nullptr, size_t row0_off8 = 0,
                                                     int ld8 = 0, float* st8 = nullptr, unsigned slot8 = 0) {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    char* t = reinterpret_cast<char*>(T);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        const f32x16& c = dt ? c1 : c0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned xa = pack_bf16(c[8 * j], c[8 * j + 1]), xb = pack_bf16(c[8 * j + 2], c[8 * j + 3]);
            const unsigned ya = pack_bf16(c[8 * j + 4], c[8 * j + 5]), yb = pack_bf16(c[8 * j + 6], c[8 * j + 7]);
            const u32x2 sa = __builtin_amdgcn_permlane32_swap(xa, ya, false, false);
            const u32x2 sb = __builtin_amdgcn_permlane32_swap(xb, yb, false, false);
            const uint4 w = {sa.x, sb.x, sa.y, sb.y};
            *reinterpret_cast<uint4*>(t + r * 144 + (2 * (2 * dt + j) + h) * 16) = w;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    unsigned short* p = reinterpret_cast<unsigned short*>(base) + row0_off + 8 * (lane & 7);
    const float s8 = base8 ? st8[0] : 1.0f;
    float am8 = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int row = 8 * j + (lane >> 3);
        const uint4 v = *reinterpret_cast<const uint4*>(t + row * 144 + (lane & 7) * 16);
        if (row < S) {
            *reinterpret_cast<uint4*>(p + (size_t)row * ld) = v;
            if (base8) {                                                  // (uniform) the fp8 copy of the same 8 values
                am8 = fmaxf(am8, amax8(v));
                *reinterpret_cast<uint2*>(base8 + row0_off8 + (size_t)row * ld8 + 8 * (lane & 7)) = quant8(v, s8);
            }
        }
    }
    if (base8) fp8_amax_note(am8, st8, slot8);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
// bias-gradient partial of one projection: sum_rows X[row][d] * w[row] for d = 32h + r, X from its pair-interleaved tile,
// w (f32, one per row) broadcast from LDS.  Equals the column sum over this sentence of the gradient the kernel stores.
__device__ __forceinline__ float weighted_colsum(const unsigned* Xt, const float* w, int r, int h) {
    float a0 = 0.f, a1 = 0.f;
#pragma unroll
    for (int rp = 0; rp < 16; ++rp) {
        const unsigned x = Xt[rp * AM_LDT + 32 * h + r];
        const float2 c = *reinterpret_cast<const float2*>(w + 2 * rp);
        a0 = __builtin_fmaf(__uint_as_float(x << 16), c.x, a0);
        a1 = __builtin_fmaf(__uint_as_float(x & 0xffff0000u), c.y, a1);
    }
    return a0 + a1;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int v = 0; v < 16; ++v) z[v] = 0.f;
    return z;
}

// Diagnostic build only (-DKVQ_NN_DIAG, tools/build_diag.sh): every wave of the two 32-token attention kernels stores s_memtime at
// entry, when its operand rows have landed, when its first store is issued and at its end: [workgroup][8] u64.
#ifdef KVQ_NN_DIAG
__device__ unsigned long long* g_nn_diag = nullptr;
#define NN_T(var) const unsigned long long var = __builtin_amdgcn_s_memtime();
#define NN_DIAG_STORE(t0, t1, t2)                                                                     \
    if (g_nn_diag != nullptr && threadIdx.x == 0) {                                                   \
        unsigned long long* o = g_nn_diag + (size_t)blockIdx.x * 8;                                   \
        o[0] = t0; o[1] = t1; o[2] = t2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = __builtin_amdgcn_s_memrealtime(); \
    }
#else
#define NN_T(var)
#define NN_DIAG_STORE(t0, t1, t2)
#endif

template <bool COAL, bool STC>
__global__ __launch_bounds__(64) void attn_fwd_mfma_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned Vt[16 * AM_LDT];
    static_assert(16 * AM_LDT * 4 == 32 * 144, "the pair-interleaved tile doubles as the 32 x 144-byte row image of rows_to_chunks()");
    NN_T(nn_t0)
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
    const bool kvalid = r < p.Sk, qvalid = r < p.Sq;
    const int rk = kvalid ? r : p.Sk - 1, rq = qvalid ? r : p.Sq - 1;
    const int64_t mv = attn_mask_element(p, b);                        // the first load out: its wait leaves the row loads in flight
    uint4 kf[4], qf[4], vf[4];
    uint4 gk[COAL ? 4 : 1], gq[COAL ? 4 : 1], gv[COAL ? 4 : 1];
    if constexpr (COAL) {
        load_rows_coalesced(p.k, (size_t)b * p.Sk * p.ldk + hd * AT_D, p.ldk, p.Sk, lane, gk);
        load_rows_coalesced(p.q, (size_t)b * p.Sq * p.ldq + hd * AT_D, p.ldq, p.Sq, lane, gq);
        load_rows_coalesced(p.v, (size_t)b * p.Sk * p.ldv + hd * AT_D, p.ldv, p.Sk, lane, gv);
    } else {
        load_row_chunks(p.k, ((size_t)b * p.Sk + rk) * p.ldk + hd * AT_D, h, kvalid, kf);
        load_row_chunks(p.q, ((size_t)b * p.Sq + rq) * p.ldq + hd * AT_D, h, qvalid, qf);
        load_row_chunks(p.v, ((size_t)b * p.Sk + rk) * p.ldv + hd * AT_D, h, kvalid, vf);
    }
    // the dropout bits need nothing that is being loaded: ~2300 cycles of Philox per wave (quarter-rate 32 x 32 multiplies) drawn
    // while the rows travel -- every wave of the launch is in the same phase, so nothing else would use the VALU then
    const unsigned long long seed = attn_seed(p);
    float keep[16];
    if (p.p_drop > 0.f) attn_keep16<1>(p, bh, r, h, keep, &seed);
    else {
#pragma unroll
        for (int v = 0; v < 16; ++v) keep[v] = 1.0f;
    }
    const unsigned long long kmask = attn_mask_ballot(p, mv);
    if constexpr (COAL) {
        rows_to_chunks(Vt, r, h, lane, gk, kf);
        rows_to_chunks(Vt, r, h, lane, gq, qf);
        rows_to_chunks(Vt, r, h, lane, gv, vf);
    }
    f32x16 acc = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) acc = mfma32(kf[s], qf[s], acc);      // S^T[key][query]
    stage_pairs_from_chunks(Vt, r, h, vf);
    NN_T(nn_t1)                                                       // (k, q, v rows have landed: the first MFMAs and the V staging consumed them)
    float s[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) s[v] = acc[v];
    float lse;
    scores_to_probs<1>(p, b, r, h, qvalid, s, lse, &kmask);
#pragma unroll
    for (int v = 0; v < 16; ++v) s[v] *= keep[v];
    uint4 pf[2];
    acc_to_frags(s, pf);
    __syncthreads();
    NN_T(nn_t2)                                                       // (softmax and dropout done: only P.V and the stores are left)
    f32x16 o2[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        o2[dt] = zero16();
#pragma unroll
        for (int st = 0; st < 2; ++st) o2[dt] = mfma32(frag_from_pairs<1>(Vt, 32 * dt + r, h, st), pf[st], o2[dt]);   // O^T[d][query]
        if (!STC && qvalid) store_ct(p.out, ((size_t)b * p.Sq + r) * p.ldo + hd * AT_D, h, 32 * dt, o2[dt]);
    }
    if constexpr (STC) store_rows_coalesced(Vt, p.out, (size_t)b * p.Sq * p.ldo + hd * AT_D, p.ldo, p.Sq, r, h, lane, o2[0], o2[1],
                                            p.out8, (size_t)b * p.Sq * p.ld8 + hd * AT_D, p.ld8, p.st8, (unsigned)bh);
    if (qvalid && h == 0 && p.lse) p.lse[((size_t)b * p.nh + hd) * p.Sq + r] = lse;
    NN_DIAG_STORE(nn_t0, nn_t1, nn_t2)
}

template <bool COAL, bool STC>
__global__ __launch_bounds__(64) void attn_bwd_mfma_kernel(AttnParams p) {
    // ONE pair-interleaved staging tile, used for K, then Q, then dO (their row chunks stay in registers): 10 KiB of LDS per
    // (sentence, head) instead of 19 KiB, so that all B * nh waves of a step-sized launch (3072 at bert-base, 12 per CU) are
    // resident at once -- with three tiles the LDS admitted 8 per CU and the launch ran one and a half rounds of a latency-bound kernel
    __shared__ __attribute__((aligned(16))) unsigned Xt[16 * AM_LDT];
    __shared__ __attribute__((aligned(16))) unsigned TDP[2 * AT_S * AM_LDX];      // dS | P~ transposed images; afterwards the row image
    unsigned* const TD = TDP;                                                     // of the three gradients' whole-line stores
    unsigned* const TP = TDP + AT_S * AM_LDX;
    static_assert(2 * AT_S * AM_LDX * 4 >= 32 * 144, "the transposed images double as the row image of store_rows_coalesced()");
    __shared__ __attribute__((aligned(16))) float Wv[3 * AT_S];
    NN_T(nn_t0)
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int bh = blockIdx.x, b = bh / p.nh, hd = bh % p.nh;
    const bool kvalid = r < p.Sk, qvalid = r < p.Sq;
    const int rk = kvalid ? r : p.Sk - 1, rq = qvalid ? r : p.Sq - 1;
    const int64_t mv = attn_mask_element(p, b);                        // (see attn_fwd_mfma_kernel)
    uint4 kf[4], qf[4], vf[4], gf[4];
    uint4 gk[COAL ? 4 : 1], gq[COAL ? 4 : 1], gv[COAL ? 4 : 1], gg[COAL ? 4 : 1];
    if constexpr (COAL) {                                              // (see load_rows_coalesced)
        load_rows_coalesced(p.k, (size_t)b * p.Sk * p.ldk + hd * AT_D, p.ldk, p.Sk, lane, gk);
        load_rows_coalesced(p.q, (size_t)b * p.Sq * p.ldq + hd * AT_D, p.ldq, p.Sq, lane, gq);
        load_rows_coalesced(p.v, (size_t)b * p.Sk * p.ldv + hd * AT_D, p.ldv, p.Sk, lane, gv);
        load_rows_coalesced(p.g_out, (size_t)b * p.Sq * p.ldo + hd * AT_D, p.ldo, p.Sq, lane, gg);
    } else {
        load_row_chunks(p.k, ((size_t)b * p.Sk + rk) * p.ldk + hd * AT_D, h, kvalid, kf);
        load_row_chunks(p.q, ((size_t)b * p.Sq + rq) * p.ldq + hd * AT_D, h, qvalid, qf);
        load_row_chunks(p.v, ((size_t)b * p.Sk + rk) * p.ldv + hd * AT_D, h, kvalid, vf);
        load_row_chunks(p.g_out, ((size_t)b * p.Sq + rq) * p.ldo + hd * AT_D, h, qvalid, gf);
    }
    const unsigned long long seed = attn_seed(p);
    float keep[16];
    if (p.p_drop > 0.f) attn_keep16<1>(p, bh, r, h, keep, &seed);      // drawn while the rows travel
    else {
#pragma unroll
        for (int v = 0; v < 16; ++v) keep[v] = 1.0f;
    }
    const unsigned long long kmask = attn_mask_ballot(p, mv);
    if constexpr (COAL) {
        rows_to_chunks(Xt, r, h, lane, gk, kf);
        rows_to_chunks(Xt, r, h, lane, gq, qf);
        rows_to_chunks(Xt, r, h, lane, gv, vf);
        rows_to_chunks(Xt, r, h, lane, gg, gf);
    }
    f32x16 accS = zero16(), accP = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) accS = mfma32(kf[s], qf[s], accS);     // S^T[key][query]
#pragma unroll
    for (int s = 0; s < 4; ++s) accP = mfma32(vf[s], gf[s], accP);     // dP~^T[key][query] = V[key] . dO[query]
    stage_pairs_from_chunks(Xt, r, h, kf);
    NN_T(nn_t1)                                                       // (all four operand row sets have landed)
    float s[16], dp[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) { s[v] = accS[v]; dp[v] = accP[v]; }
    float lse;
    scores_to_probs<1>(p, b, r, h, qvalid, s, lse, &kmask);
    float delta = 0.f;
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        dp[v] *= keep[v];
        delta += s[v] * dp[v];
    }
    delta += __shfl_xor(delta, 32, WAVE);
    float ds[16];
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        ds[v] = s[v] * (dp[v] - delta) * p.scale;
        s[v] *= keep[v];                                    // P~
    }
    stage_transposed(TD, r, h, ds);
    stage_transposed(TP, r, h, s);
    uint4 dsf[2];
    acc_to_frags(ds, dsf);
    __syncthreads();
    NN_T(nn_t2)                                                       // (P, dS ready and staged: three products and the stores are left)
    uint4 tdf[2], tpf[2];                                   // lane = key r: dS[query 16st + 8h + t][r], P~[..][r]
#pragma unroll
    for (int st = 0; st < 2; ++st) {                        // (read before dQ: the images are then free for the gradients' row image)
        tdf[st] = *reinterpret_cast<const uint4*>(TD + r * AM_LDX + 8 * st + 4 * h);
        tpf[st] = *reinterpret_cast<const uint4*>(TP + r * AM_LDX + 8 * st + 4 * h);
    }
    f32x16 o2[2];
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {                        // dQ^T[d][query] = sum_key K^T[d][key] dS^T[key][query]
        o2[dt] = zero16();
#pragma unroll
        for (int st = 0; st < 2; ++st) o2[dt] = mfma32(frag_from_pairs<1>(Xt, 32 * dt + r, h, st), dsf[st], o2[dt]);
        if (!STC && qvalid) store_ct(p.g_q, ((size_t)b * p.Sq + r) * p.ldq + hd * AT_D, h, 32 * dt, o2[dt]);
    }
    if constexpr (STC) store_rows_coalesced(TDP, p.g_q, (size_t)b * p.Sq * p.ldq + hd * AT_D, p.ldq, p.Sq, r, h, lane, o2[0], o2[1]);
    const bool partials = p.pb_q || p.pb_k || p.pb_v;       // wave-uniform
    const size_t col = (size_t)hd * AT_D + 32 * h + r;
    if (partials) {
        // column sums over this sentence of the three gradients = bias-gradient partials of the q / k / v projections:
        //   sum_q dQ[q][d] = sum_key K[key][d] cs[key],  cs[key] = sum_q dS[q][key]     (from the transposed image, lane = key)
        //   sum_k dK[k][d] = sum_q   Q[q][d]   rs[q],    rs[q]   = sum_key dS[q][key]   (lane-local)
        //   sum_k dV[k][d] = sum_q  dO[q][d]   rp[q],    rp[q]   = sum_key P~[q][key]
        float rs = 0.f, rp = 0.f, cs = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) { rs += IO<KVQ_BF16>::round(ds[v]); rp += IO<KVQ_BF16>::round(s[v]); }
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const unsigned w4[4] = {tdf[st].x, tdf[st].y, tdf[st].z, tdf[st].w};
#pragma unroll
            for (int u = 0; u < 4; ++u) cs += __uint_as_float(w4[u] << 16) + __uint_as_float(w4[u] & 0xffff0000u);
        }
        rs += __shfl_xor(rs, 32, WAVE); rp += __shfl_xor(rp, 32, WAVE); cs += __shfl_xor(cs, 32, WAVE);
        if (h == 0) { Wv[r] = cs; Wv[32 + r] = rs; Wv[64 + r] = rp; }
        __syncthreads();
        if (p.pb_q) p.pb_q[(size_t)b * p.ldp_q + col] = weighted_colsum(Xt, Wv, r, h);
    }
    __syncthreads();                                        // every read of the K tile is done: the tile becomes Q
    stage_pairs_from_chunks(Xt, r, h, qf);
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {                        // dK^T[d][key] = sum_q Q^T[d][q] dS[q][key]
        o2[dt] = zero16();
#pragma unroll
        for (int st = 0; st < 2; ++st) o2[dt] = mfma32(frag_from_pairs<0>(Xt, 32 * dt + r, h, st), tdf[st], o2[dt]);
        if (!STC && kvalid) store_ct(p.g_k, ((size_t)b * p.Sk + r) * p.ldk + hd * AT_D, h, 32 * dt, o2[dt]);
    }
    if constexpr (STC) store_rows_coalesced(TDP, p.g_k, (size_t)b * p.Sk * p.ldk + hd * AT_D, p.ldk, p.Sk, r, h, lane, o2[0], o2[1]);
    if (partials && p.pb_k) p.pb_k[(size_t)b * p.ldp_kv + col] = weighted_colsum(Xt, Wv + 32, r, h);
    __syncthreads();                                        // ... and now dO
    stage_pairs_from_chunks(Xt, r, h, gf);
    __syncthreads();
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {                        // dV^T[d][key] = sum_q dO^T[d][q] P~[q][key]
        o2[dt] = zero16();
#pragma unroll
        for (int st = 0; st < 2; ++st) o2[dt] = mfma32(frag_from_pairs<0>(Xt, 32 * dt + r, h, st), tpf[st], o2[dt]);
        if (!STC && kvalid) store_ct(p.g_v, ((size_t)b * p.Sk + r) * p.ldv + hd * AT_D, h, 32 * dt, o2[dt]);
    }
    if constexpr (STC) store_rows_coalesced(TDP, p.g_v, (size_t)b * p.Sk * p.ldv + hd * AT_D, p.ldv, p.Sk, r, h, lane, o2[0], o2[1]);
    if (partials && p.pb_v) p.pb_v[(size_t)b * p.ldp_kv + col] = weighted_colsum(Xt, Wv + 64, r, h);
    NN_DIAG_STORE(nn_t0, nn_t1, nn_t2)
}

// ---------------------------------------------------------------------------------------------------------------
// Sequences of 33 .. 128 tokens (bf16): the same wave-level products, in a loop over 32-token blocks.
//   forward : one wave per (sentence, head, query block); key blocks in order with a running maximum / sum (the probabilities
//             are rescaled, dropped and contracted with V block by block; the row is normalised once at the end)
//   backward: two kernels so that no gradient needs an atomic -- one wave per (sentence, head, query block) for dQ, one per
//             (sentence, head, key block) for dK and dV.  Both recompute P from the saved log-sum-exp and take
//             delta[q] = sum_d dO[q][d] O[q][d] from the saved forward output (a lane owns half a row of both).
// The reference tokenises to 12 - 14 tokens and BASELINE.json to 32: these kernels are the engine's cover for longer inputs,
// not part of the benchmarked step.  Dropout bits are indexed over the PADDED [queries][keys] grid of a (sentence, head), so the
// three kernels draw the same mask whatever block they look at.
// ---------------------------------------------------------------------------------------------------------------
constexpr int AT_SMAX = 128;

__device__ __forceinline__ unsigned blk_key_mask(const AttnParams& p, int b, int k0) {
    const int kg = k0 + (threadIdx.x & 31);
    const int kc = kg < p.Sk ? kg : p.Sk - 1;
    const int64_t* mp = p.mask ? p.mask + ((size_t)b * p.Sk + kc) : reinterpret_cast<const int64_t*>(p.q);   // (no load behind a branch)
    const int64_t mv = *mp;
    return (unsigned)__ballot(kg < p.Sk && (!p.mask || mv != 0));       // lanes 0..31; the upper half repeats them
}
__device__ __forceinline__ void blk_keep16(const AttnParams& p, unsigned long long seed, int bh, int SQP, int SKP, int ig, int k0, int h,
                                           float (&keep)[16]) {
    if (p.p_drop > 0.f) {
        const float inv_keep = 1.0f / (1.0f - p.p_drop);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const unsigned long long e4 = (((unsigned long long)bh * SQP + ig) * SKP + k0 + attn_key<1>(h, 4 * c)) >> 2;
            const U4 r = drop_bits(seed, p.site, e4);
            keep[4 * c] = keep_scale(r.x, p.thresh, inv_keep); keep[4 * c + 1] = keep_scale(r.y, p.thresh, inv_keep);
            keep[4 * c + 2] = keep_scale(r.z, p.thresh, inv_keep); keep[4 * c + 3] = keep_scale(r.w, p.thresh, inv_keep);
        }
    } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) keep[v] = 1.0f;
    }
}
// raw scores -> scaled, masked scores (-inf where the pair does not attend); returns the row maximum over this block's 32 keys
__device__ __forceinline__ float blk_scores(const AttnParams& p, unsigned kmask, int ig, int k0, int h, bool qvalid, float (&s)[16]) {
    float mx = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = attn_key<1>(h, jj);
        bool ok = qvalid && ((kmask >> j) & 1u);
        if (p.causal) ok = ok && (k0 + j) <= ig;
        s[jj] = ok ? s[jj] * p.scale : -INFINITY;
        mx = fmaxf(mx, s[jj]);
    }
    return fmaxf(mx, __shfl_xor(mx, 32, WAVE));
}
// sum over the lane's half row (the four chunks of load_row_chunks) of a[d] * b[d], completed with the partner lane
__device__ __forceinline__ float blk_row_dot(const uint4 (&a)[4], const uint4 (&b)[4]) {
    float acc = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const unsigned x[4] = {a[s].x, a[s].y, a[s].z, a[s].w}, y[4] = {b[s].x, b[s].y, b[s].z, b[s].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_fmaf(__uint_as_float(x[u] << 16), __uint_as_float(y[u] << 16), acc);
            acc = __builtin_fmaf(__uint_as_float(x[u] & 0xffff0000u), __uint_as_float(y[u] & 0xffff0000u), acc);
        }
    }
    return acc + __shfl_xor(acc, 32, WAVE);
}

__global__ __launch_bounds__(64) void attn_fwd_blk_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned Vt[16 * AM_LDT];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int nqb = (p.Sq + 31) / 32, nkb = (p.Sk + 31) / 32;
    const int bh = blockIdx.x / nqb, qb = blockIdx.x % nqb, b = bh / p.nh, hd = bh % p.nh;
    const int ig = 32 * qb + r;
    const bool qvalid = ig < p.Sq;
    const int rq = qvalid ? ig : p.Sq - 1;
    uint4 qf[4];
    load_row_chunks(p.q, ((size_t)b * p.Sq + rq) * p.ldq + hd * AT_D, h, qvalid, qf);
    const unsigned long long seed = attn_seed(p);
    float m_run = -INFINITY, l_run = 0.f;
    f32x16 o[2] = {zero16(), zero16()};
    const int kend = p.causal ? (qb + 1 < nkb ? qb + 1 : nkb) : nkb;
    for (int kb = 0; kb < kend; ++kb) {
        const int k0 = 32 * kb, kg = k0 + r;
        const bool kvalid = kg < p.Sk;
        const int rk = kvalid ? kg : p.Sk - 1;
        uint4 kf[4], vf[4];
        load_row_chunks(p.k, ((size_t)b * p.Sk + rk) * p.ldk + hd * AT_D, h, kvalid, kf);
        load_row_chunks(p.v, ((size_t)b * p.Sk + rk) * p.ldv + hd * AT_D, h, kvalid, vf);
        const unsigned kmask = blk_key_mask(p, b, k0);
        float keep[16];
        blk_keep16(p, seed, bh, 32 * nqb, 32 * nkb, ig, k0, h, keep);
        f32x16 acc = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = mfma32(kf[s], qf[s], acc);      // S^T[key][query]
        __syncthreads();                                                  // the previous block's reads of Vt are done
        stage_pairs_from_chunks(Vt, r, h, vf);
        float s[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) s[v] = acc[v];
        const float mx = blk_scores(p, kmask, ig, k0, h, qvalid, s);
        const float m_new = fmaxf(m_run, mx);
        const float mref = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __expf(m_run - mref);                         // 0 while nothing was seen (m_run = -inf)
        float sum = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            s[v] = __expf(s[v] - mref);
            sum += s[v];
            s[v] *= keep[v];
        }
        sum += __shfl_xor(sum, 32, WAVE);
        l_run = l_run * alpha + sum;
        m_run = m_new;
        uint4 pf[2];
        acc_to_frags(s, pf);
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
            for (int v = 0; v < 16; ++v) o[dt][v] *= alpha;
#pragma unroll
            for (int st = 0; st < 2; ++st) o[dt] = mfma32(frag_from_pairs<1>(Vt, 32 * dt + r, h, st), pf[st], o[dt]);   // O^T[d][query]
        }
    }
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
#pragma unroll
        for (int v = 0; v < 16; ++v) o[dt][v] *= inv;
        if (qvalid) store_ct(p.out, ((size_t)b * p.Sq + ig) * p.ldo + hd * AT_D, h, 32 * dt, o[dt]);
    }
    if (qvalid && h == 0 && p.lse)
        p.lse[((size_t)b * p.nh + hd) * p.Sq + ig] = (m_run == -INFINITY ? 0.f : m_run) + __logf(fmaxf(l_run, 1e-37f));
}

// what both backward kernels do with one (query block, key block) pair once K, V, Q, dO of the blocks are in registers:
// P (from the saved lse), P~ = P keep, dS = P (dP~ keep - delta) scale -- all in the accumulator layout (query r on the lane)
__device__ __forceinline__ void blk_grad_pair(const AttnParams& p, const uint4 (&kf)[4], const uint4 (&vf)[4], const uint4 (&qf)[4],
                                              const uint4 (&gf)[4], unsigned kmask, const float (&keep)[16], int ig, int k0, int h,
                                              bool qvalid, float lse, float delta, float (&pt)[16], float (&ds)[16]) {
    f32x16 accS = zero16(), accP = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) accS = mfma32(kf[s], qf[s], accS);     // S^T[key][query]
#pragma unroll
    for (int s = 0; s < 4; ++s) accP = mfma32(vf[s], gf[s], accP);     // dP~^T[key][query]
#pragma unroll
    for (int v = 0; v < 16; ++v) pt[v] = accS[v];
    blk_scores(p, kmask, ig, k0, h, qvalid, pt);
#pragma unroll
    for (int v = 0; v < 16; ++v) {
        const float pr = __expf(pt[v] - lse);                           // exp(-inf) = 0 where the pair does not attend
        ds[v] = pr * (accP[v] * keep[v] - delta) * p.scale;
        pt[v] = pr * keep[v];
    }
}

__global__ __launch_bounds__(64) void attn_bwd_blk_dq_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned Xt[16 * AM_LDT];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int nqb = (p.Sq + 31) / 32, nkb = (p.Sk + 31) / 32;
    const int bh = blockIdx.x / nqb, qb = blockIdx.x % nqb, b = bh / p.nh, hd = bh % p.nh;
    const int ig = 32 * qb + r;
    const bool qvalid = ig < p.Sq;
    const int rq = qvalid ? ig : p.Sq - 1;
    uint4 qf[4], gf[4], of[4];
    load_row_chunks(p.q, ((size_t)b * p.Sq + rq) * p.ldq + hd * AT_D, h, qvalid, qf);
    load_row_chunks(p.g_out, ((size_t)b * p.Sq + rq) * p.ldo + hd * AT_D, h, qvalid, gf);
    load_row_chunks(p.out, ((size_t)b * p.Sq + rq) * p.ldo + hd * AT_D, h, qvalid, of);
    const float lse = p.lse[((size_t)b * p.nh + hd) * p.Sq + rq];
    const unsigned long long seed = attn_seed(p);
    const float delta = blk_row_dot(gf, of);
    f32x16 dq[2] = {zero16(), zero16()};
    const int kend = p.causal ? (qb + 1 < nkb ? qb + 1 : nkb) : nkb;
    for (int kb = 0; kb < kend; ++kb) {
        const int k0 = 32 * kb, kg = k0 + r;
        const bool kvalid = kg < p.Sk;
        const int rk = kvalid ? kg : p.Sk - 1;
        uint4 kf[4], vf[4];
        load_row_chunks(p.k, ((size_t)b * p.Sk + rk) * p.ldk + hd * AT_D, h, kvalid, kf);
        load_row_chunks(p.v, ((size_t)b * p.Sk + rk) * p.ldv + hd * AT_D, h, kvalid, vf);
        const unsigned kmask = blk_key_mask(p, b, k0);
        float keep[16], pt[16], ds[16];
        blk_keep16(p, seed, bh, 32 * nqb, 32 * nkb, ig, k0, h, keep);
        blk_grad_pair(p, kf, vf, qf, gf, kmask, keep, ig, k0, h, qvalid, lse, delta, pt, ds);
        uint4 dsf[2];
        acc_to_frags(ds, dsf);
        __syncthreads();
        stage_pairs_from_chunks(Xt, r, h, kf);
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)                          // dQ^T[d][query] += sum_key K^T[d][key] dS^T[key][query]
#pragma unroll
            for (int st = 0; st < 2; ++st) dq[dt] = mfma32(frag_from_pairs<1>(Xt, 32 * dt + r, h, st), dsf[st], dq[dt]);
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
        if (qvalid) store_ct(p.g_q, ((size_t)b * p.Sq + ig) * p.ldq + hd * AT_D, h, 32 * dt, dq[dt]);
}

__global__ __launch_bounds__(64) void attn_bwd_blk_dkv_kernel(AttnParams p) {
    __shared__ __attribute__((aligned(16))) unsigned Xt[16 * AM_LDT];
    __shared__ __attribute__((aligned(16))) unsigned TD[AT_S * AM_LDX], TP[AT_S * AM_LDX];
    const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
    const int nqb = (p.Sq + 31) / 32, nkb = (p.Sk + 31) / 32;
    const int bh = blockIdx.x / nkb, kb = blockIdx.x % nkb, b = bh / p.nh, hd = bh % p.nh;
    const int k0 = 32 * kb, kg = k0 + r;
    const bool kvalid = kg < p.Sk;
    const int rk = kvalid ? kg : p.Sk - 1;
    uint4 kf[4], vf[4];
    load_row_chunks(p.k, ((size_t)b * p.Sk + rk) * p.ldk + hd * AT_D, h, kvalid, kf);
    load_row_chunks(p.v, ((size_t)b * p.Sk + rk) * p.ldv + hd * AT_D, h, kvalid, vf);
    const unsigned kmask = blk_key_mask(p, b, k0);
    const unsigned long long seed = attn_seed(p);
    f32x16 dk[2] = {zero16(), zero16()}, dv[2] = {zero16(), zero16()};
    for (int qb = p.causal ? kb : 0; qb < nqb; ++qb) {
        const int ig = 32 * qb + r;
        const bool qvalid = ig < p.Sq;
        const int rq = qvalid ? ig : p.Sq - 1;
        uint4 qf[4], gf[4], of[4];
        load_row_chunks(p.q, ((size_t)b * p.Sq + rq) * p.ldq + hd * AT_D, h, qvalid, qf);
        load_row_chunks(p.g_out, ((size_t)b * p.Sq + rq) * p.ldo + hd * AT_D, h, qvalid, gf);
        load_row_chunks(p.out, ((size_t)b * p.Sq + rq) * p.ldo + hd * AT_D, h, qvalid, of);
        const float lse = p.lse[((size_t)b * p.nh + hd) * p.Sq + rq];
        const float delta = blk_row_dot(gf, of);
        float keep[16], pt[16], ds[16];
        blk_keep16(p, seed, bh, 32 * nqb, 32 * nkb, ig, k0, h, keep);
        blk_grad_pair(p, kf, vf, qf, gf, kmask, keep, ig, k0, h, qvalid, lse, delta, pt, ds);
        __syncthreads();                                        // the previous pair's reads of the three tiles are done
        stage_transposed(TD, r, h, ds);
        stage_transposed(TP, r, h, pt);
        stage_pairs_from_chunks(Xt, r, h, qf);
        __syncthreads();
        uint4 tdf[2], tpf[2];                                   // lane = key r: dS[query 16st + 8h + t][r], P~[..][r]
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            tdf[st] = *reinterpret_cast<const uint4*>(TD + r * AM_LDX + 8 * st + 4 * h);
            tpf[st] = *reinterpret_cast<const uint4*>(TP + r * AM_LDX + 8 * st + 4 * h);
        }
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)                          // dK^T[d][key] += sum_q Q^T[d][q] dS[q][key]
#pragma unroll
            for (int st = 0; st < 2; ++st) dk[dt] = mfma32(frag_from_pairs<0>(Xt, 32 * dt + r, h, st), tdf[st], dk[dt]);
        __syncthreads();
        stage_pairs_from_chunks(Xt, r, h, gf);
        __syncthreads();
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)                          // dV^T[d][key] += sum_q dO^T[d][q] P~[q][key]
#pragma unroll
            for (int st = 0; st < 2; ++st) dv[dt] = mfma32(frag_from_pairs<0>(Xt, 32 * dt + r, h, st), tpf[st], dv[dt]);
    }
#pragma unroll
    for (int dt = 0; dt < 2; ++dt) {
        if (kvalid) store_ct(p.g_k, ((size_t)b * p.Sk + kg) * p.ldk + hd * AT_D, h, 32 * dt, dk[dt]);
        if (kvalid) store_ct(p.g_v, ((size_t)b * p.Sk + kg) * p.ldv + hd * AT_D, h, 32 * dt, dv[dt]);
    }
}

// Zero up to four byte ranges (16-byte aligned, multiples of 16 bytes) in ONE launch: the gradient tables of BertEmbeddings
// (word / position / token-type) before the embedding-gradient kernels add into them -- three fill launches per table set before.
struct ZeroRanges {
    void* p[4];
    long long n16[4];        // 16-byte units
    int n;
};
__global__ __launch_bounds__(256) void zero_ranges_kernel(ZeroRanges z) {
    const uint4 zero = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        if (r < z.n) {
            uint4* q = reinterpret_cast<uint4*>(z.p[r]);
            for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < z.n16[r]; i += (long long)gridDim.x * 256) q[i] = zero;
        }
    }
}

}  // namespace kvq

using namespace kvq;

static thread_local const unsigned long long* g_seed_off = nullptr;   // kvq_set_seed_offset: per calling thread (two engines on two threads do not see each other's)
namespace kvq {
const unsigned long long* seed_offset_ptr() { return g_seed_off; }       // (for the dropout epilogue of csrc/kvq_gemm2.hip)
}

#define DISPATCH_DT(dt, CALL_F32, CALL_BF16) \
    do {                                     \
        if ((dt) == KVQ_F32) { CALL_F32; } else { CALL_BF16; } \
    } while (0)

template <bool BWD>
static int gelu_launch(const void* h, const void* g_a, void* out, int64_t n, int io_dtype, hipStream_t st) {
    const int64_t n8 = n / 8, n4 = n / 4;
    if (n8 > 0) {
        unsigned blocks = (unsigned)((n8 + 255) / 256 > 16384 ? 16384 : (n8 + 255) / 256);
        DISPATCH_DT(io_dtype, hipLaunchKernelGGL((gelu_kernel<KVQ_F32, BWD>), dim3(blocks), dim3(256), 0, st, h, g_a, out, n8),
                    hipLaunchKernelGGL((gelu_kernel<KVQ_BF16, BWD>), dim3(blocks), dim3(256), 0, st, h, g_a, out, n8));
    }
    if (n4 > 2 * n8) {
        DISPATCH_DT(io_dtype, hipLaunchKernelGGL((gelu_tail_kernel<KVQ_F32, BWD>), dim3(1), dim3(64), 0, st, h, g_a, out, 2 * n8, n4),
                    hipLaunchKernelGGL((gelu_tail_kernel<KVQ_BF16, BWD>), dim3(1), dim3(64), 0, st, h, g_a, out, 2 * n8, n4));
    }
    return check_launch("gelu_kernel");
}

extern "C" {

#ifdef KVQ_NN_DIAG
int kvq_nn_diag_set_buffer(void* buf) {       // diagnostic library only: [workgroups of the next attention launch][8] u64, or null
    return hipMemcpyToSymbol(HIP_SYMBOL(kvq::g_nn_diag), &buf, sizeof(buf)) == hipSuccess ? 0 : -1;
}
#endif

static int drln_fwd_impl(const void* y, const void* resid, const float* gamma, const float* beta, int64_t N, int H,
                         float eps, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* out, void* pre,
                         float* mean, float* rstd, unsigned char* out8, float* st8, void* stream);

int kvq_dropout_residual_ln_fwd(const void* y, const void* resid, const float* gamma, const float* beta, int64_t N, int H,
                                float eps, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* out, void* pre,
                                float* mean, float* rstd, void* stream) {
    return drln_fwd_impl(y, resid, gamma, beta, N, H, eps, p_drop, seed, site, io_dtype, out, pre, mean, rstd, nullptr, nullptr, stream);
}

int kvq_dropout_residual_ln_fwd_fp8(const void* y, const void* resid, const float* gamma, const float* beta, int64_t N, int H,
                                    float eps, float p_drop, uint64_t seed, uint32_t site, void* out, void* pre, float* mean, float* rstd,
                                    void* out_fp8, float* fp8_state, void* stream) {
    KVQ_REQUIRE(out_fp8 && fp8_state && ((uintptr_t)out_fp8 & 3) == 0, "kvq_dropout_residual_ln_fwd_fp8: null / misaligned fp8 output or state");
    return drln_fwd_impl(y, resid, gamma, beta, N, H, eps, p_drop, seed, site, KVQ_BF16, out, pre, mean, rstd, (unsigned char*)out_fp8, fp8_state, stream);
}

static int drln_fwd_impl(const void* y, const void* resid, const float* gamma, const float* beta, int64_t N, int H,
                         float eps, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* out, void* pre,
                         float* mean, float* rstd, unsigned char* out8, float* st8, void* stream) {
    KVQ_REQUIRE(y && gamma && beta && out && N > 0 && H > 0, "kvq_dropout_residual_ln_fwd: bad argument");
    KVQ_REQUIRE(H % 4 == 0 && H <= 64 * 4 * LN_MAX_PER_LANE, "kvq_dropout_residual_ln_fwd: H=%d must be a multiple of 4 and <= 4096", H);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    KVQ_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop out of range");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((N + 3) / 4));
    const unsigned th = drop_threshold(p_drop);
#define LAUNCH_LN_FWD(DTV, PERV)                                                                                          \
    hipLaunchKernelGGL((drln_fwd_kernel<DTV, PERV>), grid, dim3(256), 0, st, y, resid, gamma, beta, N, H, eps, p_drop, th, \
                       (unsigned long long)seed, g_seed_off, site, out, pre, mean, rstd, out8, st8)
    if (!resid && H <= 768 && io_dtype == KVQ_BF16) {          // LayerNorm alone (the dense layer's epilogue added the residual)
        hipLaunchKernelGGL((drln_fwd_kernel<KVQ_BF16, 3, false>), grid, dim3(256), 0, st, y, resid, gamma, beta, N, H, eps, p_drop, th,
                           (unsigned long long)seed, g_seed_off, site, out, pre, mean, rstd, out8, st8);
    } else
    if (H <= 768) { DISPATCH_DT(io_dtype, LAUNCH_LN_FWD(KVQ_F32, 3), LAUNCH_LN_FWD(KVQ_BF16, 3)); }
    else if (H <= 1024) { DISPATCH_DT(io_dtype, LAUNCH_LN_FWD(KVQ_F32, 4), LAUNCH_LN_FWD(KVQ_BF16, 4)); }
    else { DISPATCH_DT(io_dtype, LAUNCH_LN_FWD(KVQ_F32, 16), LAUNCH_LN_FWD(KVQ_BF16, 16)); }
#undef LAUNCH_LN_FWD
    return check_launch("drln_fwd_kernel");
}

size_t kvq_ln_bwd_workspace_bytes(int64_t N, int H) {
    const int64_t blocks = (N + LNB_ROWS - 1) / LNB_ROWS;
    return (size_t)blocks * H * 3 * sizeof(float);
}

static int colsum_f32_partials(const float* part, int64_t P, int64_t C, int64_t ldp, void* out, int out_dtype, float scale,
                               int accumulate, hipStream_t st) {
    // the final sum of partial rows is the one-item case of the batched reduction: same kernel, same summation order
    kvq_reduce_item it = {};
    it.src = part; it.dst = out; it.count = P; it.cols = C; it.ld = ldp; it.scale = scale;
    it.src_dtype = KVQ_F32; it.dst_dtype = out_dtype; it.accumulate = accumulate;
    return kvq_reduce_batch(&it, 1, st);
}

int64_t kvq_ln_bwd_partial_rows(int64_t N) { return (N + LNB_ROWS - 1) / LNB_ROWS; }

static int ln_bwd_partial_impl(const void* g_out, const void* pre, const float* mean, const float* rstd,
                               const float* gamma, int64_t N, int H, float p_drop, uint64_t seed, uint32_t site,
                               int io_dtype, void* g_y, void* g_resid, int want_dbias, void* part, size_t part_bytes,
                               void* stream, int drop_on_out) {
    KVQ_REQUIRE(g_out && pre && mean && rstd && gamma && N > 0 && H > 0, "kvq_dropout_residual_ln_bwd: bad argument");
    KVQ_REQUIRE(H % 4 == 0 && H <= 3072, "kvq_dropout_residual_ln_bwd: H=%d unsupported (multiple of 4, <= 3072: 48*H bytes of LDS)", H);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    const size_t need = kvq_ln_bwd_workspace_bytes(N, H);
    if (!part || part_bytes < need) return fail(KVQ_E_WORKSPACE, "kvq_dropout_residual_ln_bwd: workspace %zu < %zu", part_bytes, need);
    KVQ_REQUIRE(!want_dbias || g_y, "kvq_dropout_residual_ln_bwd: the dense-bias partials need g_y");
    hipStream_t st = (hipStream_t)stream;
    const int64_t blocks = (N + LNB_ROWS - 1) / LNB_ROWS;
    float* pdg = (float*)part;
    const size_t lds = (size_t)4 * 3 * H * sizeof(float);
    const unsigned th = drop_threshold(p_drop);
    if (io_dtype == KVQ_BF16 && H % 8 == 0 && H <= 1024 && H >= 64 && LNB_ROWS % 8 == 0 &&
        ((((uintptr_t)g_out | (uintptr_t)pre | (uintptr_t)g_y | (uintptr_t)g_resid | (uintptr_t)gamma) & 15) == 0)) {
#define LAUNCH_LN_BWD16(PERV)                                                                                              \
    hipLaunchKernelGGL((drln_bwd16_kernel<PERV>), dim3((unsigned)blocks), dim3(256), lds, st, g_out, pre, mean, rstd, gamma,  \
                       N, H, p_drop, th, (unsigned long long)seed, g_seed_off, site, g_y, g_resid, pdg, want_dbias ? 1 : 0, drop_on_out)
        if (H <= 256) LAUNCH_LN_BWD16(1);
        else if (H <= 512) LAUNCH_LN_BWD16(2);
        else if (H <= 768) LAUNCH_LN_BWD16(3);
        else LAUNCH_LN_BWD16(4);
#undef LAUNCH_LN_BWD16
        return check_launch("drln_bwd16_kernel");
    }
#define LAUNCH_LN_BWD(DTV, PERV)                                                                                             \
    hipLaunchKernelGGL((drln_bwd_kernel<DTV, PERV>), dim3((unsigned)blocks), dim3(256), lds, st, g_out, pre, mean, rstd, gamma, \
                       N, H, p_drop, th, (unsigned long long)seed, g_seed_off, site, g_y, g_resid, pdg, want_dbias ? 1 : 0, drop_on_out)
    if (H <= 768) { DISPATCH_DT(io_dtype, LAUNCH_LN_BWD(KVQ_F32, 3), LAUNCH_LN_BWD(KVQ_BF16, 3)); }
    else if (H <= 1024) { DISPATCH_DT(io_dtype, LAUNCH_LN_BWD(KVQ_F32, 4), LAUNCH_LN_BWD(KVQ_BF16, 4)); }
    else { DISPATCH_DT(io_dtype, LAUNCH_LN_BWD(KVQ_F32, 16), LAUNCH_LN_BWD(KVQ_BF16, 16)); }
#undef LAUNCH_LN_BWD
    return check_launch("drln_bwd_kernel");
}

int kvq_dropout_residual_ln_bwd_partial(const void* g_out, const void* pre, const float* mean, const float* rstd,
                                        const float* gamma, int64_t N, int H, float p_drop, uint64_t seed, uint32_t site,
                                        int io_dtype, void* g_y, void* g_resid, int want_dbias, void* part, size_t part_bytes,
                                        void* stream) {
    return ln_bwd_partial_impl(g_out, pre, mean, rstd, gamma, N, H, p_drop, seed, site, io_dtype, g_y, g_resid, want_dbias, part,
                               part_bytes, stream, 0);
}

int kvq_ln_dropout_bwd_partial(const void* g_out, const void* pre, const float* mean, const float* rstd, const float* gamma,
                               int64_t N, int H, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* g_y,
                               void* part, size_t part_bytes, void* stream) {
    return ln_bwd_partial_impl(g_out, pre, mean, rstd, gamma, N, H, p_drop, seed, site, io_dtype, g_y, nullptr, 0, part, part_bytes,
                               stream, 1);
}

int kvq_embed_ln_fwd(const int64_t* ids, const void* word, const void* pos, const void* type_row, const float* gamma,
                     const float* beta, int64_t N, int S, int H, int64_t V, float eps, float p_drop, uint64_t seed, uint32_t site,
                     int io_dtype, void* out, void* pre, float* mean, float* rstd, void* stream) {
    KVQ_REQUIRE(ids && word && pos && type_row && gamma && beta && out && N > 0 && S > 0 && H > 0 && V > 0, "kvq_embed_ln_fwd: bad argument");
    KVQ_REQUIRE(H % 4 == 0 && H <= 64 * 4 * LN_MAX_PER_LANE, "kvq_embed_ln_fwd: H=%d must be a multiple of 4 and <= 4096", H);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    KVQ_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop out of range");
    hipStream_t st = (hipStream_t)stream;
    dim3 grid((unsigned)((N + 3) / 4));
    const unsigned th = drop_threshold(p_drop);
#define LAUNCH_EMB(DTV, PERV)                                                                                                  \
    hipLaunchKernelGGL((embed_ln_fwd_kernel<DTV, PERV>), grid, dim3(256), 0, st, ids, word, pos, type_row, gamma, beta, N, S, H, V, \
                       eps, p_drop, th, (unsigned long long)seed, g_seed_off, site, out, pre, mean, rstd)
    if (H <= 768) { DISPATCH_DT(io_dtype, LAUNCH_EMB(KVQ_F32, 3), LAUNCH_EMB(KVQ_BF16, 3)); }
    else if (H <= 1024) { DISPATCH_DT(io_dtype, LAUNCH_EMB(KVQ_F32, 4), LAUNCH_EMB(KVQ_BF16, 4)); }
    else { DISPATCH_DT(io_dtype, LAUNCH_EMB(KVQ_F32, 16), LAUNCH_EMB(KVQ_BF16, 16)); }
#undef LAUNCH_EMB
    return check_launch("embed_ln_fwd_kernel");
}

int kvq_dropout_residual_ln_bwd(const void* g_out, const void* pre, const float* mean, const float* rstd, const float* gamma,
                                int64_t N, int H, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* g_y,
                                void* g_resid, void* g_gamma, void* g_beta, void* g_bias_prev, int param_grad_dtype, int accumulate,
                                void* ws, size_t ws_bytes, void* stream) {
    int rc = kvq_dropout_residual_ln_bwd_partial(g_out, pre, mean, rstd, gamma, N, H, p_drop, seed, site, io_dtype, g_y, g_resid,
                                                 g_bias_prev ? 1 : 0, ws, ws_bytes, stream);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int64_t blocks = (N + LNB_ROWS - 1) / LNB_ROWS;
    float* pdg = (float*)ws;
    const size_t esz = param_grad_dtype == KVQ_F32 ? 4 : 2;
    const int64_t H3 = 3 * (int64_t)H;
    const bool gb_adj = g_gamma && g_beta && (char*)g_beta == (char*)g_gamma + (size_t)H * esz;
    const bool all_adj = gb_adj && g_bias_prev && (char*)g_gamma == (char*)g_bias_prev + (size_t)H * esz;
    if (all_adj) {
        // [dense bias | LN weight | LN bias] are adjacent in the engine's flat layout: ONE pass writes all three
        rc = colsum_f32_partials(pdg, blocks, H3, H3, g_bias_prev, param_grad_dtype, 1.0f, accumulate, st);
        if (rc) return rc;
    } else {
        if (g_bias_prev) { rc = colsum_f32_partials(pdg, blocks, H, H3, g_bias_prev, param_grad_dtype, 1.0f, accumulate, st); if (rc) return rc; }
        if (gb_adj) {
            rc = colsum_f32_partials(pdg + H, blocks, 2 * (int64_t)H, H3, g_gamma, param_grad_dtype, 1.0f, accumulate, st);
            if (rc) return rc;
        } else {
            if (g_gamma) { rc = colsum_f32_partials(pdg + H, blocks, H, H3, g_gamma, param_grad_dtype, 1.0f, accumulate, st); if (rc) return rc; }
            if (g_beta) { rc = colsum_f32_partials(pdg + 2 * H, blocks, H, H3, g_beta, param_grad_dtype, 1.0f, accumulate, st); if (rc) return rc; }
        }
    }
    return KVQ_OK;
}

size_t kvq_colsum_workspace_bytes(int64_t N, int64_t C) { return (size_t)((N + CS_ROWS - 1) / CS_ROWS) * C * sizeof(float); }

int64_t kvq_colsum_partial_rows(int64_t N) { return (N + CS_ROWS - 1) / CS_ROWS; }

int kvq_colsum_partial(const void* x, int64_t N, int64_t C, int64_t ld, int in_dtype, void* part, size_t part_bytes, void* stream) {
    KVQ_REQUIRE(x && N > 0 && C > 0 && ld >= C, "kvq_colsum: bad argument");
    KVQ_REQUIRE(in_dtype == KVQ_F32 || in_dtype == KVQ_BF16, "unsupported io dtype %d", in_dtype);
    const size_t need = kvq_colsum_workspace_bytes(N, C);
    if (!part || part_bytes < need) return fail(KVQ_E_WORKSPACE, "kvq_colsum: workspace %zu < %zu", part_bytes, need);
    hipStream_t st = (hipStream_t)stream;
    const int64_t P = (N + CS_ROWS - 1) / CS_ROWS;
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)P);
    DISPATCH_DT(in_dtype, hipLaunchKernelGGL(colsum_partial_kernel<KVQ_F32>, grid, dim3(256), 0, st, x, N, C, ld, (float*)part, CS_ROWS, C),
                hipLaunchKernelGGL(colsum_partial_kernel<KVQ_BF16>, grid, dim3(256), 0, st, x, N, C, ld, (float*)part, CS_ROWS, C));
    return check_launch("colsum_partial_kernel");
}

int kvq_colsum(const void* x, int64_t N, int64_t C, int64_t ld, int in_dtype, void* out, int out_dtype, float scale,
               int accumulate, void* ws, size_t ws_bytes, void* stream) {
    KVQ_REQUIRE(out, "kvq_colsum: bad argument");
    int rc = kvq_colsum_partial(x, N, C, ld, in_dtype, ws, ws_bytes, stream);
    if (rc) return rc;
    return colsum_f32_partials((const float*)ws, (N + CS_ROWS - 1) / CS_ROWS, C, C, out, out_dtype, scale, accumulate, (hipStream_t)stream);
}

int kvq_reduce_batch(const kvq_reduce_item* items, int n, void* stream) {
    KVQ_REQUIRE(items && n >= 1 && n <= KVQ_REDUCE_MAX_ITEMS, "kvq_reduce_batch: 1..%d items per call", KVQ_REDUCE_MAX_ITEMS);
    ReduceBatch rb;
    rb.n = n;
    int64_t blocks = 0;
    for (int i = 0; i < n; ++i) {
        const kvq_reduce_item& d = items[i];
        KVQ_REQUIRE(d.src && d.dst && d.count >= 1 && d.cols >= 1 && d.ld >= d.cols, "kvq_reduce_batch: item %d malformed", i);
        KVQ_REQUIRE((d.src_dtype == KVQ_F32 || d.src_dtype == KVQ_BF16) && (d.dst_dtype == KVQ_F32 || d.dst_dtype == KVQ_BF16),
                    "kvq_reduce_batch: item %d: unsupported dtype", i);
        rb.it[i] = d;
        rb.first_block[i] = (int)blocks;
        blocks += reduce_is_slab(d) ? (d.cols + RB_SLAB_COLS - 1) / RB_SLAB_COLS : (d.cols + 63) / 64;
        KVQ_REQUIRE(blocks < (1 << 30), "kvq_reduce_batch: too many workgroups");
    }
    rb.first_block[n] = (int)blocks;
    hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)blocks), dim3(1024), 0, (hipStream_t)stream, rb);
    return check_launch("reduce_batch_kernel");
}

int kvq_sum_slabs(const void* part, int S, int64_t n, int io_dtype, void* out, void* stream) {
    KVQ_REQUIRE(part && out && S >= 1 && n > 0 && n % 4 == 0, "kvq_sum_slabs: bad argument (n %% 4 == 0 required)");
    const int64_t n4 = n / 4;
    unsigned blocks = (unsigned)((n4 + 255) / 256 > 8192 ? 8192 : (n4 + 255) / 256);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(io_dtype, hipLaunchKernelGGL(sum_slabs_kernel<KVQ_F32>, dim3(blocks), dim3(256), 0, st, part, S, n4, out),
                hipLaunchKernelGGL(sum_slabs_kernel<KVQ_BF16>, dim3(blocks), dim3(256), 0, st, part, S, n4, out));
    return check_launch("sum_slabs_kernel");
}

int kvq_gelu_fwd(const void* h, void* a, int64_t n, int io_dtype, void* stream) {
    KVQ_REQUIRE(h && a && n > 0 && n % 4 == 0, "kvq_gelu_fwd: bad argument (n %% 4 == 0 required)");
    KVQ_REQUIRE(((uintptr_t)h | (uintptr_t)a) % 16 == 0, "kvq_gelu_fwd: 16-byte aligned buffers required");
    return gelu_launch<false>(h, nullptr, a, n, io_dtype, (hipStream_t)stream);
}

int64_t kvq_gelu_bwd_partial_rows(int64_t N) { return (N + GB_ROWS - 1) / GB_ROWS; }

int kvq_gelu_bwd_bias(const void* h, const void* g_a, void* g_h, int64_t N, int64_t C, int io_dtype, float* bias_part,
                      size_t part_bytes, void* stream) {
    KVQ_REQUIRE(h && g_a && g_h && bias_part && N > 0 && C > 0 && C % 8 == 0, "kvq_gelu_bwd_bias: bad argument (C %% 8 == 0 required)");
    KVQ_REQUIRE(((uintptr_t)h | (uintptr_t)g_a | (uintptr_t)g_h | (uintptr_t)bias_part) % 16 == 0, "kvq_gelu_bwd_bias: 16-byte aligned buffers required");
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    const int64_t P = (N + GB_ROWS - 1) / GB_ROWS;
    if (part_bytes < (size_t)P * C * sizeof(float)) return fail(KVQ_E_WORKSPACE, "kvq_gelu_bwd_bias: partial buffer %zu < %zu", part_bytes, (size_t)P * C * sizeof(float));
    KVQ_REQUIRE(P <= 65535, "kvq_gelu_bwd_bias: N too large");
    dim3 grid((unsigned)((C + 1023) / 1024), (unsigned)P);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(io_dtype, hipLaunchKernelGGL(gelu_bwd_bias_kernel<KVQ_F32>, grid, dim3(128), 0, st, h, g_a, g_h, N, C, bias_part),
                hipLaunchKernelGGL(gelu_bwd_bias_kernel<KVQ_BF16>, grid, dim3(128), 0, st, h, g_a, g_h, N, C, bias_part));
    return check_launch("gelu_bwd_bias_kernel");
}

int kvq_gelu_bwd(const void* h, const void* g_a, void* g_h, int64_t n, int io_dtype, void* stream) {
    KVQ_REQUIRE(h && g_a && g_h && n > 0 && n % 4 == 0, "kvq_gelu_bwd: bad argument (n %% 4 == 0 required)");
    KVQ_REQUIRE(((uintptr_t)h | (uintptr_t)g_a | (uintptr_t)g_h) % 16 == 0, "kvq_gelu_bwd: 16-byte aligned buffers required");
    return gelu_launch<true>(h, g_a, g_h, n, io_dtype, (hipStream_t)stream);
}

static int adam_launch(float* p, const void* g, float* m, float* v, float* vmax, void* shadow_bf16, int64_t n, int grad_dtype,
                       float lr, float beta1, float beta2, float eps, float weight_decay, float bc1, float bc2s, float grad_scale,
                       const float* hyper, void* stream, AdamFp8 f8 = AdamFp8{}) {
    KVQ_REQUIRE(p && g && m && v && n > 0, "kvq_adam_step: bad argument");
    KVQ_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)vmax | (uintptr_t)shadow_bf16) & 15) == 0,
                "kvq_adam_step: 16-byte aligned buffers required");
    const int64_t n4 = n / 4;
    const int n_tail = (int)(n - 4 * n4);
    const int64_t n8 = (n4 + 1) / 2;
    unsigned blocks = (unsigned)((n8 + 255) / 256 > 32768 ? 32768 : (n8 + 255) / 256);
    if (blocks == 0) blocks = 1;
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(grad_dtype,
                hipLaunchKernelGGL(adam_kernel<KVQ_F32>, dim3(blocks), dim3(256), 0, st, p, g, m, v, vmax, (unsigned short*)shadow_bf16, n4,
                                   lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, hyper, n_tail, f8),
                hipLaunchKernelGGL(adam_kernel<KVQ_BF16>, dim3(blocks), dim3(256), 0, st, p, g, m, v, vmax, (unsigned short*)shadow_bf16, n4,
                                   lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, hyper, n_tail, f8));
    return check_launch("adam_kernel");
}

int kvq_adam_step(float* p, const void* g, float* m, float* v, float* vmax, void* shadow_bf16, int64_t n, int grad_dtype,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                  void* stream) {
    KVQ_REQUIRE(step >= 1, "kvq_adam_step: step >= 1");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
    return adam_launch(p, g, m, v, vmax, shadow_bf16, n, grad_dtype, lr, beta1, beta2, eps, weight_decay, bc1, bc2s, grad_scale, nullptr, stream);
}

int kvq_adam_step_dev(float* p, const void* g, float* m, float* v, float* vmax, void* shadow_bf16, int64_t n, int grad_dtype,
                      const void* step_state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                      void* stream) {
    KVQ_REQUIRE(step_state, "kvq_adam_step_dev: null step state");
    return adam_launch(p, g, m, v, vmax, shadow_bf16, n, grad_dtype, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale,
                       reinterpret_cast<const float*>(reinterpret_cast<const char*>(step_state) + 8), stream);
}

int kvq_adam_step_dev_fp8(float* p, const void* g, float* m, float* v, float* vmax, void* shadow_bf16, int64_t n, int grad_dtype,
                          const void* step_state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                          void* w8_mirror, const int* span_segment, const float* seg_scale, const int64_t* seg_off, const int64_t* seg_n,
                          int nseg, int64_t first_element, void* stream) {
    KVQ_REQUIRE(step_state && shadow_bf16, "kvq_adam_step_dev_fp8: step state and bf16 shadow required");
    KVQ_REQUIRE(w8_mirror && span_segment && seg_scale && seg_off && seg_n && nseg > 0, "kvq_adam_step_dev_fp8: null fp8 argument");
    KVQ_REQUIRE(first_element >= 0 && first_element % 8 == 0 && n % 8 == 0 && ((uintptr_t)w8_mirror & 7) == 0,
                "kvq_adam_step_dev_fp8: the range must start and end at multiples of 8 elements");
    AdamFp8 f8;
    f8.w8 = (unsigned char*)w8_mirror; f8.gseg = span_segment; f8.scale = seg_scale; f8.seg_off = seg_off; f8.seg_n = seg_n; f8.nseg = nseg;
    f8.e_base = first_element;
    return adam_launch(p, g, m, v, vmax, shadow_bf16, n, grad_dtype, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale,
                       reinterpret_cast<const float*>(reinterpret_cast<const char*>(step_state) + 8), stream, f8);
}

static int step_state_launch(void* step_state, float lr0, float gamma, const int64_t* milestones, int n_milestones, float beta1,
                             float beta2, int phase, void* stream, const char* who) {
    KVQ_REQUIRE(step_state && n_milestones >= 0 && n_milestones <= 8 && (n_milestones == 0 || milestones),
                "%s: bad argument (at most 8 milestones)", who);
    Milestones ms = {};
    ms.n = n_milestones;
    for (int i = 0; i < n_milestones; ++i) ms.at[i] = milestones[i];
    hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (StepState*)step_state, lr0, gamma, ms, beta1,
                       beta2, phase);
    return check_launch("step_state_advance_kernel");
}
int kvq_step_state_advance(void* step_state, float lr0, float gamma, const int64_t* milestones, int n_milestones, float beta1,
                           float beta2, void* stream) {
    return step_state_launch(step_state, lr0, gamma, milestones, n_milestones, beta1, beta2, 3, stream, "kvq_step_state_advance");
}
int kvq_step_state_prepare(void* step_state, float lr0, float gamma, const int64_t* milestones, int n_milestones, float beta1,
                           float beta2, void* stream) {
    return step_state_launch(step_state, lr0, gamma, milestones, n_milestones, beta1, beta2, 1, stream, "kvq_step_state_prepare");
}
int kvq_step_state_commit(void* step_state, void* stream) {
    return step_state_launch(step_state, 0.f, 0.f, nullptr, 0, 0.f, 0.f, 2, stream, "kvq_step_state_commit");
}

size_t kvq_embed_grad_workspace_bytes(int64_t N, int H) {
    const int64_t nb = (N + EG_R - 1) / EG_R;
    return (size_t)nb * 2 * H * sizeof(float) + (size_t)nb * sizeof(int);
}

int kvq_embed_grad(const void* g, const int64_t* perm, const int64_t* sorted_ids, int64_t N, int H, int64_t V, int g_dtype,
                   void* gW, int w_dtype, int accumulate, void* ws, size_t ws_bytes, void* stream) {
    KVQ_REQUIRE(g && perm && sorted_ids && gW && N > 0 && V > 0, "kvq_embed_grad: bad argument");
    KVQ_REQUIRE(H > 0 && H % 4 == 0 && H <= 1024, "kvq_embed_grad: H=%d unsupported (multiple of 4, <= 1024)", H);
    KVQ_REQUIRE((g_dtype == KVQ_F32 || g_dtype == KVQ_BF16) && (w_dtype == KVQ_F32 || w_dtype == KVQ_BF16), "unsupported dtype");
    const size_t need = kvq_embed_grad_workspace_bytes(N, H);
    if (!ws || ws_bytes < need) return fail(KVQ_E_WORKSPACE, "kvq_embed_grad: workspace %zu < %zu", ws_bytes, need);
    const int64_t nb = (N + EG_R - 1) / EG_R;
    float* edge = (float*)ws;
    int* meta = (int*)((char*)ws + (size_t)nb * 2 * H * sizeof(float));
    hipStream_t st = (hipStream_t)stream;
    const int per = (H / 4 + WAVE - 1) / WAVE;               // 1..4
#define EG_LAUNCH(DG, DW, P)                                                                                                   \
    do {                                                                                                                       \
        hipLaunchKernelGGL((embed_grad_runs_kernel<DG, DW, P>), dim3((unsigned)nb), dim3(64), 0, st, g, perm, sorted_ids, N, H, V, \
                           gW, accumulate, edge, meta);                                                                        \
        hipLaunchKernelGGL((embed_grad_edges_kernel<DW, P>), dim3((unsigned)nb), dim3(1024), 0, st, sorted_ids, N, H, V, gW,    \
                           accumulate, (const float*)edge, (const int*)meta, nb);                                             \
    } while (0)
#define EG_PER(DG, DW)                                                                       \
    do {                                                                                     \
        if (per == 1) EG_LAUNCH(DG, DW, 1); else if (per == 2) EG_LAUNCH(DG, DW, 2);         \
        else if (per == 3) EG_LAUNCH(DG, DW, 3); else EG_LAUNCH(DG, DW, 4);                  \
    } while (0)
    if (g_dtype == KVQ_F32 && w_dtype == KVQ_F32) EG_PER(KVQ_F32, KVQ_F32);
    else if (g_dtype == KVQ_BF16 && w_dtype == KVQ_BF16) EG_PER(KVQ_BF16, KVQ_BF16);
    else if (g_dtype == KVQ_BF16 && w_dtype == KVQ_F32) EG_PER(KVQ_BF16, KVQ_F32);
    else EG_PER(KVQ_F32, KVQ_BF16);
#undef EG_PER
#undef EG_LAUNCH
    return check_launch("embed_grad_kernel");
}

#define GUMBEL_DISPATCH(KERNEL, ...)                                                                                       \
    do {                                                                                                                   \
        const int per = (K + WAVE - 1) / WAVE;                                                                             \
        dim3 grid((unsigned)((N + 3) / 4));                                                                                \
        if (io_dtype == KVQ_F32) {                                                                                         \
            if (per <= 1) hipLaunchKernelGGL((KERNEL<KVQ_F32, 1>), grid, dim3(256), 0, st, __VA_ARGS__);                   \
            else if (per <= 2) hipLaunchKernelGGL((KERNEL<KVQ_F32, 2>), grid, dim3(256), 0, st, __VA_ARGS__);              \
            else if (per <= 4) hipLaunchKernelGGL((KERNEL<KVQ_F32, 4>), grid, dim3(256), 0, st, __VA_ARGS__);              \
            else if (per <= 8) hipLaunchKernelGGL((KERNEL<KVQ_F32, 8>), grid, dim3(256), 0, st, __VA_ARGS__);              \
            else hipLaunchKernelGGL((KERNEL<KVQ_F32, 16>), grid, dim3(256), 0, st, __VA_ARGS__);                           \
        } else {                                                                                                           \
            if (per <= 1) hipLaunchKernelGGL((KERNEL<KVQ_BF16, 1>), grid, dim3(256), 0, st, __VA_ARGS__);                  \
            else if (per <= 2) hipLaunchKernelGGL((KERNEL<KVQ_BF16, 2>), grid, dim3(256), 0, st, __VA_ARGS__);             \
            else if (per <= 4) hipLaunchKernelGGL((KERNEL<KVQ_BF16, 4>), grid, dim3(256), 0, st, __VA_ARGS__);             \
            else if (per <= 8) hipLaunchKernelGGL((KERNEL<KVQ_BF16, 8>), grid, dim3(256), 0, st, __VA_ARGS__);             \
            else hipLaunchKernelGGL((KERNEL<KVQ_BF16, 16>), grid, dim3(256), 0, st, __VA_ARGS__);                          \
        }                                                                                                                  \
    } while (0)

int kvq_gumbel_forward(const void* logits, const float* noise, int64_t N, int K, float tau, int hard, uint64_t seed, uint32_t site,
                       int io_dtype, void* y, float* y_soft, int64_t* ind, float* kl_row, void* stream) {
    KVQ_REQUIRE(logits && y && ind && kl_row && N > 0, "kvq_gumbel_forward: bad argument");
    KVQ_REQUIRE(K >= 1 && K <= 1024, "kvq_gumbel_forward: K=%d unsupported (1..1024 codes)", K);
    KVQ_REQUIRE(tau > 0.f, "kvq_gumbel_forward: temperature must be positive");
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    hipStream_t st = (hipStream_t)stream;
    GUMBEL_DISPATCH(gumbel_fwd_kernel, logits, noise, N, K, 1.0f / tau, hard, (unsigned long long)seed, g_seed_off, site, y, y_soft, ind, kl_row);
    return check_launch("gumbel_fwd_kernel");
}

int kvq_gumbel_backward(const void* logits, const float* y_soft, const void* g_y, const float* g_diff, int64_t N, int K, float tau,
                        float kld_scale, int io_dtype, void* g_logits, void* stream) {
    KVQ_REQUIRE(logits && y_soft && g_logits && N > 0, "kvq_gumbel_backward: bad argument");
    KVQ_REQUIRE(K >= 1 && K <= 1024, "kvq_gumbel_backward: K=%d unsupported (1..1024 codes)", K);
    KVQ_REQUIRE(tau > 0.f, "kvq_gumbel_backward: temperature must be positive");
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    hipStream_t st = (hipStream_t)stream;
    GUMBEL_DISPATCH(gumbel_bwd_kernel, logits, y_soft, g_y, g_diff, N, K, 1.0f / tau, kld_scale, g_logits);
    return check_launch("gumbel_bwd_kernel");
}
#undef GUMBEL_DISPATCH

int kvq_set_seed_offset(const void* step_state) {
    g_seed_off = reinterpret_cast<const unsigned long long*>(step_state);
    return KVQ_OK;
}

int kvq_zero_ranges(void* const* ptrs, const int64_t* bytes, int n, void* stream) {
    KVQ_REQUIRE(ptrs && bytes && n >= 1 && n <= 4, "kvq_zero_ranges: 1..4 ranges");
    ZeroRanges z = {};
    z.n = n;
    long long most = 0;
    for (int i = 0; i < n; ++i) {
        KVQ_REQUIRE(ptrs[i] && bytes[i] > 0 && bytes[i] % 16 == 0 && ((uintptr_t)ptrs[i] & 15) == 0, "kvq_zero_ranges: range %d: 16-byte aligned pointer and size", i);
        z.p[i] = ptrs[i];
        z.n16[i] = bytes[i] / 16;
        most = z.n16[i] > most ? z.n16[i] : most;
    }
    const long long want = (most + 255) / 256;
    const unsigned blocks = (unsigned)(want > 4096 ? 4096 : want);
    hipLaunchKernelGGL(zero_ranges_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, z);
    return check_launch("zero_ranges_kernel");
}

int kvq_dropout(const void* x, int64_t n, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* out, void* stream) {
    KVQ_REQUIRE(x && out && n > 0 && n % 4 == 0, "kvq_dropout: bad argument (n %% 4 == 0 required)");
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    KVQ_REQUIRE(p_drop >= 0.f && p_drop < 1.f, "p_drop out of range");
    const int64_t n4 = n / 4;
    unsigned blocks = (unsigned)((n4 + 255) / 256 > 16384 ? 16384 : (n4 + 255) / 256);
    const unsigned th = drop_threshold(p_drop);
    hipStream_t st = (hipStream_t)stream;
    DISPATCH_DT(io_dtype, hipLaunchKernelGGL(dropout_kernel<KVQ_F32>, dim3(blocks), dim3(256), 0, st, x, out, n4, p_drop, th, (unsigned long long)seed, g_seed_off, site),
                hipLaunchKernelGGL(dropout_kernel<KVQ_BF16>, dim3(blocks), dim3(256), 0, st, x, out, n4, p_drop, th, (unsigned long long)seed, g_seed_off, site));
    return check_launch("dropout_kernel");
}

static thread_local int g_attn_variant = 2;   // per calling thread; bf16 io: 2 = MFMA kernels, 1 = packed-dot kernels (v_dot2c_f32_bf16), 0 = convert-and-fma kernels

static bool attn_coalesced() {       // KVQ_ATTN_COAL=0: A/B switch (read once): row loads of the MFMA attention kernels as row chunks per lane
    static const bool on = [] { const char* e = getenv("KVQ_ATTN_COAL"); return !(e && e[0] == '0'); }();
    return on;
}

static bool attn_store_coalesced() { // KVQ_ATTN_STC=0: A/B switch: the gradients / context rows as 16-byte pieces per lane (store_ct)
    static const bool on = [] { const char* e = getenv("KVQ_ATTN_STC"); return !(e && e[0] == '0'); }();
    return on;
}

int kvq_attn_set_variant(int variant) {
    KVQ_REQUIRE(variant >= 0 && variant <= 2, "kvq_attn_set_variant: variant %d unknown (0 fma, 1 dot2, 2 mfma)", variant);
    g_attn_variant = variant;
    return KVQ_OK;
}

static int attn_check(int B, int nh, int Sq, int Sk, int dh, int io_dtype) {
    KVQ_REQUIRE(B > 0 && nh > 0 && Sq > 0 && Sk > 0, "kvq_attn: sizes must be positive");
    KVQ_REQUIRE(Sq <= AT_SMAX && Sk <= AT_SMAX, "kvq_attn: sequence lengths (%d, %d) above the %d-token kernel limit", Sq, Sk, AT_SMAX);
    KVQ_REQUIRE(dh == AT_D, "kvq_attn: head dim %d unsupported (64 only)", dh);
    KVQ_REQUIRE(io_dtype == KVQ_F32 || io_dtype == KVQ_BF16, "unsupported io dtype %d", io_dtype);
    KVQ_REQUIRE((Sq <= AT_S && Sk <= AT_S) || io_dtype == KVQ_BF16, "kvq_attn: sequences above %d tokens are bf16 only (%d, %d)", AT_S, Sq, Sk);
    return KVQ_OK;
}
static bool attn_long(int Sq, int Sk) { return Sq > AT_S || Sk > AT_S; }

static int attn_fwd_impl(const void* q, const void* k, const void* v, const int64_t* mask, int B, int nh, int Sq, int Sk, int dh,
                         int ldq, int ldk, int ldv, int ldo, int causal, float scale, float p_drop, uint64_t seed, uint32_t site,
                         int io_dtype, void* out, float* lse, unsigned char* out8, int ld8, float* st8, void* stream);

int kvq_attn_fwd(const void* q, const void* k, const void* v, const int64_t* mask, int B, int nh, int Sq, int Sk, int dh,
                 int ldq, int ldk, int ldv, int ldo, int causal, float scale, float p_drop, uint64_t seed, uint32_t site,
                 int io_dtype, void* out, float* lse, void* stream) {
    return attn_fwd_impl(q, k, v, mask, B, nh, Sq, Sk, dh, ldq, ldk, ldv, ldo, causal, scale, p_drop, seed, site, io_dtype, out, lse,
                         nullptr, 0, nullptr, stream);
}

// 1 when kvq_attn_fwd_fp8 can run for these sizes (the MFMA kernel with whole-line stores: at most 32 tokens, default variant)
int kvq_attn_fwd_fp8_ok(int Sq, int Sk) { return !attn_long(Sq, Sk) && g_attn_variant == 2 && attn_coalesced() && attn_store_coalesced() ? 1 : 0; }

int kvq_attn_fwd_fp8(const void* q, const void* k, const void* v, const int64_t* mask, int B, int nh, int Sq, int Sk, int dh,
                     int ldq, int ldk, int ldv, int ldo, int causal, float scale, float p_drop, uint64_t seed, uint32_t site,
                     void* out, float* lse, void* out_fp8, int ld8, float* fp8_state, void* stream) {
    KVQ_REQUIRE(out_fp8 && fp8_state && ld8 >= nh * dh && ld8 % 8 == 0 && ((uintptr_t)out_fp8 & 7) == 0,
                "kvq_attn_fwd_fp8: fp8 output (8-byte aligned, row stride >= nh * dh, a multiple of 8) and state required");
    KVQ_REQUIRE(kvq_attn_fwd_fp8_ok(Sq, Sk), "kvq_attn_fwd_fp8: only the 32-token MFMA kernel emits the fp8 copy (kvq_attn_fwd_fp8_ok)");
    return attn_fwd_impl(q, k, v, mask, B, nh, Sq, Sk, dh, ldq, ldk, ldv, ldo, causal, scale, p_drop, seed, site, KVQ_BF16, out, lse,
                         (unsigned char*)out_fp8, ld8, fp8_state, stream);
}

static int attn_fwd_impl(const void* q, const void* k, const void* v, const int64_t* mask, int B, int nh, int Sq, int Sk, int dh,
                         int ldq, int ldk, int ldv, int ldo, int causal, float scale, float p_drop, uint64_t seed, uint32_t site,
                         int io_dtype, void* out, float* lse, unsigned char* out8, int ld8, float* st8, void* stream) {
    KVQ_REQUIRE(q && k && v && out, "kvq_attn_fwd: null pointer argument");
    int rc = attn_check(B, nh, Sq, Sk, dh, io_dtype);
    if (rc) return rc;
    AttnParams p = {};
    p.out8 = out8; p.ld8 = ld8; p.st8 = st8;
    p.q = q; p.k = k; p.v = v; p.out = out; p.lse = lse; p.mask = mask;
    p.B = B; p.nh = nh; p.Sq = Sq; p.Sk = Sk; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.causal = causal;
    p.scale = scale; p.p_drop = p_drop; p.thresh = drop_threshold(p_drop); p.seed = seed; p.seed_off = g_seed_off; p.site = site;
    hipStream_t st = (hipStream_t)stream;
    const bool al = (ldq % 8 == 0) && (ldk % 8 == 0) && (ldv % 8 == 0) && (ldo % 8 == 0) && (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)out) % 16 == 0);
    KVQ_REQUIRE(!out8 || al, "kvq_attn_fwd_fp8: 16-byte aligned rows required");
    if (attn_long(Sq, Sk)) {
        KVQ_REQUIRE(al, "kvq_attn_fwd: sequences above %d tokens need 16-byte aligned rows", AT_S);
        KVQ_REQUIRE(!causal || Sq == Sk, "kvq_attn_fwd: causal attention needs Sq == Sk");
        hipLaunchKernelGGL(attn_fwd_blk_kernel, dim3((unsigned)(B * nh * ((Sq + 31) / 32))), dim3(64), 0, st, p);
        return check_launch("attn_fwd_blk_kernel");
    }
    if (io_dtype == KVQ_F32) hipLaunchKernelGGL(attn_fwd_kernel<KVQ_F32>, dim3((unsigned)(B * nh)), dim3(64), 0, st, p);
    else if (al && g_attn_variant == 2) {
        const dim3 grid((unsigned)(B * nh));
        if (attn_coalesced() && attn_store_coalesced()) hipLaunchKernelGGL((attn_fwd_mfma_kernel<true, true>), grid, dim3(64), 0, st, p);
        else if (attn_coalesced()) hipLaunchKernelGGL((attn_fwd_mfma_kernel<true, false>), grid, dim3(64), 0, st, p);
        else hipLaunchKernelGGL((attn_fwd_mfma_kernel<false, false>), grid, dim3(64), 0, st, p);
    }
    else if (al && g_attn_variant == 1) hipLaunchKernelGGL(attn_fwd_bf16_kernel, dim3((unsigned)(B * nh)), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(attn_fwd_kernel<KVQ_BF16>, dim3((unsigned)(B * nh)), dim3(64), 0, st, p);
    return check_launch("attn_fwd_kernel");
}

// per-batch column sums of a [B*S, ld] gradient (head columns 0 .. nh*64): the fallback of the kernels that do not emit them
static int attn_bias_partials(const void* g, int B, int S, int C, int ld, int io_dtype, float* part, int ldp, hipStream_t st) {
    dim3 grid((unsigned)((C + 255) / 256), (unsigned)B);
    DISPATCH_DT(io_dtype, hipLaunchKernelGGL(colsum_partial_kernel<KVQ_F32>, grid, dim3(256), 0, st, g, (int64_t)B * S, C, ld, part, S, ldp),
                hipLaunchKernelGGL(colsum_partial_kernel<KVQ_BF16>, grid, dim3(256), 0, st, g, (int64_t)B * S, C, ld, part, S, ldp));
    return check_launch("colsum_partial_kernel");
}

static int attn_bwd_impl(const void* q, const void* k, const void* v, const int64_t* mask, const void* out, const float* lse,
                         const void* g_out, int B, int nh, int Sq, int Sk, int dh, int ldq, int ldk, int ldv, int ldo, int causal,
                         float scale, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* g_q, void* g_k, void* g_v,
                         float* bias_part_q, float* bias_part_k, float* bias_part_v, int ldp_q, int ldp_kv, void* stream) {
    KVQ_REQUIRE(q && k && v && g_out && g_q && g_k && g_v, "kvq_attn_bwd: null pointer argument");
    int rc = attn_check(B, nh, Sq, Sk, dh, io_dtype);
    if (rc) return rc;
    AttnParams p = {};
    p.q = q; p.k = k; p.v = v; p.mask = mask; p.g_out = g_out; p.g_q = g_q; p.g_k = g_k; p.g_v = g_v;
    KVQ_REQUIRE((!bias_part_q || ldp_q >= nh * AT_D) && ((!bias_part_k && !bias_part_v) || ldp_kv >= nh * AT_D),
                "kvq_attn_bwd: bias partial row strides must cover nh*64 columns");
    p.pb_q = bias_part_q; p.pb_k = bias_part_k; p.pb_v = bias_part_v; p.ldp_q = ldp_q; p.ldp_kv = ldp_kv;
    p.B = B; p.nh = nh; p.Sq = Sq; p.Sk = Sk; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.causal = causal;
    p.scale = scale; p.p_drop = p_drop; p.thresh = drop_threshold(p_drop); p.seed = seed; p.seed_off = g_seed_off; p.site = site;
    hipStream_t st = (hipStream_t)stream;
    const bool al = (ldq % 8 == 0) && (ldk % 8 == 0) && (ldv % 8 == 0) && (ldo % 8 == 0) &&
                    (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)g_out | (uintptr_t)g_q | (uintptr_t)g_k | (uintptr_t)g_v) % 16 == 0);
    bool emits_partials = false;
    if (attn_long(Sq, Sk)) {
        KVQ_REQUIRE(out && lse, "kvq_attn_bwd: sequences above %d tokens need the forward's output and log-sum-exp (kvq_attn_bwd_saved)", AT_S);
        KVQ_REQUIRE(al && ((uintptr_t)out % 16 == 0), "kvq_attn_bwd: sequences above %d tokens need 16-byte aligned rows", AT_S);
        KVQ_REQUIRE(!causal || Sq == Sk, "kvq_attn_bwd: causal attention needs Sq == Sk");
        p.out = const_cast<void*>(out); p.lse = const_cast<float*>(lse);
        hipLaunchKernelGGL(attn_bwd_blk_dq_kernel, dim3((unsigned)(B * nh * ((Sq + 31) / 32))), dim3(64), 0, st, p);
        rc = check_launch("attn_bwd_blk_dq_kernel");
        if (rc) return rc;
        hipLaunchKernelGGL(attn_bwd_blk_dkv_kernel, dim3((unsigned)(B * nh * ((Sk + 31) / 32))), dim3(64), 0, st, p);
    } else
    if (io_dtype == KVQ_F32) hipLaunchKernelGGL(attn_bwd_kernel<KVQ_F32>, dim3((unsigned)(B * nh)), dim3(64), 0, st, p);
    else if (al && g_attn_variant == 2) {
        const dim3 grid((unsigned)(B * nh));
        if (attn_coalesced() && attn_store_coalesced()) hipLaunchKernelGGL((attn_bwd_mfma_kernel<true, true>), grid, dim3(64), 0, st, p);
        else if (attn_coalesced()) hipLaunchKernelGGL((attn_bwd_mfma_kernel<true, false>), grid, dim3(64), 0, st, p);
        else hipLaunchKernelGGL((attn_bwd_mfma_kernel<false, false>), grid, dim3(64), 0, st, p);
        emits_partials = true;
    }
    else if (al && g_attn_variant == 1) hipLaunchKernelGGL(attn_bwd_bf16_kernel, dim3((unsigned)(B * nh)), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(attn_bwd_kernel<KVQ_BF16>, dim3((unsigned)(B * nh)), dim3(64), 0, st, p);
    rc = check_launch("attn_bwd_kernel");
    if (rc || emits_partials) return rc;
    if (bias_part_q) { rc = attn_bias_partials(g_q, B, Sq, nh * AT_D, ldq, io_dtype, bias_part_q, ldp_q, st); if (rc) return rc; }
    if (bias_part_k) { rc = attn_bias_partials(g_k, B, Sk, nh * AT_D, ldk, io_dtype, bias_part_k, ldp_kv, st); if (rc) return rc; }
    if (bias_part_v) { rc = attn_bias_partials(g_v, B, Sk, nh * AT_D, ldv, io_dtype, bias_part_v, ldp_kv, st); if (rc) return rc; }
    return KVQ_OK;
}

int kvq_attn_bwd(const void* q, const void* k, const void* v, const int64_t* mask, const void* g_out, int B, int nh, int Sq,
                 int Sk, int dh, int ldq, int ldk, int ldv, int ldo, int causal, float scale, float p_drop, uint64_t seed,
                 uint32_t site, int io_dtype, void* g_q, void* g_k, void* g_v, float* bias_part_q, float* bias_part_k,
                 float* bias_part_v, int ldp_q, int ldp_kv, void* stream) {
    return attn_bwd_impl(q, k, v, mask, nullptr, nullptr, g_out, B, nh, Sq, Sk, dh, ldq, ldk, ldv, ldo, causal, scale, p_drop, seed, site,
                         io_dtype, g_q, g_k, g_v, bias_part_q, bias_part_k, bias_part_v, ldp_q, ldp_kv, stream);
}

int kvq_attn_bwd_saved(const void* q, const void* k, const void* v, const int64_t* mask, const void* out, const float* lse,
                       const void* g_out, int B, int nh, int Sq, int Sk, int dh, int ldq, int ldk, int ldv, int ldo, int causal,
                       float scale, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* g_q, void* g_k, void* g_v,
                       float* bias_part_q, float* bias_part_k, float* bias_part_v, int ldp_q, int ldp_kv, void* stream) {
    return attn_bwd_impl(q, k, v, mask, out, lse, g_out, B, nh, Sq, Sk, dh, ldq, ldk, ldv, ldo, causal, scale, p_drop, seed, site, io_dtype,
                         g_q, g_k, g_v, bias_part_q, bias_part_k, bias_part_v, ldp_q, ldp_kv, stream);
}

}  // extern "C"
