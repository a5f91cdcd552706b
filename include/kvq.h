/*
 * kvq.h -- C ABI of libkvq.so: the MI355X (gfx950) implementation of the Kindergarten-VQ-VAE
 * Shelgon/Bagon training hot path.
 *
 * The reference has no FFI (it is pure Python on ATen); the boundary this library sits
 * behind is the reference's nn.Module / function surface.  Each entry point below names the reference
 * lines it replaces.  The Python host (kindergarten-vq-vae_amd/kvq/_ffi.py) binds exactly these symbols
 * with ctypes and hands over raw device pointers (tensor.data_ptr()) and the current HIP stream.
 *
 * Conventions
 *   - every function returns 0 on success, a negative KVQ_E_* code on failure; the message of the last
 *     failure on the calling thread is kvq_last_error().  Nothing throws, nothing is allocated or freed
 *     on behalf of the caller, no host synchronisation happens: all work is enqueued on `stream`
 *     (a hipStream_t passed as void*; NULL = the null stream).  All entry points are hipGraph-capturable and enqueue
 *     KERNELS only: no stream memset, no memcpy (a memset captured as a graph node was observed in round 4 not to keep its
 *     stream order ahead of the next kernel node, profiles/r04_fp8.md; kvq_graph_census lets a caller check what it captured).
 *   - all pointers are DEVICE pointers unless the name ends in _host.
 *   - tensors are dense row-major; "io dtype" is the storage type of activations (z, z_q, g_zq, g_z, logits):
 *     KVQ_F32 or KVQ_BF16.  The codebook E and its gradient are always f32, all arithmetic is f32
 *     (exact f32 MFMA for the distance contraction) whatever the io dtype.
 *   - G = number of independent codebooks ("factors").  G = 1 is the reference.
 *     Layouts: z[G,N,D], E[G,K,D], idx[G,N], loss[G], perplexity[G], counts[G,K].
 *
 * Numerics contract of the VQ forward ("kvq order v1", restated on the CPU in oracle/vq_oracle.c):
 *   dot(n,k)  = one f32 fmaf chain over j, visiting each group of 8 consecutive j in the order
 *               0,4,1,5,2,6,3,7 (the k-walk of v_mfma_f32_32x32x2_f32 fed by 16-byte fragments)
 *   sq(x)     = fl(p0 + p1), p_h = fmaf chain of x[j]^2 over increasing j with (j mod 8)/4 == h
 *   d(n,k)    = fl( fl(sq(z_n) + sq(e_k)) - fl(2*dot(n,k)) )            [VectorQuantizer.py:59-61]
 *   idx(n)    = first k attaining the minimum (NaN counts as smallest, as torch.argmin) [:65]
 *   z_q(n)    = fl( z_n + fl(e_idx - z_n) )                              [:72,:80]
 *   loss      = fl(m + fl(beta*m)),  m = (sum of fl(e_idx - z_n)^2 in f64) / (N*D)     [:76-77]
 *   perplexity= exp(-sum_k p_k*log(p_k + 1e-10)),  p_k = count_k / N  (f32)            [:84-85]
 */
#ifndef KVQ_H
#define KVQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KVQ_VERSION 100 /* 0.1.0 */

/* io dtypes */
#define KVQ_F32 0
#define KVQ_BF16 1

/* error codes */
#define KVQ_OK 0
#define KVQ_E_INVALID (-1)   /* bad argument (null pointer, non-positive size, unsupported dtype) */
#define KVQ_E_WORKSPACE (-2) /* workspace too small / missing */
#define KVQ_E_LAUNCH (-3)    /* HIP reported a launch error */
#define KVQ_E_NODEVICE (-4)  /* no HIP device usable */

int kvq_version(void);
const char* kvq_last_error(void);

/* What a captured hipGraph consists of (host call, no stream): counts_host[KVQ_GRAPH_NODE_KINDS] = nodes of `graph` (a hipGraph_t)
 * by kind.  The TrainEngine's step graphs must hold kernel nodes only (tests/test_graph_nodes_gpu.py asserts it on the graphs
 * the benchmark replays); no reference counterpart. */
#define KVQ_GRAPH_NODE_KERNEL 0
#define KVQ_GRAPH_NODE_MEMSET 1
#define KVQ_GRAPH_NODE_MEMCPY 2
#define KVQ_GRAPH_NODE_EMPTY 3  /* joins of forked streams */
#define KVQ_GRAPH_NODE_EVENT 4  /* event record / wait nodes */
#define KVQ_GRAPH_NODE_OTHER 5
#define KVQ_GRAPH_NODE_KINDS 6
int kvq_graph_census(void* graph, int64_t* counts_host);

/* dst[rows_padded][row_bytes] = the first `rows` rows of src (row stride src_ld_bytes), then zero rows.  All sizes and both
 * buffers in multiples of 16 bytes.  Used for the operands of a weight-gradient product gy^T . x (autograd of every nn.Linear,
 * modeling_bert.py) whose token count is not a multiple of 64, the contraction depth of one MFMA k-tile: zero rows add nothing. */
int kvq_pad_rows(const void* src, int64_t rows, int64_t row_bytes, int64_t src_ld_bytes, void* dst, int64_t rows_padded, void* stream);

/* Number of compute units / name of the device the calling thread would launch on (diagnostics only). */
int kvq_device_info(int* cu_count, char* name, size_t name_len);

/* ------------------------------------------------------------------------------------------------
 * VectorQuantizer.forward  (models/shelgon3/VectorQuantizer.py:31-93)
 *
 *   z        [G,N,D] io dtype   encoder output, N = B*S tokens          (:52-55)
 *   E        [G,K,D] f32        codebook = embedding.weight             (:25)
 *   z_q      [G,N,D] io dtype   straight-through output value           (:72,:80)
 *   idx      [G,N]   int64      min_encoding_indices                    (:65,:90)
 *   loss     [G]     f32        codebook + commitment loss              (:76-77)
 *   perplexity [G]   f32                                                (:84-85)
 *   counts   [G,K]   f32        code usage histogram = sum(min_encodings,0); may be NULL
 *   ws                         scratch of kvq_vq_workspace_bytes(N,K,D,G) bytes, 256-byte aligned
 *
 * What runs per call on the fast path (D %% 32 == 0): a small kernel that resets the minima / histogram / arrival tickets, then ONE
 * kernel for everything of :55-85 -- the f32-MFMA distance / arg-min contraction (64-bit integer atomicMin per token); the
 * workgroup that delivers the LAST code block of a token block goes on to gather, straight-through, per-token loss terms and
 * histogram for those tokens, and the token block that finishes last of all turns the partials into loss / perplexity in a
 * fixed order (no float atomics: bitwise reproducible) -- plus, in kvq_vq_forward only, the re-layout of the codebook in
 * MFMA-fragment order.  kvq_vq_set_variant(0) splits the tail off again into an epilogue and a finalize kernel (rounds 1 - 4;
 * the A/B arm and the checker of the fused tail; per calling thread, the product path never changes it).  A caller that quantises with one codebook many times between its updates (a
 * training loop: one update per optimiser step) keeps that copy itself: kvq_vq_pack_codebook after every codebook update,
 * kvq_vq_forward_packed per step.  Other D: one generic kernel (one wave per token) + the same finalize.
 * min_encodings ([N,K] one-hot, :67-68) is not produced here: see kvq_vq_one_hot.
 */
size_t kvq_vq_workspace_bytes(int64_t N, int K, int D, int G);

int kvq_vq_forward(const void* z, const float* E, int64_t N, int K, int D, int G, int io_dtype, float beta,
                   void* z_q, int64_t* idx, float* loss, float* perplexity, float* counts,
                   void* ws, size_t ws_bytes, void* stream);
int kvq_vq_set_variant(int fused);
/* packed: kvq_vq_packed_bytes(K, D, G) bytes, 16-byte aligned, written by kvq_vq_pack_codebook from the SAME E (D %% 32 == 0). */
size_t kvq_vq_packed_bytes(int K, int D, int G);
int kvq_vq_pack_codebook(const float* E, int K, int D, int G, float* packed, void* stream);
int kvq_vq_forward_packed(const void* z, const float* E, const float* packed, int64_t N, int K, int D, int G, int io_dtype,
                          float beta, void* z_q, int64_t* idx, float* loss, float* perplexity, float* counts,
                          void* ws, size_t ws_bytes, void* stream);

/* Autograd of the above (implicit in the reference; closed form in SURVEY.md §8 row A8b):
 *   g_z  = g_zq - s*fl(e_idx - z),            s = g_loss * 2/(N*D)
 *   g_E[k] = beta*s * sum_{n: idx_n = k} fl(e_k - z_n)      (deterministic: ordered slab reduction, no atomics)
 *   g_zq   [G,N,D] io dtype  upstream gradient of z_q (may be NULL = zeros)
 *   g_loss [G]     f32       upstream gradient of loss, ON DEVICE (may be NULL = ones)
 *   g_z    [G,N,D] io dtype ; g_E [G,K,D] f32 (overwritten, not accumulated); either may be NULL to skip.
 */
int kvq_vq_backward(const void* z, const float* E, const int64_t* idx, const void* g_zq, const float* g_loss,
                    int64_t N, int K, int D, int G, int io_dtype, float beta,
                    void* g_z, float* g_E, void* ws, size_t ws_bytes, void* stream);

/* min_encodings = one_hot(idx) as f32 [N,K]  (VectorQuantizer.py:67-68).  Materialised only on request. */
int kvq_vq_one_hot(const int64_t* idx, int64_t N, int K, float* enc, void* stream);

/* Test hook: the f32 distance matrix d[N,K] exactly as the fused kernel sees it (fast MFMA path when
 * `use_mfma` != 0 and the shape allows, otherwise the generic path).  Not used by the product path. */
int kvq_vq_debug_distances(const void* z, const float* E, int64_t N, int K, int D, int io_dtype,
                           int use_mfma, float* d, void* stream);

/* Kernel timing hook for bench.py's roofline line.  While enabled, every kvq_vq_forward call brackets its FUSED
 * kernel (not the memset / finalize launches) with a pair of HIP events recorded on the call's stream.
 *   kvq_prof_enable(n) : n > 0 allocates a ring of n event pairs and starts recording; n == 0 stops and frees.
 *   kvq_prof_read(ms, max) : after the caller has synchronised the stream, writes up to `max` durations in
 *                            milliseconds (oldest first), clears the ring and returns how many were written. */
int kvq_prof_enable(int n_pairs);
int kvq_prof_read(float* ms_host, int max);

/* Clock probe (measurement aid of bench.py; no reference counterpart): every one of kvq_clock_probe_rows() single-wave workgroups
 * stores out[row] = {XCC id, s_memtime, s_memrealtime, HW_ID} (4 x uint64).  Two probes on one stream bracket a region: per XCD,
 * (memtime_1 - memtime_0) / (memrealtime_1 - memrealtime_0) x 100 MHz is the shader clock the chip held there (DVFS included);
 * the peaks of MI355X_MICROARCH.md are quoted at 2.4 GHz.  Product kernels carry no stamps. */
int kvq_clock_probe_rows(void);
int kvq_clock_probe(uint64_t* out, size_t out_bytes, void* stream);

/* Which path kvq_vq_forward takes for a shape: 1 = f32-MFMA LDS-tiled kernel, 0 = generic kernel. */
int kvq_vq_uses_mfma(int64_t N, int K, int D);

/* EMA codebook update (extension named by BASELINE.json north_star; NOT in the reference -> default off):
 *   n_k <- g*n_k + (1-g)*count_k ;  m_k <- g*m_k + (1-g)*sum_{idx_n=k} z_n ;
 *   E_k <- m_k / ((n_k + eps)/(sum n + K*eps) * sum n)
 *   ema_n [G,K] f32, ema_m [G,K,D] f32 and E are updated in place. */
int kvq_vq_ema_update(const void* z, const int64_t* idx, int64_t N, int K, int D, int G, int io_dtype,
                      float decay, float eps, float* ema_n, float* ema_m, float* E,
                      void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Reconstruction loss of step()  (models/shelgon3/Trainer.py:94-101)
 *
 *   kl_div(log_softmax(logits), one_hot(ids), "batchmean")  ==  mean_n( logsumexp(logits_n) - logits_n[ids_n] )
 *   recon_ids = argmax(softmax(logits)) == argmax(logits) (first maximum)
 *
 *   logits [N,V] io dtype ; target [N] int64 ; row_loss [N] f32 ; row_lse [N] f32 ; pred [N] int64
 *   loss [1] f32 = mean(row_loss) ; acc [1] f32 = mean(pred == target)  (common/metrics.py:18-30)
 * The [N,V] one-hot of the reference (1 GB at N=8192) is never built.
 */
int kvq_ce_forward(const void* logits, const int64_t* target, int64_t N, int V, int64_t ld, int io_dtype,
                   float* row_loss, float* row_lse, int64_t* pred, float* loss, float* acc, void* stream);

/* seq_acc's per-sentence result (common/metrics.py:32-36; consumed as stats_step["metric_acc_step_per_sentence"] by
 * models/bagon/Trainer.py:107,275): per_sentence[b] = mean over s of (pred[b,s] == target[b,s]); pred / target [B,S] int64 row-major
 * (pred = kvq_ce_forward's arg-max).  The per-batch value is kvq_ce_forward's `acc`. */
int kvq_seq_acc(const int64_t* pred, const int64_t* target, int64_t B, int S, float* per_sentence, void* stream);

/* The same results from the per-tile statistics of kvq_gemm_bf16_ce (below): stats [N][tiles][4] f32 = (max, sum exp(x - max),
 * first arg-max as int bits, unused) of row n over tile t's columns < V.  Reads ONE logit per row (the target's); the [N, V]
 * logits are not read again.  SURVEY.md §8(f) rank 1. */
int kvq_ce_forward_stats(const void* logits, const int64_t* target, int64_t N, int64_t ld, int io_dtype, const float* stats, int tiles,
                         float* row_loss, float* row_lse, int64_t* pred, float* loss, float* acc, void* stream);

/* g_logits[n,v] = g_loss/N * (softmax(logits_n)[v] - [v == target_n]); may alias logits (in place). */
int kvq_ce_backward(const void* logits, const int64_t* target, const float* row_lse, const float* g_loss,
                    int64_t N, int V, int64_t ld, int io_dtype, void* g_logits, void* stream);
/* kvq_ce_backward that also leaves bias_part [kvq_ce_bwd_partial_rows(N)][ld] f32 = partial column sums of g_logits (as stored):
 * the gradient of the LM-head output bias (modeling_bert.py:483-497), finished by kvq_reduce_batch -- no second pass over
 * the [N, V] gradient.  Needs ld %% 8 == 0 and 16-byte aligned buffers; columns V .. ld of g_logits and of the sums are 0. */
int64_t kvq_ce_bwd_partial_rows(int64_t N);
int kvq_ce_backward_bias(const void* logits, const int64_t* target, const float* row_lse, const float* g_loss, int64_t N,
                         int V, int64_t ld, int io_dtype, void* g_logits, float* bias_part, size_t part_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Memory-bound pieces of the BERT blocks (HuggingFace modeling_bert.py as used by models/bagon/Bagon.py:24-31),
 * one kernel pass per block boundary; bf16 or f32 activations, f32 arithmetic.  `seed`/`site` key the counter-based
 * dropout generator (Philox4x32-10): forward and backward regenerate the same mask, nothing is stored.
 */

/* BertSelfOutput / BertOutput (:282-293, :340-352):  out = LayerNorm(dropout(y) + resid).
 *   y, resid (may be NULL), out, pre [N,H] io dtype; gamma, beta [H] f32; pre (may be NULL) = dropout(y)+resid as stored;
 *   mean, rstd [N] f32 (may be NULL).  H %% 4 == 0, H <= 4096 (backward: H <= 3072). */
int kvq_dropout_residual_ln_fwd(const void* y, const void* resid, const float* gamma, const float* beta, int64_t N, int H,
                                float eps, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* out, void* pre,
                                float* mean, float* rstd, void* stream);
size_t kvq_ln_bwd_workspace_bytes(int64_t N, int H);
/* g_y = d/dy, g_resid = d/dresid (either may be NULL); g_gamma, g_beta [H] in param_grad_dtype, overwritten or
 * accumulated into (accumulate != 0); either may be NULL.  g_bias_prev [H] (may be NULL) receives the column sums of g_y,
 * i.e. the bias gradient of the dense layer that produced y (BertSelfOutput.dense / BertOutput.dense). */
int kvq_dropout_residual_ln_bwd(const void* g_out, const void* pre, const float* mean, const float* rstd, const float* gamma,
                                int64_t N, int H, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* g_y,
                                void* g_resid, void* g_gamma, void* g_beta, void* g_bias_prev, int param_grad_dtype, int accumulate,
                                void* ws, size_t ws_bytes, void* stream);

/* out[c] (= or +=) scale * sum_n x[n,c]   (bias gradients).  x [N, ld] in_dtype, out [C] out_dtype. */
size_t kvq_colsum_workspace_bytes(int64_t N, int64_t C);
int kvq_colsum(const void* x, int64_t N, int64_t C, int64_t ld, int in_dtype, void* out, int out_dtype, float scale,
               int accumulate, void* ws, size_t ws_bytes, void* stream);

/* Batched reductions (one launch for many small sums): item i computes
 *   dst[c] = scale * sum_{p < count} src[p*ld + c] (+ dst[c] when accumulate != 0),  c < cols.
 * The backward of one BERT layer yields 8-14 of these (split-K slabs of the weight-gradient GEMMs, per-workgroup partials of
 * the LayerNorm gamma/beta and bias gradients: kvq_*_partial below); run one by one each is a launch-latency-bound kernel.
 * Rows are summed in index order, so results are deterministic.  Long rows with count <= 32 (split-K slabs) take a
 * vectorised path when cols, ld %% 8 == 0 and src/dst are 16-byte aligned. */
#define KVQ_REDUCE_MAX_ITEMS 32
typedef struct kvq_reduce_item {
    const void* src;
    void* dst;
    int64_t count, cols, ld;
    float scale;
    int32_t src_dtype, dst_dtype, accumulate;
} kvq_reduce_item;
int kvq_reduce_batch(const kvq_reduce_item* items, int n, void* stream);

/* First halves of kvq_dropout_residual_ln_bwd / kvq_colsum: everything except the final sums over the partial rows, which the
 * caller hands to kvq_reduce_batch.  LayerNorm partials: part [kvq_ln_bwd_partial_rows(N)][3H] f32 = [dbias_prev | dgamma | dbeta]
 * (the dbias_prev third only when want_dbias); column-sum partials: part [kvq_colsum_partial_rows(N)][C] f32.
 * part_bytes >= kvq_ln_bwd_workspace_bytes(N,H) / kvq_colsum_workspace_bytes(N,C). */
int64_t kvq_ln_bwd_partial_rows(int64_t N);
int kvq_dropout_residual_ln_bwd_partial(const void* g_out, const void* pre, const float* mean, const float* rstd,
                                        const float* gamma, int64_t N, int H, float p_drop, uint64_t seed, uint32_t site,
                                        int io_dtype, void* g_y, void* g_resid, int want_dbias, void* part, size_t part_bytes,
                                        void* stream);
/* BertEmbeddings (modeling_bert.py:53-110) forward in one pass and the LayerNorm half of its backward:
 *   out = dropout(LayerNorm(word[ids[n]] + (pos[n %% S] + type_row)))   -- note: dropout AFTER the LayerNorm here
 *   ids [N] int64 (an id outside [0, V) is clamped; torch raises), word [V,H], pos [>= S, H], type_row [H] in the io dtype;
 *   pre [N,H] = the LayerNorm input as stored (backward re-reads it), mean / rstd [N] f32.
 * kvq_ln_dropout_bwd_partial: g_y = d/d(LayerNorm input) for that block (the mask of (seed, site) applies to the incoming
 * gradient g_out); part as in kvq_dropout_residual_ln_bwd_partial with the dbias third unused. */
int kvq_embed_ln_fwd(const int64_t* ids, const void* word, const void* pos, const void* type_row, const float* gamma,
                     const float* beta, int64_t N, int S, int H, int64_t V, float eps, float p_drop, uint64_t seed, uint32_t site,
                     int io_dtype, void* out, void* pre, float* mean, float* rstd, void* stream);
int kvq_ln_dropout_bwd_partial(const void* g_out, const void* pre, const float* mean, const float* rstd, const float* gamma,
                               int64_t N, int H, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* g_y,
                               void* part, size_t part_bytes, void* stream);
int64_t kvq_colsum_partial_rows(int64_t N);
int kvq_colsum_partial(const void* x, int64_t N, int64_t C, int64_t ld, int in_dtype, void* part, size_t part_bytes, void* stream);

/* out[i] = sum_s part[s*n + i], f32 accumulate: combines the S split-K slabs of a weight-gradient GEMM.  n %% 4 == 0. */
int kvq_sum_slabs(const void* part, int S, int64_t n, int io_dtype, void* out, void* stream);

/* BertIntermediate activation (:325-337), erf GELU.  n elements, n %% 4 == 0. */
int kvq_gelu_fwd(const void* h, void* a, int64_t n, int io_dtype, void* stream);
int kvq_gelu_bwd(const void* h, const void* g_a, void* g_h, int64_t n, int io_dtype, void* stream);
/* kvq_gelu_bwd on a row-major [N, C] activation that also leaves bias_part [kvq_gelu_bwd_partial_rows(N)][C] f32 = partial
 * column sums of g_h (as stored, i.e. after rounding to the io dtype): the bias gradient of the dense layer in front of
 * the GELU (BertIntermediate, modeling_bert.py:298-310), finished by kvq_reduce_batch.  C %% 8 == 0, 16-byte aligned buffers. */
int64_t kvq_gelu_bwd_partial_rows(int64_t N);
int kvq_gelu_bwd_bias(const void* h, const void* g_a, void* g_h, int64_t N, int64_t C, int io_dtype, float* bias_part,
                      size_t part_bytes, void* stream);

/* BertSelfAttention / BertCrossAttention core (:111-204) for S_q, S_k <= 32 (bf16 with 16-byte aligned rows: <= 128, in 32-token
 * blocks) and head dim 64: softmax(q k^T * scale + mask) v with dropout on the probabilities.  q [B*Sq, ldq], k/v [B*Sk, ldk/ldv], out [B*Sq, ldo]; head h lives at columns h*64..;
 * mask [B,Sk] int64 (1 = attend) or NULL; causal != 0 adds key <= query.  lse [B,nh,Sq] (may be NULL). */
int kvq_attn_fwd(const void* q, const void* k, const void* v, const int64_t* mask, int B, int nh, int Sq, int Sk, int dh,
                 int ldq, int ldk, int ldv, int ldo, int causal, float scale, float p_drop, uint64_t seed, uint32_t site,
                 int io_dtype, void* out, float* lse, void* stream);
int kvq_attn_bwd(const void* q, const void* k, const void* v, const int64_t* mask, const void* g_out, int B, int nh, int Sq,
                 int Sk, int dh, int ldq, int ldk, int ldv, int ldo, int causal, float scale, float p_drop, uint64_t seed,
                 uint32_t site, int io_dtype, void* g_q, void* g_k, void* g_v, float* bias_part_q, float* bias_part_k,
                 float* bias_part_v, int ldp_q, int ldp_kv, void* stream);
/* The same backward with the forward's output `out` [B*Sq, ldo] and `lse` [B,nh,Sq] handed back: REQUIRED above 32 tokens (the
 * blocked kernels recompute the probabilities from lse and take delta = rowsum(g_out * out) from `out`; no atomics: one kernel
 * for g_q over (sentence, head, query block), one for g_k / g_v over (sentence, head, key block)); at <= 32 tokens both are
 * ignored and the call is kvq_attn_bwd. */
int kvq_attn_bwd_saved(const void* q, const void* k, const void* v, const int64_t* mask, const void* out, const float* lse,
                       const void* g_out, int B, int nh, int Sq, int Sk, int dh, int ldq, int ldk, int ldv, int ldo, int causal,
                       float scale, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* g_q, void* g_k, void* g_v,
                       float* bias_part_q, float* bias_part_k, float* bias_part_v, int ldp_q, int ldp_kv, void* stream);
/* bias_part_* (each may be NULL): per-batch column sums of g_q / g_k / g_v, [B][ldp_q] resp. [B][ldp_kv] f32, columns
 * 0 .. nh*64 -- the partial rows of the q/k/v projection bias gradients (modeling_bert.py:83-85 biases), to be finished by
 * kvq_reduce_batch over the B rows.  The MFMA kernels emit them on the way out; the other flavours run a column-sum pass. */

/* bf16 attention flavour: 2 (default) = MFMA kernels (v_mfma_f32_32x32x16_bf16 for all five products), 1 = packed-dot
 * kernels (v_dot2c_f32_bf16); both round probabilities / dS to bf16 before P.V, dS.K, dS^T.Q, P^T.dO.  0 = convert-and-fma
 * kernels (f32 probabilities).  All flavours draw the same dropout mask.  f32 io always uses the f32 kernels.  The selection is
 * per calling thread (a test / checker facility: the product path never changes it). */
int kvq_attn_set_variant(int variant);

/* ---- bf16 MFMA GEMM family, all three operand layouts of one nn.Linear's forward / backward (csrc/kvq_gemm2.hip) ----------
 * Replaces, for y = x . W^T + b of every BertSelfAttention / BertSelfOutput / BertIntermediate / BertOutput / LM-head linear
 * reached from models/bagon/Bagon.py:46-53 and models/shelgon3/Shelgon.py:52,71 (modeling_bert.py:139-352,483-497):
 *   KVQ_GEMM_NT  C[M,N] = A[M,K] . B[N,K]^T     forward        (x, W)         torch: F.linear
 *   KVQ_GEMM_NN  C[M,N] = A[M,K] . B[K,N]       input gradient (gy, W)        autograd: grad_output.mm(weight)
 *   KVQ_GEMM_TN  C[M,N] = A[K,M]^T . B[K,N]     weight gradient (gy, x)       autograd: grad_output.t().mm(input)
 * bf16 operands and result, f32 accumulation (v_mfma_f32_16x16x32_bf16), optional bias[N] (bf16) and C += (accumulate != 0).
 * K %% 64 == 0; M, N, lda, ldb, ldc %% 8 == 0; 16-byte aligned operands; row-major with the given leading dimensions.
 * `tile` picks the workgroup tile: the caller chooses it so that the tile count fills the 256 CUs (kvq.nnops.pick_tile is the caller's rule).
 * A grouped launch runs up to 16 problems of ONE layout as a single grid (e.g. the weight gradients of two BERT layers:
 * ~250 tiles of 256 x 256, one per CU over the whole token contraction -- no split-K, no partial slabs). */
#define KVQ_GEMM_NT 0
#define KVQ_GEMM_NN 1
#define KVQ_GEMM_TN 2
#define KVQ_GEMM_TILE_128x192 0   /* 8 waves; 256 tiles for [8192, 768] outputs */
#define KVQ_GEMM_TILE_128x256 1   /* 8 waves */
#define KVQ_GEMM_TILE_256x192 2   /* 8 waves */
#define KVQ_GEMM_TILE_256x256 3   /* 8 waves */
#define KVQ_GEMM_TILE_64x128 4    /* 4 waves, two workgroups per CU: outputs too small to give every CU a larger tile (the reference's own
                                   * batches: 12 tokens x 64..128 sentences, models/shelgon3/Trainer.py:82) */
#define KVQ_GEMM_TILE_128x192H 5  /* 4 waves, two ring slots, 80 KiB: TWO workgroups per CU (one wave per SIMD each) whose start-up and
                                   * epilogue run under the other's k loop.  Round-5 experiment: ahead on hot operands, behind inside the
                                   * training step (profiles/r05_gemm_ceiling.md); no caller picks it by itself */
/* OR-ed into `tile` (layout NT, one problem, no accumulate, M and N at least one tile, K >= 192): the PERSISTENT form -- one
 * workgroup per CU walks its tiles, the k-tiles of successive tiles form one uninterrupted LDS-DMA stream and the epilogue
 * goes from the accumulator registers straight to memory; pays when a CU owns two or more tiles. */
#define KVQ_GEMM_PERSISTENT 0x100
typedef struct kvq_gemm_problem {
    const void* A;
    const void* B;
    void* C;
    const void* bias;     /* bf16 [N] or NULL */
    int M, N, K;
    int lda, ldb, ldc;
    int accumulate;
} kvq_gemm_problem;
int kvq_gemm_bf16(const void* A, const void* B, const void* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                  int layout, int tile, int accumulate, void* stream);
/* The any-shape member of the family (csrc/kvq_gemm_any.hip): same layouts and meaning, NO divisibility or 16-byte alignment
 * requirement (any M, N, K, leading dimensions; 2-byte aligned operands); f32 accumulation in ascending k on the vector unit.
 * For the launch-latency-sized products that do not meet kvq_gemm_bf16's requirements (token counts that are not multiples of
 * 64, 9-code Gumbel logits, ...), so that no product of the bf16 step leaves the library. */
int kvq_gemm_any_bf16(const void* A, const void* B, const void* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc,
                      int layout, int accumulate, void* stream);
int kvq_gemm_grouped_bf16(const kvq_gemm_problem* problems, int n_problems, int layout, int tile, void* stream);
/* BertIntermediate in one kernel (modeling_bert.py:325-337), layout NT: Hout = A.B^T + bias (kept for backward) and
 * Aout = gelu(Hout as rounded to bf16) -- exact-erf GELU (erf to 1.2e-7).  tile: KVQ_GEMM_TILE_256x192 or _128x256 (optionally
 * | KVQ_GEMM_PERSISTENT). */
/* The dense layer in front of a residual LayerNorm (BertSelfOutput / BertOutput, modeling_bert.py:282-296, 339-352), layout NT:
 * C = dropout(A.B^T + bias, p_drop) + R, rounded as kvq_dropout_residual_ln_fwd rounds its `pre` (the dense output to bf16, times the
 * keep scale, plus the residual, to bf16) and with ITS masks (seed + the kvq_set_seed_offset word, `site`, element index): C is
 * bit for bit the `pre` that kernel would store, so LayerNorm forward becomes kvq_dropout_residual_ln_fwd(C, NULL, p = 0) -- half the
 * bytes -- and backward is unchanged.  R: [M, N] bf16, row stride ldc.  tile: KVQ_GEMM_TILE_128x192, _128x256 or _64x128. */
int kvq_gemm_bf16_dropres(const void* A, const void* B, const void* bias, const void* R, void* C, int M, int N, int K, int lda, int ldb,
                          int ldc, int tile, float p_drop, uint64_t seed, uint32_t site, void* stream);
int kvq_gemm_bf16_gelu(const void* A, const void* B, const void* bias, void* Hout, void* Aout, int M, int N, int K, int lda, int ldb,
                       int ldc, int tile, void* stream);
/* Its backward through the activation, layout NN: C = (A.B) * gelu'(H)  (A = gradient of BertOutput.dense's output, B = its
 * weight [K,N], H = the saved pre-activation [M,N], row stride ldc) plus part[t][n] = sum over the rows of row-tile t of the
 * bf16 values stored in C: the partial rows of BertIntermediate's bias gradient (finish with kvq_reduce_batch over
 * kvq_gemm_dgelu_partial_rows(M, tile) rows).  Replaces the dgrad GEMM + the elementwise gelu-backward + bias column sums. */
int64_t kvq_gemm_dgelu_partial_rows(int64_t M, int tile);
int kvq_gemm_bf16_dgelu(const void* A, const void* B, const void* H, void* C, float* part, size_t part_bytes, int M, int N, int K,
                        int lda, int ldb, int ldc, int tile, void* stream);

/* LM head with the forward half of the reconstruction loss in the GEMM epilogue (models/shelgon3/Trainer.py:94-101 after
 * BertLMPredictionHead.decoder, modeling_bert.py:483-497):  C[M, ldc] (bf16) = A[M,K] . B[N,K]^T + bias  (layout NT, 256 x 256
 * tiles) and stats [M][ceil(N/256)][4] f32 = per row and tile (max, sum exp(x - max), first arg-max, -) of the values as stored,
 * over the columns < V (vocabulary padding excluded).  kvq_ce_forward_stats turns them into loss / lse / arg-max / accuracy. */
size_t kvq_gemm_ce_stats_bytes(int M, int N);
int kvq_gemm_bf16_ce(const void* A, const void* B, const void* bias, void* C, int M, int N, int K, int lda, int ldb, int ldc, int V,
                     float* stats, size_t stats_bytes, void* stream);

/* ---- fp8 (OCP e4m3fn) forward GEMMs: extension named by BASELINE.json configs[4]; the reference is f32 throughout -> off by default.
 *   y = x . W^T + b of a BERT linear (modeling_bert.py:139-352) as  (sat(x sx) . sat(W sw)^T) / (sx sw) + b,  s = 448 / amax|.|
 *   per tensor, computed in the same step from the tensor that is quantised (no amax history).  Backward stays bf16.
 * kvq_fp8_quantize: one bf16 matrix [rows, cols] (row stride ld) -> dense fp8 [rows, cols]; amax, scale: device scalars (written).
 * kvq_fp8_quantize_segments: ranges [seg_off[s], +seg_n[s]) (elements; device arrays; multiples of 16) of one bf16 buffer ->
 *   the same ranges of an fp8 buffer, one amax / scale per range: all GEMM weights of the model in two launches.
 * kvq_gemm_fp8_nt: C[M,N] bf16 = (A8[M,K] . B8[N,K]^T) / (scale_a[0] scale_b[0]) + bias;  K %% 128 == 0, lda, ldb %% 16 == 0
 *   (fp8 elements), on v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (2x the bf16 matrix rate). */
int kvq_fp8_quantize(const void* x_bf16, int64_t rows, int cols, int64_t ld, void* out_fp8, float* amax, float* scale, void* stream);
/* Activations, one pass ("delayed scaling"): per call site a device record of kvq_fp8_state_floats() floats, [0] = scale
 * (initialise to 1), the rest per-workgroup amax partials (initialise to 0): quantise with the site's scale and leave this
 * tensor's amax in the partials; kvq_fp8_update_scales (once per step, nsites consecutive records) sets
 * scale = 448 / (amax * headroom) and clears the partials (headroom 0: clears only -- after a forward whose amax must not
 * count, e.g. an evaluation batch).  Values beyond the previous step's range saturate at +-448. */
int kvq_fp8_state_floats(void);
int kvq_fp8_quantize_delayed(const void* x_bf16, int64_t rows, int cols, int64_t ld, void* out_fp8, float* state, void* stream);
int kvq_fp8_update_scales(float* state, int nsites, float headroom, void* stream);
int kvq_fp8_quantize_segments(const void* src_bf16, const int64_t* seg_off, const int64_t* seg_n, int nseg, int64_t max_seg_n,
                              void* dst_fp8, float* amax, float* scale, void* stream);
/* The same inside a training step: the per-segment amax (and with it the scale) is refreshed only when the DEVICE-resident step
 * count *step_count_u64 is a multiple of `period`; on the steps between, the segments are quantised with the scale they have
 * (values that outgrew it saturate at +-448).  Weights move by ~lr per step: a period of 16 saves one of the two passes over the
 * weights on 15 steps of 16. */
/* (period < -1: ALSO the conversion pass runs only on steps that are multiples of -period -- for a caller whose optimiser kernel
 * writes the fp8 bytes itself on the other steps: kvq_adam_step_dev_fp8.) */
int kvq_fp8_quantize_segments_periodic(const void* src_bf16, const int64_t* seg_off, const int64_t* seg_n, int nseg, int64_t max_seg_n,
                                       void* dst_fp8, float* amax, float* scale, const void* step_count_u64, int period, void* stream);
int kvq_gemm_fp8_nt(const void* A8, const void* B8, const float* scale_a, const float* scale_b, const void* bias, void* C, int M, int N,
                    int K, int lda, int ldb, int ldc, void* stream);
/* Round 5 -- the fp8 copy of an activation written by the kernel that PRODUCES it, so that the fp8 GEMM reading it next needs no
 * quantisation pass: each of these writes, beside its bf16 result, exactly the bytes kvq_fp8_quantize_delayed(result, state) would
 * write (the site's scale of the previous step) and notes the result's amax in the state's partial slots (atomic maxima: any
 * number of workgroups).  The caller runs kvq_fp8_update_scales once per step as before.
 *   kvq_dropout_residual_ln_fwd_fp8 : LayerNorm output -> the QKV / cross-attention query / BertIntermediate projections
 *   kvq_attn_fwd_fp8                : attention context (at most 32 tokens, the MFMA kernel: kvq_attn_fwd_fp8_ok) -> BertSelfOutput.dense
 *   kvq_gemm_fp8_nt_gelu            : BertIntermediate on the fp8 matrix cores with the GELU epilogue of kvq_gemm_bf16_gelu, and
 *                                     (Aout_fp8 != NULL) the fp8 copy of gelu(h) -> BertOutput.dense */
int kvq_dropout_residual_ln_fwd_fp8(const void* y, const void* resid, const float* gamma, const float* beta, int64_t N, int H,
                                    float eps, float p_drop, uint64_t seed, uint32_t site, void* out, void* pre, float* mean, float* rstd,
                                    void* out_fp8, float* fp8_state, void* stream);
int kvq_attn_fwd_fp8_ok(int Sq, int Sk);
int kvq_attn_fwd_fp8(const void* q, const void* k, const void* v, const int64_t* mask, int B, int nh, int Sq, int Sk, int dh,
                     int ldq, int ldk, int ldv, int ldo, int causal, float scale, float p_drop, uint64_t seed, uint32_t site,
                     void* out, float* lse, void* out_fp8, int ld8, float* fp8_state, void* stream);
int kvq_gemm_fp8_nt_gelu(const void* A8, const void* B8, const float* scale_a, const float* scale_b, const void* bias, void* Hout, void* Aout,
                         void* Aout_fp8, int ld8, float* fp8_state, int M, int N, int K, int lda, int ldb, int ldc, void* stream);

/* torch.optim.Adam step (models/shelgon3/main.py:91: lr, weight_decay (L2, coupled), amsgrad) on flat buffers.
 *   p, m, v [, vmax] f32; g grad_dtype (scaled by grad_scale first); shadow_bf16 (may be NULL) receives bf16(p_new).
 *   step >= 1 is the 1-based step count for bias correction.  Any n >= 1 (16-byte aligned buffers; the last n %% 4 elements take a
 *   scalar path: the 9-code Gumbel bias of the reference's analysis run). */
/* kvq_adam_step_dev that also writes the fp8 (e4m3) mirror of the GEMM weights inside [first_element, first_element + n) of the flat
 * parameter buffer: w8_mirror is indexed like that buffer (one byte per element), span_segment[element >> 11] = the quantisation
 * segment covering that 2048-element span (-1 none, -2 several things: the kernel then walks seg_off / seg_n), seg_scale the segments'
 * current scales.  The bytes are kvq_fp8_quantize_segments' for the bf16 value the update stores in the shadow. */
int kvq_adam_step_dev_fp8(float* p, const void* g, float* m, float* v, float* vmax, void* shadow_bf16, int64_t n, int grad_dtype,
                          const void* step_state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                          void* w8_mirror, const int* span_segment, const float* seg_scale, const int64_t* seg_off, const int64_t* seg_n,
                          int nseg, int64_t first_element, void* stream);
int kvq_adam_step(float* p, const void* g, float* m, float* v, float* vmax, void* shadow_bf16, int64_t n, int grad_dtype,
                  float lr, float beta1, float beta2, float eps, float weight_decay, int64_t step, float grad_scale,
                  void* stream);


/* ---- device-resident step state (hipGraph-friendly training step) ------------------------------------------------------
 * step_state: 24 bytes of device memory, zero-initialised = "no optimiser step applied yet":
 *     struct { uint64_t step; float lr, bc1, bc2s, pad; }
 * kvq_step_state_advance   one-thread kernel run at the start of an optimiser step: step += 1, lr = lr0 * gamma^(number of
 *                          milestones <= step-1)  (torch MultiStepLR ticked once per finished step, Trainer.py:114-115),
 *                          bc1 = 1 - beta1^step, bc2s = sqrt(1 - beta2^step).  At most 8 milestones (host array).
 * kvq_step_state_prepare / _commit   the two halves of _advance: prepare writes lr / bc1 / bc2s of the step about to be applied
 *                          and leaves `step` (the dropout seed offset) alone, commit does step += 1.  For an optimiser whose
 *                          updates start while backward -- which still recomputes this step's dropout masks -- is running.
 * kvq_adam_step_dev        kvq_adam_step reading lr / bias corrections from the step state instead of taking them by value.
 * kvq_set_seed_offset      per calling thread: every dropout-bearing kernel that thread launches afterwards (kvq_dropout,
 *                          kvq_dropout_residual_ln_*, kvq_attn_*) uses seed + step_state->step, read on the device at run
 *                          time; NULL switches it off.  A step captured once in a hipGraph then replays with fresh masks,
 *                          the right learning rate and bias corrections, with nothing patched from the host.
 * kvq_dropout              out = x * keep/(1-p) for the Philox mask of (seed, site); element i draws from call i/4 like the
 *                          other kernels.  Forward and (applied to the gradient) backward of a plain dropout.  n %% 4 == 0. */
int kvq_step_state_advance(void* step_state, float lr0, float gamma, const int64_t* milestones, int n_milestones, float beta1,
                           float beta2, void* stream);
int kvq_step_state_prepare(void* step_state, float lr0, float gamma, const int64_t* milestones, int n_milestones, float beta1,
                           float beta2, void* stream);
int kvq_step_state_commit(void* step_state, void* stream);
int kvq_adam_step_dev(float* p, const void* g, float* m, float* v, float* vmax, void* shadow_bf16, int64_t n, int grad_dtype,
                      const void* step_state, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                      void* stream);
int kvq_set_seed_offset(const void* step_state);
int kvq_dropout(const void* x, int64_t n, float p_drop, uint64_t seed, uint32_t site, int io_dtype, void* out, void* stream);
/* Zero 1..4 byte ranges (16-byte aligned pointers and sizes) in one launch: the word / position / token-type gradient tables of
 * BertEmbeddings (modeling_bert.py:53-58) before kvq_embed_grad and the batched reductions add into them. */
int kvq_zero_ranges(void* const* ptrs, const int64_t* bytes, int n, void* stream);


/* ---- the consumer of the code indices: word x code counts ------------------------------------------------------------------
 * Replaces the per-token Python walk of analyses/unsupervised_vq_disentanglement/unsupervised_vq_disentanglement.py:166-200
 * (`vq_words_distrib[code].append(word)`, `seen_v_is.add(code)` for every token of every word :181-183;
 *  `words_of_interest_vq_distrib[word].append(v_is[0])` for a word's first token :192-199); the result files (:208-235) only use
 * counts and sets of those lists, i.e. projections of
 *     counts_all  [G][W][K] (uint32) += 1 for every token n with a word,      cell (g, slot(n), idx[n][g])
 *     counts_first[G][W][K] (uint32) += 1 for the FIRST token of every word,  same cell
 * slot_first [N] int32: -1 = the position belongs to no word (padding), else (slot << 1) | (1 if the word's first token);
 * idx [N][G] int64 = min_encoding_indices as the quantisers return them.  The tables ACCUMULATE (zero them before the first batch).
 * Positions with slot >= W or an index outside [0, K) are skipped and counted in *n_bad (optional).  Exact integer arithmetic;
 * G * W * K < 2^31. */
int kvq_code_census(const int32_t* slot_first, const int64_t* idx, int64_t N, int G, int K, int W, uint32_t* counts_all,
                    uint32_t* counts_first, uint32_t* n_bad, void* stream);


/* Word-embedding gradient (autograd of the row gather of BertEmbeddings, modeling_bert.py:53-58):
 *     gW[id][:] (= | +=) sum over the tokens n with ids[n] == id of g[n][:],   tokens added in increasing n  (deterministic)
 * The caller passes the tokens sorted by id: sorted_ids[s] ascending and perm[s] = the token at sorted position s (a STABLE
 * sort, e.g. torch.sort(ids, stable=True)); the same pair serves every embedding table looked up with these ids.
 * g [N, H] (g_dtype), gW [V, H] (w_dtype); with accumulate == 0 only the rows of ids that occur are written (zero gW first),
 * ids outside [0, V) are ignored.  H %% 4 == 0, H <= 1024.  No float atomics. */
size_t kvq_embed_grad_workspace_bytes(int64_t N, int H);
int kvq_embed_grad(const void* g, const int64_t* perm, const int64_t* sorted_ids, int64_t N, int H, int64_t V, int g_dtype,
                   void* gW, int w_dtype, int accumulate, void* ws, size_t ws_bytes, void* stream);


/* One Lloyd iteration's centroid update (the data-driven codebook initialiser of the reference runs
 * scipy.cluster.vq.kmeans2(z, K, minit='points') on encoder outputs, models/shelgon3/vq_codebook_init_weights.py:85-101):
 *     E_k <- mean of the rows z_n with idx_n == k  (f64 accumulation of per-chunk f32 sums, rows added in increasing n);
 *     a cluster without rows keeps its centroid (kmeans2's missing='warn');  counts[K] (may be NULL) receives the cluster sizes.
 * The assignment step of the iteration IS the hot kernel: kvq_vq_forward(z, E) -> idx.  ws: kvq_vq_workspace_bytes(N,K,D,1). */
int kvq_kmeans_update(const void* z, const int64_t* idx, int64_t N, int K, int D, int io_dtype, float* E, int64_t* counts,
                      void* ws, size_t ws_bytes, void* stream);


/* ---- Gumbel-softmax quantiser (the reference's other VQ_MODE, models/shelgon3/GumbelQuantizer.py:43-83) ---------------------
 * Row-wise part of GumbelQuantizer.forward between its two GEMMs (logits = proj(z), z_q = y . embed):
 *   y_soft = softmax((logits + g) / tau) with g ~ Gumbel(0,1);  ind = argmax(y_soft) (first maximum);
 *   y = y_soft, or (hard != 0) fl(fl(one_hot(ind) - y_soft) + y_soft): torch.nn.functional.gumbel_softmax's return value;
 *   kl_row[n] = sum_k q log(q K + 1e-10), q = softmax(logits)   (diff = kld_scale * mean_n kl_row, :73).
 * logits, y [N,K] io_dtype; y_soft [N,K] f32 (kept for backward; may be NULL), ind [N] int64, kl_row [N] f32.
 * noise [N,K] f32 supplies g explicitly (parity tests against torch's generator); NULL draws it from Philox4x32-10
 * (seed, site; plus the device step count when kvq_set_seed_offset is active).  K <= 1024.
 * Backward: g_logits = y_soft (g_y - <y_soft,g_y>) / tau  +  g_diff kld_scale/N * q (L - <q,L>),
 *   L = log(q K + 1e-10) + q K / (q K + 1e-10); g_y may be NULL (no gradient through y), g_diff NULL means 1. */
int kvq_gumbel_forward(const void* logits, const float* noise, int64_t N, int K, float tau, int hard, uint64_t seed, uint32_t site,
                       int io_dtype, void* y, float* y_soft, int64_t* ind, float* kl_row, void* stream);
int kvq_gumbel_backward(const void* logits, const float* y_soft, const void* g_y, const float* g_diff, int64_t N, int K, float tau,
                        float kld_scale, int io_dtype, void* g_logits, void* stream);


#ifdef __cplusplus
}
#endif
#endif /* KVQ_H */
