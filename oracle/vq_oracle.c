/*
 * vq_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C) of the reference algorithm on the hot path, used only as the checker by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing under kindergarten-vq-vae_amd/
 * may import, link or call it.
 *
 * Parity status: PINNED.  tests/test_oracle_golden.py checks this file against golden vectors produced by
 * running the reference's own module (/root/reference/models/shelgon3/VectorQuantizer.py) in the build
 * container -- see tests/golden/make_vq_golden.py.
 *
 * What is restated, and from where:
 *   kvq_oracle_distances / kvq_oracle_vq_forward   models/shelgon3/VectorQuantizer.py:55-93
 *   kvq_oracle_vq_backward                         autograd of :72-80 (closed form)
 *   kvq_oracle_ce_forward / _backward              models/shelgon3/Trainer.py:94-101, common/metrics.py:18-30
 *
 * The reference evaluates d = sum(z^2) + sum(E^2) - 2 z.E^T with ATen/MKL, whose summation order is
 * build-specific.  This file fixes ONE order ("kvq order v1", include/kvq.h) -- the order the gfx950 kernel
 * uses -- so that kernel and oracle can be compared bit for bit; against the reference itself indices are
 * identical wherever the best/second-best gap exceeds f32 rounding noise (all "separated" golden cases), and
 * near-tie tokens are checked to be minimal in fp64 within a few ulp (tests/_golden_util.py).
 *
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -mfma; fmaf must be a true fused multiply-add)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define KVQ_ORACLE_VERSION 100

int kvq_oracle_version(void) { return KVQ_ORACLE_VERSION; }

static int g_threads = 1;
void kvq_oracle_set_threads(int n) { g_threads = n > 0 ? n : 1; }
int kvq_oracle_get_threads(void) { return g_threads; }

/* position j of the contraction visited at step s of a group of 8: 0,4,1,5,2,6,3,7 */
static inline int walk8(int s) { return (s >> 1) + ((s & 1) << 2); }

/* sq(x): two half chains (j mod 8 < 4 / >= 4), each over increasing j, then one add.  kvq.h "sq(x)". */
float kvq_oracle_sq(const float* x, int D) {
    float p0 = 0.0f, p1 = 0.0f;
    for (int j = 0; j < D; ++j) {
        if ((j & 7) < 4) p0 = fmaf(x[j], x[j], p0);
        else p1 = fmaf(x[j], x[j], p1);
    }
    return p0 + p1;
}

/* dot(z,e): one fmaf chain in walk8 order; D is padded with zeros to a multiple of 8 (fmaf(0,0,a) == a). */
float kvq_oracle_dot(const float* z, const float* e, int D) {
    float acc = 0.0f;
    int G8 = (D + 7) / 8;
    for (int g = 0; g < G8; ++g)
        for (int s = 0; s < 8; ++s) {
            int j = 8 * g + walk8(s);
            if (j < D) acc = fmaf(z[j], e[j], acc);
        }
    return acc;
}

/* torch.argmin comparison: strictly smaller wins, NaN counts as smaller than everything, first index on ties */
static inline int better(float d, float best) { return (d < best) || (isnan(d) && !isnan(best)); }

/* d[N,K] (optional) and idx[N].  VectorQuantizer.py:59-65.
 * Vectorised across 16 codes at a time (independent chains), E transposed once. */
static void distances_argmin(const float* z, const float* E, int64_t N, int K, int D, float* d_out, int64_t* idx) {
    const int D8 = (D + 7) / 8 * 8;
    const int KB = 16;
    const int Kp = (K + KB - 1) / KB * KB;
    /* Et[s][k]: E in contraction-walk order, zero padded */
    float* Et = (float*)calloc((size_t)D8 * Kp, sizeof(float));
    float* e2 = (float*)calloc((size_t)Kp, sizeof(float));
    for (int k = 0; k < K; ++k) {
        e2[k] = kvq_oracle_sq(E + (size_t)k * D, D);
        for (int g = 0; g < D8 / 8; ++g)
            for (int s = 0; s < 8; ++s) {
                int j = 8 * g + walk8(s);
                Et[(size_t)(8 * g + s) * Kp + k] = j < D ? E[(size_t)k * D + j] : 0.0f;
            }
    }
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (int64_t n = 0; n < N; ++n) {
        const float* zn = z + (size_t)n * D;
        float zw[D8];
        for (int g = 0; g < D8 / 8; ++g)
            for (int s = 0; s < 8; ++s) {
                int j = 8 * g + walk8(s);
                zw[8 * g + s] = j < D ? zn[j] : 0.0f;
            }
        const float z2 = kvq_oracle_sq(zn, D);
        float best = 0.0f;
        int64_t bi = -1;
        for (int k0 = 0; k0 < Kp; k0 += KB) {
            float acc[16] = {0};
            for (int s = 0; s < D8; ++s) {
                const float zs = zw[s];
                const float* et = Et + (size_t)s * Kp + k0;
                for (int kk = 0; kk < KB; ++kk) acc[kk] = fmaf(zs, et[kk], acc[kk]);
            }
            for (int kk = 0; kk < KB && k0 + kk < K; ++kk) {
                /* (sum z^2 + sum e^2) - 2*dot : two roundings, 2*dot is exact  (VectorQuantizer.py:59-61) */
                float t = z2 + e2[k0 + kk];
                float dd = t - 2.0f * acc[kk];
                if (d_out) d_out[(size_t)n * K + k0 + kk] = dd;
                if (bi < 0 || better(dd, best)) { best = dd; bi = k0 + kk; }
            }
        }
        if (idx) idx[n] = bi;
    }
    free(Et);
    free(e2);
}

int kvq_oracle_distances(const float* z, const float* E, int64_t N, int K, int D, float* d) {
    if (!z || !E || !d || N <= 0 || K <= 0 || D <= 0) return -1;
    distances_argmin(z, E, N, K, D, d, NULL);
    return 0;
}

/* VectorQuantizer.forward, one codebook.  counts may be NULL. */
int kvq_oracle_vq_forward(const float* z, const float* E, int64_t N, int K, int D, float beta, float* z_q,
                          int64_t* idx, float* loss, float* perplexity, float* counts) {
    if (!z || !E || !z_q || !idx || !loss || !perplexity || N <= 0 || K <= 0 || D <= 0) return -1;
    distances_argmin(z, E, N, K, D, NULL, idx);
    double sumsq = 0.0;
    float* cnt = (float*)calloc((size_t)K, sizeof(float));
    for (int64_t n = 0; n < N; ++n) {
        const float* zn = z + (size_t)n * D;
        const float* e = E + (size_t)idx[n] * D;   /* min_encodings @ E == E[idx]   (:72) */
        float* q = z_q + (size_t)n * D;
        for (int j = 0; j < D; ++j) {
            float diff = e[j] - zn[j];              /* z_q - z                        (:76-77,:80) */
            q[j] = zn[j] + diff;                    /* z + (z_q - z).detach()         (:80) */
            sumsq += (double)diff * (double)diff;
        }
        cnt[idx[n]] += 1.0f;
    }
    float m = (float)(sumsq / ((double)N * (double)D));
    *loss = m + beta * m;                           /* mean(.) + beta*mean(.)         (:76-77) */
    /* The terms are the reference's f32 values; their SUM is kept in f64: torch.sum adds in a vectorised pairwise order whose
       error stays near one ulp, a sequential f32 sum over K = 8192 terms drifts by 2e-4 (golden case k8192_default). */
    double ent = 0.0;
    for (int k = 0; k < K; ++k) {
        float p = cnt[k] / (float)N;                /* e_mean                         (:84) */
        ent += (double)(p * logf(p + 1e-10f));
    }
    *perplexity = expf(-(float)ent);                 /* (:85) */
    if (counts) memcpy(counts, cnt, (size_t)K * sizeof(float));
    free(cnt);
    return 0;
}

/* Closed-form autograd of forward.  g_zq / g_loss may be NULL (zeros / 1). */
int kvq_oracle_vq_backward(const float* z, const float* E, const int64_t* idx, const float* g_zq,
                           const float* g_loss, int64_t N, int K, int D, float beta, float* g_z, float* g_E) {
    if (!z || !E || !idx || N <= 0 || K <= 0 || D <= 0) return -1;
    const double c = g_loss ? (double)*g_loss : 1.0;
    const double s = c * 2.0 / ((double)N * (double)D);
    double* acc = g_E ? (double*)calloc((size_t)K * D, sizeof(double)) : NULL;
    for (int64_t n = 0; n < N; ++n) {
        const float* zn = z + (size_t)n * D;
        const float* e = E + (size_t)idx[n] * D;
        for (int j = 0; j < D; ++j) {
            float diff = e[j] - zn[j];
            if (g_z) g_z[(size_t)n * D + j] = (float)((g_zq ? (double)g_zq[(size_t)n * D + j] : 0.0) - s * (double)diff);
            if (acc) acc[(size_t)idx[n] * D + j] += (double)diff;
        }
    }
    if (g_E) {
        for (size_t i = 0; i < (size_t)K * D; ++i) g_E[i] = (float)((double)beta * s * acc[i]);
        free(acc);
    }
    return 0;
}

/* EMA codebook update, textbook form (van den Oord et al. 2017, App. A.1; NOT in the reference:
 * "parity unpinned" for this one function -- it is an extension and defaults off). */
int kvq_oracle_vq_ema_update(const float* z, const int64_t* idx, int64_t N, int K, int D, float decay, float eps,
                             float* ema_n, float* ema_m, float* E) {
    if (!z || !idx || !ema_n || !ema_m || !E) return -1;
    double* cnt = (double*)calloc((size_t)K, sizeof(double));
    double* sum = (double*)calloc((size_t)K * D, sizeof(double));
    for (int64_t n = 0; n < N; ++n) {
        cnt[idx[n]] += 1.0;
        for (int j = 0; j < D; ++j) sum[(size_t)idx[n] * D + j] += (double)z[(size_t)n * D + j];
    }
    double tot = 0.0;
    for (int k = 0; k < K; ++k) {
        ema_n[k] = (float)((double)decay * ema_n[k] + (1.0 - (double)decay) * cnt[k]);
        tot += (double)ema_n[k];
    }
    for (int k = 0; k < K; ++k) {
        double nk = ((double)ema_n[k] + (double)eps) / (tot + (double)K * (double)eps) * tot;
        for (int j = 0; j < D; ++j) {
            size_t i = (size_t)k * D + j;
            ema_m[i] = (float)((double)decay * ema_m[i] + (1.0 - (double)decay) * sum[i]);
            E[i] = (float)((double)ema_m[i] / nk);
        }
    }
    free(cnt);
    free(sum);
    return 0;
}

/* Trainer.py:94-101: kl_div(log_softmax(logits), one_hot(ids), "batchmean") and argmax(softmax(logits)).
 * With a one-hot target the KL sum collapses to  -log_softmax(logits)[target]  per row. */
int kvq_oracle_ce_forward(const float* logits, const int64_t* target, int64_t N, int V, float* row_loss,
                          float* row_lse, int64_t* pred, float* loss, float* acc) {
    if (!logits || !target || N <= 0 || V <= 0) return -1;
    double tot = 0.0;
    int64_t hit = 0;
    for (int64_t n = 0; n < N; ++n) {
        const float* x = logits + (size_t)n * V;
        float mx = x[0];
        int64_t am = 0;
        for (int v = 1; v < V; ++v)
            if (x[v] > mx) { mx = x[v]; am = v; }          /* first maximum */
        double se = 0.0;
        for (int v = 0; v < V; ++v) se += exp((double)x[v] - (double)mx);
        double lse = (double)mx + log(se);
        double l = lse - (double)x[target[n]];
        if (row_loss) row_loss[n] = (float)l;
        if (row_lse) row_lse[n] = (float)lse;
        if (pred) pred[n] = am;
        tot += l;
        hit += (am == target[n]);
    }
    if (loss) *loss = (float)(tot / (double)N);
    if (acc) *acc = (float)((double)hit / (double)N);   /* seq_acc per batch: common/metrics.py:25-30 */
    return 0;
}

int kvq_oracle_ce_backward(const float* logits, const int64_t* target, const float* g_loss, int64_t N, int V,
                           float* g_logits) {
    if (!logits || !target || !g_logits) return -1;
    const double c = (g_loss ? (double)*g_loss : 1.0) / (double)N;
    for (int64_t n = 0; n < N; ++n) {
        const float* x = logits + (size_t)n * V;
        float mx = x[0];
        for (int v = 1; v < V; ++v) mx = x[v] > mx ? x[v] : mx;
        double se = 0.0;
        for (int v = 0; v < V; ++v) se += exp((double)x[v] - (double)mx);
        for (int v = 0; v < V; ++v) {
            double p = exp((double)x[v] - (double)mx) / se;
            g_logits[(size_t)n * V + v] = (float)(c * (p - (v == target[n] ? 1.0 : 0.0)));
        }
    }
    return 0;
}
