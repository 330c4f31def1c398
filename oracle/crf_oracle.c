/*
 * oracle/crf_oracle.c -- CPU restatement of the reference CTC-CRF loss path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under cat_amd/ or ctc_crf/ may include, link, import or
 * execute this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.
 *
 * Parity status: the reference (thu-spmi/CAT v3.0.1) ships NO golden vectors for this path
 * (src/ctc_crf/test/main.py:35 only prints the loss) and has NO CPU implementation
 * (src/ctc_crf/setup.py:15-16), so this restatement is pinned by
 *   (1) an independent fp64 brute-force path enumerator (oracle/brute.py) on the reference's own
 *       test fixture (src/ctc_crf/test/main.py:15-28 + test/den_lm.fst, re-created from text by
 *       tests/golden/make_golden.py) and on random tiny graphs,
 *   (2) torch's CPU ctc_loss (an unrelated implementation) for the numerator, and
 *   (3) on the GPU box, the reference's own kernels compiled in place for gfx950 from /root/reference
 *       (oracle/Makefile target `ref`): the denominator (oracle/_ref/libden_ref.so: alpha table entry by
 *       entry, tests/test_gpu_parity.py::test_denominator_vs_reference_kernels) and, since round 6, the
 *       numerator (oracle/_ref/libctc_ref.so: compute_ctc_loss's costs, gradients and alpha workspace,
 *       ::test_numerator_vs_reference_kernels -- fp32 build of this file within 2.3e-7 of the reference's
 *       forward table, same -inf pattern, on six cases incl. repeats, an empty label sequence and an
 *       invalid utterance).
 *
 * Each function cites the reference file:line it restates.  The file is compiled twice:
 *   -DREAL=double -DSUF=_f64   parity oracle
 *   -DREAL=float  -DSUF=_f32   same arithmetic type as the reference CUDA build; the timed
 *                              "cpu_baseline" (kind "port")
 * Parallelism: OpenMP over utterances (the reference uses one CUDA block per utterance,
 * den_calculate.cu:443-446, gpu_ctc.h:258-270), serial in t and in states.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL double
#define SUF _f64
#endif
#define CAT2(a, b) a##b
#define CAT(a, b) CAT2(a, b)
#define FN(name) CAT(name, SUF)

typedef REAL real;
#define NEG_INF (-(real)INFINITY)

static inline real r_exp(real x) { return sizeof(real) == 4 ? (real)expf((float)x) : (real)exp((double)x); }
static inline real r_log1p(real x) { return sizeof(real) == 4 ? (real)log1pf((float)x) : (real)log1p((double)x); }
static inline real r_fabs(real x) { return x < 0 ? -x : x; }

/* den_calculate.cu:29-35 and ctc_helper.h:47-59 (identical definitions). */
static inline real log_plus(real a, real b) {
    if (a == NEG_INF) return b;
    if (b == NEG_INF) return a;
    real m = a > b ? a : b;
    return r_log1p(r_exp(-r_fabs(a - b))) + m;
}

/* ------------------------------------------------------------------------------------------
 * Graph in the layout den_calculate.cu:309-355 builds from fst_read.cc:40-60:
 * per-state in-arc lists ("alpha" side, gathered while iterating source states ascending and
 * their arcs in file order) and out-arc lists ("beta" side, file order).
 * label = ilabel-1, weight = -tropical cost, end_weight = -Final (fst_read.cc:43-59).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int S, A;
    int *in_off, *in_src, *in_lab;   /* in-arcs of state s: [in_off[s], in_off[s+1])  */
    real *in_w;
    int *out_off, *out_dst, *out_lab; /* out-arcs of state s                            */
    real *out_w;
    real *start_w, *end_w;
} graph_t;

static void graph_build(graph_t *g, int S, int A, const int *src, const int *dst, const int *lab,
                        const float *w, const float *start_w, const float *end_w) {
    g->S = S; g->A = A;
    g->in_off = calloc((size_t)S + 1, sizeof(int)); g->out_off = calloc((size_t)S + 1, sizeof(int));
    g->in_src = malloc(sizeof(int) * (size_t)(A + 1)); g->in_lab = malloc(sizeof(int) * (size_t)(A + 1));
    g->in_w = malloc(sizeof(real) * (size_t)(A + 1));
    g->out_dst = malloc(sizeof(int) * (size_t)(A + 1)); g->out_lab = malloc(sizeof(int) * (size_t)(A + 1));
    g->out_w = malloc(sizeof(real) * (size_t)(A + 1));
    g->start_w = malloc(sizeof(real) * (size_t)S); g->end_w = malloc(sizeof(real) * (size_t)S);
    for (int s = 0; s < S; ++s) { g->start_w[s] = (real)start_w[s]; g->end_w[s] = (real)end_w[s]; }
    for (int k = 0; k < A; ++k) { g->in_off[dst[k] + 1]++; g->out_off[src[k] + 1]++; }
    for (int s = 0; s < S; ++s) { g->in_off[s + 1] += g->in_off[s]; g->out_off[s + 1] += g->out_off[s]; }
    int *ci = calloc((size_t)S, sizeof(int)), *co = calloc((size_t)S, sizeof(int));
    /* arcs arrive ordered by (source state, file order) -- the StateIterator/ArcIterator order of
     * fst_read.cc:42-60 -- so appending in input order reproduces the reference's list order. */
    for (int k = 0; k < A; ++k) {
        int pi = g->in_off[dst[k]] + ci[dst[k]]++;
        g->in_src[pi] = src[k]; g->in_lab[pi] = lab[k]; g->in_w[pi] = (real)w[k];
        int po = g->out_off[src[k]] + co[src[k]]++;
        g->out_dst[po] = dst[k]; g->out_lab[po] = lab[k]; g->out_w[po] = (real)w[k];
    }
    free(ci); free(co);
}

static void graph_free(graph_t *g) {
    free(g->in_off); free(g->in_src); free(g->in_lab); free(g->in_w);
    free(g->out_off); free(g->out_dst); free(g->out_lab); free(g->out_w);
    free(g->start_w); free(g->end_w);
}

/* ------------------------------------------------------------------------------------------
 * Denominator forward-backward for ONE utterance.
 * Restates compute_alpha (den_calculate.cu:427-451; kernels :63-161) and
 * compute_beta_and_grad (:453-481; kernels :163-261).
 *   logits : [T][V] log-probs of this utterance (float, as the reference takes them)
 *   grad   : [T][V] out, gamma_den[t][v]; rows t >= lx are left 0 (copy_grad returns early, :239)
 * returns logZ via *lz_alpha (alpha_lld_kernal) and *lz_beta (beta_lld_kernal, generalised to
 * LSE_s(beta_0[s]+start_w[s]); the reference reads state 0 only, :255-261).
 * ------------------------------------------------------------------------------------------ */
static int den_one(const graph_t *g, const float *logits, int T, int V, int lx, float *grad,
                   double *lz_alpha, double *lz_beta, double *alpha_out) {
    const int S = g->S;
    (void)T;
    real *alpha = malloc(sizeof(real) * (size_t)(lx + 1) * (size_t)S);
    real *beta = malloc(sizeof(real) * 2 * (size_t)S);
    real *gs = malloc(sizeof(real) * (size_t)V);
    if (!alpha || !beta || !gs) { free(alpha); free(beta); free(gs); return 1; }

    /* alpha_first_kernel :63-73 */
    for (int s = 0; s < S; ++s) alpha[s] = g->start_w[s];
    /* alpha_kernel :75-103, t = 1..lx */
    for (int t = 1; t <= lx; ++t) {
        const real *prev = alpha + (size_t)(t - 1) * S;
        real *cur = alpha + (size_t)t * S;
        const float *lg = logits + (size_t)(t - 1) * V;
        for (int s = 0; s < S; ++s) {
            real result = NEG_INF;
            for (int k = g->in_off[s]; k < g->in_off[s + 1]; ++k)
                result = log_plus(prev[g->in_src[k]] + g->in_w[k] + (real)lg[g->in_lab[k]], result);
            cur[s] = result;
        }
    }
    /* alpha_last_kernel :105-119 (+end_weight) and alpha_lld_kernal :122-161 (LSE over states).
     * We do not modify alpha[lx] in place: beta_kernel only reads rows t < lx. */
    real lz = NEG_INF;
    for (int s = 0; s < S; ++s) lz = log_plus(lz, alpha[(size_t)lx * S + s] + g->end_w[s]);
    *lz_alpha = (double)lz;
    /* debug entry oracle_den_alpha: the table as the reference leaves it in its `alpha` buffer -- rows 0..lx, row lx
     * with end_weight added in place by alpha_last_kernel :105-119; rows beyond lx are never written (:86). */
    if (alpha_out) {
        for (size_t i = 0; i < (size_t)(lx + 1) * S; ++i) alpha_out[i] = (double)alpha[i];
        for (int s = 0; s < S; ++s) alpha_out[(size_t)lx * S + s] = (double)(alpha[(size_t)lx * S + s] + g->end_w[s]);
    }

    /* beta_last_kernel :163-175 */
    for (int s = 0; s < S; ++s) beta[(size_t)(lx % 2) * S + s] = g->end_w[s];
    /* beta_kernel :189-227 + copy_grad :229-253, t = lx-1..0 */
    for (int t = lx - 1; t >= 0; --t) {
        const real *nxt = beta + (size_t)((t + 1) % 2) * S;
        real *cur = beta + (size_t)(t % 2) * S;
        const real *al = alpha + (size_t)t * S;
        const float *lg = logits + (size_t)t * V;
        for (int v = 0; v < V; ++v) gs[v] = NEG_INF;
        for (int s = 0; s < S; ++s) {
            real br = NEG_INF;
            for (int k = g->out_off[s]; k < g->out_off[s + 1]; ++k) {
                real tmp = nxt[g->out_dst[k]] + g->out_w[k] + (real)lg[g->out_lab[k]];
                br = log_plus(tmp, br);
                /* atomic_log_plus into grad_storage[label][tid%32] :221-223, reduced over the 32
                 * slots by copy_grad :245-249 -- order-free log-sum here. */
                gs[g->out_lab[k]] = log_plus(gs[g->out_lab[k]], al[s] + tmp);
            }
            cur[s] = br;
        }
        float *gr = grad + (size_t)t * V;
        for (int v = 0; v < V; ++v) gr[v] = (float)r_exp(gs[v] - lz); /* copy_grad :251 */
    }
    /* beta_first_kernel :177-187, beta_lld_kernal :255-261 */
    real lzb = NEG_INF;
    for (int s = 0; s < S; ++s) lzb = log_plus(lzb, beta[s] + g->start_w[s]);
    *lz_beta = (double)lzb;
    free(alpha); free(beta); free(gs);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * CTC numerator for ONE utterance on LOG-PROBS with blank = 0 (gpu_ctc/README.txt:2).
 * Restates compute_alpha_kernel (gpu_ctc_kernels.h:87-213) and
 * compute_betas_and_grad_kernel (:218-458): both alpha and beta include the emission at t,
 * gamma[t][k] = exp(LSE_{s: l'_s = k}(alpha+beta) - probs[t][k] - loglike) (:431-435);
 * labels absent from the utterance and rows t >= T_b stay 0 (grads are pre-zeroed,
 * ctc_crf/__init__.py:71).
 * Validity rule L + repeats <= T (gpu_ctc.h:166-174; kernels return early :108-109,261-262 and
 * the reference then reports uninitialised workspace).  DEFINED here: *valid = 0, loglike = 0,
 * gamma = 0 ("skip the utterance"), mirrored by the HIP path.
 * ------------------------------------------------------------------------------------------ */
static int ctc_one(const float *probs, int T, int V, const int *lab, int L, float *grad,
                   double *loglike, int *valid, double *alpha_out) {   /* alpha_out: NULL, or [T][2L+1], the forward table as the reference's
                                                                          workspace holds it (gpu_ctc_kernels.h:134-196: alpha[s + t * S]) */
    const int S = 2 * L + 1, blank = 0;
    int repeats = 0;
    for (int i = 1; i < L; ++i) repeats += (lab[i] == lab[i - 1]); /* gpu_ctc.h:161-165 */
    *valid = 1; *loglike = 0.0;
    if (T <= 0 || L + repeats > T) { *valid = (T <= 0 && L == 0) ? 1 : 0; return 0; }
    int *lb = malloc(sizeof(int) * (size_t)S);
    real *alpha = malloc(sizeof(real) * (size_t)T * (size_t)S);
    real *beta = malloc(sizeof(real) * (size_t)S), *bnew = malloc(sizeof(real) * (size_t)S);
    real *acc = malloc(sizeof(real) * (size_t)V);
    if (!lb || !alpha || !beta || !bnew || !acc) { free(lb); free(alpha); free(beta); free(bnew); free(acc); return 1; }
    for (int i = 0; i < L; ++i) { lb[2 * i] = blank; lb[2 * i + 1] = lab[i]; } /* :112-122 */
    lb[2 * L] = blank;

    /* forward :134-196 */
    for (int s = 0; s < S; ++s) alpha[s] = NEG_INF;
    int start = (L + repeats < T) ? 0 : 1, end = S > 1 ? 2 : 1;
    for (int i = start; i < end; ++i) alpha[i] = (real)probs[lb[i]];
    for (int t = 1; t < T; ++t) {
        const real *pa = alpha + (size_t)(t - 1) * S;
        real *ca = alpha + (size_t)t * S;
        const float *pr = probs + (size_t)t * V;
        ca[0] = (start == 0) ? pa[0] + (real)pr[blank] : pa[0]; /* :165-173 */
        for (int s = 1; s < S; ++s) {
            real prev = log_plus(pa[s], pa[s - 1]);
            if (lb[s] != blank && s != 1 && lb[s] != lb[s - 2]) prev = log_plus(prev, pa[s - 2]);
            ca[s] = prev + (real)pr[lb[s]];
        }
    }
    real ll = NEG_INF; /* :198-212 */
    {
        const int val = 2 * (L - 1) + 1 - ((L + repeats) == T ? 1 : 0);
        int s0 = val * (L != 0) + start, s1 = val * (L != 0) + end;
        for (int i = s0; i < s1; ++i) ll = log_plus(ll, alpha[(size_t)(T - 1) * S + i]);
    }
    *loglike = (double)ll;
    if (alpha_out) for (size_t i = 0; i < (size_t)T * S; ++i) alpha_out[i] = (double)alpha[i];

    /* backward + posteriors :264-436 */
    int bstart = S > 1 ? S - 2 : 0, bend = (L + repeats < T) ? S : S - 1;
    for (int s = 0; s < S; ++s) beta[s] = NEG_INF;
    for (int i = bstart; i < bend; ++i) beta[i] = (real)probs[(size_t)(T - 1) * V + lb[i]];
    for (int t = T - 1; t >= 0; --t) {
        const float *pr = probs + (size_t)t * V;
        if (t < T - 1) {
            for (int s = 0; s < S - 1; ++s) { /* :343-353 */
                real nx = log_plus(beta[s], beta[s + 1]);
                if (lb[s] != blank && s != S - 2 && lb[s] != lb[s + 2]) nx = log_plus(nx, beta[s + 2]);
                bnew[s] = nx + (real)pr[lb[s]];
            }
            bnew[S - 1] = (bend == S) ? beta[S - 1] + (real)pr[blank] : beta[S - 1]; /* :359-361 */
            memcpy(beta, bnew, sizeof(real) * (size_t)S);
        }
        for (int v = 0; v < V; ++v) acc[v] = NEG_INF;
        const real *al = alpha + (size_t)t * S;
        for (int s = 0; s < S; ++s) acc[lb[s]] = log_plus(acc[lb[s]], al[s] + beta[s]); /* :395-402 */
        float *gr = grad + (size_t)t * V;
        for (int v = 0; v < V; ++v)
            gr[v] = acc[v] == NEG_INF ? 0.0f : (float)r_exp(acc[v] - (real)pr[v] - ll); /* :431-435 */
    }
    free(lb); free(alpha); free(beta); free(bnew); free(acc);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Public entry points (ctypes).  All tensors are plain host arrays.
 * ------------------------------------------------------------------------------------------ */

/* gpu_den (binding.cpp:65-84): grad_den [B][T][V] (zero-filled here, __init__.py:67),
 * costs_alpha[B], costs_beta[B]. */
int FN(oracle_den)(int S, int A, const int *src, const int *dst, const int *lab, const float *w,
                   const float *start_w, const float *end_w, const float *logits, int B, int T,
                   int V, const int *lx, float *grad_den, double *costs_alpha, double *costs_beta) {
    graph_t g; graph_build(&g, S, A, src, dst, lab, w, start_w, end_w);
    memset(grad_den, 0, sizeof(float) * (size_t)B * T * V);
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
    for (int b = 0; b < B; ++b)
        err |= den_one(&g, logits + (size_t)b * T * V, T, V, lx[b], grad_den + (size_t)b * T * V,
                       costs_alpha + b, costs_beta + b, NULL);
    graph_free(&g);
    return err;
}

/* Debug entry (tests only): oracle_den plus the forward table in the layout of the reference's `alpha` buffer,
 * [B][T+1][S] (den_calculate.cu:70, 88-89: alpha[b * S * (T+1) + t * S + s]), as double; rows t > lx[b] are left
 * untouched.  The reference fills each entry with a SERIAL loop over the state's in-arcs (:96-100), in the order
 * graph_build reproduces, so its table is deterministic and comparable entry by entry. */
int FN(oracle_den_alpha)(int S, int A, const int *src, const int *dst, const int *lab, const float *w,
                         const float *start_w, const float *end_w, const float *logits, int B, int T,
                         int V, const int *lx, float *grad_den, double *costs_alpha, double *costs_beta,
                         double *alpha_out) {
    graph_t g; graph_build(&g, S, A, src, dst, lab, w, start_w, end_w);
    memset(grad_den, 0, sizeof(float) * (size_t)B * T * V);
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
    for (int b = 0; b < B; ++b)
        err |= den_one(&g, logits + (size_t)b * T * V, T, V, lx[b], grad_den + (size_t)b * T * V,
                       costs_alpha + b, costs_beta + b, alpha_out + (size_t)b * (T + 1) * S);
    graph_free(&g);
    return err;
}

/* gpu_ctc (binding.cpp:86-117) but on the [B][T][V] layout (the reference transposes to
 * [T][B][V] first, __init__.py:70, and back, :77): grad_ctc [B][T][V], costs_ctc[B] = +loglike. */
int FN(oracle_ctc)(const float *logits, int B, int T, int V, const int *labels, const int *lx,
                   const int *ly, float *grad_ctc, double *costs_ctc, int *valid) {
    memset(grad_ctc, 0, sizeof(float) * (size_t)B * T * V);
    int *off = malloc(sizeof(int) * (size_t)(B + 1));
    off[0] = 0;
    for (int b = 0; b < B; ++b) off[b + 1] = off[b] + ly[b];
    int err = 0;
#pragma omp parallel for schedule(dynamic, 1) reduction(| : err)
    for (int b = 0; b < B; ++b)
        err |= ctc_one(logits + (size_t)b * T * V, lx[b], V, labels + off[b], ly[b],
                       grad_ctc + (size_t)b * T * V, costs_ctc + b, valid + b, NULL);
    free(off);
    return err;
}

/* ... and the forward table of every utterance, alpha [B][T][Smax] (row t of utterance b: its own 2 ly[b] + 1 entries, the rest untouched),
 * for the entry-by-entry comparison with the reference's own kernels (tests/test_gpu_parity.py::test_numerator_vs_reference_kernels). */
int FN(oracle_ctc_alpha)(const float *logits, int B, int T, int V, const int *labels, const int *lx,
                         const int *ly, float *grad_ctc, double *costs_ctc, int *valid, double *alpha, int Smax) {
    memset(grad_ctc, 0, sizeof(float) * (size_t)B * T * V);
    int *off = malloc(sizeof(int) * (size_t)(B + 1));
    off[0] = 0;
    for (int b = 0; b < B; ++b) off[b + 1] = off[b] + ly[b];
    int err = 0;
    for (int b = 0; b < B; ++b) {
        const int S = 2 * ly[b] + 1, Tb = lx[b];
        double *tmp = malloc(sizeof(double) * (size_t)(Tb > 0 ? Tb : 1) * S);
        err |= ctc_one(logits + (size_t)b * T * V, Tb, V, labels + off[b], ly[b], grad_ctc + (size_t)b * T * V, costs_ctc + b, valid + b, tmp);
        if (valid[b] && Tb > 0 && S <= Smax)
            for (int t = 0; t < Tb; ++t) memcpy(alpha + ((size_t)b * T + t) * Smax, tmp + (size_t)t * S, sizeof(double) * (size_t)S);
        free(tmp);
    }
    free(off);
    return err;
}

/* _CTC_CRF.forward (ctc_crf/__init__.py:60-90): loss = sum_b(costs_alpha_den - (1+lamb)*costs_ctc),
 * grad = grad_den - (1+lamb)*grad_ctc, both / B when size_average.  grad [B][T][V] out. */
int FN(oracle_ctc_crf)(int S, int A, const int *src, const int *dst, const int *lab, const float *w,
                       const float *start_w, const float *end_w, const float *logits, int B, int T,
                       int V, const int *labels, const int *lx, const int *ly, double lamb,
                       int size_average, float *grad, double *loss, double *costs_den,
                       double *costs_ctc) {
    size_t n = (size_t)B * T * V;
    float *gd = malloc(sizeof(float) * n), *gc = malloc(sizeof(float) * n);
    double *cb = malloc(sizeof(double) * (size_t)B);
    int *valid = malloc(sizeof(int) * (size_t)B);
    int err = FN(oracle_den)(S, A, src, dst, lab, w, start_w, end_w, logits, B, T, V, lx, gd, costs_den, cb);
    err |= FN(oracle_ctc)(logits, B, T, V, labels, lx, ly, gc, costs_ctc, valid);
    double tot = 0.0;
    for (int b = 0; b < B; ++b) tot += costs_den[b] - (1.0 + lamb) * costs_ctc[b];
    const double sc = size_average ? 1.0 / (double)B : 1.0;
    for (size_t i = 0; i < n; ++i) grad[i] = (float)(((double)gd[i] - (1.0 + lamb) * (double)gc[i]) * sc);
    *loss = tot * sc;
    free(gd); free(gc); free(cb); free(valid);
    return err;
}
