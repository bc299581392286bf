/*
 * oracle/erm_oracle.c -- TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED.
 *
 * CPU restatement (plain C, fp64, single thread, un-fused reference schedule) of the Gibbs hot
 * path of ExtendedRtIrtModeling.jl: every full conditional of /root/reference/src/Draw.pl.jl and
 * the sweep orders / collection / log-likelihoods of src/GibbsRtIrt.pl.jl,
 * src/GibbsRtIrtCross.pl.jl and src/GibbsRtIrtLatent.pl.jl.  Each function cites the reference
 * lines it follows.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * The product (libertirt.so) never links, includes or calls anything in oracle/.
 *
 * "Parity unpinned": the reference's test-suite holds no golden vectors for this path
 * (test/test-basic-test.jl:1-3 calls an undefined function), Julia is not installable here, and
 * the scalar samplers live in un-vendored Julia packages (see orc_rng.h).  The oracle is pinned
 * only by build-owned checks: Philox known-answer vectors, closed-form moments of every sampler,
 * and numpy re-evaluations of the reference's broadcast expressions (tests/test_oracle_*.py).
 *
 * Data layout: column-major (Julia) N x J for Y (uint8 0/1) and logT, N x nFeat for X.
 * Random numbers: counter-based streams keyed by (seed, chain, site, i, j, sweep), orc_rng.h.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "orc_rng.h"

enum { ORC_MLIRT = 0, ORC_RTIRT = 1, ORC_CROSSQR = 2, ORC_LATENTQR = 3,
       /* the non-quantile variants */
       ORC_NULL = 4,    /* GibbsRtIrtNull   src/GibbsRtIrt.pl.jl:151-183, sample! :367-426 */
       ORC_CROSS = 5,   /* GibbsRtIrtCross  src/GibbsRtIrtCross.pl.jl:77-110, sample! :176-235 */
       ORC_LATENT = 6   /* GibbsRtIrtLatent src/GibbsRtIrtLatent.pl.jl:70-102, sample! :168-233 */ };

/* OpenMP is used only for the multi-threaded CPU baseline of bench.py: every parallel loop runs over independent subjects,
 * items or cells whose draws are counter-addressed and whose sums stay inside one iteration, so results are bit-identical
 * for any thread count (the log-likelihood's reduction is the one exception: its association changes with the team size).
 * orc_threads defaults to 1, the reference's own execution model. */
#define ORC_PRAGMA(x) _Pragma(#x)
#define ORC_OMP_FOR ORC_PRAGMA(omp parallel for schedule(static) num_threads(orc_threads))
static int orc_threads = 1;
void orc_set_threads(int n) { orc_threads = n > 0 ? n : 1; }
int orc_get_threads(void) { return orc_threads; }

typedef struct {
    int32_t model;
    int32_t nItem;
    int64_t nSubj;
    int32_t nFeat;
    int32_t intercept;   /* sample! kwarg, default false */
    int32_t onepl;       /* itemtype == "1pl" */
    int32_t cov2one;     /* sample! kwarg */
    int32_t chain;
    int32_t sigp_mode;   /* LatentQr a19: 0 = reference's closed form of the N x N '/' quirk, 1 = intended sum r^2/(2 k2 nu) */
    double qRt;
    uint64_t seed;
} orc_config;

typedef struct {
    const uint8_t* Y;     /* N x J col-major */
    const double* logT;   /* N x J col-major (may be NULL for MlIrt) */
    const double* X;      /* N x nFeat col-major (may be NULL) */
} orc_data;

typedef struct {
    double* theta;   /* N */
    double* a;       /* J */
    double* b;       /* J */
    double* zeta;    /* N */
    double* lambda;  /* J */
    double* sig2t;   /* J */
    double* beta;    /* MlIrt: p ; RtIrt: p x 2 col-major ; LatentQr: nFeat+2 */
    double* Sigp;    /* 2 x 2 col-major */
    double* rho;     /* J */
    double* nu;      /* LatentQr: N ; CrossQr: N x J col-major */
    double* omega;   /* N x J col-major (scratch/output) */
} orc_state;

#define IDX(i, j, N) ((size_t)(j) * (size_t)(N) + (size_t)(i))

/* ------------------------------------------------------------------ small dense helpers */
static void chol_lower(int n, const double* A, double* L) /* A col-major symmetric -> L col-major lower */
{
    memset(L, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; ++j) {
        double d = A[j + j * n];
        for (int k = 0; k < j; ++k) d -= L[j + k * n] * L[j + k * n];
        d = sqrt(d);
        L[j + j * n] = d;
        for (int i = j + 1; i < n; ++i) {
            double v = A[i + j * n];
            for (int k = 0; k < j; ++k) v -= L[i + k * n] * L[j + k * n];
            L[i + j * n] = v / d;
        }
    }
}
static void mat_inverse(int n, const double* A, double* Ainv) /* Gauss-Jordan, partial pivoting, col-major */
{
    double* M = (double*)malloc(sizeof(double) * n * 2 * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) { M[i * 2 * n + j] = A[i + j * n]; M[i * 2 * n + n + j] = (i == j); }
    for (int c = 0; c < n; ++c) {
        int piv = c;
        for (int r = c + 1; r < n; ++r) if (fabs(M[r * 2 * n + c]) > fabs(M[piv * 2 * n + c])) piv = r;
        if (piv != c) for (int j = 0; j < 2 * n; ++j) { double t = M[c * 2 * n + j]; M[c * 2 * n + j] = M[piv * 2 * n + j]; M[piv * 2 * n + j] = t; }
        double d = M[c * 2 * n + c];
        for (int j = 0; j < 2 * n; ++j) M[c * 2 * n + j] /= d;
        for (int r = 0; r < n; ++r) if (r != c) {
            double f = M[r * 2 * n + c];
            if (f != 0.0) for (int j = 0; j < 2 * n; ++j) M[r * 2 * n + j] -= f * M[c * 2 * n + j];
        }
    }
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ainv[i + j * n] = M[i * 2 * n + n + j];
    free(M);
}
static void solve_spd(int n, const double* A, const double* rhs, double* x) /* A x = rhs */
{
    double* Ai = (double*)malloc(sizeof(double) * n * n);
    mat_inverse(n, A, Ai);
    for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += Ai[i + j * n] * rhs[j]; x[i] = s; }
    free(Ai);
}
/* design row x_i = [1, X_i1..X_inFeat] (+ theta_i when with_theta) */
static inline double xrow(const orc_config* c, const orc_data* d, const orc_state* s, int64_t i, int col, int with_theta)
{
    if (col == 0) return 1.0;
    if (col <= c->nFeat) return d->X[IDX(i, col - 1, c->nSubj)];
    (void)with_theta;
    return s->theta[i];
}
static void xtx(const orc_config* c, const orc_data* d, const orc_state* s, int p, int with_theta, double* out)
{
    for (int u = 0; u < p; ++u) for (int v = 0; v < p; ++v) {
        double acc = 0;
        for (int64_t i = 0; i < c->nSubj; ++i) acc += xrow(c, d, s, i, u, with_theta) * xrow(c, d, s, i, v, with_theta);
        out[u + v * p] = acc;
    }
}
static inline double k1_of(double q) { return (1.0 - 2.0 * q) / (q * (1.0 - q)); }
static inline double k2_of(double q) { return 2.0 / (q * (1.0 - q)); }
static inline double log1pexp(double x) { return x > 0 ? x + log1p(exp(-x)) : log1p(exp(x)); }

/* ------------------------------------------------------------------ hoisted data constants */
/* mean(Data.logT), std(Data.logT) -- kwargs of drawItemIntensity, src/Draw.pl.jl:215,239 */
static void logT_mean_std(const orc_config* c, const orc_data* d, double* mu, double* sd)
{
    size_t n = (size_t)c->nSubj * c->nItem;
    double s = 0;
    for (size_t k = 0; k < n; ++k) s += d->logT[k];
    double m = s / (double)n, ss = 0;
    for (size_t k = 0; k < n; ++k) { double e = d->logT[k] - m; ss += e * e; }
    *mu = m; *sd = sqrt(ss / (double)(n - 1));
}

/* ------------------------------------------------------------------ full conditionals */

/* drawRaPgRandomVariable, src/Draw.pl.jl:36-40 */
static void draw_omega(const orc_config* c, orc_state* s, uint32_t sweep)
{
    for (int j = 0; j < c->nItem; ++j)
        ORC_OMP_FOR
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double eta = s->a[j] * (s->theta[i] - s->b[j]);
            orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_OMEGA, (uint32_t)i, (uint32_t)j, sweep);
            s->omega[IDX(i, j, c->nSubj)] = orc_pg1(&st, eta);
        }
}

/* drawSubjAbility (src/Draw.pl.jl:49-62) and drawSubjAbilityNull (:67-80).
 * prior: 0 = Null (mu0 = 0), 1 = x*beta[:,1]; sigma0^2 = Sigp[1,1] (1.0 for GibbsMlIrt, whose
 * setInitialValues never sets Sigp -- src/GibbsRtIrt.pl.jl:85-91,228; likelihood uses Normal(mu,1.) :201). */
static void moments_theta(const orc_config* c, const orc_data* d, const orc_state* s, int prior, double* parM, double* parV)
{
    double s0 = (c->model == ORC_MLIRT) ? 1.0 : s->Sigp[0];
    int p = c->nFeat + 1;
    ORC_OMP_FOR
    for (int64_t i = 0; i < c->nSubj; ++i) {
        double mu0 = 0.0;
        if (prior) for (int u = 0; u < p; ++u) mu0 += xrow(c, d, s, i, u, 0) * s->beta[u]; /* beta[:,1] = first p entries */
        double sa = 0, sb = 0;
        for (int j = 0; j < c->nItem; ++j) {
            double w = s->omega[IDX(i, j, c->nSubj)], kap = (double)d->Y[IDX(i, j, c->nSubj)] - 0.5;
            sa += s->a[j] * s->a[j] * w;
            sb += s->a[j] * (kap + s->a[j] * s->b[j] * w);
        }
        parV[i] = 1.0 / (1.0 / s0 + sa);
        parM[i] = parV[i] * (mu0 / s0 + sb);
    }
}
static void draw_theta(const orc_config* c, const orc_data* d, orc_state* s, int prior, uint32_t sweep)
{
    double* m = (double*)malloc(sizeof(double) * c->nSubj), *v = (double*)malloc(sizeof(double) * c->nSubj);
    moments_theta(c, d, s, prior, m, v);
    ORC_OMP_FOR
    for (int64_t i = 0; i < c->nSubj; ++i) {
        orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_THETA, (uint32_t)i, 0, sweep);
        s->theta[i] = m[i] + sqrt(v[i]) * orc_normal(&st);
    }
    free(m); free(v);
}

/* drawItemDiscrimination, src/Draw.pl.jl:88-93 (mu_a0 = 1, sigma_a0 = 1) */
static void moments_a(const orc_config* c, const orc_data* d, const orc_state* s, double* parM, double* parV)
{
    ORC_OMP_FOR
    for (int j = 0; j < c->nItem; ++j) {
        double sv = 0, sm = 0;
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double e = s->theta[i] - s->b[j];
            sv += e * e * s->omega[IDX(i, j, c->nSubj)];
            sm += ((double)d->Y[IDX(i, j, c->nSubj)] - 0.5) * e;
        }
        parV[j] = 1.0 / (1.0 + sv);
        parM[j] = parV[j] * (1.0 + sm);
    }
}
static void draw_a(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double m[c->nItem], v[c->nItem];
    moments_a(c, d, s, m, v);
    for (int j = 0; j < c->nItem; ++j) {
        orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_A, 0, (uint32_t)j, sweep);
        s->a[j] = orc_truncnorm0(&st, m[j], sqrt(v[j]));
    }
    if (c->onepl) for (int j = 0; j < c->nItem; ++j) s->a[j] = 1.0; /* src/GibbsRtIrt.pl.jl:234-236,303-305 */
}

/* drawItemDifficulty, src/Draw.pl.jl:98-105 (mu_b0 = 0, sigma_b0 = 1, clamp to [-4,4]) */
static void moments_b(const orc_config* c, const orc_data* d, const orc_state* s, double* parM, double* parV)
{
    ORC_OMP_FOR
    for (int j = 0; j < c->nItem; ++j) {
        double sv = 0, sm = 0;
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double w = s->omega[IDX(i, j, c->nSubj)];
            sv += s->a[j] * s->a[j] * w;
            sm += s->a[j] * (((double)d->Y[IDX(i, j, c->nSubj)] - 0.5) - s->theta[i] * s->a[j] * w);
        }
        parV[j] = 1.0 / (1.0 + sv);
        parM[j] = parV[j] * (0.0 - sm);
    }
}
static void draw_b(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double m[c->nItem], v[c->nItem];
    moments_b(c, d, s, m, v);
    for (int j = 0; j < c->nItem; ++j) {
        orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_B, 0, (uint32_t)j, sweep);
        double b = m[j] + sqrt(v[j]) * orc_normal(&st);
        s->b[j] = b < -4.0 ? -4.0 : (b > 4.0 ? 4.0 : b);
    }
}

/* drawSubjSpeed (src/Draw.pl.jl:132-141), drawSubjSpeedLatentQr (:161-174), drawSubjSpeedCrossQr (:192-206),
 * drawSubjSpeedNull (:119-127: prior N(0, 1), Sigp NOT used), drawSubjSpeedLatent (:147-156), drawSubjSpeedCross (:179-187) */
static void moments_zeta(const orc_config* c, const orc_data* d, const orc_state* s, double* parM, double* parV)
{
    const double k1 = k1_of(c->qRt), k2 = k2_of(c->qRt);
    ORC_OMP_FOR
    for (int64_t i = 0; i < c->nSubj; ++i) {
        double mu0 = 0.0, s0 = s->Sigp[3];
        double sv = 0, sm = 0;
        if (c->model == ORC_RTIRT) {
            int p = c->nFeat + 1;
            for (int u = 0; u < p; ++u) mu0 += xrow(c, d, s, i, u, 0) * s->beta[p + u]; /* beta[:,2] */
        } else if (c->model == ORC_LATENTQR) {
            int p = c->nFeat + 2;
            for (int u = 0; u < p; ++u) mu0 += xrow(c, d, s, i, u, 1) * s->beta[u];
            mu0 += k1 * s->nu[i];
            s0 = s->Sigp[3] * (k2 * s->nu[i]);
        } else if (c->model == ORC_LATENT) {
            int p = c->nFeat + 2;
            for (int u = 0; u < p; ++u) mu0 += xrow(c, d, s, i, u, 1) * s->beta[u];
        } else if (c->model == ORC_NULL) {
            s0 = 1.0;
        }
        if (c->model == ORC_CROSS) {
            for (int j = 0; j < c->nItem; ++j) {
                sv += 1.0 / s->sig2t[j];
                sm += (s->lambda[j] - d->logT[IDX(i, j, c->nSubj)] - s->theta[i] * s->rho[j]) / s->sig2t[j];
            }
        } else if (c->model == ORC_CROSSQR) {
            for (int j = 0; j < c->nItem; ++j) {
                double nu = s->nu[IDX(i, j, c->nSubj)];
                double den = s->sig2t[j] * (k2 * nu);
                sv += 1.0 / den;
                sm += (s->lambda[j] - d->logT[IDX(i, j, c->nSubj)] - s->theta[i] * s->rho[j] + k1 * nu) / den;
            }
        } else {
            for (int j = 0; j < c->nItem; ++j) {
                sv += 1.0 / s->sig2t[j];
                sm += (s->lambda[j] - d->logT[IDX(i, j, c->nSubj)]) / s->sig2t[j];
            }
        }
        parV[i] = 1.0 / (1.0 / s0 + sv);
        parM[i] = parV[i] * (mu0 / s0 + sm);
    }
}
static void draw_zeta(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double* m = (double*)malloc(sizeof(double) * c->nSubj), *v = (double*)malloc(sizeof(double) * c->nSubj);
    moments_zeta(c, d, s, m, v);
    ORC_OMP_FOR
    for (int64_t i = 0; i < c->nSubj; ++i) {
        /* one block per subject and sweep feeds both row draws: words 0,1 -> theta's normal, words 2,3 -> zeta's */
        orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_THETA, (uint32_t)i, 0, sweep);
        (void)orc_u32(&st); (void)orc_u32(&st);
        s->zeta[i] = m[i] + sqrt(v[i]) * orc_normal(&st);
    }
    free(m); free(v);
}

/* drawItemIntensity (src/Draw.pl.jl:215-220), drawItemIntensityCrossQr (:239-251), drawItemIntensityCross (:225-231) */
static void moments_lambda(const orc_config* c, const orc_data* d, const orc_state* s, double* parM, double* parV)
{
    double mu, sd; logT_mean_std(c, d, &mu, &sd);
    const double k1 = k1_of(c->qRt), k2 = k2_of(c->qRt);
    ORC_OMP_FOR
    for (int j = 0; j < c->nItem; ++j) {
        if (c->model == ORC_CROSSQR) {
            double sv = 0, sm = 0;
            for (int64_t i = 0; i < c->nSubj; ++i) {
                double nu = s->nu[IDX(i, j, c->nSubj)], den = s->sig2t[j] * (k2 * nu);
                sv += 1.0 / den;
                sm += (d->logT[IDX(i, j, c->nSubj)] + s->zeta[i] + s->theta[i] * s->rho[j] - k1 * nu) / den;
            }
            parV[j] = 1.0 / (1.0 / (sd * sd) + sv);
            parM[j] = parV[j] * (mu / (sd * sd) + sm);
        } else if (c->model == ORC_CROSS) {
            double sm = 0;
            for (int64_t i = 0; i < c->nSubj; ++i) sm += (d->logT[IDX(i, j, c->nSubj)] + s->zeta[i] + s->theta[i] * s->rho[j]) / s->sig2t[j];
            parV[j] = 1.0 / (1.0 / (sd * sd) + (double)c->nSubj / s->sig2t[j]);
            parM[j] = parV[j] * (mu / (sd * sd) + sm);
        } else {
            double sm = 0;
            for (int64_t i = 0; i < c->nSubj; ++i) sm += d->logT[IDX(i, j, c->nSubj)] + s->zeta[i];
            parV[j] = 1.0 / (1.0 / (sd * sd) + (double)c->nSubj / s->sig2t[j]);
            parM[j] = parV[j] * (mu / (sd * sd) + sm / s->sig2t[j]);
        }
    }
}
static void draw_lambda(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double m[c->nItem], v[c->nItem];
    moments_lambda(c, d, s, m, v);
    for (int j = 0; j < c->nItem; ++j) {
        orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_LAMBDA, 0, (uint32_t)j, sweep);
        s->lambda[j] = orc_truncnorm0(&st, m[j], sqrt(v[j]));
    }
}

/* drawItemTimeResidual (src/Draw.pl.jl:257-262), drawItemTimeResidualCrossQr (:278-288), drawItemTimeResidualCross (:267-273);
 * delta_a = delta_b = 1e-3.
 * out: InverseGamma(shape, scale) parameters */
static void moments_sig2t(const orc_config* c, const orc_data* d, const orc_state* s, double* shape, double* scale)
{
    const double k1 = k1_of(c->qRt), k2 = k2_of(c->qRt);
    ORC_OMP_FOR
    for (int j = 0; j < c->nItem; ++j) {
        if (c->model == ORC_CROSSQR) {
            double sq = 0, sn = 0;
            for (int64_t i = 0; i < c->nSubj; ++i) {
                double nu = s->nu[IDX(i, j, c->nSubj)];
                double r = d->logT[IDX(i, j, c->nSubj)] - s->lambda[j] + s->zeta[i] + s->theta[i] * s->rho[j] - k1 * nu;
                sq += r * r / (2.0 * (k2 * nu));
                sn += nu;
            }
            shape[j] = 1e-3 + (double)c->nSubj * 3.0 / 2.0;
            scale[j] = 1e-3 + sq + sn;
        } else if (c->model == ORC_CROSS) {
            double sq = 0;
            for (int64_t i = 0; i < c->nSubj; ++i) { double r = d->logT[IDX(i, j, c->nSubj)] - s->lambda[j] + s->zeta[i] + s->theta[i] * s->rho[j]; sq += r * r; }
            shape[j] = 1e-3 + (double)c->nSubj / 2.0;
            scale[j] = 1e-3 + sq / 2.0;
        } else {
            double sq = 0;
            for (int64_t i = 0; i < c->nSubj; ++i) { double r = d->logT[IDX(i, j, c->nSubj)] - s->lambda[j] + s->zeta[i]; sq += r * r; }
            shape[j] = 1e-3 + (double)c->nSubj / 2.0;
            scale[j] = 1e-3 + sq / 2.0;
        }
    }
}
static void draw_sig2t(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double sh[c->nItem], sc[c->nItem];
    moments_sig2t(c, d, s, sh, sc);
    for (int j = 0; j < c->nItem; ++j) {
        orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_SIG2T, 0, (uint32_t)j, sweep);
        s->sig2t[j] = orc_invgamma(&st, sh[j], sc[j]);
    }
}

/* drawQrWeightsCrossQr (src/Draw.pl.jl:303-320) / drawQrWeightsLatentQr (:325-343):
 * mu = clamp(parB/parA, 1e-10, Inf); nu = clamp(1/IG(mu, parB^2), 1e-10, 1e10) */
static inline double qr_weight(orc_stream* st, double parA, double parB)
{
    double mu = parB / parA;
    if (mu < 1e-10) mu = 1e-10;
    double nu = 1.0 / orc_invgauss(st, mu, parB * parB);
    return nu < 1e-10 ? 1e-10 : (nu > 1e10 ? 1e10 : nu);
}
static void draw_nu(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    const double k1 = k1_of(c->qRt), k2 = k2_of(c->qRt);
    if (c->model == ORC_CROSSQR) {
        for (int j = 0; j < c->nItem; ++j) {
            double parB = sqrt(2.0 * k2 + k1 * k1) / sqrt(s->sig2t[j] * k2);
            ORC_OMP_FOR
            for (int64_t i = 0; i < c->nSubj; ++i) {
                double parA = fabs(d->logT[IDX(i, j, c->nSubj)] - s->lambda[j] + s->zeta[i] + s->theta[i] * s->rho[j]) / sqrt(s->sig2t[j] * k2);
                orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_NU, (uint32_t)i, (uint32_t)j, sweep);
                s->nu[IDX(i, j, c->nSubj)] = qr_weight(&st, parA, parB);
            }
        }
    } else { /* LatentQr */
        int p = c->nFeat + 2;
        double parB = sqrt(2.0 * k2 + k1 * k1) / sqrt(s->Sigp[3] * k2);
        ORC_OMP_FOR
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double xb = 0;
            for (int u = 0; u < p; ++u) xb += xrow(c, d, s, i, u, 1) * s->beta[u];
            double parA = fabs(s->zeta[i] - xb) / sqrt(s->Sigp[3] * k2);
            orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_NU, (uint32_t)i, 0, sweep);
            s->nu[i] = qr_weight(&st, parA, parB);
        }
    }
}

/* getSubjCoefficientsMlIrt, src/Draw.pl.jl:351-357: beta = (x'x) \ x'theta */
static void get_beta_mlirt(const orc_config* c, const orc_data* d, orc_state* s)
{
    int p = c->nFeat + 1;
    double A[p * p], r[p];
    xtx(c, d, s, p, 0, A);
    for (int u = 0; u < p; ++u) { double acc = 0; for (int64_t i = 0; i < c->nSubj; ++i) acc += xrow(c, d, s, i, u, 0) * s->theta[i]; r[u] = acc; }
    solve_spd(p, A, r, s->beta);
    if (!c->intercept) s->beta[0] = 0.0; /* src/GibbsRtIrt.pl.jl:225-227 */
}

/* drawSubjCoefficients, src/Draw.pl.jl:380-393.  Note `1/sigma_b0^2 .+ M` adds 1.0 to EVERY element of
 * the 2p x 2p precision (reference quirk, reproduced). */
static void moments_beta_rtirt(const orc_config* c, const orc_data* d, const orc_state* s, double* parM, double* parV)
{
    int p = c->nFeat + 1, n = 2 * p;
    double A[p * p], xe[p * 2], iO[4], P[n * n], t[n];
    xtx(c, d, s, p, 0, A);
    for (int u = 0; u < p; ++u) {
        double a0 = 0, a1 = 0;
        for (int64_t i = 0; i < c->nSubj; ++i) { double x = xrow(c, d, s, i, u, 0); a0 += x * s->theta[i]; a1 += x * s->zeta[i]; }
        xe[u] = a0; xe[p + u] = a1;
    }
    mat_inverse(2, s->Sigp, iO);
    for (int i1 = 0; i1 < 2; ++i1) for (int j1 = 0; j1 < 2; ++j1)
        for (int i2 = 0; i2 < p; ++i2) for (int j2 = 0; j2 < p; ++j2)
            P[(i1 * p + i2) + (j1 * p + j2) * n] = 1.0 + iO[i1 + j1 * 2] * A[i2 + j2 * p];
    mat_inverse(n, P, parV);
    /* vec(x'eta * invOmega') : M[r,cc] = sum_k xe[r,k] * iO[cc,k] */
    for (int cc = 0; cc < 2; ++cc) for (int r = 0; r < p; ++r) t[cc * p + r] = 0.0 + xe[r] * iO[cc + 0 * 2] + xe[p + r] * iO[cc + 1 * 2];
    for (int i = 0; i < n; ++i) { double acc = 0; for (int j = 0; j < n; ++j) acc += parV[i + j * n] * t[j]; parM[i] = acc; }
}
static void draw_beta_rtirt(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    int p = c->nFeat + 1, n = 2 * p;
    double parM[n], parV[n * n], L[n * n], z[n];
    moments_beta_rtirt(c, d, s, parM, parV);
    /* Symmetric(parV) reads the upper triangle */
    for (int i = 0; i < n; ++i) for (int j = 0; j < i; ++j) parV[i + j * n] = parV[j + i * n];
    chol_lower(n, parV, L);
    orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_BETA, 0, 0, sweep);
    for (int i = 0; i < n; ++i) z[i] = orc_normal(&st);
    for (int i = 0; i < n; ++i) { double acc = parM[i]; for (int j = 0; j <= i; ++j) acc += L[i + j * n] * z[j]; s->beta[i] = acc; }
    if (!c->intercept) { s->beta[0] = 0.0; s->beta[p] = 0.0; } /* src/GibbsRtIrt.pl.jl:293-295 */
}

/* getSubjCoefficientsLatentQr, src/Draw.pl.jl:446-458.  (w (x) x'x) beta = vec(x'(zeta - k1 nu) w') with w an
 * N-vector stacks N scaled copies of the same p x p system; its least-squares solution is
 * beta = (x'x)^-1 x'(zeta - k1 nu), x = [1 X theta]. */
static void get_beta_latentqr(const orc_config* c, const orc_data* d, orc_state* s)
{
    int p = c->nFeat + 2;
    const double k1 = k1_of(c->qRt);
    double A[p * p], r[p];
    xtx(c, d, s, p, 1, A);
    for (int u = 0; u < p; ++u) { double acc = 0; for (int64_t i = 0; i < c->nSubj; ++i) acc += xrow(c, d, s, i, u, 1) * (s->zeta[i] - k1 * s->nu[i]); r[u] = acc; }
    solve_spd(p, A, r, s->beta);
    if (!c->intercept) s->beta[0] = 0.0; /* src/GibbsRtIrtLatent.pl.jl:288-290 */
}

static void cov2one_rescale(double* S) /* src/Draw.pl.jl:507-511 */
{
    double d1 = 1.0 / sqrt(S[0]);
    S[0] *= d1 * d1; S[1] *= d1; S[2] *= d1;
    double d2 = 1.0 / sqrt(S[3]);
    S[3] *= d2 * d2; S[1] *= d2; S[2] *= d2;
    S[0] = 1.0; S[3] = 1.0;
}

/* drawSubjCovariance, src/Draw.pl.jl:499-515: s ~ InverseWishart(N+3, e'e + I2) (inverse of a Bartlett Wishart);
 * drawSubjCovarianceNull, :522-535, is the same draw with e = eta (no regression) */
static void scale_sigp_rtirt(const orc_config* c, const orc_data* d, const orc_state* s, double* Psi)
{
    int p = c->nFeat + 1;
    double e00 = 0, e01 = 0, e11 = 0;
    for (int64_t i = 0; i < c->nSubj; ++i) {
        double m0 = 0, m1 = 0;
        if (c->model != ORC_NULL) for (int u = 0; u < p; ++u) { double x = xrow(c, d, s, i, u, 0); m0 += x * s->beta[u]; m1 += x * s->beta[p + u]; }
        double e0 = s->theta[i] - m0, e1 = s->zeta[i] - m1;
        e00 += e0 * e0; e01 += e0 * e1; e11 += e1 * e1;
    }
    Psi[0] = e00 + 1.0; Psi[1] = e01; Psi[2] = e01; Psi[3] = e11 + 1.0;
}
static void draw_sigp_rtirt(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double Psi[4], Pi[4], L[4];
    scale_sigp_rtirt(c, d, s, Psi);
    mat_inverse(2, Psi, Pi);
    chol_lower(2, Pi, L);
    double df = (double)c->nSubj + 3.0;
    orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_SIGP, 0, 0, sweep);
    double c1 = sqrt(orc_chisq(&st, df));
    double n21 = orc_normal(&st);
    double c2 = sqrt(orc_chisq(&st, df - 1.0));
    /* Z = L * A, A = [c1 0; n21 c2];  W = Z Z' */
    double z00 = L[0] * c1, z10 = L[1] * c1 + L[3] * n21, z11 = L[3] * c2;
    double W[4] = { z00 * z00, z10 * z00, z00 * z10, z10 * z10 + z11 * z11 };
    mat_inverse(2, W, s->Sigp);
    if (c->cov2one) cov2one_rescale(s->Sigp);
}

/* drawSubjCovarianceCross, src/Draw.pl.jl:542-557 */
static void draw_sigp_cross(const orc_config* c, orc_state* s, uint32_t sweep)
{
    double sq = 0;
    for (int64_t i = 0; i < c->nSubj; ++i) sq += s->zeta[i] * s->zeta[i];
    orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_SIGP, 0, 0, sweep);
    double v = orc_invgamma(&st, 1e-3 + (double)c->nSubj / 2.0, 1e-3 + sq / 2.0);
    s->Sigp[0] = 1.0; s->Sigp[1] = 0.0; s->Sigp[2] = 0.0; s->Sigp[3] = v;
    if (c->cov2one) cov2one_rescale(s->Sigp);
}

/* drawSubjCovarianceLatentQr, src/Draw.pl.jl:585-606.  `r.^2 / (2*k2e)` at :594 is a vector/vector matrix
 * division: an N x N rank-one matrix r2 w'/(w'w) whose sum is (sum r2)(sum w)/(sum w^2), w = 2 k2 nu.
 * sigp_mode 0 reproduces that closed form; sigp_mode 1 is the evidently intended sum r_i^2/(2 k2 nu_i). */
static double scale_sigp_latentqr(const orc_config* c, const orc_data* d, const orc_state* s)
{
    int p = c->nFeat + 2;
    const double k1 = k1_of(c->qRt), k2 = k2_of(c->qRt);
    double sr2 = 0, sw = 0, sw2 = 0, sintended = 0, snu = 0;
    for (int64_t i = 0; i < c->nSubj; ++i) {
        double xb = 0;
        for (int u = 0; u < p; ++u) xb += xrow(c, d, s, i, u, 1) * s->beta[u];
        double r = s->zeta[i] - xb - k1 * s->nu[i], w = 2.0 * (k2 * s->nu[i]);
        sr2 += r * r; sw += w; sw2 += w * w; sintended += r * r / w; snu += s->nu[i];
    }
    double q = c->sigp_mode ? sintended : sr2 * sw / sw2;
    return 1e-3 + q + snu;
}
static void draw_sigp_latentqr(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double parB = scale_sigp_latentqr(c, d, s);
    orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_SIGP, 0, 0, sweep);
    double v = orc_invgamma(&st, 1e-3 + (double)c->nSubj * 3.0 / 2.0, parB);
    s->Sigp[0] = 1.0; s->Sigp[1] = 0.0; s->Sigp[2] = 0.0; s->Sigp[3] = v;
    if (c->cov2one) cov2one_rescale(s->Sigp);
}

/* drawSubjCoefficientsLatent, src/Draw.pl.jl:399-416: invO = 1/Sigp[2,2]; parV = inv(1/sb0^2 .+ invO x'x) (the `.+` adds 1 to
 * EVERY element, as in drawSubjCoefficients); parM = parV (0 .+ x'zeta invO); beta = parM + chol(parV).L randn(nFeat+2); x = [1 X theta] */
static void moments_beta_latent(const orc_config* c, const orc_data* d, const orc_state* s, double* parM, double* parV)
{
    int q = c->nFeat + 2;
    double A[q * q], P[q * q], t[q];
    const double iO = 1.0 / s->Sigp[3];
    xtx(c, d, s, q, 1, A);
    for (int u = 0; u < q; ++u) { double acc = 0; for (int64_t i = 0; i < c->nSubj; ++i) acc += xrow(c, d, s, i, u, 1) * s->zeta[i]; t[u] = 0.0 + acc * iO; }
    for (int e = 0; e < q * q; ++e) P[e] = 1.0 + iO * A[e];
    mat_inverse(q, P, parV);
    for (int i = 0; i < q; ++i) { double acc = 0; for (int j = 0; j < q; ++j) acc += parV[i + j * q] * t[j]; parM[i] = acc; }
}
static void draw_beta_latent(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    int q = c->nFeat + 2;
    double parM[q], parV[q * q], L[q * q], z[q];
    moments_beta_latent(c, d, s, parM, parV);
    for (int i = 0; i < q; ++i) for (int j = 0; j < i; ++j) parV[i + j * q] = parV[j + i * q];   /* Symmetric(parV): upper triangle */
    chol_lower(q, parV, L);
    orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_BETA, 0, 0, sweep);
    for (int i = 0; i < q; ++i) z[i] = orc_normal(&st);
    for (int i = 0; i < q; ++i) { double acc = parM[i]; for (int j = 0; j <= i; ++j) acc += L[i + j * q] * z[j]; s->beta[i] = acc; }
    if (!c->intercept) s->beta[0] = 0.0; /* src/GibbsRtIrtLatent.pl.jl:184-186 */
}

/* drawSubjCovarianceLatent, src/Draw.pl.jl:563-579: s ~ InverseGamma(da + N/2, db + sum((zeta - x beta)^2)/2), Sigp = [1 0; 0 s] */
static double scale_sigp_latent(const orc_config* c, const orc_data* d, const orc_state* s)
{
    int q = c->nFeat + 2;
    double sq = 0;
    for (int64_t i = 0; i < c->nSubj; ++i) {
        double xb = 0;
        for (int u = 0; u < q; ++u) xb += xrow(c, d, s, i, u, 1) * s->beta[u];
        sq += (s->zeta[i] - xb) * (s->zeta[i] - xb);
    }
    return 1e-3 + sq / 2.0;
}
static void draw_sigp_latent(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double parB = scale_sigp_latent(c, d, s);
    orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_SIGP, 0, 0, sweep);
    double v = orc_invgamma(&st, 1e-3 + (double)c->nSubj / 2.0, parB);
    s->Sigp[0] = 1.0; s->Sigp[1] = 0.0; s->Sigp[2] = 0.0; s->Sigp[3] = v;
    if (c->cov2one) cov2one_rescale(s->Sigp);
}

/* drawSubjCorrCrossQr, src/Draw.pl.jl:474-489, and drawSubjCorrCross, :463-469 (sigma_rho = 1) */
static void moments_rho(const orc_config* c, const orc_data* d, const orc_state* s, double* parM, double* parV)
{
    const double k1 = k1_of(c->qRt), k2 = k2_of(c->qRt);
    ORC_OMP_FOR
    for (int j = 0; j < c->nItem; ++j) {
        double sv = 0, sm = 0;
        if (c->model == ORC_CROSS) {
            for (int64_t i = 0; i < c->nSubj; ++i) {
                sv += s->theta[i] * s->theta[i] / s->sig2t[j];
                sm += s->theta[i] * (s->lambda[j] - s->zeta[i] - d->logT[IDX(i, j, c->nSubj)]) / s->sig2t[j];
            }
        } else
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double nu = s->nu[IDX(i, j, c->nSubj)], den = s->sig2t[j] * (k2 * nu);
            sv += s->theta[i] * s->theta[i] / den;
            sm += s->theta[i] * (s->lambda[j] - s->zeta[i] - d->logT[IDX(i, j, c->nSubj)] + k1 * nu) / den;
        }
        parV[j] = 1.0 / (1.0 + sv);
        parM[j] = parV[j] * (0.0 + sm);
    }
}
static void draw_rho(const orc_config* c, const orc_data* d, orc_state* s, uint32_t sweep)
{
    double m[c->nItem], v[c->nItem];
    moments_rho(c, d, s, m, v);
    for (int j = 0; j < c->nItem; ++j) {
        orc_stream st = orc_stream_make(c->seed, c->chain, ORC_SITE_RHO, 0, (uint32_t)j, sweep);
        s->rho[j] = m[j] + sqrt(v[j]) * orc_normal(&st);
    }
}

/* ------------------------------------------------------------------ log-likelihoods
 * getLogLikelihoodMlIrt src/GibbsRtIrt.pl.jl:195-204; ...RtIrt :262-272;
 * ...CrossQr src/GibbsRtIrtCross.pl.jl:240-258; ...LatentQr src/GibbsRtIrtLatent.pl.jl:243-264 */
static const double LOG_2PI = 1.8378770664093454836;
static inline double logpdf_normal(double x, double mu, double sd) { double z = (x - mu) / sd; return -0.5 * LOG_2PI - log(sd) - 0.5 * z * z; }

double orc_loglik(const orc_config* c, const orc_data* d, const orc_state* s)
{
    const double k1 = k1_of(c->qRt), k2 = k2_of(c->qRt);
    double lb = 0, lt = 0, ls = 0;
    ORC_PRAGMA(omp parallel for reduction(+:lb,lt) schedule(static) num_threads(orc_threads))
    for (int j = 0; j < c->nItem; ++j)
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double eta = s->a[j] * (s->theta[i] - s->b[j]);
            lb += (d->Y[IDX(i, j, c->nSubj)] ? eta : 0.0) - log1pexp(eta);
            if (c->model == ORC_RTIRT || c->model == ORC_LATENTQR || c->model == ORC_NULL || c->model == ORC_LATENT)
                lt += logpdf_normal(d->logT[IDX(i, j, c->nSubj)], s->lambda[j] - s->zeta[i], sqrt(s->sig2t[j]));
            else if (c->model == ORC_CROSS)   /* src/GibbsRtIrtCross.pl.jl:158-170 */
                lt += logpdf_normal(d->logT[IDX(i, j, c->nSubj)], s->lambda[j] - s->zeta[i] - s->theta[i] * s->rho[j], sqrt(s->sig2t[j]));
            else if (c->model == ORC_CROSSQR) {
                double nu = s->nu[IDX(i, j, c->nSubj)];
                lt += logpdf_normal(d->logT[IDX(i, j, c->nSubj)], s->lambda[j] - s->zeta[i] - s->theta[i] * s->rho[j] + k1 * nu,
                                    sqrt(s->sig2t[j] * (k2 * nu)));
            }
        }
    if (c->model == ORC_MLIRT) {
        int p = c->nFeat + 1;
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double mu = 0; for (int u = 0; u < p; ++u) mu += xrow(c, d, s, i, u, 0) * s->beta[u];
            ls += logpdf_normal(s->theta[i], mu, 1.0);
        }
    } else if (c->model == ORC_RTIRT || c->model == ORC_CROSSQR || c->model == ORC_NULL || c->model == ORC_CROSS) {
        /* Null / Cross: zero mean (src/GibbsRtIrt.pl.jl:351-362, src/GibbsRtIrtCross.pl.jl:158-170) */
        int p = c->nFeat + 1;
        double Si[4]; mat_inverse(2, s->Sigp, Si);
        double logdet = log(s->Sigp[0] * s->Sigp[3] - s->Sigp[1] * s->Sigp[2]);
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double m0 = 0, m1 = 0;
            if (c->model == ORC_RTIRT) for (int u = 0; u < p; ++u) { double x = xrow(c, d, s, i, u, 0); m0 += x * s->beta[u]; m1 += x * s->beta[p + u]; }
            double e0 = s->theta[i] - m0, e1 = s->zeta[i] - m1;
            ls += -LOG_2PI - 0.5 * logdet - 0.5 * (e0 * (Si[0] * e0 + Si[2] * e1) + e1 * (Si[1] * e0 + Si[3] * e1));
        }
    } else if (c->model == ORC_LATENT) { /* src/GibbsRtIrtLatent.pl.jl:151-162 */
        int p = c->nFeat + 2;
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double mu = 0; for (int u = 0; u < p; ++u) mu += xrow(c, d, s, i, u, 1) * s->beta[u];
            ls += logpdf_normal(s->zeta[i], mu, sqrt(s->Sigp[3]));
        }
    } else { /* LatentQr */
        int p = c->nFeat + 2;
        for (int64_t i = 0; i < c->nSubj; ++i) {
            double mu = 0; for (int u = 0; u < p; ++u) mu += xrow(c, d, s, i, u, 1) * s->beta[u];
            ls += logpdf_normal(s->zeta[i], mu + k1 * s->nu[i], sqrt(s->Sigp[3] * (k2 * s->nu[i])));
        }
    }
    return lb + lt + ls;
}

/* ------------------------------------------------------------------ sweeps (reference order) */
static void sweep_once(const orc_config* c, const orc_data* d, orc_state* s, uint32_t t)
{
    switch (c->model) {
    case ORC_MLIRT: /* src/GibbsRtIrt.pl.jl:221-246: beta -> omega -> a -> b -> theta */
        get_beta_mlirt(c, d, s);
        draw_omega(c, s, t);
        draw_a(c, d, s, t);
        draw_b(c, d, s, t);
        draw_theta(c, d, s, 1, t);
        break;
    case ORC_RTIRT: /* src/GibbsRtIrt.pl.jl:289-324: beta -> Sigp -> omega -> b -> a -> theta -> lambda -> sig2t -> zeta */
        draw_beta_rtirt(c, d, s, t);
        draw_sigp_rtirt(c, d, s, t);
        draw_omega(c, s, t);
        draw_b(c, d, s, t);
        draw_a(c, d, s, t);
        draw_theta(c, d, s, 1, t);
        draw_lambda(c, d, s, t);
        draw_sig2t(c, d, s, t);
        draw_zeta(c, d, s, t);
        break;
    case ORC_CROSSQR: /* src/GibbsRtIrtCross.pl.jl:276-302: nu -> rho -> Sigp -> omega -> b -> a -> theta(Null) -> lambda -> sig2t -> zeta */
        draw_nu(c, d, s, t);
        draw_rho(c, d, s, t);
        draw_sigp_cross(c, s, t);
        draw_omega(c, s, t);
        draw_b(c, d, s, t);
        draw_a(c, d, s, t);
        draw_theta(c, d, s, 0, t);
        draw_lambda(c, d, s, t);
        draw_sig2t(c, d, s, t);
        draw_zeta(c, d, s, t);
        break;
    case ORC_LATENTQR: /* src/GibbsRtIrtLatent.pl.jl:282-314: nu -> beta(get) -> Sigp -> omega -> b -> a -> theta(Null) -> lambda -> sig2t -> zeta */
        draw_nu(c, d, s, t);
        get_beta_latentqr(c, d, s);
        draw_sigp_latentqr(c, d, s, t);
        draw_omega(c, s, t);
        draw_b(c, d, s, t);
        draw_a(c, d, s, t);
        draw_theta(c, d, s, 0, t);
        draw_lambda(c, d, s, t);
        draw_sig2t(c, d, s, t);
        draw_zeta(c, d, s, t);
        break;
    case ORC_NULL: /* src/GibbsRtIrt.pl.jl:378-404: beta = 0 -> Sigp(Null) -> omega -> b -> a -> theta(Null) -> lambda -> sig2t -> zeta(Null) */
        memset(s->beta, 0, sizeof(double) * 2 * (c->nFeat + 1));
        draw_sigp_rtirt(c, d, s, t);
        draw_omega(c, s, t);
        draw_b(c, d, s, t);
        draw_a(c, d, s, t);
        draw_theta(c, d, s, 0, t);
        draw_lambda(c, d, s, t);
        draw_sig2t(c, d, s, t);
        draw_zeta(c, d, s, t);
        break;
    case ORC_CROSS: /* src/GibbsRtIrtCross.pl.jl:187-213: rho -> Sigp -> omega -> b -> a -> theta(Null) -> lambda -> sig2t -> zeta */
        draw_rho(c, d, s, t);
        draw_sigp_cross(c, s, t);
        draw_omega(c, s, t);
        draw_b(c, d, s, t);
        draw_a(c, d, s, t);
        draw_theta(c, d, s, 0, t);
        draw_lambda(c, d, s, t);
        draw_sig2t(c, d, s, t);
        draw_zeta(c, d, s, t);
        break;
    case ORC_LATENT: /* src/GibbsRtIrtLatent.pl.jl:179-211: beta(draw) -> Sigp -> omega -> b -> a -> theta(Null) -> lambda -> sig2t -> zeta */
        draw_beta_latent(c, d, s, t);
        draw_sigp_latent(c, d, s, t);
        draw_omega(c, s, t);
        draw_b(c, d, s, t);
        draw_a(c, d, s, t);
        draw_theta(c, d, s, 0, t);
        draw_lambda(c, d, s, t);
        draw_sig2t(c, d, s, t);
        draw_zeta(c, d, s, t);
        break;
    }
}

int orc_qr_width(const orc_config* c, int with_nu)
{
    switch (c->model) {
    case ORC_MLIRT: return c->nFeat + 1;                                   /* src/GibbsRtIrt.pl.jl:45 */
    case ORC_RTIRT: return 2 * (c->nFeat + 1) + 4;                         /* :67 */
    case ORC_CROSSQR: return c->nItem + 4 + (with_nu ? (int)(c->nSubj * c->nItem) : 0); /* src/GibbsRtIrtCross.pl.jl:65 */
    case ORC_LATENTQR: return c->nFeat + 2 + 4 + (with_nu ? (int)c->nSubj : 0);          /* src/GibbsRtIrtLatent.pl.jl:60 */
    case ORC_NULL: return 2 * (c->nFeat + 1) + 4;                          /* OutputPost, src/GibbsRtIrt.pl.jl:67 */
    case ORC_CROSS: return c->nItem + 4;                                   /* OutputPostCross, src/GibbsRtIrtCross.pl.jl:36-50 */
    case ORC_LATENT: return c->nFeat + 2 + 4;                              /* OutputPostRtIrtLatent, src/GibbsRtIrtLatent.pl.jl:33-47 */
    }
    return 0;
}

/* Run sweeps sweep0+1 .. sweep0+nsweeps.  Traces (any may be NULL) are [sweep][param] row-major:
 * ra = [theta; a; b], rt = [zeta; lambda; sig2t], qr per model (src/GibbsRtIrt.pl.jl:241-242,319-321 etc.),
 * ll = log-likelihood after each sweep. */
int orc_run(const orc_config* c, const orc_data* d, orc_state* s, int64_t sweep0, int64_t nsweeps,
            double* tr_ra, double* tr_rt, double* tr_qr, int qr_with_nu, double* tr_ll)
{
    const int64_t N = c->nSubj; const int J = c->nItem;
    const int wq = orc_qr_width(c, qr_with_nu);
    for (int64_t k = 0; k < nsweeps; ++k) {
        uint32_t t = (uint32_t)(sweep0 + k + 1);
        sweep_once(c, d, s, t);
        if (tr_ra) { double* r = tr_ra + (size_t)k * (N + 2 * J); memcpy(r, s->theta, sizeof(double) * N); memcpy(r + N, s->a, sizeof(double) * J); memcpy(r + N + J, s->b, sizeof(double) * J); }
        if (tr_rt && c->model != ORC_MLIRT) { double* r = tr_rt + (size_t)k * (N + 2 * J); memcpy(r, s->zeta, sizeof(double) * N); memcpy(r + N, s->lambda, sizeof(double) * J); memcpy(r + N + J, s->sig2t, sizeof(double) * J); }
        if (tr_qr) {
            double* r = tr_qr + (size_t)k * wq; int o = 0;
            if (c->model == ORC_MLIRT) { memcpy(r, s->beta, sizeof(double) * (c->nFeat + 1)); }
            else if (c->model == ORC_RTIRT || c->model == ORC_NULL) { int nb = 2 * (c->nFeat + 1); memcpy(r, s->beta, sizeof(double) * nb); memcpy(r + nb, s->Sigp, sizeof(double) * 4); }
            else if (c->model == ORC_CROSSQR || c->model == ORC_CROSS) { memcpy(r, s->rho, sizeof(double) * J); o = J; memcpy(r + o, s->Sigp, sizeof(double) * 4); o += 4; if (qr_with_nu && c->model == ORC_CROSSQR) memcpy(r + o, s->nu, sizeof(double) * N * J); }
            else { int nb = c->nFeat + 2; memcpy(r, s->beta, sizeof(double) * nb); o = nb; memcpy(r + o, s->Sigp, sizeof(double) * 4); o += 4; if (qr_with_nu && c->model == ORC_LATENTQR) memcpy(r + o, s->nu, sizeof(double) * N); }
        }
        if (tr_ll) tr_ll[k] = orc_loglik(c, d, s);
    }
    return 0;
}

/* ------------------------------------------------------------------ exported unit hooks for tests */
void orc_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) { orc_philox4x32_10(ctr, key, out); }

/* which: 0 u32->unif, 1 normal, 2 expo, 3 pg1(par0[k]), 4 invgauss(par0[k], par1[k]), 5 truncnorm0(par0,par1),
 * 6 gamma(par0), 7 pg tail weight(par0) (no rng), 8 qr_weight(parA=par0, parB=par1), 9 ndtri(par0) (no rng)
 * element k uses stream (seed, chain 0, site, i = k, j = 0, sweep) */
void orc_sample_batch(int which, uint64_t seed, int site, uint32_t sweep, int64_t n, const double* par0, const double* par1, double* out)
{
    for (int64_t k = 0; k < n; ++k) {
        orc_stream st = orc_stream_make(seed, 0, site, (uint32_t)k, 0, sweep);
        switch (which) {
        case 0: out[k] = orc_unif(&st); break;
        case 1: out[k] = orc_normal(&st); break;
        case 2: out[k] = orc_expo(&st); break;
        case 3: out[k] = orc_pg1(&st, par0[k]); break;
        case 4: out[k] = orc_invgauss(&st, par0[k], par1[k]); break;
        case 5: out[k] = orc_truncnorm0(&st, par0[k], par1[k]); break;
        case 6: out[k] = orc_gamma(&st, par0[k]); break;
        case 7: out[k] = orc_pg_tail_weight(par0[k]); break;
        case 8: out[k] = qr_weight(&st, par0[k], par1[k]); break;
        case 9: out[k] = orc_ndtri(par0[k]); break;
        }
    }
}

/* n draws of GIG(p, a, b); element k uses stream (seed, chain 0, site, i = k, j = 0, sweep) */
void orc_sample_gig(uint64_t seed, int site, uint32_t sweep, int64_t n, double p, double a, double b, double* out)
{
    for (int64_t k = 0; k < n; ++k) {
        orc_stream st = orc_stream_make(seed, 0, site, (uint32_t)k, 0, sweep);
        out[k] = orc_gig(&st, p, a, b);
    }
}

/* conditional moments for numpy cross-checks.  which: 0 theta(prior=x*beta) 1 theta(Null) 2 a 3 b 4 zeta 5 lambda
 * 6 sig2t(shape,scale) 7 rho 8 beta_rtirt (parM[2p], parV[2p*2p]) 9 Sigp scale matrix Psi (out1[4]) 10 latentqr Sigp scale (out1[1]) */
void orc_moments(const orc_config* c, const orc_data* d, const orc_state* s, int which, double* out1, double* out2)
{
    switch (which) {
    case 0: moments_theta(c, d, s, 1, out1, out2); break;
    case 1: moments_theta(c, d, s, 0, out1, out2); break;
    case 2: moments_a(c, d, s, out1, out2); break;
    case 3: moments_b(c, d, s, out1, out2); break;
    case 4: moments_zeta(c, d, s, out1, out2); break;
    case 5: moments_lambda(c, d, s, out1, out2); break;
    case 6: moments_sig2t(c, d, s, out1, out2); break;
    case 7: moments_rho(c, d, s, out1, out2); break;
    case 8: moments_beta_rtirt(c, d, s, out1, out2); break;
    case 9: scale_sigp_rtirt(c, d, s, out1); break;
    case 10: out1[0] = scale_sigp_latentqr(c, d, s); break;
    case 11: moments_beta_latent(c, d, s, out1, out2); break;
    case 12: out1[0] = scale_sigp_latent(c, d, s); break;
    }
}

/* single draw steps, for fine-grained device parity tests.  step ids follow sweep_once. */
void orc_step(const orc_config* c, const orc_data* d, orc_state* s, int step, uint32_t t)
{
    switch (step) {
    case 0: draw_omega(c, s, t); break;
    case 1: draw_theta(c, d, s, 1, t); break;
    case 2: draw_theta(c, d, s, 0, t); break;
    case 3: draw_a(c, d, s, t); break;
    case 4: draw_b(c, d, s, t); break;
    case 5: draw_zeta(c, d, s, t); break;
    case 6: draw_lambda(c, d, s, t); break;
    case 7: draw_sig2t(c, d, s, t); break;
    case 8: draw_nu(c, d, s, t); break;
    case 9: draw_rho(c, d, s, t); break;
    case 10: draw_beta_rtirt(c, d, s, t); break;
    case 11: draw_sigp_rtirt(c, d, s, t); break;
    case 12: get_beta_mlirt(c, d, s); break;
    case 13: get_beta_latentqr(c, d, s); break;
    case 14: draw_sigp_latentqr(c, d, s, t); break;
    case 15: draw_sigp_cross(c, s, t); break;
    case 16: draw_beta_latent(c, d, s, t); break;
    case 17: draw_sigp_latent(c, d, s, t); break;
    }
}


/* ---------------------------------------------------------------------------------------------------------------------
 * Synthetic data: the five generators of /root/reference/src/SimTools.jl restated with the DEVICE's stream addressing
 * (csrc/erm_kernels.hpp gen_kernel), so that erm_simulate_data can be compared value by value.
 *   subject stream (DATA_SUBJ = 12, i, 0, sweep 0), consumed in this order: the nFeat covariates X[i, f] ~ N(0,1)
 *   (setDataMlIrt: X[i, 1] ~ Bernoulli(1/2) instead -- its uniform is drawn AFTER that column's unused normal), then z0, z1 ~ N(0,1),
 *   then (Latent only) the zeta noise;  cell stream (DATA_CELL = 13, i, j, sweep 0): the uniform of Y[i, j], then the logT variate(s).
 *   gen 0  setDataMlIrt        src/SimTools.jl:349-368   theta = X beta + z0 (trueStd = 1)
 *   gen 1  setDataRtIrt        src/SimTools.jl:149-178   (theta, zeta) = X beta + L (z0, z1), L = chol(Sigp); logT ~ N(lambda_j - zeta_i, sig2t_j) truncated to (0, inf)
 *   gen 2  setDataRtIrtNull    src/SimTools.jl:117-144   (theta, zeta) = L (z0, z1); logT as gen 1
 *   gen 3  setDataRtIrtCross   src/SimTools.jl:220-255   (theta, zeta) = L (z0, z1); logT = lambda_j - zeta_i - theta_i rho_j + e,
 *                                                        e ~ N(0, 0.3) ("norm") | t_5 ("tail") | Gamma(1/2, 1) - 1 ("skew")
 *   gen 4  setDataRtIrtLatent  src/SimTools.jl:304-343   theta = z0; zeta = [X theta] beta + e (same three types); logT = lambda_j - zeta_i + N(0,1)
 * Y[i, j] ~ Bernoulli(logistic(a_j (theta_i - b_j))) for every generator (BernoulliLogit).  Like the host generators this is the
 * reference's DISTRIBUTION; Julia's Random.seed! stream cannot be reproduced.
 * beta: gen 0 [nFeat]; gen 1 [nFeat][2] column-major (theta column, then zeta column); gen 4 [nFeat + 1].
 * Outputs column-major: X [N x nFeat], Y [N x J] bytes, logT [N x J]; theta, zeta [N].
 * --------------------------------------------------------------------------------------------------------------------- */
enum { ORC_SITE_DATA_SUBJ = 12, ORC_SITE_DATA_CELL = 13 };

static double gen_noise(orc_stream* s, int kind)        /* src/SimTools.jl:238-247, 322-328 */
{
    if (kind == 0) return 0.3 * orc_normal(s);                                             /* Normal(0, 0.3) */
    if (kind == 1) { double zn = orc_normal(s); return zn / sqrt(orc_chisq(s, 5.0) / 5.0); }   /* TDist(5) */
    double u = orc_unif(s);                                                                /* Gamma(1/2, 1) - 1: Gamma(a) = Gamma(a + 1) U^(1/a) */
    return orc_gamma(s, 1.5) * u * u - 1.0;
}

void orc_simulate_data(int gen, int noise, uint64_t seed, int64_t N, int J, int F,
                       const double* a, const double* b, const double* lambda, const double* sig2t, const double* rho,
                       const double* Sigp /* vec, may be NULL = I */, const double* beta,
                       double* X, double* theta, double* zeta, uint8_t* Y, double* logT)
{
    double L0 = 1.0, L1 = 0.0, L2 = 1.0;
    if (Sigp) { L0 = sqrt(Sigp[0]); L1 = Sigp[1] / L0; L2 = sqrt(Sigp[3] - L1 * L1); }
    ORC_OMP_FOR
    for (int64_t i = 0; i < N; ++i) {
        orc_stream ss = orc_stream_make(seed, 0, ORC_SITE_DATA_SUBJ, (uint32_t)i, 0u, 0u);
        double mt = 0.0, mz = 0.0;
        for (int f = 0; f < F; ++f) {
            double x = orc_normal(&ss);
            if (gen == 0 && f == 0) x = orc_unif(&ss) < 0.5 ? 1.0 : 0.0;
            X[(size_t)f * N + i] = x;
            if (gen == 0) mt += x * beta[f];
            else if (gen == 1) { mt += x * beta[f]; mz += x * beta[F + f]; }
            else if (gen == 4) mz += x * beta[f];
        }
        double z0 = orc_normal(&ss), z1 = orc_normal(&ss), th, ze;
        if (gen == 0) { th = mt + z0; ze = 0.0; }
        else if (gen == 4) { th = z0; ze = mz + th * beta[F] + gen_noise(&ss, noise); }
        else { th = mt + L0 * z0; ze = mz + L1 * z0 + L2 * z1; }
        theta[i] = th; zeta[i] = ze;
        for (int j = 0; j < J; ++j) {
            orc_stream sc = orc_stream_make(seed, 0, ORC_SITE_DATA_CELL, (uint32_t)i, (uint32_t)j, 0u);
            double eta = a[j] * (th - b[j]);
            Y[(size_t)j * N + i] = orc_unif(&sc) < 1.0 / (1.0 + exp(-eta)) ? 1 : 0;
            if (gen == 0) continue;
            double lt;
            if (gen == 1 || gen == 2) lt = orc_truncnorm0(&sc, lambda[j] - ze, sqrt(sig2t[j]));
            else if (gen == 3) lt = lambda[j] - ze - th * rho[j] + gen_noise(&sc, noise);
            else lt = lambda[j] - ze + orc_normal(&sc);
            logT[(size_t)j * N + i] = lt;
        }
    }
}
