/*
 * erm_cli.c -- plain-C driver of libertirt.so (SURVEY.md 8(b) "non-Julia driver"): everything a host needs goes through
 * include/ertirt.h with plain pointers, exactly as the Julia shim's ccall would.  Generates a small synthetic data set in the
 * style of setDataRtIrt (/root/reference/src/SimTools.jl:149-178), runs sample!'s loop on the GPU and prints posterior
 * summaries and recovery statistics.
 *
 *   erm_cli [--model mlirt|rtirt|crossqr|latentqr|null|cross|latent] [--nsubj N] [--nitem J] [--nfeat F] [--niter K] [--nchain C]
 *           [--seed S] [--precision f32|f64]  [--device D] [--qrt Q] [--dry-run]
 *
 * Build: gcc -O2 -I include tools/erm_cli.c -L extendedrtirtmodeling.jl_amd -lertirt -lm -Wl,-rpath,'$ORIGIN/../extendedrtirtmodeling.jl_amd' -o tools/erm_cli
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "ertirt.h"

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double unif(void)   /* splitmix64 -> (0,1) */
{
    uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return ((double)(z >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static double gauss(void) { return sqrt(-2.0 * log(unif())) * cos(6.283185307179586 * unif()); }
/* N(m, s) truncated to (0, inf): plain rejection near the bulk, Robert's (1995) exponential proposal in the tail (a rejection loop
 * on the untruncated normal would spin ~1/Phi(m/s) times for a fast subject on a slow item) */
static double pos_gauss(double m, double s)
{
    const double a = -m / s;                 /* standardised lower bound */
    if (a < 0.5) { double v; do v = m + s * gauss(); while (v <= 0.0); return v; }
    const double al = 0.5 * (a + sqrt(a * a + 4.0));
    for (;;) {
        const double z = a - log(unif()) / al;
        if (unif() < exp(-0.5 * (z - al) * (z - al))) return m + s * z;
    }
}

static double corr(const double* x, const double* y, long n)
{
    double mx = 0, my = 0, sxx = 0, syy = 0, sxy = 0;
    for (long i = 0; i < n; ++i) { mx += x[i]; my += y[i]; }
    mx /= n; my /= n;
    for (long i = 0; i < n; ++i) { sxx += (x[i] - mx) * (x[i] - mx); syy += (y[i] - my) * (y[i] - my); sxy += (x[i] - mx) * (y[i] - my); }
    return sxy / sqrt(sxx * syy);
}
static double rmse(const double* x, const double* y, long n)
{
    double s = 0;
    for (long i = 0; i < n; ++i) s += (x[i] - y[i]) * (x[i] - y[i]);
    return sqrt(s / n);
}

#define CHECK(call) do { int rc_ = (call); if (rc_ != 0) { fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, erm_last_error()); return 2; } } while (0)

int main(int argc, char** argv)
{
    const char* model = "rtirt"; const char* prec = "f32"; int dry = 0;
    long N = 2000; int J = 15, F = 3, niter = 400, nchain = 1, device = 0; uint64_t seed = 1234; double qrt = 0.5;
    for (int k = 1; k < argc; ++k) {
        if (!strcmp(argv[k], "--version")) { printf("%s\n", erm_version()); return 0; }
        if (!strcmp(argv[k], "--dry-run")) { dry = 1; continue; }        /* generate the data, print its summary, do not touch the GPU */
        if (!strcmp(argv[k], "--help") || k + 1 >= argc) { printf("usage: erm_cli [--model m] [--nsubj N] [--nitem J] [--nfeat F] [--niter K] [--nchain C] [--seed S] [--precision f32|f64] [--device D] [--qrt Q] | --version\n"); return !strcmp(argv[k], "--help") ? 0 : 1; }
        const char* v = argv[k + 1];
        if (!strcmp(argv[k], "--model")) model = v;
        else if (!strcmp(argv[k], "--nsubj")) N = atol(v);
        else if (!strcmp(argv[k], "--nitem")) J = atoi(v);
        else if (!strcmp(argv[k], "--nfeat")) F = atoi(v);
        else if (!strcmp(argv[k], "--niter")) niter = atoi(v);
        else if (!strcmp(argv[k], "--nchain")) nchain = atoi(v);
        else if (!strcmp(argv[k], "--seed")) seed = strtoull(v, NULL, 10);
        else if (!strcmp(argv[k], "--precision")) prec = v;
        else if (!strcmp(argv[k], "--device")) device = atoi(v);
        else if (!strcmp(argv[k], "--qrt")) qrt = atof(v);
        else { fprintf(stderr, "unknown option %s\n", argv[k]); return 1; }
        ++k;
    }
    int mid = !strcmp(model, "mlirt") ? ERM_MODEL_MLIRT : !strcmp(model, "rtirt") ? ERM_MODEL_RTIRT
            : !strcmp(model, "crossqr") ? ERM_MODEL_CROSSQR : !strcmp(model, "latentqr") ? ERM_MODEL_LATENTQR
            : !strcmp(model, "null") ? ERM_MODEL_NULL : !strcmp(model, "cross") ? ERM_MODEL_CROSS : !strcmp(model, "latent") ? ERM_MODEL_LATENT : -1;
    if (mid < 0) { fprintf(stderr, "unknown model %s\n", model); return 1; }
    rng_state ^= seed;

    /* ---- true parameters and data (column-major, as Julia hands them over) */
    double* a = malloc(sizeof(double) * J), *b = malloc(sizeof(double) * J), *lam = malloc(sizeof(double) * J), *sg = malloc(sizeof(double) * J);
    double* beta = malloc(sizeof(double) * 2 * (F > 0 ? F : 1));
    double* X = malloc(sizeof(double) * N * (F > 0 ? F : 1));
    double* theta = malloc(sizeof(double) * N), *zeta = malloc(sizeof(double) * N);
    uint8_t* Y = malloc((size_t)N * J); double* logT = malloc(sizeof(double) * N * J);
    for (int j = 0; j < J; ++j) { a[j] = pos_gauss(1.0, 0.2); b[j] = 0.5 * gauss(); lam[j] = pos_gauss(4.0, 0.2); sg[j] = exp(log(0.3) + 0.2 * gauss()); }
    for (int f = 0; f < 2 * F; ++f) beta[f] = (mid == ERM_MODEL_NULL) ? 0.0 : gauss();    /* the Null model has no covariate effects */
    for (long e = 0; e < N * F; ++e) X[e] = gauss();
    for (long i = 0; i < N; ++i) {
        double mt = 0, mz = 0;
        for (int f = 0; f < F; ++f) { mt += X[i + (long)f * N] * beta[f]; mz += X[i + (long)f * N] * beta[F + f]; }
        /* only MlIrt / RtIrt regress theta on X; the other models fix theta's scale at N(0, 1) (drawSubjAbilityNull) */
        theta[i] = ((mid == ERM_MODEL_MLIRT || mid == ERM_MODEL_RTIRT) ? mt : 0.0) + gauss(); zeta[i] = mz + gauss();
    }
    for (int j = 0; j < J; ++j)
        for (long i = 0; i < N; ++i) {
            const double eta = a[j] * (theta[i] - b[j]);
            Y[i + (long)j * N] = unif() < 1.0 / (1.0 + exp(-eta));
            logT[i + (long)j * N] = pos_gauss(lam[j] - zeta[i], sqrt(sg[j]));
        }

    if (dry) {
        double sy = 0, st = 0;
        for (long e = 0; e < N * J; ++e) { sy += Y[e]; st += logT[e]; }
        printf("dry run      mean(Y)=%.4f mean(logT)=%.4f\n", sy / (double)(N * J), st / (double)(N * J));
        return 0;
    }
    /* ---- the sample! path through the C ABI */
    erm_config cfg; memset(&cfg, 0, sizeof cfg);
    cfg.model = mid; cfg.n_item = J; cfg.n_subj = N; cfg.n_feat = (mid == ERM_MODEL_CROSSQR || mid == ERM_MODEL_CROSS) ? 0 : F; cfg.n_iter = niter; cfg.n_chain = nchain;
    cfg.n_burnin = (int)floor(niter / 2.0 + 0.5);
    cfg.cov2one = (mid == ERM_MODEL_LATENTQR || mid == ERM_MODEL_LATENT) ? 0 : 1; cfg.q_rt = qrt; cfg.seed = seed; cfg.device = device;
    cfg.precision = !strcmp(prec, "f64") ? ERM_PREC_F64 : ERM_PREC_F32; cfg.trace_mode = ERM_TRACE_SUMMARY;
    erm_handle h = NULL;
    CHECK(erm_create(&cfg, &h));
    CHECK(erm_set_data(h, Y, mid == ERM_MODEL_MLIRT ? NULL : logT, cfg.n_feat ? X : NULL));
    /* initial values as the constructors' setInitialValues: theta, zeta ~ N(0,1), a = 1, b = 0, lambda = 0, sig2t = 1 */
    double* th0 = malloc(sizeof(double) * N), *ze0 = malloc(sizeof(double) * N), *one = malloc(sizeof(double) * J), *zero = calloc(J, sizeof(double));
    double* nu0 = malloc(sizeof(double) * (mid == ERM_MODEL_CROSSQR ? (size_t)N * J : (size_t)N));
    for (long i = 0; i < N; ++i) { th0[i] = gauss(); ze0[i] = gauss(); }
    for (int j = 0; j < J; ++j) one[j] = 1.0;
    for (size_t e = 0; e < (mid == ERM_MODEL_CROSSQR ? (size_t)N * J : (size_t)N); ++e) nu0[e] = 1.0;
    double sigp0[4] = { 1, 0, 0, 1 }; double beta0[64] = { 0 };
    erm_state st; memset(&st, 0, sizeof st);
    st.theta = th0; st.a = one; st.b = zero; st.zeta = ze0; st.lambda = zero; st.sig2t = one; st.sigp = sigp0; st.beta = beta0; st.rho = zero;
    if (mid == ERM_MODEL_CROSSQR || mid == ERM_MODEL_LATENTQR) st.nu = nu0;
    CHECK(erm_set_state(h, &st));
    CHECK(erm_run(h, (int64_t)niter * nchain));
    erm_timing tm; CHECK(erm_get_timing(h, &tm));

    double* mth = malloc(sizeof(double) * N), *mze = malloc(sizeof(double) * N), *ma = malloc(sizeof(double) * J), *mb = malloc(sizeof(double) * J);
    double* ml = malloc(sizeof(double) * J), *ms = malloc(sizeof(double) * J); double mbeta[64], msig[4];
    erm_state mean; memset(&mean, 0, sizeof mean);
    mean.theta = mth; mean.zeta = mze; mean.a = ma; mean.b = mb; mean.lambda = ml; mean.sig2t = ms; mean.beta = mbeta; mean.sigp = msig;
    CHECK(erm_get_mean(h, &mean));
    printf("library      %s\n", erm_version());
    printf("model        %s  nSubj=%ld nItem=%d nFeat=%d nIter=%d nChain=%d precision=%s\n", model, N, J, cfg.n_feat, niter, nchain, prec);
    printf("device time  %.3f ms for %lld sweeps (%.1f us/sweep, %.3g cell-updates/s)\n", tm.run_ms, (long long)tm.sweeps,
           1e3 * tm.run_ms / (double)tm.sweeps, (double)N * J * (double)tm.sweeps / (tm.run_ms * 1e-3));
    printf("post rows    %lld\n", (long long)erm_post_count(h));
    printf("recovery     cor(theta)=%.3f rmse(a)=%.3f rmse(b)=%.3f", corr(theta, mth, N), rmse(a, ma, J), rmse(b, mb, J));
    if (mid != ERM_MODEL_MLIRT) printf(" cor(zeta)=%.3f rmse(lambda)=%.3f rmse(sig2t)=%.3f", corr(zeta, mze, N), rmse(lam, ml, J), rmse(sg, ms, J));
    printf("\n");
    erm_destroy(h);
    return 0;
}
