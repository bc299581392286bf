/*
 * oracle/orc_rng.h -- TEST INFRASTRUCTURE ONLY (CPU oracle; "parity unpinned", see erm_oracle.c header).
 *
 * Counter-based random streams and the scalar samplers behind every full conditional of
 * /root/reference/src/Draw.pl.jl.  The reference takes these from un-vendored Julia packages
 * (PolyaGammaSamplers 0.1, Distributions 0.25, Random stdlib; Project.toml:14-15,37-38) whose
 * sources are absent from /root/reference, and Julia's Xoshiro stream cannot be reproduced
 * outside Julia.  They are therefore restated from the published algorithms:
 *
 *   Philox4x32-10      Salmon, Moraes, Dror, Shaw (SC'11); constants as in Random123 / rocRAND.
 *   PG(1,c)            Polson, Scott & Windle (2013) Alg. for J*(1,z), truncation t = 0.64
 *                      (call site src/Draw.pl.jl:38, PolyaGammaPSWSampler(1, eta)).
 *   IG(mu,lambda)      Michael, Schucany & Haas (1976)   (call sites src/Draw.pl.jl:312,335)
 *   TN(m,s;0,inf)      rejection from N(0,1) / Robert (1995) exponential tail
 *                      (call sites src/Draw.pl.jl:91,218,248)
 *   Gamma(shape>=1)    Marsaglia & Tsang (2000)          (InverseGamma: src/Draw.pl.jl:260,286,546,596)
 *
 * A "stream" is the sequence of 32-bit words philox(key; c0=i, c1=j, c2=sweep,
 * c3 = site<<24 | chain<<16 | k) for k = 0,1,2,...; words are consumed strictly in order, so the
 * HIP kernels (which restate the same definitions on the device) consume the same variates for
 * the same (site, i, j, sweep) regardless of launch geometry.
 */
#ifndef ORC_RNG_H
#define ORC_RNG_H
#include <stdint.h>
#include <math.h>

#define ORC_PI 3.14159265358979323846

/* draw-site ids (shared by oracle and device; part of the sampling specification) */
enum {
    ORC_SITE_OMEGA = 1, ORC_SITE_THETA = 2, ORC_SITE_ZETA = 3, ORC_SITE_NU = 4,
    ORC_SITE_B = 5, ORC_SITE_A = 6, ORC_SITE_LAMBDA = 7, ORC_SITE_SIG2T = 8,
    ORC_SITE_BETA = 9, ORC_SITE_SIGP = 10, ORC_SITE_RHO = 11, ORC_SITE_TEST = 15
};

static inline void orc_philox4x32_10(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

typedef struct {
    uint32_t key[2];
    uint32_t c0, c1, c2, c3base;
    uint32_t k;      /* next block index */
    uint32_t buf[4];
    int pos;         /* next unread word in buf (4 = empty) */
} orc_stream;

static inline orc_stream orc_stream_make(uint64_t seed, int chain, int site, uint32_t i, uint32_t j, uint32_t sweep)
{
    orc_stream s;
    s.key[0] = (uint32_t)seed; s.key[1] = (uint32_t)(seed >> 32);
    s.c0 = i; s.c1 = j; s.c2 = sweep;
    s.c3base = ((uint32_t)site << 24) | (((uint32_t)chain & 0xFFu) << 16);
    s.k = 0; s.pos = 4;
    return s;
}

static inline uint32_t orc_u32(orc_stream* s)
{
    if (s->pos == 4) {
        uint32_t ctr[4] = { s->c0, s->c1, s->c2, s->c3base | (s->k & 0xFFFFu) };
        orc_philox4x32_10(ctr, s->key, s->buf);
        s->k++; s->pos = 0;
    }
    return s->buf[s->pos++];
}

/* U(0,1): (x + 1/2) 2^-32, never 0 or 1.  (The fp32 device path uses ((x>>9)+1/2) 2^-23, the
 * same value truncated to 23 bits so that it is exact in fp32.) */
static inline double orc_unif(orc_stream* s) { return ((double)orc_u32(s) + 0.5) * (1.0 / 4294967296.0); }
static inline double orc_expo(orc_stream* s) { return -log(orc_unif(s)); }
/* N(0,1): Box-Muller cosine branch, two words per variate (the sine partner is discarded). */
static inline double orc_normal(orc_stream* s)
{
    double u1 = orc_unif(s), u2 = orc_unif(s);
    return sqrt(-2.0 * log(u1)) * cos(2.0 * ORC_PI * u2);
}

/* log of the standard normal cdf, stable in both tails */
static inline double orc_log_pnorm(double x)
{
    if (x > 0.0) return log1p(-0.5 * erfc(x * M_SQRT1_2));
    return log(0.5 * erfc(-x * M_SQRT1_2));
}

/* ---- Inverse Gaussian IG(mu, lambda), Michael-Schucany-Haas.  With y = N^2 and w = mu y the smaller root
 * mu + mu/(2 lambda) (w - sqrt(w (4 lambda + w))) is written 4 lambda / (y (1 + sqrt(1 + 4 lambda / w))^2):
 * algebraically identical, free of cancellation and overflow, with the limits lambda/y (mu -> inf) and mu (y -> 0).
 * The first root is kept with probability mu/(mu + x1) = 1/(1 + x1/mu). */
static inline double orc_invgauss(orc_stream* s, double mu, double lambda)
{
    double n = orc_normal(s);
    double y = n * n;
    double w = mu * y;
    double t = 1.0 + sqrt(1.0 + 4.0 * lambda / w);
    double x1 = 4.0 * lambda / (y * t * t);
    double u = orc_unif(s);
    return (u >= 1.0 / (1.0 + x1 / mu)) ? mu * mu / x1 : x1;
}

/* ---- standard normal quantile: Giles' (2010) single-precision erfinv polynomial as a starting point,
 * polished by Newton steps on Phi(x) = p evaluated through erfc in the lower tail (accurate to ~1e-15).  The device evaluates the
 * same function directly by Wichura's AS 241 (csrc/erm_rng.hpp); tests/test_gpu_samplers.py compares both with each other and with Cephes. */
static inline double orc_ndtri(double p)
{
    int upper = p > 0.5;
    double q = upper ? 1.0 - p : p;                 /* exact for our p = (k + 1/2) 2^-32 */
    double xx = 2.0 * q - 1.0;                      /* erfinv argument in (-1, 0] */
    double w = -log(4.0 * q * (1.0 - q)), pl;
    if (w < 5.0) {
        w -= 2.5;
        pl = 2.81022636e-08; pl = 3.43273939e-07 + pl * w; pl = -3.5233877e-06 + pl * w; pl = -4.39150654e-06 + pl * w;
        pl = 0.00021858087 + pl * w; pl = -0.00125372503 + pl * w; pl = -0.00417768164 + pl * w; pl = 0.246640727 + pl * w;
        pl = 1.50140941 + pl * w;
    } else {
        w = sqrt(w) - 3.0;
        pl = -0.000200214257; pl = 0.000100950558 + pl * w; pl = 0.00134934322 + pl * w; pl = -0.00367342844 + pl * w;
        pl = 0.00573950773 + pl * w; pl = -0.0076224613 + pl * w; pl = 0.00943887047 + pl * w; pl = 1.00167406 + pl * w;
        pl = 2.83297682 + pl * w;
    }
    double x = M_SQRT2 * pl * xx;                   /* <= 0 */
    /* three steps reach rounding level wherever the polynomial is a single-precision approximation (q > ~1e-7); beyond that it is
     * extrapolated, so keep stepping until the step no longer changes x (at most 8 steps) */
    for (int it = 0; it < 8; ++it) {
        double cdf = 0.5 * erfc(-x * M_SQRT1_2);
        double pdf = 0.3989422804014327 * exp(-0.5 * x * x);
        double dx = (cdf - q) / pdf;
        x -= dx;
        if (it >= 2 && fabs(dx) <= 4e-16 * fabs(x)) break;
    }
    return upper ? -x : x;
}

/* ---- Polya-Gamma PG(1, c): Devroye / Polson-Scott-Windle sampler for J*(1, z = |c|/2), t = 0.64, restated as a
 * SINGLE-LEVEL rejection sampler in which one attempt consumes exactly one Philox block (u0..u3) and has no inner loop.
 * Target (unnormalised): f(x) = e^{-z^2 x/2} sum_n (-1)^n a_n(x), a_n as in PSW (2013).  Envelope, two pieces per z:
 *   x > t  : a_0(x) e^{-z^2 x/2} = (pi/2) e^{-K x},  K = pi^2/8 + z^2/2, mass p = pi/(2K) e^{-K t}; proposal X = t + E/K, E = -log u1.
 *   x <= t : a_0(x) e^{-z^2 x/2} = 4 phi(Z) e^{-z^2/(2 Z^2)} dZ in terms of Z = x^{-1/2} >= a = 1/sqrt(t)  (a_0 is twice the Levy density,
 *            i.e. X = 1/Z^2 with Z a half-normal).  Two proposals:
 *     z < 8  : Robert's (1995) exponential proposal for a normal tail, tilted per z:  Z = a + E/lam, E = -log u1.  With
 *              g(Z) = -(Z - lam)^2/2 - z^2/(2 Z^2):   4 phi(Z) e^{-z^2/(2Z^2)} = [4/(lam sqrt(2 pi)) e^{lam^2/2 - lam a + M}] . lam e^{-lam (Z-a)} . e^{g(Z) - M},
 *              so for any M >= max_{Z >= a} g the bracket is the piece's envelope mass q and e^{g(Z) - M} <= 1 the acceptance probability.
 *              g is concave in Z and decreasing in z.  (lam, M) come from a table over z-bins of width 1/16: for the bin with lower edge z_k,
 *              Z*^2 = sqrt(10.6 + 1.07 z_k^2) (a fit of the mass-minimising choice), d = z_k^2 / Z*^3, lam = Z* - d (=> g'(Z*; z_k) = 0),
 *              M = g(Z*; z_k) = -d^2/2 - z_k^2/(2 Z*^2): exact maximum for z = z_k, an upper bound for every z in the bin.
 *              One logarithm serves both pieces; no normal quantile, no inner loop.  Acceptance of the whole attempt before the series
 *              test: 0.95 at z = 0, 0.92 at z = 1, 0.80 at z = 2, 0.65 at z = 3, 0.27 at z = 8.
 *     z >= 8 : a_0(x) e^{-z^2 x/2} extended to ALL x > 0 (the IG(1/z, 1) kernel), mass 2 e^{-z}; X ~ IG(1/z, 1) by Michael-Schucany-Haas
 *              with N = Phi^-1(u1), root choice u2; draws with X > t are rejected (|eta| >= 16: rare; acceptance -> 1).
 *   The tail is proposed with probability r = p / (p + q).  Then the alternating-series test with V = u3 in ratio form
 *   S_n/a_0 = 1 - rho_1 + rho_2 - ..., rho_n = a_n/a_0 = (2n+1) e^{-pi^2 n(n+1) X/2} (X > t)  or  (2n+1) e^{-2 n(n+1)/X} (X <= t).
 * Same law as PolyaGammaPSWSampler(1, eta) (src/Draw.pl.jl:38): PG(1,c) = J*(1,|c|/2)/4 -- tests/test_oracle_psw.py compares the draws
 * with an independently written two-level PSW/Devroye sampler. */
#define ORC_PG_T 0.64
#define ORC_PG_A 1.25            /* 1/sqrt(t) */
#define ORC_PG_NBIN 128
#define ORC_PG_ZMAX 8.0          /* NBIN bins of width 1/16 */

static inline double orc_u32_to_unif(uint32_t x) { return ((double)x + 0.5) * (1.0 / 4294967296.0); }

/* left-piece proposal parameters of z-bin k: rate lam, c = 1/lam, bound M, envelope mass q */
static inline void orc_pg_bin(int k, double* lam, double* c, double* M, double* q)
{
    double zk = (double)k / 16.0;
    double Zs2 = sqrt(10.6 + 1.07 * zk * zk), Zs = sqrt(Zs2);
    double d = zk * zk / (Zs * Zs2);
    double l = Zs - d, m = -0.5 * d * d - 0.5 * zk * zk / Zs2;
    *lam = l; *c = 1.0 / l; *M = m;
    *q = 4.0 / (l * sqrt(2.0 * ORC_PI)) * exp(0.5 * l * l - ORC_PG_A * l + m);
}

/* envelope mass of the left piece at z */
static inline double orc_pg_left_mass(double z)
{
    if (z < ORC_PG_ZMAX) { double lam, c, M, q; orc_pg_bin((int)(z * 16.0), &lam, &c, &M, &q); return q; }
    return 2.0 * exp(-z);
}

/* probability that an attempt proposes from the exponential tail */
static inline double orc_pg_tail_weight(double z)
{
    const double t = ORC_PG_T;
    double K = 0.125 * ORC_PI * ORC_PI + 0.5 * z * z;
    double p = ORC_PI / (2.0 * K) * exp(-K * t);
    return p / (p + orc_pg_left_mass(z));
}

/* one attempt; returns 1 and sets *out = X/4 on acceptance */
static inline int orc_pg1_attempt(double z, const uint32_t w[4], double* out)
{
    const double t = ORC_PG_T;
    double K = 0.125 * ORC_PI * ORC_PI + 0.5 * z * z;
    double r = orc_pg_tail_weight(z);
    double u0 = orc_u32_to_unif(w[0]), u1 = orc_u32_to_unif(w[1]), u2 = orc_u32_to_unif(w[2]), V = orc_u32_to_unif(w[3]);
    double x;
    if (u0 < r) {
        x = t + (-log(u1)) / K;
    } else if (z < ORC_PG_ZMAX) {
        double lam, c, M, q;
        orc_pg_bin((int)(z * 16.0), &lam, &c, &M, &q);
        double Z = ORC_PG_A + c * (-log(u1));
        x = 1.0 / (Z * Z);
        if (u2 > exp(-0.5 * (Z - lam) * (Z - lam) - 0.5 * z * z * x - M)) return 0;
    } else {
        double mu = 1.0 / z, n = orc_ndtri(u1);
        double ww = mu * n * n;
        double sq = sqrt(ww) * sqrt(4.0 + ww), den = sq + ww;
        double q = den > 0.0 ? 2.0 * sqrt(ww) / den : 1.0;
        double x1 = mu * q * q;
        x = (u2 >= mu / (mu + x1)) ? mu * mu / x1 : x1;
        if (x > t) return 0;
    }
    double S = 1.0;
    for (int n = 1; n <= 200; ++n) {
        double nn = (double)n * (double)(n + 1);
        double rho = (2.0 * n + 1.0) * (x > t ? exp(-0.5 * ORC_PI * ORC_PI * nn * x) : exp(-2.0 * nn / x));
        if (n & 1) { S -= rho; if (V <= S) { *out = 0.25 * x; return 1; } }
        else       { S += rho; if (V > S) return 0; }
    }
    *out = 0.25 * x; return 1; /* unreachable guard */
}

/* draw for stream (site, i, j, sweep): attempt k uses Philox block k of that stream */
static inline double orc_pg1(orc_stream* s, double c)
{
    double z = 0.5 * fabs(c);
    for (;;) {
        uint32_t w[4];
        w[0] = orc_u32(s); w[1] = orc_u32(s); w[2] = orc_u32(s); w[3] = orc_u32(s);
        double out;
        if (orc_pg1_attempt(z, w, &out)) return out;
    }
}

/* ---- truncated normal on (0, inf): TN(m, s; 0, inf) ---- */
static inline double orc_truncnorm0(orc_stream* s, double m, double sd)
{
    double alpha = -m / sd, z;
    if (alpha <= 0.0) {
        do { z = orc_normal(s); } while (z < alpha);
    } else {
        double lam = 0.5 * (alpha + sqrt(alpha * alpha + 4.0));
        for (;;) {
            z = alpha + orc_expo(s) / lam;
            double u = orc_unif(s);
            if (u <= exp(-0.5 * (z - lam) * (z - lam))) break;
        }
    }
    return m + sd * z;
}

/* ---- Gamma(shape >= 1, scale 1), Marsaglia-Tsang without the squeeze ---- */
static inline double orc_gamma(orc_stream* s, double shape)
{
    double d = shape - 1.0 / 3.0, c = 1.0 / sqrt(9.0 * d);
    for (;;) {
        double x, v;
        do { x = orc_normal(s); v = 1.0 + c * x; } while (v <= 0.0);
        v = v * v * v;
        double u = orc_unif(s);
        if (log(u) < 0.5 * x * x + d - d * v + d * log(v)) return d * v;
    }
}
/* InverseGamma(shape, scale) in Distributions.jl's parametrisation: scale / Gamma(shape,1) */
static inline double orc_invgamma(orc_stream* s, double shape, double scale) { return scale / orc_gamma(s, shape); }
static inline double orc_chisq(orc_stream* s, double k) { return 2.0 * orc_gamma(s, 0.5 * k); }


/* Generalized inverse Gaussian GIG(p, a, b), density proportional to x^(p-1) exp(-(a x + b/x)/2) -- the distribution of
 * /root/reference/src/GenInvGaussian.jl (params p, a, b; :17-30), whose own sampler (:76-106, gamma-proposal rejection) is dead code in
 * the reference.  Restated with Devroye's (2014, "Random variate generation for the generalized inverse Gaussian
 * distribution", Statistics and Computing 24) uniformly efficient rejection sampler for the log-concave density of log X:
 * X = sqrt(b/a) Y, Y ~ GIG(|p|, omega = sqrt(a b)) (inverted for p < 0).  One attempt uses three uniforms. */
static inline double orc_gig_psi(double x, double alpha, double lam) { return -alpha * (cosh(x) - 1.0) - lam * (exp(x) - x - 1.0); }
static inline double orc_gig_dpsi(double x, double alpha, double lam) { return -alpha * sinh(x) - lam * (exp(x) - 1.0); }
static inline double orc_gig(orc_stream* st, double p, double a, double b)
{
    const double omega = sqrt(a * b);
    const int inv = p < 0.0;
    const double lam = fabs(p);
    const double alpha = sqrt(omega * omega + lam * lam) - lam;
    double x = -orc_gig_psi(1.0, alpha, lam), t, s;
    if (x >= 0.5 && x <= 2.0) t = 1.0; else if (x > 2.0) t = sqrt(2.0 / (alpha + lam)); else t = log(4.0 / (alpha + 2.0 * lam));
    x = -orc_gig_psi(-1.0, alpha, lam);
    if (x >= 0.5 && x <= 2.0) s = 1.0;
    else if (x > 2.0) s = sqrt(4.0 / (alpha * cosh(1.0) + lam));
    else { double s1 = 1.0 / lam, s2 = log(1.0 + 1.0 / alpha + sqrt(1.0 / (alpha * alpha) + 2.0 / alpha)); s = s1 < s2 ? s1 : s2; }
    const double eta = -orc_gig_psi(t, alpha, lam), zeta = -orc_gig_dpsi(t, alpha, lam);
    const double theta = -orc_gig_psi(-s, alpha, lam), xi = orc_gig_dpsi(-s, alpha, lam);
    const double pp = 1.0 / xi, r = 1.0 / zeta, td = t - r * eta, sd = s - pp * theta, q = td + sd;
    double rnd = 0.0;
    for (int tries = 0; tries < 4096; ++tries) {
        const double U = orc_unif(st), V = orc_unif(st), W = orc_unif(st);
        if (U < q / (pp + q + r)) rnd = -sd + q * V;
        else if (U < (q + r) / (pp + q + r)) rnd = td - r * log(V);
        else rnd = -sd + pp * log(V);
        const double f1 = exp(-eta - zeta * (rnd - t)), f2 = exp(-theta + xi * (rnd + s));
        const double g = (rnd >= -sd && rnd <= td) ? 1.0 : (rnd > td ? f1 : f2);
        if (W * g <= exp(orc_gig_psi(rnd, alpha, lam))) break;
    }
    double y = exp(rnd) * (lam / omega + sqrt(1.0 + lam * lam / (omega * omega)));
    if (inv) y = 1.0 / y;
    return y * sqrt(b / a);
}

#endif
