// erm_kernels.hpp -- the Gibbs sweep as gfx950 kernels.
//
// One reference sweep (/root/reference/src/GibbsRtIrt.pl.jl:289-324 and siblings) is re-scheduled, without changing any
// conditional or its conditioning values, into
//     tiny step  : one workgroup; reduces the previous pass's per-workgroup statistics slabs in a fixed order and makes
//                  every item-level / structural draw (beta, Sigma_p, b, a, lambda, sigma2_t, rho) from sufficient statistics;
//     row pass   : ONE streaming pass over (omega, Y, logT[, nu]) that draws theta_i, zeta_i for every subject, then the
//                  NEXT sweep's omega_ij (and nu) -- whose conditioning values are final at that point -- and accumulates
//                  the statistics the next tiny step needs plus this sweep's log-likelihood.
// GibbsRtIrtCrossQr needs two passes per sweep because lambda_t depends on theta_t (src/Draw.pl.jl:239-251).
//
// Thread mapping of a pass: a wave holds R = 64/W subjects x W lanes; lane (r, s) owns items j = s + W k, k < ceil(J/W).
// Row sums are W-lane butterflies, item sums are R-lane butterflies followed by wave-private fp64 LDS accumulators, so all
// reductions are order-deterministic: a chain is bit-reproducible run to run.
// HBM layout: Y u8 [N][J], centred logT / omega / nu `real` [N][J] (row-major), theta/zeta `real` [N].
#pragma once
#include <type_traits>
#include "erm_layout.hpp"
#include "erm_rng.hpp"

namespace erm {

struct Ctl {
    uint32_t sweep;       // global index of the sweep whose item draws are current
    uint32_t row;         // trace row of that sweep within the current sample! call
    uint32_t burn_rows;   // rows < burn_rows are burn-in (not accumulated into Post.mean)
    uint32_t err;         // sticky non-finite flag
    uint32_t first;       // 1 until the first sweep of the current erm_run has been drawn (that sweep writes trace row `row` itself instead of row + 1, and no
                          // log-likelihood of a preceding pass): device-side, so that every sweep of a run is the same launch and whole runs replay from graphs
    uint32_t pad_;
    unsigned long long dbg_attempts, dbg_trips, dbg_cells;   // -DERM_DIAG_BUILD with ERM_PASS_STOP=9: PG attempts, wave trips, cells
};

template <typename real> struct PassArgs {
    const uint8_t* Y; const real* C; real* omega; real* nu; const real* X;
    real* theta; real* zeta;
    const double* par; const double* cst; double* slab; const Ctl* ctl;
    const double* pgtab;                            // [PG_NBIN][4] proposal table of the Polya-Gamma sampler (erm_rng.hpp, pg_bin)
    double* gslab; unsigned int* gcnt;              // per-group reduced slabs and arrival counters (GROUP consecutive workgroups)
    double* sum_theta; double* sum_zeta; double* sum_nu;
    real* tr_theta; real* tr_zeta; real* tr_nu;     // [rows][N] (CrossQr's tr_nu: [rows][N][J]) or nullptr
    long long N; long long rows_per_block;          // each workgroup owns rows [b*rpb, (b+1)*rpb)
    int rows_per_wave;                              // sizes the row-sum region of LDS: nWaves * 4 * rows_per_wave >= 4 * rows_per_block values (erm_geometry.hpp)
    int J; int nFeat; int W; int logW; int IPL;
    int mode;             // 0 = prologue (no theta/zeta draws, no LL, no trace), 1 = full sweep pass
    int ngx;              // extra global statistics inserted before the log-likelihood slot (LatentQr sigp_mode 1: the 1/nu-weighted Gram entries)
    uint32_t chain; uint64_t seed; double k1, k2;
    int dbg_stop;         // -DERM_DIAG_BUILD only: skip everything after stage k (0 = run everything); ignored by the shipped library
    uint32_t dbg_sweep;   // -DERM_DIAG_BUILD only: 0 = the stop applies to every launch, else only to the launch that draws this sweep
    unsigned long long* dbg_ts;   // diagnostics only (ERM_TIMELINE): [2 workgroups][16 waves][16 checkpoints] of the 100 MHz wall clock
    int acc_off;          // byte offset in dynamic LDS of the per-wave item accumulators [nWaves][NSTAT][J], the LAST region of a launch's LDS
    uint32_t row_base;    // subject index of local row 0 in the whole data set (subject-sharded chains; 0 otherwise): the random streams are
                          // addressed by the GLOBAL subject index, so a chain does not depend on how its subjects are spread over devices
    // PERSIST kernels (small data sets: K sweeps in one launch): both halves of the double buffers, the number of sweeps and the parity of the
    // first one; the statistics rows travel between the sweeps of a launch as tagged packets (xbuf, see persist_* below), tags tag0+1 .. tag0+nsweeps-1
    double* parB[2]; Ctl* ctlB[2]; double* gslabB[2];
    uint32_t nsweeps; uint32_t cur0;
    unsigned long long* xbuf; uint32_t tag0; unsigned int* tmo;
};

// Statistics exchange between the sweeps of a PERSIST launch (every workgroup resident: the host launches at most one per compute unit).
// A grid barrier (arrive on a counter, poll it, acquire, then load the rows) is four dependent trips to memory and measured 4.8 us on 32
// workgroups -- more than the kernel boundary it replaces.  Instead every 64-bit word of a row carries its own validity: a double travels as two
// packets {tag : 32 | half : 32}, each ONE 8-byte agent-scope store (atomic by size), and the readers poll the packets themselves: one store trip
// plus one load trip, no counter, no fence, no ordering between packets needed.  xbuf[parity][workgroup][2 * NS]; a row written after sweep k goes
// to parity k & 1 with tag tag0 + k + 1.  No workgroup can overwrite a row another still waits for: writing the row of sweep k + 2 takes every
// workgroup's row of sweep k + 1, which a workgroup writes only after it has read all rows of sweep k.  The wait is bounded by the wall clock: on
// time-out the launch sets *tmo, every workgroup leaves the sweep loop at its next head, and erm_run replays the call on the per-sweep schedule.
__device__ __forceinline__ void persist_put(unsigned long long* row, int e, double v, uint32_t tag)
{
    const unsigned long long b = (unsigned long long)__double_as_longlong(v), t = (unsigned long long)tag << 32;
    __hip_atomic_store(row + 2 * e, t | (b & 0xffffffffull), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(row + 2 * e + 1, t | (b >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// entry e of the sum over all workgroups' rows, in the association of the per-sweep path (group sums of GROUP rows in workgroup order, then the
// groups in order: the epilogue's group reduction followed by reduce_rows), so that the chain does not depend on the schedule.
// tmo[0]: the launch's time-out flag, tmo[1]: the bound of one wait in ticks of the 100 MHz wall clock (1 s; set by the host with every erm_run),
// tmo[2]: fault injection for the tests (ERM_FLAG_TEST_PERSIST_TIMEOUT).  A wait that fails takes the clock once and from then on compares; a launch that
// has timed out anywhere gives up at once everywhere (every workgroup leaves the sweep loop at its next head: pass_kernel), and erm_run restores the state
// it saved and replays the call on the per-sweep schedule.
__device__ __forceinline__ double persist_get(const unsigned long long* par_rows, int nblocks, int NS, int e, uint32_t tag, unsigned int* tmo)
{
    double t = 0.0;
    unsigned long long t_end = 0ull;
    for (int g0 = 0; g0 < nblocks; g0 += GROUP) {
        unsigned long long lo[GROUP], hi[GROUP];
        for (;;) {
            bool ok = true;
#pragma unroll
            for (int u = 0; u < GROUP; ++u) {
                const unsigned long long* q = par_rows + (size_t)(g0 + u < nblocks ? g0 + u : g0) * 2 * NS + 2 * e;
                lo[u] = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                hi[u] = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int u = 0; u < GROUP; ++u) ok = ok && (uint32_t)(lo[u] >> 32) == tag && (uint32_t)(hi[u] >> 32) == tag;
            if (ok) break;
            if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;
            const unsigned long long now = wall_clock64();
            if (t_end == 0ull) t_end = now + (unsigned long long)__hip_atomic_load(tmo + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else if (now > t_end) { __hip_atomic_store(tmo, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
            __builtin_amdgcn_s_sleep(2);
        }
        double tg = 0.0;
#pragma unroll
        for (int u = 0; u < GROUP; ++u) {
            const double v = __longlong_as_double((long long)((lo[u] & 0xffffffffull) | (hi[u] << 32)));
            tg += (g0 + u < nblocks) ? v : 0.0;
        }
        t += tg;
    }
    return t;
}

// Lanes of ONE wave exchange data through LDS: DS instructions of a wave execute in order, so a compiler-level fence is all
// that is needed between a phase that writes and a phase that reads.
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename real> __device__ __forceinline__ real row_normal(uint32_t wa, uint32_t wb, [[maybe_unused]] const double2* tab = nullptr)
{
    if constexpr (sizeof(real) == 8) return fm::sqrt(-2.0 * (tab ? fm::log(word_to_unif<double>(wa), tab) : fm::log(word_to_unif<double>(wa)))) * fm::cos2pi(word_to_unif<double>(wb));
    else return r_sqrt(real(-2) * r_log(word_to_unif<real>(wa))) * r_cos2pi(word_to_unif<real>(wb));
}

template <typename T> __device__ __forceinline__ T bfly_sum(T v, int lo, int hi)   // sum over lanes differing in bits [lo, hi)
{
    for (int m = lo; m < hi; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
}

// log(1 + e^x)
__device__ __forceinline__ float  log1pexp_r(float x)  { return fmaxf(x, 0.f) + r_log(1.f + r_exp(-fabsf(x))); }
__device__ __forceinline__ double log1pexp_r(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }

constexpr double LOG_2PI = 1.8378770664093454836;
// Response-time log-likelihood of the single-pass models from SUFFICIENT STATISTICS instead of per cell (round 4).  With c = logT - column mean,
//   sum_i (c_ij + zeta_i - lc_j)^2 = csq_j + 2 G_j + zz - 2 lc_j sz + N lc_j^2      (lc_j = lambda_j - mean_j, G_j = sum_i c_ij zeta_i, sz = sum zeta, zz = sum zeta^2),
// so  LL_rt = -1/2 sum_j [ N (log 2 pi + log sig2t_j) + (csq_j + N lc_j^2) / sig2t_j ]                  (constant in the subjects: the tiny step, par derived[1])
//             - sum_j G_j / sig2t_j - zz/2 sum_j 1/sig2t_j + sz sum_j lc_j / sig2t_j                     (linear in the statistics: every workgroup, from its own sums)
// -- statistics the column phase accumulates anyway for the lambda / sig2t draws (tiny_items forms the same expansion).  The column phase loses the residual,
// its square and the per-item constants of every cell (it is VALU-issue-bound), the head the logarithm of sig2t per item.  Same log-likelihood to ~1e-14.
#ifndef ERM_RTLL_STATS
#define ERM_RTLL_STATS 1
#endif
template <int MODEL, int PHASE> constexpr bool rtll_stats() { return ERM_RTLL_STATS != 0 && PHASE == 0 && (fam_rt(MODEL) || fam_lq(MODEL)); }
constexpr int KB = 4;     // items per lane whose loads are in flight together in the row-sum phase

// Stage-timing / counting diagnostics (early returns that leave GARBAGE results, PG attempt counters) exist only in a library built with
// -DERM_DIAG_BUILD (tools/tiny_stages.sh); the shipped library has no such code path, so no environment variable can corrupt a fit.
#ifdef ERM_DIAG_BUILD
// (with dbg_sweep != 0 the early return applies to that ONE sweep only: every launch before it ran in full, so the truncated launch works on a valid
// chain state -- tools/stage_budget.sh reads its counters and its duration)
#define ERM_DIAG_STOP(args, k) do { if ((args).dbg_stop == (k) && ((args).dbg_sweep == 0u || sweep == (args).dbg_sweep)) __builtin_amdgcn_endpgm(); } while (0)
#define ERM_DIAG_ON(args, k) ((args).dbg_stop == (k))
#else
#define ERM_DIAG_STOP(args, k) ((void)0)
#define ERM_DIAG_ON(args, k) false
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Tiny step
// ---------------------------------------------------------------------------------------------------------------------
// out[e] = sum over the nb rows of `slab` (row length NS) in row order, 16 loads in flight; the same order wherever statistics are
// reduced, so a sweep's log-likelihood does not depend on which kernel happened to reduce it
__device__ __forceinline__ void reduce_rows(const double* slab, int nb, int NS, double* out, int tid, int nthreads)
{
    for (int e = tid; e < NS; e += nthreads) {
        double t = 0.0;
        for (int b0 = 0; b0 < nb; b0 += 16) {
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = slab[(size_t)(b0 + u < nb ? b0 + u : b0) * NS + e];
#pragma unroll
            for (int u = 0; u < 16; ++u) t += (b0 + u < nb) ? v[u] : 0.0;
        }
        out[e] = t;
    }
}

struct TinyArgs {
    const double* par; double* par_out;             // parameter block read / written (the same buffer unless the step runs redundantly in every workgroup)
    const double* cst; const double* slab0; const double* slab1;
    const Ctl* ctl; Ctl* ctl_out; Ctl* ctl_err;     // counters read / written; sticky error flag
    double* tr_item;      // [rows][4J + NQ] : a, b, lambda, sig2t, then the small part of qr
    double* tr_ll;        // [rows]
    long long N; int J; int nFeat; int nb0, nb1;
    int mode;             // 0 = draw sweep (advance ctl), 1 = final (only reduce the last pass's log-likelihood)
    int intercept, onepl, cov2one, sigp_mode;
    uint32_t chain; uint64_t seed; double k1, k2;
    int nq;               // number of small qr entries recorded per sweep
    int ngx;              // extra global statistics of slab0 (see PassArgs::ngx)
    int dbg_stop;         // -DERM_DIAG_BUILD only: return after stage k (0 = run everything); ignored by the shipped library
    uint32_t dbg_sweep;   // -DERM_DIAG_BUILD only: see PassArgs::dbg_sweep
};

__device__ inline void d_cov2one(double* S)   // src/Draw.pl.jl:507-511
{
    const double d1 = q_rcp(q_sqrt(S[0]));
    S[0] *= d1 * d1; S[1] *= d1; S[2] *= d1;
    const double d2 = q_rcp(q_sqrt(S[3]));
    S[3] *= d2 * d2; S[1] *= d2; S[2] *= d2;
    S[0] = 1.0; S[3] = 1.0;
}

// 2 x 2 InverseWishart(df, Psi) of drawSubjCovariance (src/Draw.pl.jl:499-515): W ~ Wishart(df, Psi^-1) by its Bartlett factor (L = chol(Psi^-1),
// A = [c1 0; n21 c2] with c1^2 ~ chi2(df), n21 ~ N(0,1), c2^2 ~ chi2(df - 1): W = L A A' L'), then W^-1.  The three variates are drawn apart from the
// matrix arithmetic (in the sweep kernel by another thread, while the statistics they do not depend on are still being reduced); the unit test
// (erm_debug_invwishart) runs the two functions back to back.
__device__ __forceinline__ void bartlett2_variates(Stream& ss, double df, double* v3)
{
    v3[0] = q_sqrt(chisq(ss, df));
    v3[1] = normal<double>(ss);
    v3[2] = q_sqrt(chisq(ss, df - 1.0));
}
__device__ __forceinline__ void invwishart2(const double* Psi, const double* v3, double* S)
{
    const double ipdet = q_rcp(Psi[0] * Psi[3] - Psi[1] * Psi[2]);
    const double Pi[4] = { Psi[3] * ipdet, -Psi[1] * ipdet, -Psi[2] * ipdet, Psi[0] * ipdet };
    const double l00 = q_sqrt(Pi[0]), l10 = q_div(Pi[1], l00), l11 = q_sqrt(Pi[3] - l10 * l10);
    const double c1 = v3[0], n21 = v3[1], c2 = v3[2];
    const double z00 = l00 * c1, z10 = l10 * c1 + l11 * n21, z11 = l11 * c2;
    const double Wm[4] = { z00 * z00, z10 * z00, z00 * z10, z10 * z10 + z11 * z11 };
    const double idet = q_rcp(Wm[0] * Wm[3] - Wm[1] * Wm[2]);
    S[0] = Wm[3] * idet; S[1] = -Wm[1] * idet; S[2] = -Wm[2] * idet; S[3] = Wm[0] * idet;
}

// lower Cholesky factor of the n x n matrix V (column-major, leading dimension n) into L: one column per step, rows in parallel
// across the lanes of ONE wave; all lanes of the wave must call it
__device__ inline void chol_lower_wave(int n, const double* V, double* L, int lane)
{
    if (n <= 8) {
        // the usual sizes (2 (nFeat + 1) = 8 for the default nFeat = 3): lane i keeps row i of L in registers and reads the pivot row by
        // lane broadcast, so a column step costs no LDS round trips; same operations in the same order as the general loop below
        double row[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) row[k] = 0.0;
        const int li = lane < n ? lane : n - 1;
        for (int jj = 0; jj < n; ++jj) {
            double t = V[li + jj * n];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const double bj = __shfl(row[k], jj, 64);
                t = fma(-(k < jj ? row[k] : 0.0), bj, t);
            }
            const double piv = __shfl(t, jj, 64);
            double r = __builtin_amdgcn_rsq(piv);
            r = r * fma(-0.5 * piv * r, r, 1.5);
            r = r * fma(-0.5 * piv * r, r, 1.5);
            const double v = (lane >= jj && lane < n) ? t * r : 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) row[k] = (k == jj) ? v : row[k];
        }
        if (lane < n) {
#pragma unroll
            for (int k = 0; k < 8; ++k) if (k <= lane && k < n) L[lane + k * n] = row[k];
        }
        wave_sync();
        return;
    }
    for (int jj = 0; jj < n; ++jj) {
        double t = 0.0;
        if (lane >= jj && lane < n) {
            t = V[lane + jj * n];
            for (int k = 0; k < jj; ++k) t -= L[lane + k * n] * L[jj + k * n];
        }
        // 1/sqrt(pivot) by the hardware estimate and two Newton steps (full double precision), then multiplications only: the square
        // root and the division of the textbook column step are the longest links of this one-wave dependent chain
        const double piv = __shfl(t, jj, 64);
        double r = __builtin_amdgcn_rsq(piv);
        r = r * fma(-0.5 * piv * r, r, 1.5);
        r = r * fma(-0.5 * piv * r, r, 1.5);
        if (lane >= jj && lane < n) L[lane + jj * n] = t * r;       // the diagonal: piv / sqrt(piv)
        wave_sync();
    }
}
// i-th standard normal of stream (BETA, 0, 0, sweep): words 2i, 2i+1, i.e. block i/2 -- the oracle draws them consecutively
__device__ inline double beta_normal(uint64_t seed, uint32_t chain, uint32_t sweep, int i)
{
    uint32_t w0, w1, w2, w3;
    philox4x32_10(0u, 0u, sweep, ((uint32_t)SITE_BETA << 24) | ((chain & 0xFFu) << 16) | (uint32_t)(i >> 1), (uint32_t)seed, (uint32_t)(seed >> 32), w0, w1, w2, w3);
    const double u1 = word_to_unif<double>((i & 1) ? w2 : w0), u2 = word_to_unif<double>((i & 1) ? w3 : w1);
    return q_sqrt(-2.0 * q_log(u1)) * q_cos2pi(u2);
}


// One tiny step = tiny_items (every thread of the workgroup; no barrier inside), a workgroup barrier, tiny_struct (ONE wave; only
// wave-level synchronisation inside).  Both work on an LDS-resident parameter block `par` (read: the previous values, written in
// place: this sweep's).  st0 / st1: reduced statistics of the previous pass(es); part: >= J doubles of scratch; work: TINY_WORK
// doubles; sh_x: x'x and its inverse.  STEP: 0 = the per-sweep step of single-pass models / Cross-family step 1; 1 = Cross-family
// step 2 (lambda, sig2t).
#define ERM_TINY_COMMON                                                                                          \
    const int J = T.J, p = T.nFeat + 1;                                                                          \
    const double Nd = (double)T.N;                                                                               \
    constexpr int NSTAT0 = Stats<MODEL, 0>::NSTAT;                                                               \
    constexpr int NSTAT1 = fam_cq(MODEL) ? Stats<CROSSQR, 1>::NSTAT : 0;                                         \
    const double* K0 = cstp + cst_off_k0(J);                /* cstp: the item constants (global or an LDS copy) */ \
    const double* cm = cstp + cst_off_m(J);                                                                      \
    const double* csq = cstp + cst_off_csq(J);                                                                   \
    const double muLam = cstp[cst_off_mu(J)], sdLam = cstp[cst_off_mu(J) + 1];                                   \
    double* Sigp = par + par_off_sigp(J);                                                                        \
    double* beta = par + par_off_beta(J);                                                                        \
    const double* G0 = st0 + NSTAT0 * J;                    /* global statistics of slab0 */                     \
    double* spd = work + TINY_WORK - 4;                     /* pre-drawn variates of the Sigma_p draw */          \
    double* bn = work + TINY_WORK - 4 - 2 * PMAX;           /* beta_t, contiguous [2p] / [p+1] */                 \
    double* qf = work + TINY_WORK - 12 - 2 * PMAX;          /* quadratic forms for Sigma_p (RtIrt) */             \
    (void)Nd; (void)NSTAT1; (void)K0; (void)cm; (void)csq; (void)muLam; (void)sdLam; (void)Sigp; (void)beta; (void)G0; (void)spd; (void)bn; (void)qf;

// item draws (threads 128..) and the variates of the Sigma_p draw (thread 64): neither depends on beta_t
template <int MODEL, int STEP>
__device__ __forceinline__ void tiny_items(const TinyArgs& T, double* par, const double* st0, const double* st1, double* part, double* work, uint32_t sweep,
                                           const double* cstp)
{
    ERM_TINY_COMMON
    const int tid = threadIdx.x, nthreads = blockDim.x;
    //   RtIrt   : c1^2 ~ chi2(N+3), n21 ~ N(0,1), c2^2 ~ chi2(N+2) (Bartlett factor of the Wishart; same stream order as the oracle)
    //   others  : g ~ Gamma(shape) of the InverseGamma
    if (tid == (nthreads > 64 ? 64 : 0) && STEP == 0 && MODEL != MLIRT) {
        Stream ss(T.seed, T.chain, SITE_SIGP, 0u, 0u, sweep);
        if (fam_rt(MODEL)) {
            bartlett2_variates(ss, Nd + 3.0, spd);
        } else {
            spd[0] = gamma_mt(ss, 1e-3 + (MODEL == LATENTQR ? Nd * 3.0 / 2.0 : Nd / 2.0));
        }
    }
    // =========================================================== item draws: one thread per item, in waves 2..; when the workgroup has
    // enough threads the response-time draws (lambda, sig2t) of an item run on a second thread, concurrently with its (b, a) draws
    // (a workgroup of fewer than 256 threads has no waves to spare: every thread takes items)
    const int ioff = nthreads >= 256 ? 128 : 0;
    const int nit = nthreads - ioff;
    const int Jw = (J + 63) & ~63;                          // the second thread of an item starts at a wave boundary: lanes of one wave in
    const int rt_off = (nit >= 2 * Jw) ? Jw : 0;            // both roles would run the two chains one after the other
    for (int j = tid - ioff; tid >= ioff && j < J; j += (rt_off ? 2 * Jw : nit)) {
        if (STEP == 0) {
            const double S0 = st0[0 * J + j], S1 = st0[1 * J + j], S2 = st0[2 * J + j], K1 = st0[3 * J + j];
            double a = par[j], b = par[J + j];
            if (fam_cq(MODEL)) {
                // rho_t: drawSubjCorrCrossQr src/Draw.pl.jl:474-489 / drawSubjCorrCross :463-469 (uses sig2t_{t-1})
                const double sg = par[3 * J + j];
                const double R0 = st1[0 * J + j], R1 = st1[1 * J + j];
                const double isk = q_rcp(sg * T.k2);
                const double parV = q_rcp(1.0 + R0 * isk);
                const double parM = parV * (0.0 + R1 * isk);
                Stream sr(T.seed, T.chain, SITE_RHO, 0u, (uint32_t)j, sweep);
                par[4 * J + j] = parM + q_sqrt(parV) * normal<double>(sr);
            }
            auto draw_b = [&]() {   // drawItemDifficulty src/Draw.pl.jl:98-105
                const double parV = q_rcp(1.0 + a * a * S0);
                const double parM = parV * (0.0 - (a * K0[j] - a * a * S1));
                Stream sb(T.seed, T.chain, SITE_B, 0u, (uint32_t)j, sweep);
                double v = parM + q_sqrt(parV) * normal<double>(sb);
                b = v < -4.0 ? -4.0 : (v > 4.0 ? 4.0 : v);
            };
            auto draw_a = [&]() {   // drawItemDiscrimination src/Draw.pl.jl:88-93
                const double parV = q_rcp(1.0 + (S2 - 2.0 * b * S1 + b * b * S0));
                const double parM = parV * (1.0 + (K1 - b * K0[j]));
                Stream sa(T.seed, T.chain, SITE_A, 0u, (uint32_t)j, sweep);
                a = truncnorm0(sa, parM, q_sqrt(parV));
                if (T.onepl) a = 1.0;
            };
            if (MODEL == MLIRT) { draw_a(); draw_b(); }   // src/GibbsRtIrt.pl.jl:233-237
            else { draw_b(); draw_a(); }                  // :301-305
            par[j] = a; par[J + j] = b;
        }
    }
    for (int j = tid - ioff - rt_off; tid >= ioff + rt_off && j < J; j += (rt_off ? 2 * Jw : nit)) {
        if ((fam_rt(MODEL) || fam_lq(MODEL)) && STEP == 0) {
            // lambda: drawItemIntensity src/Draw.pl.jl:215-220 ; sig2t: drawItemTimeResidual :257-262
            // sum zeta, sum zeta^2 over subjects (RtIrt: (x'zeta)[0] and zz; LatentQr: tracked explicitly)
            const double sz = fam_rt(MODEL) ? G0[p] : G0[2 * p + 5];
            const double zz = fam_rt(MODEL) ? G0[2 * p + 2] : G0[2 * p + 6];
            const double Gj = st0[4 * J + j];
            const double sg_old = par[3 * J + j];
            const double isd2 = q_rcp(sdLam * sdLam), isg = q_rcp(sg_old);
            const double parV = q_rcp(isd2 + Nd * isg);
            const double parM = parV * (muLam * isd2 + (Nd * cm[j] + sz) * isg);
            Stream sl(T.seed, T.chain, SITE_LAMBDA, 0u, (uint32_t)j, sweep);
            const double lam = truncnorm0(sl, parM, q_sqrt(parV));
            const double lc = lam - cm[j];
            const double ssq = csq[j] + 2.0 * Gj + zz - 2.0 * lc * sz + Nd * lc * lc;
            Stream sv(T.seed, T.chain, SITE_SIG2T, 0u, (uint32_t)j, sweep);
            const double sg = invgamma(sv, 1e-3 + Nd / 2.0, 1e-3 + ssq / 2.0);
            par[2 * J + j] = lam; par[3 * J + j] = sg;
            part[j] = q_rcp(sg);
        }
        if (fam_cq(MODEL) && STEP == 1) {
            // lambda: drawItemIntensityCrossQr src/Draw.pl.jl:239-251 / ...Cross :225-231 ; sig2t: drawItemTimeResidualCrossQr :278-288 / ...Cross :267-273
            const double W0 = st0[4 * J + j], W1 = st0[5 * J + j], W2 = st0[6 * J + j], V = st0[7 * J + j];
            const double sg_old = par[3 * J + j];
            const double isd2 = q_rcp(sdLam * sdLam), isk = q_rcp(sg_old * T.k2);
            const double parV = q_rcp(isd2 + W0 * isk);
            const double parM = parV * (muLam * isd2 + (W1 + cm[j] * W0) * isk);
            Stream sl(T.seed, T.chain, SITE_LAMBDA, 0u, (uint32_t)j, sweep);
            const double lam = truncnorm0(sl, parM, q_sqrt(parV));
            const double lc = lam - cm[j];
            const double ssq = (W2 - 2.0 * lc * W1 + lc * lc * W0) / (2.0 * T.k2);
            Stream sv(T.seed, T.chain, SITE_SIG2T, 0u, (uint32_t)j, sweep);
            const double sg = (MODEL == CROSSQR) ? invgamma(sv, 1e-3 + Nd * 3.0 / 2.0, 1e-3 + ssq + V) : invgamma(sv, 1e-3 + Nd / 2.0, 1e-3 + ssq);
            par[2 * J + j] = lam; par[3 * J + j] = sg;
            part[j] = q_rcp(sg);
        }
        if (MODEL == MLIRT || (fam_cq(MODEL) && STEP == 0)) part[j] = q_rcp(par[3 * J + j]);   // sig2t not drawn in this step
    }
}

// GibbsRtIrt: covariance of beta_t and its Cholesky factor (drawSubjCoefficients src/Draw.pl.jl:380-393) by ONE wave -- they depend on Sigma_p_{t-1} and
// the constant (x'x)^-1 only, not on the previous pass's statistics, so a persistent launch has them made by an idle wave while the statistics
// are still on their way (pass_kernel) instead of at the head of wave 0's chain.  Posterior precision = 11' + kron(inv(Sigp), x'x) (the reference's
// `1/sigma^2 .+ M` adds 1 to EVERY element), so by Sherman-Morrison parV = Minv - v v'/(1 + 1'v), Minv = kron(Sigp, (x'x)^-1), v = Minv 1.
// work: V [n*n], L [n*n], tv [n], vv [n], rs [n], pmv [n], zb [n]  (n = 2p)
__device__ __forceinline__ void rtirt_beta_cov(int p, const double* Sigp, const double* Xinv, double* work, int lane)
{
    const int n = 2 * p;
    double* V = work, *L = V + n * n, *vv = L + n * n + n, *rs = vv + n;
    if (lane < p) { double t = 0.0; for (int w = 0; w < p; ++w) t += Xinv[lane + w * PMAX]; rs[lane] = t; }
    wave_sync();
    if (lane < n) { const int a_ = lane / p, u = lane % p; vv[lane] = (Sigp[a_] + Sigp[a_ + 2]) * rs[u]; }
    wave_sync();
    double cden = 1.0;
    for (int i = 0; i < n; ++i) cden += vv[i];
    const double icden = q_rcp(cden);
    for (int e = lane; e < n * n; e += 64) {
        int i = e % n, jj = e / n;
        if (i > jj) { const int t_ = i; i = jj; jj = t_; }              // Symmetric(parV): upper triangle
        const int a_ = i / p, u = i % p, b_ = jj / p, w = jj % p;
        V[e] = Sigp[a_ + 2 * b_] * Xinv[u + w * PMAX] - vv[i] * vv[jj] * icden;
    }
    wave_sync();
    chol_lower_wave(n, V, L, lane);
}

// structural draws by ONE wave (lane = its lane index): beta_t (lanes cooperate through LDS), sum_j 1/sig2t_j, Sigma_p_t | beta_t.
// PART 0: only the leading part of the beta chain, which needs neither the item draws nor Sigma_p's variates (so it can run while
// other waves draw the items); PART 1: the rest; PART 2: everything.
template <int MODEL, int STEP, int PART>
__device__ __forceinline__ void tiny_struct(const TinyArgs& T, double* par, const double* st0, const double* st1, double* part, double* work,
                                            const double* sh_x, uint32_t sweep, int lane, bool have_cov = false, bool have_z = false)
{
    // have_cov: GibbsRtIrt's V and L are already in `work` (rtirt_beta_cov by another wave); have_z: beta's standard normals are already in `work`
    // (zb, drawn by another wave during the head).  Same values either way.
    const double* cstp = T.cst;
    ERM_TINY_COMMON
    const double* XtX = sh_x;                               // p x p, column-major with leading dimension PMAX
    const double* Xinv = sh_x + PMAX * PMAX;                // (x'x)^-1, same layout
    (void)XtX; (void)Xinv;
    if (STEP == 0) {
        if (MODEL == MLIRT) {
            if (PART != 0)
            // getSubjCoefficientsMlIrt src/Draw.pl.jl:351-357 : beta = (x'x) \ x'theta ; beta[1] = 0 unless intercept
            if (lane < p) {
                double t = 0.0;
                for (int v = 0; v < p; ++v) t += Xinv[lane + v * PMAX] * G0[v];
                beta[lane] = (lane == 0 && !T.intercept) ? 0.0 : t;
            }
        } else if (MODEL == NULLM) {
            // src/GibbsRtIrt.pl.jl:380: Para.beta = zeros(nFeat+1, 2) every sweep
            if (PART != 0 && lane < 2 * p) { bn[lane] = 0.0; beta[(lane / p) * PMAX + (lane % p)] = 0.0; }
            if (PART != 0 && lane < 8) qf[lane] = 0.0;
        } else if (MODEL == RTIRT) {
            // drawSubjCoefficients src/Draw.pl.jl:380-393.  Posterior precision = 11' + kron(inv(Sigp), x'x) (the reference's
            // `1/sigma^2 .+ M` adds 1 to EVERY element), so by Sherman-Morrison
            //   parV = Minv - v v'/(1 + 1'v),  Minv = kron(Sigp, (x'x)^-1),  v = Minv 1.
            const int n = 2 * p;
            double* V = work, *L = V + n * n, *tv = L + n * n, *pmv = tv + 3 * n, *zb = pmv + n;
            const double* xt = G0, *xz = G0 + p;
            if (PART != 1) {
            if (!have_cov) rtirt_beta_cov(p, Sigp, Xinv, work, lane);
            const double s00 = Sigp[0], s10 = Sigp[1], s01 = Sigp[2], s11 = Sigp[3];
            const double isdet = q_rcp(s00 * s11 - s10 * s01);
            const double iO[4] = { s11 * isdet, -s10 * isdet, -s01 * isdet, s00 * isdet };
            if (lane < n) {
                const int a_ = lane / p, u = lane % p;
                tv[lane] = 0.0 + xt[u] * iO[a_] + xz[u] * iO[a_ + 2];      // vec(x'eta * inv(Sigp)')
            }
            wave_sync();
            double pm = 0.0;
            if (lane < n) for (int jj = 0; jj < n; ++jj) pm += V[lane + jj * n] * tv[jj];
            if (lane < n) pmv[lane] = pm;
            }   // ---- part A ends (everything above depends only on the previous pass's statistics and Sigma_p_{t-1})
            if (PART != 0) {
            const double pm = lane < n ? pmv[lane] : 0.0;
            const double zi = lane < n ? (have_z ? zb[lane] : beta_normal(T.seed, T.chain, sweep, lane)) : 0.0;     // z_i drawn by lane i
            double t = pm;
            for (int jj = 0; jj < n; ++jj) {
                const double zj = __shfl(zi, jj, 64);
                if (lane < n && jj <= lane) t += L[lane + jj * n] * zj;
            }
            if (lane < n) {
                if (!T.intercept && (lane == 0 || lane == p)) t = 0.0;          // src/GibbsRtIrt.pl.jl:293-295
                bn[lane] = t;
                beta[(lane / p) * PMAX + (lane % p)] = t;
            }
            wave_sync();
            // quadratic forms needed by Sigma_p: qf[a + 2b] = beta_a' x'x beta_b, qf[4 + a + 2b] = beta_a' x'eta_b; lane f computes form f
            if (lane < 8) {
                const int a_ = lane & 1, b_ = (lane >> 1) & 1;
                double v = 0.0;
                if (lane < 4) { for (int u = 0; u < p; ++u) for (int w = 0; w < p; ++w) v += bn[a_ * p + u] * XtX[u + w * PMAX] * bn[b_ * p + w]; }
                else { const double* xe = b_ == 0 ? G0 : G0 + p; for (int u = 0; u < p; ++u) v += bn[a_ * p + u] * xe[u]; }
                qf[lane] = v;
            }
            }   // PART != 0
        } else if (MODEL == LATENT) {
            // drawSubjCoefficientsLatent src/Draw.pl.jl:399-416, x = [1 X theta] (q = p + 1 columns):
            //   parV = inv(11' + x'x / Sigp22)  (`1/sb0^2 .+ M` adds 1 to EVERY element), parM = parV x'zeta / Sigp22,
            //   beta = parM + chol(Symmetric(parV)).L z
            const int q = p + 1;
            double* Mx = work, *L = Mx + q * q, *V = L + q * q, *tv = V + q * q;
            const double* xt = G0; const double tt = G0[p]; const double* xz = G0 + p + 1; const double tz = G0[2 * p + 1];
            const double iO = q_rcp(Sigp[3]);
            if (PART != 1) {
            for (int e = lane; e < q * q; e += 64) {
                const int i = e % q, jj = e / q;
                const double a_ = (i < p && jj < p) ? XtX[i + jj * PMAX] : ((i == p && jj == p) ? tt : xt[i < jj ? i : jj]);
                Mx[e] = 1.0 + iO * a_;
            }
            if (lane < q) tv[lane] = 0.0 + (lane < p ? xz[lane] : tz) * iO;
            wave_sync();
            chol_lower_wave(q, Mx, L, lane);                     // precision = L L'
            if (lane < q) {                                      // lane k: column k of the inverse (forward, then backward substitution)
                const int k = lane;
                for (int i = 0; i < q; ++i) {
                    double t = (i == k) ? 1.0 : 0.0;
                    for (int m = 0; m < i; ++m) t -= L[i + m * q] * V[m + k * q];
                    V[i + k * q] = q_div(t, L[i + i * q]);
                }
                for (int i = q - 1; i >= 0; --i) {
                    double t = V[i + k * q];
                    for (int m = i + 1; m < q; ++m) t -= L[m + i * q] * V[m + k * q];
                    V[i + k * q] = q_div(t, L[i + i * q]);
                }
            }
            wave_sync();
            }   // ---- part A ends
            if (PART != 0) {
            for (int e = lane; e < q * q; e += 64) {             // Symmetric(parV): upper triangle
                const int i = e % q, jj = e / q;
                Mx[e] = V[(i < jj ? i : jj) + (i < jj ? jj : i) * q];
            }
            wave_sync();
            double pm = 0.0;
            if (lane < q) for (int jj = 0; jj < q; ++jj) pm += Mx[lane + jj * q] * tv[jj];
            chol_lower_wave(q, Mx, L, lane);
            const double zi = lane < q ? (have_z ? tv[q + lane] : beta_normal(T.seed, T.chain, sweep, lane)) : 0.0;      // zb = tv + q
            double t = pm;
            for (int jj = 0; jj < q; ++jj) {
                const double zj = __shfl(zi, jj, 64);
                if (lane < q && jj <= lane) t += L[lane + jj * q] * zj;
            }
            if (lane < q) {
                if (!T.intercept && lane == 0) t = 0.0;          // src/GibbsRtIrtLatent.pl.jl:184-186
                bn[lane] = t; beta[lane] = t;
            }
            }   // PART != 0
        } else if (MODEL == LATENTQR) {
            // getSubjCoefficientsLatentQr src/Draw.pl.jl:446-458 : beta = (x'x)^-1 x'(zeta - k1 nu), x = [1 X theta];
            // block inverse with the constant (X~'X~)^-1 and the Schur complement of the theta column.
            if (PART != 0 && lane == 0) {
                const int q = p + 1;
                const double* xt = G0; const double tt = G0[p]; const double* xu = G0 + p + 1;
                const double tu = G0[2 * p + 1];
                double* h = work, *g = h + PMAX;
                double hx = 0.0, hu = 0.0;
                for (int u = 0; u < p; ++u) {
                    double t1 = 0.0, t2 = 0.0;
                    for (int v = 0; v < p; ++v) { t1 += Xinv[u + v * PMAX] * xt[v]; t2 += Xinv[u + v * PMAX] * xu[v]; }
                    h[u] = t1; g[u] = t2;
                }
                for (int u = 0; u < p; ++u) { hx += xt[u] * h[u]; hu += h[u] * xu[u]; }
                const double b2 = q_div(tu - hu, tt - hx);
                for (int u = 0; u < p; ++u) bn[u] = g[u] - h[u] * b2;
                bn[p] = b2;
                if (!T.intercept) bn[0] = 0.0;                                   // src/GibbsRtIrtLatent.pl.jl:288-290
                for (int u = 0; u < q; ++u) beta[u] = bn[u];
            }
        }
    }

    wave_sync();
    if (PART == 0) return;
    // derived scalar for the row pass: sum_j 1/sig2t_j (part[j] = 1/sig2t_j; fixed order: lane l sums j = l, l+64, ..., then a butterfly)
    {
        double t = 0.0;
        for (int jj = lane; jj < J; jj += 64) t += part[jj];
        t = bfly_sum(t, 1, 64);
        if (lane == 0) par[par_off_derived(J)] = t;
    }
    if constexpr (rtll_stats<MODEL, 0>() && STEP == 0) {
        // the part of this sweep's response-time log-likelihood that does not depend on the subjects (see ERM_RTLL_STATS): derived[1]
        // (explicit fma everywhere in this statistic: every instantiation of the kernels must form the same bits, whatever the compiler would contract)
        double cc = 0.0;
        for (int jj = lane; jj < J; jj += 64) {
            const double lc = par[2 * J + jj] - cm[jj];
            cc = fma(Nd, LOG_2PI + q_log(par[3 * J + jj]), cc);
            cc = fma(part[jj], fma(Nd * lc, lc, csq[jj]), cc);
        }
        cc = bfly_sum(cc, 1, 64);
        if (lane == 0) par[par_off_derived(J) + 1] = -0.5 * cc;
    }
    // =========================================================== Sigma_p_t | beta_t (thread 0; its random numbers were pre-drawn above)
    if (lane == 0 && STEP == 0 && MODEL != MLIRT) {
        double S[4] = { 1.0, 0.0, 0.0, 1.0 };
        if (fam_rt(MODEL)) {
            // drawSubjCovariance src/Draw.pl.jl:499-515 (Null: drawSubjCovarianceNull :522-535, the same draw with beta = 0) : InverseWishart(N+3, e'e + I), e'e from sufficient statistics
            // (the quadratic forms beta_a' x'x beta_b and beta_a' x'eta_b were reduced by wave 0 just above: qf[0..7])
            const double tt = G0[2 * p], tz = G0[2 * p + 1], zz = G0[2 * p + 2];
            const double* bAb = qf; const double* bx = qf + 4;
            const double ee00 = tt - 2.0 * bx[0] + bAb[0];
            const double ee01 = tz - bx[0 + 2 * 1] - bx[1 + 2 * 0] + bAb[0 + 2 * 1];
            const double ee11 = zz - 2.0 * bx[3] + bAb[3];
            const double Psi[4] = { ee00 + 1.0, ee01, ee01, ee11 + 1.0 };
            invwishart2(Psi, spd, S);
        } else if (MODEL == LATENT) {
            // drawSubjCovarianceLatent src/Draw.pl.jl:563-579 : InverseGamma(da + N/2, db + sum((zeta - x beta)^2)/2), x = [1 X theta]
            const double* xt = G0; const double tt = G0[p]; const double* xz = G0 + p + 1;
            const double tz = G0[2 * p + 1], zz = G0[2 * p + 2];
            double sr2 = zz;
            for (int u = 0; u < p; ++u) sr2 -= 2.0 * bn[u] * xz[u];
            sr2 -= 2.0 * bn[p] * tz;
            for (int u = 0; u < p; ++u) for (int v = 0; v < p; ++v) sr2 += bn[u] * XtX[u + v * PMAX] * bn[v];
            for (int u = 0; u < p; ++u) sr2 += 2.0 * bn[u] * xt[u] * bn[p];
            sr2 += bn[p] * bn[p] * tt;
            S[3] = q_div(1e-3 + sr2 / 2.0, spd[0]);
        } else if (MODEL == LATENTQR) {
            // drawSubjCovarianceLatentQr src/Draw.pl.jl:585-606 with the N x N '/' quirk in closed form
            const double* xt = G0; const double tt = G0[p]; const double* xu = G0 + p + 1;
            const double tu = G0[2 * p + 1], uu = G0[2 * p + 2], snu = G0[2 * p + 3], snu2 = G0[2 * p + 4];
            double sr2 = uu;
            for (int u = 0; u < p; ++u) sr2 -= 2.0 * bn[u] * xu[u];
            sr2 -= 2.0 * bn[p] * tu;
            for (int u = 0; u < p; ++u) for (int v = 0; v < p; ++v) sr2 += bn[u] * XtX[u + v * PMAX] * bn[v];
            for (int u = 0; u < p; ++u) sr2 += 2.0 * bn[u] * xt[u] * bn[p];
            sr2 += bn[p] * bn[p] * tt;
            const double sw = 2.0 * T.k2 * snu, sw2 = 4.0 * T.k2 * T.k2 * snu2;
            double quirk = q_div(sr2 * sw, sw2);
            if (T.sigp_mode == 1) {
                // the evidently intended sum_i r_i^2 / (2 k2 nu_i), r = u - x~ beta, from the 1/nu-weighted Gram statistics
                const int q = p + 1, ntri = q * (q + 1) / 2;
                const double* Wg = G0 + 2 * p + 7;
                double sw_r2 = Wg[ntri + q];
                for (int u = 0; u < q; ++u) sw_r2 -= 2.0 * bn[u] * Wg[ntri + u];
                int e = 0;
                for (int u = 0; u < q; ++u) for (int v = u; v < q; ++v, ++e) sw_r2 += (u == v ? 1.0 : 2.0) * bn[u] * Wg[e] * bn[v];
                quirk = sw_r2 / (2.0 * T.k2);
            }
            const double parB = 1e-3 + quirk + snu;
            S[3] = q_div(parB, spd[0]);
        } else {
            // drawSubjCovarianceCross src/Draw.pl.jl:542-557
            const double zz = st1[NSTAT1 * J + 0];
            S[3] = q_div(1e-3 + zz / 2.0, spd[0]);
        }
        if (T.cov2one) d_cov2one(S);
        for (int e = 0; e < 4; ++e) Sigp[e] = S[e];
    }
    wave_sync();
}

// the whole step in a single workgroup
template <int MODEL, int STEP>
__device__ __forceinline__ void tiny_draws(const TinyArgs& T, double* par, const double* st0, const double* st1, double* part, double* work,
                                           const double* sh_x, uint32_t sweep)
{
    tiny_items<MODEL, STEP>(T, par, st0, st1, part, work, sweep, T.cst);
    __syncthreads();
    if (threadIdx.x < 64) tiny_struct<MODEL, STEP, 2>(T, par, st0, st1, part, work, sh_x, sweep, (int)threadIdx.x);
    __syncthreads();
}

// Bookkeeping of one tiny step by ONE workgroup: item-level trace row, finite check, the updated parameter block and the sweep / row
// counters go to global memory (par_out / ctl_out may alias the inputs when a single workgroup runs the step).
template <int MODEL, int STEP>
__device__ __forceinline__ void tiny_publish(const TinyArgs& T, const double* par, uint32_t sweep, uint32_t row, int tid, int nthreads,
                                             double* par_out = nullptr, Ctl* ctl_out = nullptr, uint32_t burn_rows = 0u)
{
    // (a persistent launch passes this sweep's halves of the double buffers and the burn-in row count it carries; otherwise T's)
    if (!par_out) { par_out = T.par_out; ctl_out = T.ctl_out; burn_rows = T.ctl->burn_rows; }
    const int J = T.J, p = T.nFeat + 1;
    const double* Sigp = par + par_off_sigp(J);
    const double* beta = par + par_off_beta(J);
    const bool last_step = (MODEL != CROSSQR) || STEP == 1;
    if (last_step && T.tr_item) {
        const int wrow = 4 * J + T.nq;
        double* tr = T.tr_item + (size_t)row * wrow;
        for (int e = tid; e < 4 * J; e += nthreads) tr[e] = par[e];
        for (int k = tid; k < T.nq; k += nthreads) {
            double v;
            if (MODEL == MLIRT) v = beta[k];
            else if (fam_rt(MODEL)) { const int nb = 2 * p; v = k < nb ? (k < p ? beta[k] : beta[PMAX + k - p]) : Sigp[k - nb]; }
            else if (fam_cq(MODEL)) v = k < J ? par[4 * J + k] : Sigp[k - J];
            else { const int nb = p + 1; v = k < nb ? beta[k] : Sigp[k - nb]; }
            tr[4 * J + k] = v;
        }
    }
    for (int e = tid; e < par_size(J); e += nthreads) {
        if (!(fabs(par[e]) < 1e300)) atomicCAS(&T.ctl_err->err, 0u, 1u + (uint32_t)e);   // a non-finite entry of the parameter block
        par_out[e] = par[e];
    }
    if (tid == 0 && STEP == 0) { ctl_out->sweep = sweep; ctl_out->row = row; ctl_out->burn_rows = burn_rows; ctl_out->first = 0u; }
}


// ---------------------------------------------------------------------------------------------------------------------
// Row pass.  blockDim.x = 64 * nWaves; workgroup b owns a contiguous range of subjects.
//   phase 1 (lane (r, s) of a wave: subject r of the wave's group, items s, s+W, ...):
//       row sums over omega_t / Y / logT, theta_t and zeta_t draws, per-subject outputs and global statistics, then the
//       persistent-lane loop that draws omega_{t+1};
//   barrier (everything the workgroup wrote for its own subjects is visible to all its waves);
//   phase 2 (lane = item j, waves stride over the workgroup's subjects): per-cell log-likelihood terms and the item
//       statistics, accumulated in fp64 REGISTERS with no cross-lane traffic; CrossQr's per-cell nu_{t+1} draw lives here.
//   epilogue: fixed-order sum of the waves' accumulators -> this workgroup's slab row.
// ---------------------------------------------------------------------------------------------------------------------
// FUSED (single-pass models): the kernel first runs this sweep's tiny step itself -- every workgroup redundantly, from the previous
// launch's group-reduced statistics (T.slab0) and parameter block (T.par), bit-identically; workgroup 0 publishes the results
// (T.par_out, T.ctl_out, traces).  Inputs and outputs are distinct (double-buffered) allocations, so a workgroup that starts late
// never sees a half-updated block.  That removes one kernel boundary and the tiny kernel's cold start from every sweep.
// PERSIST (FUSED only; small data sets): A.nsweeps sweeps in ONE launch, a packet exchange of the statistics rows between them instead of a kernel boundary; the double buffers
// alternate inside the launch exactly as consecutive launches alternate them, so the chain is the per-sweep schedule's bit for bit.
template <int MODEL, typename real, int PHASE, bool FUSED, bool PERSIST = false>
__global__ void __launch_bounds__(PERSIST ? PERSIST_THREADS : max_block_threads(MODEL, sizeof(real) == 8)) pass_kernel(const PassArgs<real> A_, const TinyArgs T_)
{
    const PassArgs<real>& A = A_; const TinyArgs& T = T_;
    static_assert(!PERSIST || (FUSED && PHASE == 0), "persistent launches exist for the fused single-pass sweep only");
    using ST = Stats<MODEL, PHASE>;
    constexpr int NSTAT = ST::NSTAT;
    // the PG phase hands the workgroup's cells out from ONE dynamic queue (round 2: one queue per wave made the waves of a SIMD finish up to
    // 50 us apart -- the hardware favours a SIMD's oldest wave)
    const int J = A.J, W = A.W, R = 64 / W, IPL = A.IPL;
    const int F = A.nFeat, p = F + 1;                // design [1 X]
    const int NG = ST::ng(p) + A.ngx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nWaves = blockDim.x >> 6;
    const int s = lane & (W - 1), r = lane >> A.logW;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* sh_struct = reinterpret_cast<double*>(smem);        // 8 + 2*PMAX doubles
    // the per-wave item accumulators close the launch's dynamic LDS (A.acc_off)
    double* sh_acc = reinterpret_cast<double*>(smem + A.acc_off);    // [nWaves][NSTAT][J]
    double* sh_gacc = sh_struct + 8 + 2 * PMAX;                 // [nWaves][NG]
    real* sh_item = reinterpret_cast<real*>(sh_gacc + (((size_t)nWaves * NG + 1) & ~(size_t)1));    // [NITEMARR][JS], JS = item_stride(J) >= J (erm_layout.hpp); 16-byte aligned
    const int JS = item_stride(J);
    real* sh_a = sh_item, *sh_b = sh_item + JS, *sh_a2 = sh_item + 2 * JS, *sh_a2b = sh_item + 3 * JS;
    real* sh_lamc = sh_item + 4 * JS, *sh_isig = sh_item + 5 * JS, *sh_lsig = sh_item + 6 * JS, *sh_rho = sh_item + 7 * JS;
    real* sh_rs = sh_item + NITEMARR * JS;                                         // [rows_per_block][3] row sums, indexed by the subject's position in the workgroup
                                                                                   // (the region holds nWaves * 4 * rows_per_wave >= 4 * rows_per_block values)
    // the per-item product of the cell streams' Philox blocks (philox4x32_10_vk_cell), one uint2 per item, in the two item arrays these models never read
    // (log sig2t: the response-time log-likelihood comes from statistics; rho: the Cross family's) -- 2 J reals = J uint2 in either engine
#ifndef ERM_PHILOX_HOIST
#define ERM_PHILOX_HOIST 1
#endif
    // (not in the persistent small-data kernel: a workgroup draws about one cell per lane there, so the item products are not amortised and their two multiplications
    // sit in the head's dependent chain -- 1 000 x 15 GibbsMlIrt 13.6 against 13.4 us per sweep)
    constexpr bool PHX = ERM_PHILOX_HOIST != 0 && !PERSIST && PHASE == 0 && !fam_cq(MODEL) && (MODEL == MLIRT || rtll_stats<MODEL, PHASE>());
    [[maybe_unused]] uint2* sh_phx = reinterpret_cast<uint2*>(sh_lsig);
    const int NV = nv_of(MODEL, A.nFeat);
    real* sh_val = sh_item + NITEMARR * JS + (size_t)nWaves * 4 * A.rows_per_wave;   // [rows_per_block][NV] per-subject values of the global statistics

    // diagnostics: per-wave phase timeline of workgroups 0 and gridDim/2 (lane 0 of each wave stamps the constant-rate wall clock)
    // (compiled in only with -DERM_TIMELINE_BUILD: even the disabled checks cost registers and ~3 us per sweep)
    auto stamp = [&](int k) {
#ifdef ERM_TIMELINE_BUILD
        if (A.dbg_ts && lane == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x / 2) && wave < 16)
            A.dbg_ts[((blockIdx.x == 0 ? 0 : 1) * 16 + wave) * 16 + k] = wall_clock64();
#else
        (void)k;
#endif
    };
    stamp(0);
    // The launch's inputs are requested in ONE trip: every source's first element per thread is loaded (clamped index, so the loads are
    // unconditional and the compiler issues them back to back) before anything waits.  As separate load -> wait -> LDS-store loops the head
    // made six dependent trips to L2 / HBM before its first barrier.  And the waves SHARE the work by role: the time to that barrier is
    // instruction issue -- the sixteen waves of a starting workgroup, four to a SIMD, each ran all ~600 instructions of this prologue (3.8 us) --
    // so the waves that hold an element of the statistics rows (role A) do nothing else, and the others (role B) take the tables, x'x, the item
    // constants and the parameter block.  Too few waves for that (small workgroups, long tests): every wave does everything, as before.
    const int tid0 = (int)threadIdx.x, nthr0 = (int)blockDim.x;
    const int NSh = NSTAT * J + NG;
    const int kA = FUSED ? (NSh + 63) >> 6 : 0;
    const bool split = FUSED && nWaves >= kA + 4;
    const bool roleA = FUSED && (!split || wave < kA);
    const bool roleB = !split || wave >= kA;
    const int tB = split ? tid0 - 64 * kA : tid0, nB = split ? nthr0 - 64 * kA : nthr0;       // role B's thread index and count
    // the Polya-Gamma proposal table: {lam, 1/lam, M, q} per z-bin rounded to fp32 (decisions), 1/lam in fp64 (the fp64 engine's values);
    // published by the first barrier below (pass_static_lds() in erm_layout.hpp counts these arrays)
    [[maybe_unused]] const float4* sh_pgf = nullptr;
    [[maybe_unused]] const double* sh_pgc = nullptr;
    [[maybe_unused]] double2 pg_lo = make_double2(0.0, 0.0), pg_hi = make_double2(0.0, 0.0);
    // FUSED: the (first) sweep's head inputs -- x'x and its inverse, the item constants, the parameter block, the first GROUP group rows of the statistics
    [[maybe_unused]] double hd_x = 0.0, hd_c = 0.0, hd_p = 0.0, hd_r[16];
    [[maybe_unused]] const double* par0 = T.par; [[maybe_unused]] const double* slab00 = T.slab0;
    if constexpr (PERSIST) { par0 = (A.cur0 & 1u) ? A.parB[1] : A.parB[0]; slab00 = (A.cur0 & 1u) ? A.gslabB[1] : A.gslabB[0]; }
    if (roleB) {
        if constexpr (PHASE == 0) {
            const double2* b = reinterpret_cast<const double2*>(A.pgtab) + 2 * (tB < PG_NBIN ? tB : 0);
            pg_lo = b[0]; pg_hi = b[1];
        }
        if constexpr (FUSED) {
            hd_x = T.cst[cst_off_xtx(J) + (tB < 2 * PMAX * PMAX ? tB : 0)];
            hd_c = T.cst[tB < 3 * J + 2 ? tB : 0];
            hd_p = par0[tB < par_size(J) ? tB : 0];
        }
    }
    if constexpr (FUSED) {
        if (roleA) {
            // (the host allocates at least GROUP rows: rows beyond nb0 are requested too and masked in the sum; per-row clamps cost ~110 scalar
            // instructions per wave)
            const double* rp = slab00 + (tid0 < NSh ? tid0 : 0);
#pragma unroll
            for (int u = 0; u < 16; ++u) hd_r[u] = rp[(size_t)u * NSh];
        }
    }
    uint32_t c_sweep = A.ctl->sweep, c_row = A.ctl->row;        // the chain's counters: read once, carried in registers through a persistent launch
    const uint32_t c_burn = A.ctl->burn_rows;
    [[maybe_unused]] const uint32_t c_first = A.ctl->first;     // no sweep of this erm_run has been drawn yet
    asm volatile("" ::: "memory");                  // the loads above stay above the table arithmetic below
    // fp64 engine: the 2 KB table of fm::log (filled here; the first barrier below -- the head's, or the staging barrier -- publishes it)
    [[maybe_unused]] const double2* logtab = nullptr;
    if constexpr (sizeof(real) == 8) {
        __shared__ double2 sh_logtab[128];
        if (roleB) fm::fill_log_table(sh_logtab, tB, nB);
        logtab = sh_logtab;
    }
    if constexpr (PHASE == 0) {
        __shared__ float4 sh_pgf_[PG_NBIN];
        if (roleB) {
            if (tB < PG_NBIN) sh_pgf_[tB] = make_float4((float)pg_lo.x, (float)pg_lo.y, (float)pg_hi.x, (float)pg_hi.y);
            for (int k = tB + nB; k < PG_NBIN; k += nB) {
                const double* b = A.pgtab + 4 * k;
                sh_pgf_[k] = make_float4((float)b[0], (float)b[1], (float)b[2], (float)b[3]);
            }
        }
        sh_pgf = sh_pgf_;
        if constexpr (sizeof(real) == 8) {
            __shared__ double sh_pgc_[PG_NBIN];
            if (roleB) {
                if (tB < PG_NBIN) sh_pgc_[tB] = pg_lo.y;
                for (int k = tB + nB; k < PG_NBIN; k += nB) sh_pgc_[k] = A.pgtab[4 * k + 1];
            }
            sh_pgc = sh_pgc_;
        }
    }
    const uint8_t* __restrict__ gY = A.Y;
    const real* __restrict__ gC = A.C;
    const real* __restrict__ gX = A.X;

    // FUSED only: LDS scratch of the tiny step, appended to the pass layout
    const int NS0 = NSTAT * J + NG;
    double* st0 = reinterpret_cast<double*>(sh_val + (((size_t)A.rows_per_block * NV + 1) & ~(size_t)1));   // 8-byte aligned
    double* part = st0 + NS0;                                   // J doubles (1/sig2t_j)
    double* work = part + J;
    double* sh_x = work + TINY_WORK;
    double* lp = sh_x + 2 * PMAX * PMAX;
    double* lcst = lp + par_size(J);                            // K0, column means, csq, muLam, sdLam (3J + 2)
    const bool writer = blockIdx.x == 0;
    if constexpr (FUSED) {
        // the head inputs requested at the top of the kernel go to LDS (a persistent launch: its first sweep's; the later ones keep x'x, the
        // constants and the parameter block in LDS and receive the statistics as packets), then whatever a thread's first element did not
        // cover (long tests, large grids); the statistics in reduce_rows' order
        if (roleB) {
            if (tB < 2 * PMAX * PMAX) sh_x[tB] = hd_x;
            for (int e = tB + nB; e < 2 * PMAX * PMAX; e += nB) sh_x[e] = T.cst[cst_off_xtx(J) + e];
            if (tB < 3 * J + 2) lcst[tB] = hd_c;
            for (int e = tB + nB; e < 3 * J + 2; e += nB) lcst[e] = T.cst[e];
            if (tB < par_size(J)) lp[tB] = hd_p;
            for (int e = tB + nB; e < par_size(J); e += nB) lp[e] = par0[e];
        }
        if (roleA) {
            const int nA = split ? 64 * kA : nthr0;
            if (tid0 < NS0) {
                double t = 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u) t += (u < T.nb0) ? hd_r[u] : 0.0;
                for (int b0 = 16; b0 < T.nb0; b0 += 16) {
                    double v[16];
#pragma unroll
                    for (int u = 0; u < 16; ++u) v[u] = slab00[(size_t)(b0 + u < T.nb0 ? b0 + u : b0) * NS0 + tid0];
#pragma unroll
                    for (int u = 0; u < 16; ++u) t += (b0 + u < T.nb0) ? v[u] : 0.0;
                }
                st0[tid0] = t;
            }
            reduce_rows(slab00, T.nb0, NS0, st0, tid0 + nA, nA);
        }
    }
    const uint32_t n_loop = PERSIST ? A.nsweeps : 1u;
    for (uint32_t ks = 0; ks < n_loop; ++ks) {
    // PERSIST, the response-time models: inside the sweep loop the arguments are read through a pointer to the kernel-argument segment that the compiler
    // cannot see through, so that nothing derived from an argument is hoisted out of the loop and kept (spilled) across it -- every field is an s_load
    // where it is used.  GibbsRtIrt's kernel: 594 -> 366 spilled SGPRs, 25 -> 0 spilled VGPRs, 1 000 x 15 20.9 -> 18.8 us per sweep.  GibbsMlIrt's has
    // less to spill and lost 2 % to the loads' latency: it keeps the plain arguments.
    constexpr bool LAUNDER = PERSIST && MODEL != MLIRT;
    const char __attribute__((address_space(4)))* kap = (const char __attribute__((address_space(4)))*)__builtin_amdgcn_kernarg_segment_ptr();
    if constexpr (LAUNDER) asm volatile("" : "+s"(kap));
    const PassArgs<real>& A = LAUNDER ? *reinterpret_cast<const PassArgs<real>*>((const char*)kap) : A_;
    const TinyArgs& T = LAUNDER ? *reinterpret_cast<const TinyArgs*>((const char*)kap + ((sizeof(PassArgs<real>) + 7) & ~(size_t)7)) : T_;
    // this sweep's halves of the double buffers (a persistent launch alternates them as consecutive launches do; A and T themselves stay untouched --
    // as modified private copies every field of the two argument structs lived in a scalar register for the whole loop: 250 more SGPR spills)
    double* k_par_out = T.par_out; Ctl* k_ctl_out = T.ctl_out; double* k_gslab_out = A.gslab; int k_first = (int)c_first;
    if constexpr (PERSIST) {
        const bool odd = ((A.cur0 + ks) & 1u) != 0u;
        k_par_out = odd ? A.parB[0] : A.parB[1];
        k_ctl_out = odd ? A.ctlB[0] : A.ctlB[1];
        k_gslab_out = odd ? A.gslabB[0] : A.gslabB[1];
        if (ks > 0) { k_first = 0; stamp(0); }
    }
    uint32_t sweep = c_sweep, trow = c_row;
    [[maybe_unused]] bool have_cov = false;                                     // GibbsRtIrt, later sweeps of a persistent launch: see the head
    // (persistent launches only: elsewhere wave 0's chain hides behind the other waves' row sums, and the extra block cost the fp32 large-data kernel 0.9 %)
    [[maybe_unused]] const bool have_z = PERSIST && (MODEL == RTIRT || MODEL == LATENT) && nWaves >= 5;
    const double* parsrc = A.par;
    if constexpr (FUSED) {
        // ------------------------------------------------------------------------------------------------ this sweep's tiny step
        const int tid = threadIdx.x, nthr = blockDim.x;
        if (PERSIST && ks > 0) {
            // GibbsRtIrt: while waves 0.. wait for the statistics, the last wave makes beta's covariance and its Cholesky factor from Sigma_p_{t-1}
            // (in lp since the previous sweep) -- 3.5 us that would otherwise head wave 0's chain after the barrier
            if constexpr (MODEL == RTIRT) {
                have_cov = nWaves >= 3;
                if (have_cov && wave == nWaves - 1) rtirt_beta_cov(p, lp + par_off_sigp(J), sh_x + PMAX * PMAX, work, lane);
            }
            // (x'x and the item constants are still in LDS: nothing writes them)
            // later sweeps of a persistent launch: lp still holds the parameter block this workgroup drew (every workgroup runs the tiny step),
            // the statistics arrive as packets
            const unsigned long long* rows = A.xbuf + (size_t)((ks - 1u) & 1u) * gridDim.x * 2 * NS0;
            for (int e = tid; e < NS0; e += nthr) st0[e] = persist_get(rows, (int)gridDim.x, NS0, e, A.tag0 + ks, A.tmo);
            // the launch's time-out flag as this workgroup sees it now (the upper half of the row-group counter's slot is free): published by the barrier below
            if (tid == 0) reinterpret_cast<unsigned int*>(sh_struct + 7)[1] = __hip_atomic_load(A.tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if constexpr (PERSIST) {
            // a launch that has timed out is abandoned by every workgroup (uniformly: all threads read the same LDS word); erm_run restores the saved state
            if (ks > 0 && reinterpret_cast<const unsigned int*>(sh_struct + 7)[1] != 0u) __builtin_amdgcn_endpgm();
        }
        stamp(1);
        const uint32_t prev_row = c_row;
        sweep = c_sweep + 1u;
        ERM_DIAG_STOP(A, 30);
        trow = k_first ? prev_row : prev_row + 1u;
        // log-likelihood of the sweep the last pass completed (+ the subject-free part its tiny step left in the parameter block: ERM_RTLL_STATS)
        if (writer && tid == 0 && !k_first && T.tr_ll) T.tr_ll[prev_row] = st0[NS0 - 1] + (rtll_stats<MODEL, PHASE>() ? lp[par_off_derived(J) + 1] : 0.0);
        // item draws now; the structural chain (beta_t -> Sigma_p_t, ~7 us of dependent fp64 work on one wave) runs on wave 0 AFTER the
        // staging barrier below, concurrently with the other waves' row sums, which do not need it (see `sh_ready`)
        // wave 0 starts the pre-barrier part of the structural chain at once (it is the longest strand of the head and needs nothing of
        // this sweep's item draws); in workgroups of 256+ threads it owns no item threads and skips tiny_items altogether
        // beta's standard normals (GibbsRtIrt: 2p of them, GibbsRtIrtLatent: p + 1) need only the sweep number: wave 4, idle during the head, draws them
        // into `work` (zb) for wave 0's chain after the barrier
        if (have_z && wave == 4) {
            const int nz = MODEL == RTIRT ? 2 * p : p + 1;
            double* zb = MODEL == RTIRT ? work + 2 * nz * nz + 4 * nz : work + 3 * nz * nz + nz;
            if (lane < nz) zb[lane] = beta_normal(T.seed, T.chain, sweep, lane);
        }
        if (wave == 0) tiny_struct<MODEL, 0, 0>(T, lp, st0, nullptr, part, work, sh_x, sweep, lane, have_cov, have_z);
        stamp(14);
        if (wave != 0 || nthr < 256) tiny_items<MODEL, 0>(T, lp, st0, nullptr, part, work, sweep, lcst);
        stamp(15);
        // every item thread stages what it has just drawn for the row pass (same thread mapping as tiny_items), so ONE barrier ends the head
        {
            const int ioff = nthr >= 256 ? 128 : 0, nit = nthr - ioff, Jw = (J + 63) & ~63, rt_off = (nit >= 2 * Jw) ? Jw : 0;
            const int jstep = rt_off ? 2 * Jw : nit;
            for (int j = tid - ioff; tid >= ioff && j < J; j += jstep) {
                const double a = lp[j], b = lp[J + j];
                sh_a[j] = (real)a; sh_b[j] = (real)b; sh_a2[j] = (real)(a * a); sh_a2b[j] = (real)(a * a * b);
                if constexpr (PHX) sh_phx[j] = philox_item_product((uint32_t)j, sweep + 1u, (uint32_t)A.seed); else sh_rho[j] = (real)lp[4 * J + j];
            }
            for (int j = tid - ioff - rt_off; tid >= ioff + rt_off && j < J; j += jstep) {
                const double lam = lp[2 * J + j], sg = lp[3 * J + j];
                sh_lamc[j] = (real)(lam - lcst[cst_off_m(J) + j]); sh_isig[j] = (real)(1.0 / sg);
                if constexpr (!rtll_stats<MODEL, PHASE>() && !PHX) sh_lsig[j] = (real)log(sg);
            }
            if (tid == 0) { *reinterpret_cast<int*>(sh_struct + 5) = 0; *reinterpret_cast<unsigned int*>(sh_struct + 7) = 0u; }       // sh_ready, row-group counter
            for (int e = tid; e < nWaves * NG; e += nthr) sh_gacc[e] = 0.0;
        }
        __syncthreads();
        ERM_DIAG_STOP(A, 31);
        parsrc = lp;
    }
    const bool post_burn = trow >= c_burn;
    c_sweep = sweep; c_row = trow;                            // what tiny_publish stores as the next sweep's counters
    int* sh_ready = reinterpret_cast<int*>(sh_struct + 5);      // FUSED: set by wave 0 once sh_struct holds Sigma_p_t, beta_t, sum 1/sig2t

    // ---- stage item parameters and structural scalars (stand-alone row pass; a FUSED kernel has done it above)
    if (!FUSED) {
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        const double a = parsrc[j], b = parsrc[J + j], lam = parsrc[2 * J + j], sg = parsrc[3 * J + j], rho = parsrc[4 * J + j];
        sh_a[j] = (real)a; sh_b[j] = (real)b; sh_a2[j] = (real)(a * a); sh_a2b[j] = (real)(a * a * b);
        sh_lamc[j] = (real)(lam - (FUSED ? lcst[cst_off_m(J) + j] : A.cst[cst_off_m(J) + j])); sh_isig[j] = (real)(1.0 / sg);
        if constexpr (PHX) sh_phx[j] = philox_item_product((uint32_t)j, sweep + 1u, (uint32_t)A.seed); else { sh_lsig[j] = (real)log(sg); sh_rho[j] = (real)rho; }
    }
    if (FUSED && threadIdx.x == 0) *sh_ready = 0;
    if (!FUSED && threadIdx.x < 8 + 2 * PMAX) {
        double v = 0.0;
        if (threadIdx.x < 4) v = parsrc[par_off_sigp(J) + threadIdx.x];
        else if (threadIdx.x >= 8) v = parsrc[par_off_beta(J) + threadIdx.x - 8];
        else if (threadIdx.x == 4) v = parsrc[par_off_derived(J)];                // sum_j 1/sig2t_j
        sh_struct[threadIdx.x] = v;
    }
    for (int e = threadIdx.x; e < nWaves * NG; e += blockDim.x) sh_gacc[e] = 0.0;
    __syncthreads();
    }

    stamp(2);
    ERM_DIAG_STOP(A, 32);
    if constexpr (FUSED) {
        if (wave == 0) {
            // structural chain of this sweep's tiny step on wave 0 while waves 1.. stream their row sums; its results (Sigma_p_t, beta_t,
            // sum 1/sig2t) are first needed by phase 1 (ii), behind the barrier that ends the row sums; wave 0 joins the row sums (groups from a counter) when it is done.
            tiny_struct<MODEL, 0, 1>(T, lp, st0, nullptr, part, work, sh_x, sweep, lane, false, have_z);
            if (lane < 8 + 2 * PMAX) {
                double v = 0.0;
                if (lane < 4) v = lp[par_off_sigp(J) + lane];
                else if (lane >= 8) v = lp[par_off_beta(J) + lane - 8];
                else if (lane == 4) v = lp[par_off_derived(J)];
                if (lane < 5 || lane >= 8) sh_struct[lane] = v;     // slots 5-7: sh_ready, the PG phase's cell counter, the row-sum phase's group counter
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            if (lane == 0) __hip_atomic_store(sh_ready, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (writer) tiny_publish<MODEL, 0>(T, lp, sweep, trow, lane, 64, k_par_out, k_ctl_out, c_burn);
        }
    }
    stamp(3);
    const real k1 = (real)A.k1, k2 = (real)A.k2;
    double* acc = sh_acc + (size_t)wave * NSTAT * J;
    double ll = 0.0;

    const long long row0 = (long long)blockIdx.x * A.rows_per_block;
    const long long row1 = (row0 + A.rows_per_block < A.N) ? row0 + A.rows_per_block : A.N;
    const int nrows_blk = (int)(row1 - row0);
    // the workgroup's first cell / subject: offsets inside the workgroup fit 32 bits (< 2^22 cells), so no 64-bit index arithmetic per row
    [[maybe_unused]] const real* blk_omega = A.omega + (size_t)row0 * J;
    [[maybe_unused]] const uint8_t* blk_Y = gY + (size_t)row0 * J;
    [[maybe_unused]] const real* blk_C = (MODEL != MLIRT) ? gC + (size_t)row0 * J : nullptr;
    [[maybe_unused]] const real* blk_theta = A.theta + row0;
    [[maybe_unused]] const real* blk_zeta = A.zeta + row0;
    [[maybe_unused]] const real* blk_nu = (has_nu(MODEL) && fam_cq(MODEL)) ? A.nu + (size_t)row0 * J : nullptr;
    // (no wave owns subjects: the row sums take groups of subjects from a counter, the subject draws and the PG phase run over the whole
    // workgroup, the column phase deals batches of four subjects round-robin)
    ERM_DIAG_STOP(A, 1);

    // =================================================================================================== phase 1 (i)
    // sums over each subject's items; lane (r, s): subject r of the group, items s, s+W, ...
    // Groups of R = 64 / W consecutive subjects of the WORKGROUP are handed to the waves from an LDS counter (a wave's own slice of ~24 subjects
    // filled only 25 of the 32 row slots of its four groups; and a FUSED kernel's wave 0, busy with the structural chain, simply joins late).
    // Which wave sums a subject does not matter: the sums go to sh_rs by the subject's position in the workgroup.
    const int ngroups = (nrows_blk + R - 1) / R;
    auto next_group = [&]() -> int {
        unsigned int g = 0u;
        if (lane == 0) g = atomicAdd(reinterpret_cast<unsigned int*>(sh_struct + 7), 1u);
        return (int)__builtin_amdgcn_readfirstlane(g);
    };
#ifndef ERM_PAIRS_F32
#define ERM_PAIRS_F32 1       // round 3: the fp32 engine takes item pairs too (8-byte loads; fewer address computations per cell): 53.4 -> 52.9 us in an A/B
#endif
    if (A.mode == 1 && (sizeof(real) == 8 || ERM_PAIRS_F32) && (J & 1) == 0) {
        // fp64 engine, even test lengths: a lane takes PAIRS of neighbouring items (2s, 2s+1), (2(s+W), ...), so that omega and logT come in
        // 16-byte loads and a wave-instruction covers whole 128-byte lines of a row instead of 64-byte halves (120.4 -> 113.9 us per sweep;
        // the fp32 engine's 8-byte pairs gained nothing and stay on the scalar path)
        using real2 = typename std::conditional<sizeof(real) == 8, double2, float2>::type;
        const int P = J >> 1, IPP = (P + W - 1) / W;
        // the loop twice: with the item arrays' stride a compile-time constant (test lengths up to ITEM_STRIDE: the arrays' reads share one address) and as a variable
        auto pair_sums = [&](auto jsc) {
        const int js = jsc;
        for (;;) {
            const int g = next_group();
            if (g >= ngroups) break;
            const int qrow = g * R + r;                        // the subject's position in the workgroup
            const bool rowok = qrow < nrows_blk;
            const long long i = row0 + qrow;
            const unsigned int base = (unsigned int)(rowok ? qrow : 0) * (unsigned int)J;      // 32-bit offsets from the workgroup's first cell (< 2^22)
            real s0 = 0, s1 = 0, s2 = 0;
            const real thr = (PHASE == 1 && rowok) ? A.theta[i] : real(0);
            for (int k0 = 0; k0 < IPP; k0 += KB) {
                real2 wv[KB], cv[KB]; unsigned int yv[KB]; int jv4[KB]; bool ok4[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int q = s + W * (k0 + u);
                    ok4[u] = rowok && (k0 + u) < IPP && q < P;
                    jv4[u] = ok4[u] ? 2 * q : 0;
                    const unsigned int e = base + (unsigned int)jv4[u];
                    // (plain element offsets here: 32-bit BYTE offsets from the scalar base, which pay in the column phase below, cost the fp32 engine 1.2 us per
                    // sweep in this loop and change nothing in the fp64 one -- profiles/round4_ab_address_arithmetic.log)
                    if constexpr (PHASE == 0) { wv[u] = *reinterpret_cast<const real2*>(blk_omega + e); yv[u] = *reinterpret_cast<const unsigned short*>(blk_Y + e); }
                    else { yv[u] = 0u; if constexpr (has_nu(MODEL)) wv[u] = *reinterpret_cast<const real2*>(blk_nu + e); else { wv[u].x = real(1); wv[u].y = real(1); } }
                    if constexpr (MODEL != MLIRT) cv[u] = *reinterpret_cast<const real2*>(blk_C + e); else { cv[u].x = real(0); cv[u].y = real(0); }
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int j = jv4[u];
                    const real m = ok4[u] ? real(1) : real(0);
                    // the item pair's entries of every array, each ONE aligned real2 read: it + k * js (a, b, a^2, a^2 b, lambda - mean, 1/sig2t, log sig2t, rho); j is even,
                    // the arrays' base and stride are multiples of 16 bytes
                    const real* it = reinterpret_cast<const real*>(__builtin_assume_aligned(sh_item + j, sizeof(real2)));
                    auto pair_of = [&](int k) -> real2 { return *reinterpret_cast<const real2*>(it + k * js); };
                    if constexpr (PHASE == 0) {
                        const real kap0 = (yv[u] & 0xFFu) ? real(0.5) : real(-0.5), kap1 = (yv[u] >> 8) ? real(0.5) : real(-0.5);      // Y is 0/1 (erm_set_data checks): y - 1/2 by a select, not a conversion
                        const real2 pa = pair_of(0), pa2 = pair_of(2), pa2b = pair_of(3);
                        s0 += m * (pa2.x * wv[u].x + pa2.y * wv[u].y);
                        s1 += m * ((pa.x * kap0 + pa2b.x * wv[u].x) + (pa.y * kap1 + pa2b.y * wv[u].y));
                        if (fam_rt(MODEL) || fam_lq(MODEL)) { const real2 pl = pair_of(4), ps = pair_of(5); s2 += m * ((pl.x - cv[u].x) * ps.x + (pl.y - cv[u].y) * ps.y); }
                    } else {   // Cross family pass B: zeta sums with per-cell nu weights (src/Draw.pl.jl:201-202)
                        const real2 pl = pair_of(4), ps = pair_of(5), pr = pair_of(7);
                        const real nu0 = ok4[u] ? wv[u].x : real(1), nu1 = ok4[u] ? wv[u].y : real(1);
                        const real id0 = r_div(ps.x, k2 * nu0), id1 = r_div(ps.y, k2 * nu1);
                        s0 += m * (id0 + id1);
                        s2 += m * ((pl.x - cv[u].x - thr * pr.x + k1 * nu0) * id0 + (pl.y - cv[u].y - thr * pr.y + k1 * nu1) * id1);
                    }
                }
            }
            s0 = bfly_sum(s0, 1, W); s2 = bfly_sum(s2, 1, W);
            if (PHASE == 0) s1 = bfly_sum(s1, 1, W);
            if (rowok && s == 0) { real* o = sh_rs + 3 * qrow; o[0] = s0; o[1] = s1; o[2] = s2; }
        }
        };
        if (ITEM_STRIDE > 0 && JS == ITEM_STRIDE) pair_sums(std::integral_constant<int, ITEM_STRIDE>{}); else pair_sums(JS);
    } else if (A.mode == 1) {
        for (;;) {
            const int g = next_group();
            if (g >= ngroups) break;
            const int qrow = g * R + r;
            const bool rowok = qrow < nrows_blk;
            const long long i = row0 + qrow;
            const unsigned int base = (unsigned int)(rowok ? qrow : 0) * (unsigned int)J;
            real s0 = 0, s1 = 0, s2 = 0;
            // batches of 4 items per lane with every load issued before any use (clamped index + mask: no branches)
            const real thr = (PHASE == 1 && rowok) ? A.theta[i] : real(0);
            for (int k0 = 0; k0 < IPL; k0 += KB) {
                real wv[KB], cv[KB], yv[KB]; int jv4[KB]; bool ok4[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int j = s + W * (k0 + u);
                    ok4[u] = rowok && (k0 + u) < IPL && j < J;
                    jv4[u] = ok4[u] ? j : 0;
                    const unsigned int e = base + (unsigned int)jv4[u];
                    wv[u] = (PHASE == 0) ? blk_omega[e] : (has_nu(MODEL) ? blk_nu[e] : real(1));
                    yv[u] = (PHASE == 0) ? (real)blk_Y[e] : real(0);
                    cv[u] = (MODEL != MLIRT) ? blk_C[e] : real(0);
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int j = jv4[u];
                    const real m = ok4[u] ? real(1) : real(0);
                    if (PHASE == 0) {
                        const real kap = yv[u] - real(0.5);
                        s0 += m * (sh_a2[j] * wv[u]);
                        s1 += m * (sh_a[j] * kap + sh_a2b[j] * wv[u]);
                        if (fam_rt(MODEL) || fam_lq(MODEL)) s2 += m * ((sh_lamc[j] - cv[u]) * sh_isig[j]);
                    } else {   // CrossQr pass B: zeta sums with per-cell nu weights (src/Draw.pl.jl:201-202)
                        const real nu = ok4[u] ? wv[u] : real(1);
                        const real iden = r_div(sh_isig[j], k2 * nu);
                        s0 += m * iden;
                        s2 += m * ((sh_lamc[j] - cv[u] - thr * sh_rho[j] + k1 * nu) * iden);
                    }
                }
            }
            s0 = bfly_sum(s0, 1, W); s2 = bfly_sum(s2, 1, W);
            if (PHASE == 0) s1 = bfly_sum(s1, 1, W);
            if (rowok && s == 0) { real* o = sh_rs + 3 * qrow; o[0] = s0; o[1] = s1; o[2] = s2; }
        }
    }
    // every wave's row sums (and, FUSED, wave 0's structural results in sh_struct: it stored them before its own row sums) are visible to
    // the whole workgroup from here on
    __syncthreads();
    stamp(4);
    ERM_DIAG_STOP(A, 5);

    // =================================================================================================== phase 1 (ii)
    // one lane per subject: theta_t / zeta_t draws, per-subject outputs, structural log-likelihood, LatentQr's nu_{t+1}.  The workgroup's
    // subjects are dealt to consecutive lanes of the WORKGROUP (subject q of the block on lane q mod 64 of wave q / 64), not of the wave that
    // summed them: a wave's own slice is ~24 subjects, i.e. sixteen waves each paid a whole trip of this ~300-instruction phase with a third of
    // their lanes -- issue-bound, 7.2 us of the fp64 sweep; packed, seven waves make the trip and the others go straight to the next barrier.
    stamp(5);
    const real sig11 = (MODEL == MLIRT) ? real(1) : (real)sh_struct[0];
    const real sig22 = (real)sh_struct[3];
    const real sum_isig = (real)sh_struct[4];
    const double* beta = sh_struct + 8;
    // bivariate-normal log-density constants of Sigma_p (src/GibbsRtIrt.pl.jl:268-269)
    const double sp_det = sh_struct[0] * sh_struct[3] - sh_struct[1] * sh_struct[2];
    const double sp_idet = q_rcp(sp_det);
    // LatentQr's quantile weights (fp64): cB = sqrt(2 k2 + k1^2), lambda = parB^2 = (2 k2 + k1^2) / (Sigp22 k2), once per wave
    [[maybe_unused]] const double lq_cB = sqrt((double)(real(2) * k2 + k1 * k1));
    [[maybe_unused]] const double lq_lam = (double)(real(2) * k2 + k1 * k1) / ((double)sig22 * (double)k2);
    const double sp_c0 = -LOG_2PI - 0.5 * q_log(sp_det);
    const double sp_q00 = sh_struct[3] * sp_idet, sp_q01 = -(sh_struct[1] + sh_struct[2]) * sp_idet, sp_q11 = sh_struct[0] * sp_idet;
    for (int q0 = (int)threadIdx.x & ~63; q0 < nrows_blk; q0 += (int)blockDim.x) {
        const bool rok = q0 + lane < nrows_blk;
        const long long i = row0 + (rok ? q0 + lane : 0);  // clamped: loads are unconditional, stores masked
        const int li = (int)(i - row0);
        real th = A.theta[i];
        real ze = (MODEL != MLIRT) ? A.zeta[i] : real(0);
        real nu_row = real(1);
        if (MODEL == LATENTQR) nu_row = A.nu[i];
        real mu0a = 0, mu0b = 0;
        real xr[8];                                         // first 8 covariates of the subject, loaded once (all loads up front)
#pragma unroll
        for (int u = 0; u < 8; ++u) xr[u] = (PHASE == 0 && !fam_cq(MODEL) && u < F) ? gX[(size_t)i * F + u] : real(0);
        auto xcol = [&](int u) -> real {                    // column u of [1 X]
            if (u == 0) return real(1);
            real v = real(0);
            if (u <= 8) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v = (q == u - 1) ? xr[q] : v;
                return v;
            }
            return gX[(size_t)i * F + (u - 1)];
        };
        if (PHASE == 0 && !fam_cq(MODEL) && MODEL != NULLM) {
            for (int u = 0; u < p; ++u) {
                const real xu = xcol(u);
                mu0a += xu * (real)beta[u];
                if (MODEL == RTIRT) mu0b += xu * (real)beta[PMAX + u];
            }
        }
        real xb5 = 0, nu_next = real(0);
        if (A.mode == 1) {
            const real sA = sh_rs[3 * li], sB = sh_rs[3 * li + 1], sC = sh_rs[3 * li + 2];
            // one Philox block per subject and sweep feeds both row draws: words 0,1 -> theta's normal, words 2,3 -> zeta's
            uint32_t rw0, rw1, rw2, rw3;
            philox4x32_10((uint32_t)i + A.row_base, 0u, sweep, ((uint32_t)SITE_THETA << 24) | ((A.chain & 0xFFu) << 16), (uint32_t)A.seed, (uint32_t)(A.seed >> 32), rw0, rw1, rw2, rw3);
            if (PHASE == 0) {
                // theta: src/Draw.pl.jl:49-62 (prior x*beta[:,1]) / :67-80 (Null prior)
                const real mu0 = (MODEL == MLIRT || MODEL == RTIRT) ? mu0a : real(0);
                const real parV = q_rcp(q_rcp(sig11) + sA);
                const real parM = parV * (q_div(mu0, sig11) + sB);
                th = parM + q_sqrt(parV) * row_normal<real>(rw0, rw1, logtab);
            }
            if (fam_rt(MODEL) || fam_lq(MODEL)) {
                // zeta: src/Draw.pl.jl:132-141 / :161-174 (LatentQr) / :147-156 (Latent) / :119-127 (Null: prior N(0,1), Sigp unused)
                real mu0 = mu0b, s0 = sig22;
                if (MODEL == NULLM) { mu0 = real(0); s0 = real(1); }
                if (fam_lq(MODEL)) {
                    xb5 = mu0a + th * (real)beta[p];
                    mu0 = xb5 + k1 * nu_row;
                    s0 = sig22 * (k2 * nu_row);
                }
                const real parV = q_rcp(q_rcp(s0) + sum_isig);
                const real parM = parV * (q_div(mu0, s0) + sC);
                ze = parM + q_sqrt(parV) * row_normal<real>(rw2, rw3, logtab);
            }
            if (fam_cq(MODEL) && PHASE == 1) {
                // zeta: src/Draw.pl.jl:192-206 (CrossQr) / :179-187 (Cross) (zero prior mean, prior variance Sigp[2,2]); sA = sum of weights here
                const real parV = q_rcp(q_rcp(sig22) + sA);
                const real parM = parV * sC;
                ze = parM + q_sqrt(parV) * row_normal<real>(rw2, rw3, logtab);
            }
            if (rok) {
                if (PHASE == 0) {
                    A.theta[i] = th;
                    if (A.tr_theta) A.tr_theta[(size_t)trow * A.N + i] = th;
                    if (post_burn) A.sum_theta[i] += (double)th;
                }
                if (fam_rt(MODEL) || fam_lq(MODEL) || (fam_cq(MODEL) && PHASE == 1)) {
                    A.zeta[i] = ze;
                    if (A.tr_zeta) A.tr_zeta[(size_t)trow * A.N + i] = ze;
                    if (post_burn) A.sum_zeta[i] += (double)ze;
                }
                // structural log-likelihood terms (src/GibbsRtIrt.pl.jl:201,269; src/GibbsRtIrtLatent.pl.jl:261)
                if (MODEL == MLIRT) {
                    const real e = th - mu0a;
                    ll += -0.5 * LOG_2PI - 0.5 * (double)(e * e);
                } else if (fam_rt(MODEL) || (fam_cq(MODEL) && PHASE == 1)) {
                    const double e0 = (double)(th - mu0a), e1 = (double)(ze - mu0b);
                    ll += sp_c0 - 0.5 * (sp_q00 * e0 * e0 + sp_q01 * e0 * e1 + sp_q11 * e1 * e1);
                } else if (fam_lq(MODEL)) {
                    const double var = (double)sig22 * ((double)k2 * (double)nu_row);
                    const double e = (double)(ze - (xb5 + k1 * nu_row));
                    ll += -0.5 * LOG_2PI - 0.5 * q_log(var) - 0.5 * q_div(e * e, var);
                    if (MODEL == LATENTQR) {
                        if (A.tr_nu) A.tr_nu[(size_t)trow * A.N + i] = nu_row;
                        if (post_burn) A.sum_nu[i] += (double)nu_row;
                    }
                }
            }
        } else if (MODEL == LATENTQR) {
            xb5 = mu0a + th * (real)beta[p];
        }
        if (MODEL == LATENTQR) {
            // nu_{t+1}: src/Draw.pl.jl:325-343 (depends on zeta_t, theta_t, beta_t, Sigp_t only)
            Stream st(A.seed, A.chain, SITE_NU, (uint32_t)i + A.row_base, 0u, sweep + 1u);
            if constexpr (sizeof(real) == 8) nu_next = qr_weight_q(st, fabs(ze - xb5), lq_cB, lq_lam, logtab);
            else {
                const real den = r_sqrt(sig22 * k2);
                nu_next = qr_weight<real>(st, r_div(r_abs(ze - xb5), den), r_div(r_sqrt(real(2) * k2 + k1 * k1), den), logtab);
            }
            if (rok) A.nu[i] = nu_next;
        }

        // ---- per-subject values of the global statistics (sums over subjects of products of two of them), parked in LDS and reduced
        // by the whole workgroup after the barrier below (a 64-lane fp64 butterfly per statistic here cost 6 us of the pass):
        // slot c-1 holds value code c: 1..F -> X columns, F+1 theta, F+2 zeta, F+3 u = zeta - k1 nu_{t+1}, F+4 nu_{t+1}
        if ((NG > 1 || PHASE == 0) && rok) {
            real* o = sh_val + (size_t)(i - row0) * NV;
            for (int u = 1; u <= F; ++u) o[u - 1] = xcol(u);
            o[F] = th; o[F + 1] = ze;
            if constexpr (fam_lq(MODEL)) { o[F + 2] = ze - k1 * nu_next; o[F + 3] = nu_next; }
        }
    }
    stamp(6);
    ERM_DIAG_STOP(A, 2);

    // ---------------- omega_{t+1} | theta_t, a_t, b_t  (src/Draw.pl.jl:36-40), persistent lanes over the wave's flattened cells:
    // lane l owns cells l, l+64, l+128, ... of the slice (cell c = (subject ra + c / J, item c % J), which is also its offset in
    // the row-major omega slice, so stores are fully coalesced).  Each trip makes ONE single-block PG attempt; a lane moves on
    // to its next cell as soon as a draw is accepted, so a wave pays the max over lanes of the TOTAL attempts of ~equal queues.
    // Attempt k of cell (i, j) uses Philox block k of stream (OMEGA, i, j, sweep+1).
    if constexpr (PHASE == 0) {
        if (threadIdx.x == 0) *reinterpret_cast<unsigned int*>(sh_struct + 6) = blockDim.x;
        __syncthreads();                              // theta_t of every subject of the workgroup is parked in sh_val
        const long long qrow0 = row0;                 // first subject of the queue's slice
        const int ncell = nrows_blk * J;
        auto theta_of = [&](int rr_) -> real { return sh_val[(size_t)rr_ * NV + F]; };
        const float invJ = 1.0f / (float)J;
        // cells are handed out dynamically from a workgroup-shared LDS counter: a lane that finishes a cell grabs the next index, so
        // every lane stays busy until the slice is exhausted (which lane draws which cell does not matter: draws are addressed
        // by (i, j, sweep), never by lane)
        unsigned int* qhead = reinterpret_cast<unsigned int*>(sh_struct + 6);
        auto locate = [&](int c, int& rr, int& j) {      // c -> (row within slice, item); exact for c < 2^22
            rr = (int)(((float)c + 0.5f) * invJ);
            j = c - rr * J;
            if (j < 0) { j += J; --rr; } else if (j >= J) { j -= J; ++rr; }
        };
        int c = (int)threadIdx.x, rr, j;
        locate(c, rr, j);
        bool active = c < ncell;
        uint32_t att = 0;
        real th = active ? theta_of(rr) : real(0);
        real z = active ? real(0.5) * r_abs(sh_a[j] * (th - sh_b[j])) : real(0);
        int kb = pg_bin_index(z);                     // the cell's row of the proposal table
        real* om = A.omega + (size_t)qrow0 * J;
        const uint32_t c3 = ((uint32_t)SITE_OMEGA << 24) | ((A.chain & 0xFFu) << 16);
        // PHX: what a cell fixes of its attempts' Philox blocks (philox4x32_10_vk_cell), made when the cell is taken
        [[maybe_unused]] uint32_t ph_hi = 0u, ph_m2 = 0u, ph_m3 = 0u;
        [[maybe_unused]] const uint32_t ph_n1k = (uint32_t)((uint64_t)0xCD9E8D57u * (sweep + 1u)) ^ ((uint32_t)A.seed + 0x9E3779B9u);
        auto cell_words = [&](int rr_, int j_) {
            if constexpr (PHX) {
                const uint64_t p0 = (uint64_t)0xD2511F53u * ((uint32_t)(qrow0 + rr_) + A.row_base);
                const uint2 pj = sh_phx[j_];
                ph_hi = (uint32_t)(p0 >> 32); ph_m2 = pj.x ^ (uint32_t)p0 ^ ((uint32_t)(A.seed >> 32) + 0xBB67AE85u); ph_m3 = pj.y;
            }
        };
        if (active) cell_words(rr, j);
        [[maybe_unused]] unsigned int n_att = 0, n_trip = 0;
#ifndef ERM_PG_VKEYS
#define ERM_PG_VKEYS 20      // (8 until the end of round 4; with two rounds of the block hoisted out of the attempt the loop has the registers for all of them: 67.8 -> 67.6 us, 6 / 12 lose)
#endif
        constexpr int NVK = PERSIST ? 20 : ERM_PG_VKEYS;         // round keys of the attempts' Philox blocks kept in vector registers (philox4x32_10_vk)
        uint32_t pgk[NVK > 0 ? NVK : 1];
        philox_vector_keys<NVK>((uint32_t)A.seed, (uint32_t)(A.seed >> 32), pgk);
        // (letting a wave whose queue ran dry serve other waves' queues was tried: the hardware favours a SIMD's oldest wave, so the
        // four waves of a SIMD finish up to 17 us apart -- but the phase is VALU-throughput-bound, the SIMD is busy until the last
        // one ends either way, and the stealing logic only added instructions: 78.5 vs 75.3 us per sweep)
        // Both engines run the same flat loop.  fp64: the attempt is DECIDED in fp32 behind guard bands and its value -- the fp64 logarithm
        // of u1 and one reciprocal, whichever piece proposed -- is evaluated in the same trip for every lane (pg1_attempt_f64); the rare lane
        // inside a guard band, or with z >= 8, repeats the attempt through the reference form (a real call).  Round 2's proposal by the
        // inverse normal cdf needed a different 30-70-instruction fp64 evaluation per quantile range and therefore per-wave value queues
        // sorted by piece (four ballot rounds per trip, 64 KB of LDS, scattered 8-byte stores); this form needs none of it.
        while (PERSIST ? __all(active) : __any(active)) {    // PERSIST: while every lane holds a cell (a wave with an idle lane has found the queue empty: the loop below)
#ifdef ERM_DIAG_COUNTERS
            ++n_trip; n_att += active ? 1u : 0u;
#endif
            if (active) {
                uint32_t w0, w1, w2, w3;
                if constexpr (PHX) philox4x32_10_vk_cell<NVK>(ph_hi, ph_m2, ph_m3, c3 | att, ph_n1k, (uint32_t)A.seed, (uint32_t)(A.seed >> 32), pgk, w0, w1, w2, w3);
                else philox4x32_10_vk<NVK>((uint32_t)(qrow0 + rr) + A.row_base, (uint32_t)j, sweep + 1u, c3 | att, (uint32_t)A.seed, (uint32_t)(A.seed >> 32), pgk, w0, w1, w2, w3);
                real w;
                bool acc_, unsure;
                if constexpr (sizeof(real) == 8) acc_ = pg1_attempt_f64<true>(z, w0, w1, w2, w3, sh_pgf[kb], sh_pgc[kb], logtab, w, unsure);
                else { acc_ = pg1_attempt(z, w0, w1, w2, w3, sh_pgf[kb], w); unsure = !(z < (real)PG_ZMAX); }
                if (__any(unsure)) {                             // inside a guard band / z >= 8: the reference form decides
                    double o2;
                    const bool a2 = pg1_attempt_ref_call((double)z, w0, w1, w2, w3, A.pgtab, &o2);
                    if (unsure) { acc_ = a2; w = (real)o2; }
                }
                if (acc_ || att + 1u >= (uint32_t)MAX_TRIES) {
                    // write-through (sc1) store: omega_{t+1} is next read by the column phase (behind a barrier) and by the next launch, and nothing of it
                    // stays dirty in L2 for the end-of-kernel write-back (A/B on one box: 75.9-76.2 -> 75.4-75.7 us per sweep)
                    __hip_atomic_store(reinterpret_cast<real*>(reinterpret_cast<char*>(om) + (unsigned int)c * (unsigned int)sizeof(real)), w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    c = (int)atomicAdd(qhead, 1u);
                    att = 0;
                    active = c < ncell;
                    if (active) { locate(c, rr, j); th = theta_of(rr); z = real(0.5) * r_abs(sh_a[j] * (th - sh_b[j])); kb = pg_bin_index(z); cell_words(rr, j); }
                } else ++att;
            }
        }
        // (persistent launches -- small data sets -- only: in the large-data kernel the second copy of the attempt raised the register pressure of the
        // loop above, 21 more moves and SGPR reloads per trip, +7 % instructions for one trip saved in twenty-three)
        if constexpr (PERSIST)
        // The queue is empty (some lane found no cell): the wave's idle lanes now attempt AHEAD for the cells still open.  With n open cells the
        // wave's lanes form n teams of S = 2^floor(log2(64 / n)); member m of a team makes attempt att + m of its cell (attempts are addressed by
        // (cell, attempt), so which lane makes one does not matter) and the cell takes the accepted attempt of lowest index -- the draw the
        // sequential loop would have made, a trip or several earlier.  A small data set has one cell per lane and its PG phase is the longest
        // rejection chain of the workgroup (about four trips); with teams it is two.
        for (;;) {
            const unsigned long long am = __ballot(active);
            if (am == 0ull) break;
            const int nact = __popcll(am);
            int lgS = 0;
            while ((nact << (lgS + 1)) <= 64) ++lgS;
            const int S = 1 << lgS;
            // compact the open cells' lanes: lane k < nact learns the k-th open lane (one forward permute of a full permutation)
            const int rank_a = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)am, 0u));
            const int rank_i = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(~am >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)~am, 0u));
            const int owners = __builtin_amdgcn_ds_permute((active ? rank_a : nact + rank_i) << 2, lane);
            const int team = lane >> lgS, member = lane & (S - 1);
            const int own = __builtin_amdgcn_ds_bpermute(team << 2, owners) << 2;
            const bool work = team < nact;
            const int rr_h = __builtin_amdgcn_ds_bpermute(own, rr), j_h = __builtin_amdgcn_ds_bpermute(own, j), kb_h = __builtin_amdgcn_ds_bpermute(own, kb);
            const uint32_t att_h = (uint32_t)__builtin_amdgcn_ds_bpermute(own, (int)att) + (uint32_t)member;
            real z_h;
            if constexpr (sizeof(real) == 8) {
                const long long zb = __double_as_longlong(z);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(own, (int)(uint32_t)zb), hi = (uint32_t)__builtin_amdgcn_ds_bpermute(own, (int)(uint32_t)((unsigned long long)zb >> 32));
                z_h = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            } else z_h = __int_as_float(__builtin_amdgcn_ds_bpermute(own, __float_as_int(z)));
#ifdef ERM_DIAG_COUNTERS
            ++n_trip; n_att += active ? 1u : 0u;
#endif
            bool acc_h = false;
            real w_h = real(0);
            if (work) {
                uint32_t w0, w1, w2, w3;
                philox4x32_10_vk<NVK>((uint32_t)(qrow0 + rr_h) + A.row_base, (uint32_t)j_h, sweep + 1u, c3 | att_h, (uint32_t)A.seed, (uint32_t)(A.seed >> 32), pgk, w0, w1, w2, w3);
                bool unsure;
                if constexpr (sizeof(real) == 8) acc_h = pg1_attempt_f64<true>(z_h, w0, w1, w2, w3, sh_pgf[kb_h], sh_pgc[kb_h], logtab, w_h, unsure);
                else { acc_h = pg1_attempt(z_h, w0, w1, w2, w3, sh_pgf[kb_h], w_h); unsure = !(z_h < (real)PG_ZMAX); }
                if (__any(unsure)) {
                    double o2;
                    const bool a2 = pg1_attempt_ref_call((double)z_h, w0, w1, w2, w3, A.pgtab, &o2);
                    if (unsure) { acc_h = a2; w_h = (real)o2; }
                }
                acc_h = acc_h || att_h + 1u >= (uint32_t)MAX_TRIES;
            }
            const unsigned long long hm = __ballot(acc_h);
            // the owner reads its team's verdicts: the accepted attempt of lowest index, if any
            const unsigned long long mine = (hm >> ((rank_a << lgS) & 63)) & (S == 64 ? ~0ull : ((1ull << S) - 1ull));
            const int src = (((rank_a << lgS) + (mine ? __builtin_ctzll(mine) : 0)) & 63) << 2;
            real w_o;
            if constexpr (sizeof(real) == 8) {
                const long long wb = __double_as_longlong(w_h);
                const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)wb), hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)((unsigned long long)wb >> 32));
                w_o = __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
            } else w_o = __int_as_float(__builtin_amdgcn_ds_bpermute(src, __float_as_int(w_h)));
            if (active) {
                if (mine) { __hip_atomic_store(om + c, w_o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); active = false; }
                else att += (uint32_t)S;
            }
        }
#ifdef ERM_DIAG_COUNTERS
        if (ERM_DIAG_ON(A, 9)) {
            Ctl* cw = const_cast<Ctl*>(A.ctl);
            atomicAdd(&cw->dbg_attempts, (unsigned long long)n_att);
            if (lane == 0) { atomicAdd(&cw->dbg_trips, (unsigned long long)n_trip); atomicAdd(&cw->dbg_cells, (unsigned long long)ncell); }
        }
#endif
    }
    stamp(7);
    ERM_DIAG_STOP(A, 3);
    __syncthreads();
    stamp(8);

    // ---------------- CrossQr pass B: nu_{t+1} for every cell (src/Draw.pl.jl:303-320) -- and, with nu_t still in hand, the cell's RT log-likelihood
    // term and the nu traces -- over the workgroup's FLATTENED cells, like the PG phase (cell c = subject row0 + c / J, item c % J = its offset in
    // the row-major slice: coalesced).  The column phase below keeps one lane per item for its register accumulators, which leaves 14 of 64 lanes idle
    // at 50 items; this is the expensive part of the pass (an inverse-Gaussian draw: ~90 instructions per cell) and needs no accumulator per item.
    // (fp64 engine only: in fp32 the draw is a handful of hardware transcendentals and the extra pass over C and nu cost more than the idle lanes: 106.7 -> 109.9 us)
    constexpr bool NU_FLAT = MODEL == CROSSQR && PHASE == 1 && sizeof(real) == 8;
    if constexpr (NU_FLAT) {
        const int ncell = nrows_blk * J;
        const float invJ = 1.0f / (float)J;
        const double qr_cB = sqrt((double)(real(2) * k2 + k1 * k1));
        double llf = 0.0;
        for (int c = (int)threadIdx.x; c < ncell; c += (int)blockDim.x) {
            int rr = (int)(((float)c + 0.5f) * invJ);
            int j = c - rr * J;
            if (j < 0) { j += J; --rr; } else if (j >= J) { j -= J; ++rr; }
            const long long i = row0 + rr;
            const size_t e = (size_t)i * J + j;
            const real th = sh_val[(size_t)rr * NV + F], ze = sh_val[(size_t)rr * NV + F + 1];
            const real cc = blk_C[c], nu = blk_nu[c];
            const real lamc = sh_lamc[j], isig = sh_isig[j], rho = sh_rho[j];
            if (A.mode == 1) {
                const real var_ = k2 * nu;                                 // times sig2t_j
                const real er = cc - lamc + ze + th * rho - k1 * nu;     // logT - mu_t
                real lv, qv;
                if constexpr (sizeof(real) == 8) { lv = fm::log(var_, logtab); qv = fm::div(er * er * isig, var_); }      // var_ = k2 nu in [1e-10 k2, 1e10 k2]
                else { lv = r_log(var_); qv = r_div(er * er * isig, var_); }
                llf += (double)(real(-0.5) * ((real)LOG_2PI + sh_lsig[j] + lv + qv));
                if (post_burn && A.sum_nu) A.sum_nu[e] += (double)nu;
                if (A.tr_nu) A.tr_nu[(size_t)trow * (size_t)A.N * J + e] = nu;     // Post.qr's vec(nu_t) (src/GibbsRtIrtCross.pl.jl:296)
            }
            Stream st(A.seed, A.chain, SITE_NU, (uint32_t)i + A.row_base, (uint32_t)j, sweep + 1u);
            real nun;
            if constexpr (sizeof(real) == 8) nun = qr_weight_q(st, fabs(cc - lamc + ze + th * rho), qr_cB, (double)(real(2) * k2 + k1 * k1) * (double)isig / (double)k2, logtab);
            else {
                const real den = r_div(r_sqrt(k2), r_sqrt(isig));    // sqrt(sig2t k2)
                nun = qr_weight<real>(st, r_div(r_abs(cc - lamc + ze + th * rho), den), r_div(r_sqrt(real(2) * k2 + k1 * k1), den), logtab);
            }
            __hip_atomic_store(A.nu + e, nun, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // write-through: read back by the column phase behind the barrier
        }
        ll += llf;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- global statistics for the next tiny step: statistic g = sum over the workgroup's subjects of va * vb (* 1/nu for the
    // sigp_mode-1 block); one wave per statistic (statistic g on wave g mod nWaves, so the work is spread over the whole workgroup),
    // lane l sums subjects l, l+64, ... in order, then a 64-lane butterfly: fixed order
    if (NG > 1) {
        const int cT = F + 1, cZ = F + 2, cU = F + 3, cN = F + 4;
        const int tg = wave, tl = lane, ngrp = nWaves;
        for (int g = tg; g < NG - 1; g += ngrp) {
            int ca = 0, cb2 = 0; bool weighted = false;
            if (MODEL == MLIRT) { ca = g; cb2 = cT; }
            else if (fam_rt(MODEL)) {
                if (g < p) { ca = g; cb2 = cT; }
                else if (g < 2 * p) { ca = g - p; cb2 = cZ; }
                else if (g == 2 * p) { ca = cT; cb2 = cT; }
                else if (g == 2 * p + 1) { ca = cT; cb2 = cZ; }
                else { ca = cZ; cb2 = cZ; }
            } else if (fam_lq(MODEL)) {
                if (g < p) { ca = g; cb2 = cT; }
                else if (g == p) { ca = cT; cb2 = cT; }
                else if (g < 2 * p + 1) { ca = g - p - 1; cb2 = cU; }
                else if (g == 2 * p + 1) { ca = cT; cb2 = cU; }
                else if (g == 2 * p + 2) { ca = cU; cb2 = cU; }
                else if (g == 2 * p + 3) { ca = 0; cb2 = cN; }
                else if (g == 2 * p + 4) { ca = cN; cb2 = cN; }
                else if (g == 2 * p + 5) { ca = 0; cb2 = cZ; }
                else if (g == 2 * p + 6) { ca = cZ; cb2 = cZ; }
                else {
                    // sigp_mode 1: entries of x~' W x~ (upper triangle, row-major), x~' W u, u' W u with x~ = [1 X theta], W = diag(1/nu_{t+1})
                    const int q = p + 1, ntri = q * (q + 1) / 2;
                    int e = g - (2 * p + 7);
                    auto col = [&](int u) { return u < p ? u : cT; };
                    if (e < ntri) { int u = 0; while (e >= q - u) { e -= q - u; ++u; } ca = col(u); cb2 = col(u + e); }
                    else if (e < ntri + q) { ca = col(e - ntri); cb2 = cU; }
                    else { ca = cU; cb2 = cU; }
                    weighted = true;
                }
            } else { ca = cZ; cb2 = cZ; }      // CrossQr pass B: sum zeta^2
            double accg = 0.0;
            for (int li = tl; li < nrows_blk; li += 64) {
                const real* o = sh_val + (size_t)li * NV;
                const double va = ca == 0 ? 1.0 : (double)o[ca - 1], vb = cb2 == 0 ? 1.0 : (double)o[cb2 - 1];
                accg += weighted ? va * vb / (double)o[cN - 1] : va * vb;
            }
            accg = bfly_sum(accg, 1, 64);
            if (tl == 0) sh_gacc[g] = accg;          // wave 0's slot of the per-wave table summed by the epilogue (the other waves' stay 0)
            if constexpr (rtll_stats<MODEL, PHASE>()) {
                // ERM_RTLL_STATS: this workgroup's share of the response-time log-likelihood that is linear in sum zeta and sum zeta^2
                const int g_sz = fam_rt(MODEL) ? p : 2 * p + 5, g_zz = fam_rt(MODEL) ? 2 * p + 2 : 2 * p + 6;
                if (A.mode == 1 && g == g_sz) {
                    double sl = 0.0;
                    for (int jj = tl; jj < J; jj += 64) sl = fma((double)sh_lamc[jj], (double)sh_isig[jj], sl);
                    sl = bfly_sum(sl, 1, 64);
                    if (tl == 0) ll = fma(accg, sl, ll);
                }
                if (A.mode == 1 && g == g_zz && tl == 0) ll = fma(-0.5 * accg, sh_struct[4], ll);
            }
        }
    }

    stamp(9);
    // =================================================================================================== phase 2
    // lane = item j; batches of four consecutive subjects are dealt to the waves round-robin (batch b on wave b mod nWaves: a fixed assignment, so
    // the per-wave partial sums -- and the chain -- are reproducible bit for bit; a wave's own contiguous slice of ~24 subjects filled 24 of the 28
    // slots of its seven batches), each wave from the LAST batch to the first, so that the phase ends on the rows the
    // next sweep's row sums read first (A/B on one box: 109.7 -> 108.9 us per sweep; FETCH_SIZE is unchanged -- the L2s of a multi-XCD part
    // are written back and invalidated between launches -- so the gain is the memory side's); accumulators live in fp64 registers
    bool p2_done = false;
    const int nbatch = (nrows_blk + 3) >> 2;
    if constexpr (sizeof(real) == 8 && PHASE == 0 && !fam_cq(MODEL)) {
        // fp64 engine, even test lengths: a lane takes the item PAIR (2l, 2l+1), the two half-waves take two subjects at a time, so that omega
        // and logT come in 16-byte loads (the 8-byte loads of one item per lane reach 0.5-0.7 of that rate); lanes l and l + 32 then hold the
        // sums of different subjects for the same two items and are added at the end
        if ((J & 1) == 0 && !ERM_DIAG_ON(A, 7)) {
            p2_done = true;
            const int half = lane >> 5, l32 = lane & 31;
            for (int cb = 0; cb * 64 < J; ++cb) {
                const int j0 = cb * 64 + 2 * l32;
                const bool jv = j0 < J;
                const int jc = jv ? j0 : 0;
                const double a0 = sh_a[jc], a1 = sh_a[jc + 1], b0 = sh_b[jc], b1 = sh_b[jc + 1];
                constexpr bool RTLL = rtll_stats<MODEL, PHASE>();      // the response-time log-likelihood comes from the statistics: no residual per cell
                [[maybe_unused]] const double lamc0 = RTLL ? 0.0 : (double)sh_lamc[jc], lamc1 = RTLL ? 0.0 : (double)sh_lamc[jc + 1];
                [[maybe_unused]] const double isig0 = sh_isig[jc], isig1 = sh_isig[jc + 1];
                [[maybe_unused]] const double lsig0 = RTLL ? 0.0 : (double)sh_lsig[jc], lsig1 = RTLL ? 0.0 : (double)sh_lsig[jc + 1];
                double S0[NSTAT], S1[NSTAT];
#pragma unroll
                for (int q = 0; q < NSTAT; ++q) { S0[q] = 0.0; S1[q] = 0.0; }
                // The phase is VALU-issue-bound (tools/stage_budget.py: 7.5 M wave-instructions, 14 of its 16 us), so the per-cell work is pared down:
                // statistics go straight into the fp64 accumulators by fma (theta^2 and theta/2 once per subject); the cell log-likelihood
                //   y eta - log(1 + e^eta) - (log 2 pi + log sig2t_j + er^2 / sig2t_j) / 2
                // is kept as three per-lane partial sums -- -max(s, 0) with s = eta or -eta by y, the PRODUCT of the factors 1 + e^{-|eta|} (each in
                // (1, 2]: one logarithm per lane at the end, or every 512 factors), and sum er^2 / sig2t_j -- and the per-item constants enter
                // once, times the number of cells.
                double lmax = 0.0, bprod = 1.0, rtq = 0.0, asum = 0.0;
                int ncells = 0, nfac = 0, ny0 = 0, ny1 = 0;
                for (int bt = nbatch - 1 - ((nbatch - 1 - wave) % nWaves + nWaves) % nWaves; bt >= 0; bt -= nWaves) {      // a batch: 2 slots x 2 half-waves = 4 subjects, 4 cells per lane
                    double thv[2], zev[2]; double2 wv[2], cv[2]; unsigned int yv[2]; bool okv[2];
                    const int q0 = 4 * bt + 3;                         // the batch's last subject (position in the workgroup); rows beyond the workgroup's are masked
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int q = q0 - (2 * u + half);
                        okv[u] = jv && q < nrows_blk;
                        const int qc = q < nrows_blk ? q : 0;
                        // (24-bit multiplications: full rate, and a position in the workgroup, nItem and NV are all far below 2^24; the 32-bit multiply-add
                        // the compiler otherwise takes is a v_mad_u64_u32, three times the issue cost)
                        const unsigned int e = __umul24((unsigned int)qc, (unsigned int)J) + (unsigned int)jc;      // < 2^22: 32-bit offsets from the workgroup's first cell
                        // theta_t, zeta_t of the subject: parked in LDS by the subject draws (a broadcast read per half-wave instead of two global loads);
                        // omega and logT at 32-bit BYTE offsets from the workgroup's (scalar) base -- one shift instead of three 64-bit vector operations per load
                        const real* sv = sh_val + __umul24((unsigned int)qc, (unsigned int)NV) + F;
                        thv[u] = sv[0];
                        zev[u] = (MODEL != MLIRT) ? sv[1] : 0.0;
                        const unsigned int e8 = e << 3;
                        wv[u] = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(blk_omega) + e8);
                        yv[u] = *reinterpret_cast<const unsigned short*>(blk_Y + e);
                        if constexpr (MODEL != MLIRT) cv[u] = *reinterpret_cast<const double2*>(reinterpret_cast<const char*>(blk_C) + e8); else { cv[u].x = 0.0; cv[u].y = 0.0; }
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (!okv[u]) continue;
                        const double th = thv[u], ze = zev[u], th2 = th * th, hth = 0.5 * th;
                        auto cell = [&](double w, bool y, double c, double a, double b, double lamc, double isig, double* S, int& ny) {
                            S[0] += w; S[1] = fma(w, th, S[1]); S[2] = fma(w, th2, S[2]); S[3] += y ? hth : -hth;
                            if constexpr (fam_rt(MODEL) || fam_lq(MODEL)) S[4] = fma(c, ze, S[4]);
                            {   // (also in the statistics-only pass that opens a chain, A.mode == 0, which discards it: a test of the mode here is a branch per cell and
                                // a basic-block boundary between the four cells' independent polynomial chains)
                                // y eta - max(eta, 0) = -max(s, 0) with s = (1 - 2y) eta = -2 kappa eta, and max(s, 0) = (|eta| + s) / 2: the sum over a lane's cells is
                                // sum |eta| / 2 - sum kappa eta, and sum kappa eta = a_j (sum kappa theta - b_j sum kappa) comes from S[3] and the count of ones --
                                // one add (|eta| is an operand modifier) and a count per cell instead of a sign flip, a select, a maximum and an add
                                const double eta = a * (th - b);
                                asum += fabs(eta);
                                ny += y ? 1 : 0;
                                bprod *= 1.0 + fm::exp_neg_ll(fabs(eta));
                                if constexpr (!RTLL && (fam_rt(MODEL) || fam_lq(MODEL))) {
                                    const double er = c + ze - lamc;
                                    rtq = fma(er * er, isig, rtq);
                                }
                            }
                        };
                        cell(wv[u].x, (yv[u] & 0xFFu) != 0u, cv[u].x, a0, b0, lamc0, isig0, S0, ny0);
                        cell(wv[u].y, (yv[u] >> 8) != 0u, cv[u].y, a1, b1, lamc1, isig1, S1, ny1);
                        ++ncells;
                    }
                    nfac += 4;
                    if (A.mode == 1 && nfac >= 512) { lmax += fm::log(bprod, logtab); bprod = 1.0; nfac = 0; }     // (wave-uniform) keeps the product below 2^1023
                }
                double llc = 0.0;
                if (A.mode == 1) {
                    const double hn = 0.5 * (double)ncells;
                    const double sketa = fma(a0, fma(-b0, (double)ny0 - hn, S0[3]), a1 * fma(-b1, (double)ny1 - hn, S1[3]));      // sum kappa eta of this lane's cells (S[3]: before the half-waves are added)
                    llc = -(fma(0.5, asum, -sketa) + lmax + fm::log(bprod, logtab));
                    if constexpr (!RTLL && (fam_rt(MODEL) || fam_lq(MODEL))) llc -= 0.5 * (rtq + (double)ncells * (2.0 * LOG_2PI + lsig0 + lsig1));
                }
#pragma unroll
                for (int q = 0; q < NSTAT; ++q) { S0[q] += __shfl_xor(S0[q], 32, 64); S1[q] += __shfl_xor(S1[q], 32, 64); }
                if (jv && half == 0) {
#pragma unroll
                    for (int q = 0; q < NSTAT; ++q) { acc[q * J + j0] = S0[q]; acc[q * J + j0 + 1] = S1[q]; }
                    if constexpr (RTLL) { if (A.mode == 1) llc = fma(-isig1, S1[4], fma(-isig0, S0[4], llc)); }      // - sum_j G_j / sig2t_j of this wave's subjects
                }
                ll += llc;
            }
        }
    }
    if constexpr (sizeof(real) == 4 && PHASE == 0 && !fam_cq(MODEL)) {
        // fp32 fast mode, even test lengths: the same item-pair / half-wave mapping (8-byte loads of omega and logT, one address computation and one
        // theta / zeta load per TWO cells): the phase is VALU-issue-bound, and the one-item-per-lane path below spends as much on addresses, masks
        // and conversions as on the cells (83 lane-instructions per cell-update against 45 here).  Cell arithmetic in fp32, the four cells of a lane's
        // batch summed in fp32 and added to the fp64 accumulators once per batch, as everywhere in this mode.
        if ((J & 1) == 0 && !ERM_DIAG_ON(A, 7)) {
            p2_done = true;
            const int half = lane >> 5, l32 = lane & 31;
            for (int cb = 0; cb * 64 < J; ++cb) {
                const int j0 = cb * 64 + 2 * l32;
                const bool jv = j0 < J;
                const int jc = jv ? j0 : 0;
                const float a0 = sh_a[jc], a1 = sh_a[jc + 1], b0 = sh_b[jc], b1 = sh_b[jc + 1];
                constexpr bool RTLL = rtll_stats<MODEL, PHASE>();
                [[maybe_unused]] const float lamc0 = RTLL ? 0.0f : (float)sh_lamc[jc], lamc1 = RTLL ? 0.0f : (float)sh_lamc[jc + 1];
                [[maybe_unused]] const float isig0 = sh_isig[jc], isig1 = sh_isig[jc + 1];
                [[maybe_unused]] const float lconst = RTLL ? 0.0f : 2.0f * (float)LOG_2PI + (float)sh_lsig[jc] + (float)sh_lsig[jc + 1];
                double S0[NSTAT], S1[NSTAT];
#pragma unroll
                for (int q = 0; q < NSTAT; ++q) { S0[q] = 0.0; S1[q] = 0.0; }
                double llc = 0.0;
                for (int bt = nbatch - 1 - ((nbatch - 1 - wave) % nWaves + nWaves) % nWaves; bt >= 0; bt -= nWaves) {
                    float thv[2], zev[2]; float2 wv[2], cv[2]; unsigned int yv[2]; bool okv[2];
                    const int q0 = 4 * bt + 3;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int q = q0 - (2 * u + half);
                        okv[u] = jv && q < nrows_blk;
                        const int qc = q < nrows_blk ? q : 0;
                        const unsigned int e = __umul24((unsigned int)qc, (unsigned int)J) + (unsigned int)jc;
                        const real* sv = sh_val + __umul24((unsigned int)qc, (unsigned int)NV) + F;      // (as in the fp64 loop above)
                        thv[u] = sv[0];
                        zev[u] = (MODEL != MLIRT) ? sv[1] : 0.0f;
                        const unsigned int e4 = e << 2;
                        wv[u] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(blk_omega) + e4);
                        yv[u] = *reinterpret_cast<const unsigned short*>(blk_Y + e);
                        if constexpr (MODEL != MLIRT) cv[u] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(blk_C) + e4); else { cv[u].x = 0.0f; cv[u].y = 0.0f; }
                    }
                    float bs0[NSTAT], bs1[NSTAT], bl = 0.0f;
#pragma unroll
                    for (int q = 0; q < NSTAT; ++q) { bs0[q] = 0.0f; bs1[q] = 0.0f; }
                    int nrow = 0;
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        if (!okv[u]) continue;
                        const float th = thv[u], ze = zev[u], th2 = th * th, hth = 0.5f * th;
                        auto cell = [&](float w, bool y, float c, float a, float b, float lamc, float isig, float* bs) {
                            bs[0] += w; bs[1] = fmaf(w, th, bs[1]); bs[2] = fmaf(w, th2, bs[2]); bs[3] += y ? hth : -hth;
                            if constexpr (fam_rt(MODEL) || fam_lq(MODEL)) bs[4] = fmaf(c, ze, bs[4]);
                            {   // (in every mode, as in the fp64 loop: no branch per cell)
                                const float eta = a * (th - b);
                                float t = (y ? eta : 0.0f) - log1pexp_r(eta);
                                if constexpr (!RTLL && (fam_rt(MODEL) || fam_lq(MODEL))) {
                                    const float er = c + ze - lamc;
                                    t = fmaf(-0.5f * er * er, isig, t);
                                }
                                bl += t;
                            }
                        };
                        cell(wv[u].x, (yv[u] & 0xFFu) != 0u, cv[u].x, a0, b0, lamc0, isig0, bs0);
                        cell(wv[u].y, (yv[u] >> 8) != 0u, cv[u].y, a1, b1, lamc1, isig1, bs1);
                        ++nrow;
                    }
#pragma unroll
                    for (int q = 0; q < NSTAT; ++q) { S0[q] += (double)bs0[q]; S1[q] += (double)bs1[q]; }
                    if constexpr (!RTLL) { if (A.mode == 1 && (fam_rt(MODEL) || fam_lq(MODEL))) bl = fmaf(-0.5f * (float)nrow, lconst, bl); }
                    llc += (double)bl;
                }
#pragma unroll
                for (int q = 0; q < NSTAT; ++q) { S0[q] += __shfl_xor(S0[q], 32, 64); S1[q] += __shfl_xor(S1[q], 32, 64); }
                if (jv && half == 0) {
#pragma unroll
                    for (int q = 0; q < NSTAT; ++q) { acc[q * J + j0] = S0[q]; acc[q * J + j0 + 1] = S1[q]; }
                    if constexpr (RTLL) { if (A.mode == 1) llc = fma(-(double)isig1, S1[4], fma(-(double)isig0, S0[4], llc)); }
                }
                if (A.mode == 1) ll += llc;
            }
        }
    }
    for (int cb = 0; cb * 64 < J && !ERM_DIAG_ON(A, 7) && !p2_done; ++cb) {
        const int j = cb * 64 + lane;
        const bool jv = j < J;
        const real a = jv ? sh_a[j] : real(0), b = jv ? sh_b[j] : real(0);
        const real lamc = jv ? sh_lamc[j] : real(0), isig = jv ? sh_isig[j] : real(1), lsig = jv ? sh_lsig[j] : real(0);
        const real rho = jv ? sh_rho[j] : real(0);
        // quantile weights (CrossQr pass B, fp64): the scale-free constants of the inverse-Gaussian draw, once per item
        [[maybe_unused]] const double qr_cB = sqrt((double)(real(2) * k2 + k1 * k1));
        [[maybe_unused]] const double qr_lam = (double)(real(2) * k2 + k1 * k1) * (double)isig / (double)k2;       // parB^2 = (2 k2 + k1^2) / (sig2t k2)
        double S[NSTAT];
#pragma unroll
        for (int q = 0; q < NSTAT; ++q) S[q] = 0.0;
        double llc = 0.0;
        const int jc = jv ? j : 0;
        for (int bt = nbatch - 1 - ((nbatch - 1 - wave) % nWaves + nWaves) % nWaves; bt >= 0; bt -= nWaves) {
            real thv[4], zev[4], wv[4], cv[4], nv[4]; bool yv[4], okv[4]; long long iv[4];
            const int q0 = 4 * bt + 3;
#pragma unroll
            for (int u = 0; u < 4; ++u) {                 // every load of the batch is issued before any use
                const int q = q0 - u;
                okv[u] = jv && q < nrows_blk;
                const int qc = q < nrows_blk ? q : 0;
                iv[u] = row0 + qc;
                const unsigned int e = (unsigned int)qc * (unsigned int)J + (unsigned int)jc;      // 32-bit offset from the workgroup's first cell
                thv[u] = blk_theta[qc];
                zev[u] = (MODEL != MLIRT) ? blk_zeta[qc] : real(0);
                wv[u] = (PHASE == 0) ? blk_omega[e] : real(0);
                yv[u] = (PHASE == 0) ? (blk_Y[e] != 0) : false;
                cv[u] = (MODEL != MLIRT) ? blk_C[e] : real(0);
                nv[u] = (MODEL == CROSSQR) ? blk_nu[e] : real(1);
            }
            // the 4 cells of a batch are summed in `real` and enter the fp64 accumulators once per batch (fp64 VALU work is what bounds
            // this phase; a 4-term fp32 sum costs ~1 ulp of its terms' own rounding)
            real bs[NSTAT]; real bl = real(0);
            // fp64: sum_u log(1 + e^{-|eta_u|}) of a batch = log prod_u (1 + e^{-|eta_u|}) -- one logarithm per four cells (each factor is in (1, 2])
            [[maybe_unused]] double bprod = 1.0;
            [[maybe_unused]] double vprod = 1.0;         // pass B, fp64: product of the batch's k2 nu (the RT log-likelihood's log-variance terms)
#pragma unroll
            for (int q = 0; q < NSTAT; ++q) bs[q] = real(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (!okv[u]) continue;
                const long long i = iv[u];
                const size_t e = (size_t)i * J + j;
                const real th = thv[u], ze = zev[u];
                if constexpr (PHASE == 0) {
                    const real w = wv[u];
                    const bool y = yv[u];
                    const real c = cv[u];
                    const real wt = w * th;
                    bs[0] += w; bs[1] += wt; bs[2] += wt * th; bs[3] += y ? real(0.5) * th : real(-0.5) * th;
                    if constexpr (fam_rt(MODEL) || fam_lq(MODEL)) bs[4] += c * ze;
                    if (A.mode == 1) {
                        const real eta = a * (th - b);
                        real t;
                        if constexpr (sizeof(real) == 8) {
                            t = (y ? eta : 0.0) - (eta > 0.0 ? eta : 0.0);
                            bprod *= 1.0 + fm::exp_neg_ll(fabs(eta));
                        } else t = (y ? eta : real(0)) - log1pexp_r(eta);
                        if constexpr (!rtll_stats<MODEL, PHASE>() && (fam_rt(MODEL) || fam_lq(MODEL))) {
                            const real er = c + ze - lamc;
                            t += real(-0.5) * ((real)LOG_2PI + lsig + er * er * isig);
                        }
                        bl += t;
                    }
                    if constexpr (fam_cq(MODEL)) {
                        // statistics for lambda_t, sig2t_t (src/Draw.pl.jl:246-247, 285) with nu_t, zeta_{t-1}, theta_t, rho_t
                        const real nu = nv[u];
                        const real rr = c + ze + th * rho - k1 * nu;
                        const real inu = r_rcp(nu), ri = rr * inu;
                        bs[4] += inu; bs[5] += ri; bs[6] += rr * ri; bs[7] += nu;
                    }
                } else if constexpr (NU_FLAT) {
                    // CrossQr pass B: rho statistics (src/Draw.pl.jl:484-485) with the nu_{t+1} the flat phase above has just drawn (nv: read behind its barrier)
                    const real c = cv[u];
                    const real nun = nv[u];
                    real ti;
                    if constexpr (sizeof(real) == 8) ti = th * fm::rcp(nun); else ti = th * r_rcp(nun);      // nu_{t+1} is clamped to [1e-10, 1e10]
                    bs[0] += th * ti;
                    bs[1] += (lamc - ze - c + k1 * nun) * ti;
                } else {
                    // Cross pass B (nu == 1): RT log-likelihood and rho statistics (src/Draw.pl.jl:484-485)
                    const real c = cv[u];
                    const real nu = nv[u];
                    if (A.mode == 1) {
                        const real var_ = k2 * nu;                               // times sig2t_j
                        const real er = c - lamc + ze + th * rho - k1 * nu;    // logT - mu_t
                        real lv;
                        if constexpr (sizeof(real) == 8) { lv = real(0); vprod *= var_; }       // fp64: ONE logarithm per batch of four cells, of the product (each factor in [1e-10 k2, 1e10 k2])
                        else lv = r_log(var_);
                        real qv;
                        if constexpr (sizeof(real) == 8) qv = fm::div(er * er * isig, var_); else qv = r_div(er * er * isig, var_);       // var_ is a normal, finite number
                        bl += real(-0.5) * ((real)LOG_2PI + lsig + lv + qv);
                        if (has_nu(MODEL) && post_burn && A.sum_nu) A.sum_nu[e] += (double)nu;
                        if (has_nu(MODEL) && A.tr_nu) A.tr_nu[(size_t)trow * (size_t)A.N * J + e] = nu;     // Post.qr's vec(nu_t) (src/GibbsRtIrtCross.pl.jl:296)
                    }
                    real nun = real(1);
                    if constexpr (has_nu(MODEL)) {
                        Stream st(A.seed, A.chain, SITE_NU, (uint32_t)i + A.row_base, (uint32_t)j, sweep + 1u);
                        if constexpr (sizeof(real) == 8) nun = qr_weight_q(st, fabs(c - lamc + ze + th * rho), qr_cB, qr_lam, logtab);      // parB / parA = cB / |residual|, lambda = parB^2 (per item, hoisted)
                        else {
                            const real den = r_div(r_sqrt(k2), r_sqrt(isig));    // sqrt(sig2t k2)
                            nun = qr_weight<real>(st, r_div(r_abs(c - lamc + ze + th * rho), den), r_div(r_sqrt(real(2) * k2 + k1 * k1), den), logtab);
                        }
                        A.nu[e] = nun;
                    }
                    real ti;
                    if constexpr (sizeof(real) == 8) ti = th * fm::rcp(nun); else ti = th * r_rcp(nun);      // nu_{t+1} is clamped to [1e-10, 1e10]
                    bs[0] += th * ti;
                    bs[1] += (lamc - ze - c + k1 * nun) * ti;
                }
            }
#pragma unroll
            for (int q = 0; q < NSTAT; ++q) S[q] += (double)bs[q];
            if constexpr (sizeof(real) == 8 && PHASE == 0) { if (A.mode == 1) bl -= fm::log(bprod, logtab); }
            if constexpr (sizeof(real) == 8 && PHASE == 1) { if (A.mode == 1) bl -= 0.5 * fm::log(vprod, logtab); }
            llc += (double)bl;
        }
        if (jv) {
#pragma unroll
            for (int q = 0; q < NSTAT; ++q) acc[q * J + j] = S[q];
            if constexpr (rtll_stats<MODEL, PHASE>()) { if (A.mode == 1) llc = fma(-(double)isig, S[4], llc); }
        }
        ll += llc;
    }

    stamp(10);
    ERM_DIAG_STOP(A, 4);

    // ---------------- block epilogue: fixed-order reduction of the wave accumulators into this block's slab row
    ll = bfly_sum(ll, 1, 64);
    if (lane == 0) sh_gacc[(size_t)wave * NG + NG - 1] = ll;
    __syncthreads();
    const int NS = NSTAT * J + NG;
    double* out = A.slab + (size_t)blockIdx.x * NS;
    // (tests, ERM_FLAG_TEST_PERSIST_TIMEOUT: workgroup 1 loses its first statistics row, so that the launch times out and erm_run falls back)
    [[maybe_unused]] bool lose_row = false;
    if constexpr (PERSIST) lose_row = ks == 0u && blockIdx.x == 1u && A.tmo[2] != 0u;
    for (int e = threadIdx.x; e < NS; e += blockDim.x) {
        double t = 0.0;
        if (e < NSTAT * J) { for (int w = 0; w < nWaves; ++w) t += sh_acc[(size_t)w * NSTAT * J + e]; }
        else { const int gi = e - NSTAT * J; for (int w = 0; w < nWaves; ++w) t += sh_gacc[(size_t)w * NG + gi]; }
        if (PERSIST && ks + 1u < n_loop) { if (!lose_row) persist_put(A.xbuf + ((size_t)(ks & 1u) * gridDim.x + blockIdx.x) * 2 * NS, e, t, A.tag0 + ks + 1u); }
        else __hip_atomic_store(out + e, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);    // write-through (sc1) store: no release fence needed
    }
    if (PERSIST && ks + 1u < n_loop) {
        // not the launch's last sweep: the next head polls the packets; no ticket, no group rows.  The barrier keeps the head's LDS writes behind
        // this epilogue's LDS reads.
        __syncthreads();
        stamp(13);
        continue;
    }
    // ---- hierarchical reduction: the LAST workgroup of each group of GROUP consecutive ones to arrive sums the group's slab rows
    // in workgroup order (fixed order => deterministic), so the tiny step reads ceil(grid/GROUP) rows instead of grid rows.
    // Hand-off per cdna_hip_programming.md Guideline 16 (form R1): the slab row is stored write-through (sc1), every storing
    // wave drains its stores, workgroup barrier, one lane takes a ticket with an agent-scope atomic; the last arriver acquires
    // at agent scope (L1 invalidate) before any of its waves loads.  No L2 write-back (release fence) is needed, which matters
    // because each workgroup has just dirtied its whole omega slice.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    stamp(11);
    int* sh_flag = reinterpret_cast<int*>(sh_struct);       // sh_struct is dead by now
    const int grp = blockIdx.x / GROUP;
    const int gfirst = grp * GROUP;
    const int gcount = ((int)gridDim.x - gfirst < GROUP) ? (int)gridDim.x - gfirst : GROUP;
    if (threadIdx.x == 0) {
        const unsigned int ticket = __hip_atomic_fetch_add(A.gcnt + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (ticket % (unsigned int)gcount == (unsigned int)(gcount - 1)) ? 1 : 0;   // counters only ever grow (zeroed by the host per erm_run)
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        sh_flag[0] = last;
    }
    __syncthreads();
    stamp(12);
    if (sh_flag[0]) {
        double* gout = k_gslab_out + (size_t)grp * NS;
        for (int e = threadIdx.x; e < NS; e += blockDim.x) {
            double v[GROUP];
#pragma unroll
            for (int b = 0; b < GROUP; ++b) v[b] = A.slab[(size_t)(gfirst + (b < gcount ? b : 0)) * NS + e];   // all loads in flight
            double t = 0.0;
#pragma unroll
            for (int b = 0; b < GROUP; ++b) t += (b < gcount) ? v[b] : 0.0;
            gout[e] = t;
        }
    }
    stamp(13);
    }   // sweeps of a persistent launch
}


template <int MODEL, int STEP>
__global__ void __launch_bounds__(TINY_THREADS) tiny_kernel(TinyArgs T)
{
    const int J = T.J, p = T.nFeat + 1;
    constexpr int NSTAT0 = Stats<MODEL, 0>::NSTAT;
    const int NG0 = Stats<MODEL, 0>::ng(p) + T.ngx;
    const int NS0 = NSTAT0 * J + NG0;
    constexpr int NSTAT1 = fam_cq(MODEL) ? Stats<CROSSQR, 1>::NSTAT : 0;
    const int NG1 = fam_cq(MODEL) ? 2 : 0;
    const int NS1 = NSTAT1 * J + NG1;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* st0 = reinterpret_cast<double*>(smem);          // NS0 reduced statistics of slab0
    double* st1 = st0 + NS0;                                // NS1 reduced statistics of slab1
    double* part = st1 + NS1;                               // J doubles (1/sig2t_j)
    double* work = part + J;                                // TINY_WORK scratch
    double* sh_x = work + TINY_WORK;                        // x'x and its inverse (2 * PMAX * PMAX), staged once
    double* lp = sh_x + 2 * PMAX * PMAX;                    // the parameter block, updated in place and written back at the end
    const int tid = threadIdx.x;
    // every input's first elements are requested before anything waits (as separate load -> wait -> store loops the kernel made six dependent trips to
    // L2 / HBM, half of its 7 us): two elements per thread of x'x | its inverse, of the parameter block and of the statistics of pass A, one of pass
    // B's; GROUP group rows each (the host allocates at least GROUP rows; those beyond nb are masked in the sum); reduce_rows' order
    constexpr int TT = TINY_THREADS;
    constexpr bool HAS1 = fam_cq(MODEL) && STEP == 0;
    const int nx = 2 * PMAX * PMAX, npar = par_size(J);
    const double* cx = T.cst + cst_off_xtx(J);
    const double x0 = cx[tid < nx ? tid : 0], x1 = cx[tid + TT < nx ? tid + TT : 0];
    const double p0 = T.par[tid < npar ? tid : 0], p1 = T.par[tid + TT < npar ? tid + TT : 0];
    double ra[GROUP], rb[GROUP], rc[GROUP];
    const int w0 = tid & ~63;                                // (whole waves skip what none of their lanes needs: sixteen waves issuing 50 loads each is the kernel's time)
#pragma unroll
    for (int u = 0; u < GROUP; ++u) { ra[u] = 0.0; rb[u] = 0.0; rc[u] = 0.0; }
    if (w0 < NS0) {
        const double* qa = T.slab0 + (tid < NS0 ? tid : 0);
#pragma unroll
        for (int u = 0; u < GROUP; ++u) ra[u] = qa[(size_t)u * NS0];
    }
    if (w0 + TT < NS0) {
        const double* qb = T.slab0 + (tid + TT < NS0 ? tid + TT : 0);
#pragma unroll
        for (int u = 0; u < GROUP; ++u) rb[u] = qb[(size_t)u * NS0];
    }
    if (HAS1 && w0 < NS1) {
        const double* qc = T.slab1 + (tid < NS1 ? tid : 0);
#pragma unroll
        for (int u = 0; u < GROUP; ++u) rc[u] = qc[(size_t)u * NS1];
    }
    if (tid < nx) sh_x[tid] = x0;
    if (tid + TT < nx) sh_x[tid + TT] = x1;
    for (int e = tid + 2 * TT; e < nx; e += TT) sh_x[e] = cx[e];
    if (tid < npar) lp[tid] = p0;
    if (tid + TT < npar) lp[tid + TT] = p1;
    for (int e = tid + 2 * TT; e < npar; e += TT) lp[e] = T.par[e];
    auto finish = [&](const double (&r)[GROUP], const double* slab, int nb, int NS, int e, double* out) {
        if (e >= NS) return;
        double t = 0.0;
#pragma unroll
        for (int u = 0; u < GROUP; ++u) t += (u < nb) ? r[u] : 0.0;
        for (int b0 = GROUP; b0 < nb; b0 += GROUP) {
            double v[GROUP];
#pragma unroll
            for (int u = 0; u < GROUP; ++u) v[u] = slab[(size_t)(b0 + u < nb ? b0 + u : b0) * NS + e];
#pragma unroll
            for (int u = 0; u < GROUP; ++u) t += (b0 + u < nb) ? v[u] : 0.0;
        }
        out[e] = t;
    };
    finish(ra, T.slab0, T.nb0, NS0, tid, st0);
    finish(rb, T.slab0, T.nb0, NS0, tid + TT, st0);
    reduce_rows(T.slab0, T.nb0, NS0, st0, tid + 2 * TT, TT);
    if constexpr (HAS1) {
        finish(rc, T.slab1, T.nb1, NS1, tid, st1);
        reduce_rows(T.slab1, T.nb1, NS1, st1, tid + TT, TT);
    }
    __syncthreads();

    const uint32_t prev_row = T.ctl->row;
    const bool first = T.ctl->first != 0u;                  // no sweep of this erm_run precedes this step
    const uint32_t sweep = T.ctl->sweep + ((T.mode == 0 && STEP == 0) ? 1u : 0u);   // the sweep being drawn
    const uint32_t row = (T.mode == 0 && STEP == 0 && !first) ? prev_row + 1u : prev_row;

    // ---- log-likelihood of the sweep the last full pass completed
    if (STEP == 0 && tid == 0 && !first && T.tr_ll) {
        double llv = st0[NS0 - 1];
        if (fam_cq(MODEL)) llv += st1[NS1 - 1];
        if (rtll_stats<MODEL, 0>()) llv += lp[par_off_derived(J) + 1];         // ERM_RTLL_STATS: the subject-free part, left by the tiny step that drew the sweep's lambda / sig2t
        T.tr_ll[prev_row] = llv;
    }
    if (T.mode == 1) return;
    ERM_DIAG_STOP(T, 1);

    tiny_draws<MODEL, STEP>(T, lp, st0, st1, part, work, sh_x, sweep);
    tiny_publish<MODEL, STEP>(T, lp, sweep, row, tid, TINY_THREADS);
}

// ---------------------------------------------------------------------------------------------------------------------
// Convergence diagnostics on the device-resident traces (the reference pulls Post.ra/rt/qr through MCMCChains'
// ess_rhat in checkConvergence, src/SimTools.jl:419-443): split-R-hat and the effective sample size by Geyer's initial monotone
// sequence over the split chains (Gelman et al., BDA3 sec. 11.4-11.5; the non-rank-normalised estimator), one thread per parameter.
//   draws: trace row (m * nChain + l), m >= nBurnin; each chain l is split into its first and last n = floor((nIter - nBurnin)/2)
//   draws  => M = 2 nChain sequences.  W = mean of the sequences' variances (n-1 denominator), B/n = variance of their means,
//   var+ = (n-1)/n W + B/n, rhat = sqrt(var+ / W), rho_t = 1 - (W - mean_c acov_c(t)) / var+ with acov_c(t) = 1/n sum_i (x_i - mu_c)
//   (x_{i+t} - mu_c); P_k = rho_{2k} + rho_{2k+1} summed while positive and made non-increasing; ess = M n / (-1 + 2 sum_k P_k).
// A parameter that never moves (beta[1] = 0, Sigma_p[1,1] = 1 ...) has W = 0 and gets NaN, as MCMCChains reports it.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int DIAG_MAXSEQ = 32;
// Subject-sharded chains: a device's reduced statistics (the nb group rows of a pass) summed into ONE row, the unit the devices
// all-gather before every tiny step; reduce_rows order, so the result does not depend on the launch geometry of this kernel
__global__ void __launch_bounds__(256) shard_pack_kernel(const double* gslab, int nb, int NS, double* out)
{
    reduce_rows(gslab, nb, NS, out, (int)threadIdx.x, (int)blockDim.x);
}

// erm_set_data on the device.  The caller's arrays are column-major (Julia): uploaded as they are, then
//   colstats_cm_kernel : one workgroup per column j: K0_j = sum_i (Y_ij - 1/2), sum_i logT_ij (fp64, fixed order), validity flags
//                        (bit 0: a Y that is not 0/1, bit 1: a non-finite logT);
//   to_rows_kernel     : 32 x 32 tiles through LDS, dst[i][j] = (T)(src[j][i] - shift[j]) -- Y bytes, logT centred by its column mean
//                        (subtracted in fp64 BEFORE the value is rounded to the engine's cell type), X;
//   colsq_kernel       : per-workgroup partial sums over rows of the squared centred values ([block][J]).
__global__ void __launch_bounds__(256) colstats_cm_kernel(const uint8_t* Y, const double* L, long long N, int has_l, double* out, int J, unsigned int* flags)
{
    const int j = blockIdx.x, tid = threadIdx.x;
    double sk = 0.0, sl = 0.0;
    unsigned int bad = 0u;
    for (long long i = tid; i < N; i += 256) {
        const uint8_t y = Y[(size_t)j * N + i];
        bad |= (y > 1) ? 1u : 0u;
        sk += (double)y - 0.5;
        if (has_l) { const double v = L[(size_t)j * N + i]; bad |= (fabs(v) < 1.79e308) ? 0u : 2u; sl += v; }
    }
    __shared__ double shk[256], shl[256];
    shk[tid] = sk; shl[tid] = sl;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) { shk[tid] += shk[tid + w]; shl[tid] += shl[tid + w]; } __syncthreads(); }
    if (tid == 0) { out[j] = shk[0]; out[J + j] = shl[0]; }
    if (bad) atomicOr(flags, bad);
}
template <typename S, typename T>
__global__ void __launch_bounds__(256) to_rows_kernel(const S* src, long long N, int J, const double* shift, T* dst)
{
    __shared__ double tile[32][33];
    const long long i0 = (long long)blockIdx.x * 32;
    const int j0 = (int)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int j = j0 + r; const long long i = i0 + tx;
        if (j < J && i < N) tile[r][tx] = (double)src[(size_t)j * N + i] - (shift ? shift[j] : 0.0);
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long long i = i0 + r; const int j = j0 + tx;
        if (i < N && j < J) dst[(size_t)i * J + j] = (T)tile[tx][r];
    }
}
template <typename real>
__global__ void __launch_bounds__(128) colsq_kernel(const real* C, long long N, int J, double* part)
{
    const long long per = (N + gridDim.x - 1) / gridDim.x, r0 = (long long)blockIdx.x * per, r1 = (r0 + per < N) ? r0 + per : N;
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        double sq = 0.0;
        for (long long i = r0; i < r1; ++i) { const double c = (double)C[(size_t)i * J + j]; sq += c * c; }
        part[(size_t)blockIdx.x * J + j] = sq;
    }
}

// Post.ra / rt / qr in Julia layout: the device keeps a subject-level trace as [row = m * nChain + l][subject] (coalesced stores, one row per
// sweep); Julia's array is [nIter][width][nChain] with the iteration fastest.  dst[i * nIter + m] = (double) src[(m * nChain + l) * ld + i] for ONE
// chain l, 32 x 32 tiles through LDS so that both the reads (along subjects) and the writes (along iterations) are coalesced.
template <typename T>
__global__ void __launch_bounds__(256) trace_transpose_kernel(const T* src, long long ld, long long ncol, int nIter, int nChain, int l, double* dst)
{
    __shared__ double tile[32][33];
    const long long i0 = (long long)blockIdx.x * 32;
    const int m0 = (int)blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int m = m0 + r; const long long i = i0 + tx;
        if (m < nIter && i < ncol) tile[r][tx] = (double)src[((long long)m * nChain + l) * ld + i];
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const long long i = i0 + r; const int m = m0 + tx;
        if (i < ncol && m < nIter) dst[i * nIter + m] = tile[tx][r];
    }
}

// dst[i] += src[i] (chain farms: post-burn-in sums of the chains that share a device, before the RCCL all-reduce over the devices)
__global__ void __launch_bounds__(256) acc_kernel(double* dst, const double* src, long long n)
{
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dst[i] += src[i];
}

template <typename T>
__global__ void diag_kernel(const T* tr, long long ncol, long long ld, int nIter, int nChain, int nBurnin, double* ess, double* rhat)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= ncol) return;
    const int Tn = nIter - nBurnin, n = Tn / 2, M = 2 * nChain;
    auto at = [&](int c, int i) -> double {                      // draw i of split sequence c
        const int l = c >> 1, m0 = nBurnin + ((c & 1) ? Tn - n : 0);
        return (double)tr[((long long)(m0 + i) * nChain + l) * ld + k];
    };
    double mu[DIAG_MAXSEQ];
    double W = 0.0, mbar = 0.0;
    for (int c = 0; c < M; ++c) {
        double s1 = 0.0;
        for (int i = 0; i < n; ++i) s1 += at(c, i);
        mu[c] = s1 / n; mbar += mu[c];
        double s2 = 0.0;
        for (int i = 0; i < n; ++i) { const double d = at(c, i) - mu[c]; s2 += d * d; }
        W += s2 / (n - 1);
    }
    W /= M; mbar /= M;
    double Bn = 0.0;
    for (int c = 0; c < M; ++c) Bn += (mu[c] - mbar) * (mu[c] - mbar);
    Bn /= (M - 1);
    const double varp = W * (n - 1) / n + Bn;
    if (!(W > 0.0)) { ess[k] = __builtin_nan(""); rhat[k] = __builtin_nan(""); return; }
    rhat[k] = sqrt(varp / W);
    auto rho = [&](int t) -> double {
        double a = 0.0;
        for (int c = 0; c < M; ++c) {
            double s = 0.0;
            for (int i = 0; i + t < n; ++i) s += (at(c, i) - mu[c]) * (at(c, i + t) - mu[c]);
            a += s / n;
        }
        return 1.0 - (W - a / M) / varp;
    };
    double sum = 0.0, prev = 1e300;
    for (int t = 0; t + 1 < n; t += 2) {
        double P = (t == 0 ? 1.0 - (W - W * (n - 1) / n) / varp : rho(t)) + rho(t + 1);
        if (!(P > 0.0)) break;
        if (P > prev) P = prev;
        prev = P;
        sum += P;
    }
    ess[k] = (double)M * n / (-1.0 + 2.0 * sum);
}

// ---------------------------------------------------------------------------------------------------------------------
// Synthetic data on the device: the generators of src/SimTools.jl -- setDataRtIrt :149-178, setDataRtIrtNull
// :117-144, setDataMlIrt :349-368, setDataRtIrtCross :220-255, setDataRtIrtLatent :304-343 -- written straight into the engine's
// resident buffers, one thread per subject.  Streams (DATA_SUBJ, i) / (DATA_CELL, i, j) of the data seed: like the host generators,
// this is the reference's distribution, not Julia's Random.seed! stream.
//   gen 0 MlIrt : X[:,1] ~ Bernoulli(1/2), X[:,2:] ~ N(0,1), theta ~ N(X beta, 1)
//   gen 1 RtIrt : X ~ N(0,1), (theta, zeta) = X beta + N2(0, Sigp), logT ~ N(lambda_j - zeta_i, sig2t_j) truncated to (0, inf)
//   gen 2 Null  : (theta, zeta) ~ N2(0, Sigp), logT as RtIrt
//   gen 3 Cross : (theta, zeta) ~ N2(0, Sigp), logT = lambda_j - zeta_i - theta_i rho_j + e
//   gen 4 Latent: theta ~ N(0,1), X ~ N(0,1), zeta = [X theta] beta + e, logT = lambda_j - zeta_i + N(0,1)
//   e ("noise"): 0 N(0, 0.3), 1 t_5, 2 Gamma(1/2, 1) - 1   (the 0.3 belongs to the normal type only: src/SimTools.jl:238-247)
// Y_ij ~ Bernoulli(logistic(a_j (theta_i - b_j))) always.  logT is written raw; center_kernel subtracts the column means afterwards.
// ---------------------------------------------------------------------------------------------------------------------
struct GenArgs {
    uint8_t* Y; void* C; void* X; double* theta; double* zeta;   // C, X in the engine's cell type
    const double* truth;       // a[J] b[J] lambda[J] sig2t[J] rho[J] | Sigp chol L00 L10 L11 | beta (RtIrt: [F][2] row-major; MlIrt [F]; Latent [F+1])
    long long N; int J, F, gen, noise; uint64_t seed;
};
__device__ inline double gen_noise(Stream& s, int kind)      // src/SimTools.jl:238-247, 322-328: Normal(0, 0.3) | TDist(5) | Gamma(1/2, 1) - 1
{
    if (kind == 0) return 0.3 * normal<double>(s);
    if (kind == 1) { const double zn = normal<double>(s); return zn / sqrt(chisq(s, 5.0) / 5.0); }
    const double u = uniform<double>(s);
    const double g = gamma_mt(s, 1.5) * u * u;                            // Gamma(a) = Gamma(a + 1) U^(1/a), a = 1/2 (Marsaglia-Tsang needs a >= 1)
    return g - 1.0;
}
template <typename real>
__global__ void gen_kernel(GenArgs G)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.N) return;
    const int J = G.J, F = G.F;
    const double* a = G.truth, *b = a + J, *lam = b + J, *sg = lam + J, *rho = sg + J, *L = rho + J, *beta = L + 3;
    real* X = reinterpret_cast<real*>(G.X);
    real* C = reinterpret_cast<real*>(G.C);
    Stream ss(G.seed, 0u, SITE_DATA_SUBJ, (uint32_t)i, 0u, 0u);
    double mt = 0.0, mz = 0.0;
    for (int f = 0; f < F; ++f) {
        double x = normal<double>(ss);
        if (G.gen == 0 && f == 0) x = uniform<double>(ss) < 0.5 ? 1.0 : 0.0;
        const real xr = (real)x;                          // the model sees the stored value
        X[(size_t)i * F + f] = xr;
        if (G.gen == 0) mt += (double)xr * beta[f];
        else if (G.gen == 1) { mt += (double)xr * beta[2 * f]; mz += (double)xr * beta[2 * f + 1]; }
        else if (G.gen == 4) mz += (double)xr * beta[f];
    }
    const double z0 = normal<double>(ss), z1 = normal<double>(ss);
    double th, ze;
    if (G.gen == 0) { th = mt + z0; ze = 0.0; }
    else if (G.gen == 4) { th = z0; ze = mz + th * beta[F] + gen_noise(ss, G.noise); }
    else { th = mt + L[0] * z0; ze = mz + L[1] * z0 + L[2] * z1; }
    G.theta[i] = th; G.zeta[i] = ze;
    for (int j = 0; j < J; ++j) {
        Stream sc(G.seed, 0u, SITE_DATA_CELL, (uint32_t)i, (uint32_t)j, 0u);
        const double eta = a[j] * (th - b[j]);
        G.Y[(size_t)i * J + j] = uniform<double>(sc) < 1.0 / (1.0 + exp(-eta)) ? 1 : 0;
        if (G.gen == 0) continue;
        double lt;
        if (G.gen == 1 || G.gen == 2) lt = truncnorm0(sc, lam[j] - ze, sqrt(sg[j]));
        else if (G.gen == 3) lt = lam[j] - ze - th * rho[j] + gen_noise(sc, G.noise);
        else lt = lam[j] - ze + normal<double>(sc);
        C[(size_t)i * J + j] = (real)lt;
    }
}
// per-workgroup partial column sums of the generated data: [block][3][J] = sum kappa, sum logT, sum logT^2 (fp64), then x'x partials
template <typename real>
__global__ void colsum_kernel(const uint8_t* Y, const real* C, const real* X, long long N, int J, int F, int has_c, double* part)
{
    const long long per = (N + gridDim.x - 1) / gridDim.x, r0 = (long long)blockIdx.x * per, r1 = (r0 + per < N) ? r0 + per : N;
    const int p = F + 1;
    double* out = part + (size_t)blockIdx.x * (3 * J + p * p);
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        double sk = 0.0, s1 = 0.0, s2 = 0.0;
        for (long long i = r0; i < r1; ++i) {
            sk += (double)Y[(size_t)i * J + j] - 0.5;
            if (has_c) { const double c = (double)C[(size_t)i * J + j]; s1 += c; s2 += c * c; }
        }
        out[j] = sk; out[J + j] = s1; out[2 * J + j] = s2;
    }
    for (int e = threadIdx.x; e < p * p; e += blockDim.x) {
        const int u = e % p, v = e / p;
        double t = 0.0;
        for (long long i = r0; i < r1; ++i) {
            const double xu = u == 0 ? 1.0 : (double)X[(size_t)i * F + u - 1], xv = v == 0 ? 1.0 : (double)X[(size_t)i * F + v - 1];
            t += xu * xv;
        }
        out[3 * J + e] = t;
    }
}
// logT -> logT - column mean, and the centred sums of squares per workgroup ([block][J])
template <typename real>
__global__ void center_kernel(real* C, long long N, int J, const double* mean, double* part)
{
    const long long per = (N + gridDim.x - 1) / gridDim.x, r0 = (long long)blockIdx.x * per, r1 = (r0 + per < N) ? r0 + per : N;
    for (int j = threadIdx.x; j < J; j += blockDim.x) {
        const double m = mean[j];
        double sq = 0.0;
        for (long long i = r0; i < r1; ++i) {
            const double c = (double)C[(size_t)i * J + j] - m;
            const real cr = (real)c;
            C[(size_t)i * J + j] = cr;
            sq += (double)cr * (double)cr;
        }
        part[(size_t)blockIdx.x * J + j] = sq;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// erm_run's bookkeeping in two small launches.  As separate stream operations (two host -> device copies of the counters, a fill of the tickets, three
// device -> host copies at the end) they were ~45 us of device time per erm_run -- more than two microseconds per sweep of a 20-sweep call.
//   run_begin_kernel : both copies of the chain's counters, the group tickets zeroed, the persistent launch's wait bound and test hook;
//   run_end_kernel   : the counters of both buffers and the time-out word into PINNED HOST memory (visible to the host once the stream has drained).
// ---------------------------------------------------------------------------------------------------------------------
// (the call's parameters come from PINNED HOST memory the host fills before it enqueues the call: the kernel's arguments never change, so it can sit at the head of a
// replayed graph that holds the whole call -- erm_run: whole_graph)
struct RunParams { Ctl v; unsigned int tmo_ticks, tmo_fault; };
__global__ void __launch_bounds__(256) run_begin_kernel(Ctl* c0, Ctl* c1, const RunParams* hp, unsigned int* gcnt, int n_gcnt)
{
    const int t = (int)threadIdx.x;
    const RunParams rp = *hp;
    if (t == 0) { *c0 = rp.v; *c1 = rp.v; }
    for (int k = t; k < n_gcnt; k += (int)blockDim.x) gcnt[k] = (k == n_gcnt - 3) ? rp.tmo_ticks : ((k == n_gcnt - 2) ? rp.tmo_fault : 0u);      // [tickets | tmo flag, ticks, fault, pad]
}
__global__ void __launch_bounds__(64) run_end_kernel(const Ctl* c0, const Ctl* c1, const unsigned int* tmo, Ctl* host_out, unsigned int* host_tmo)
{
    if (threadIdx.x == 0) { host_out[0] = *c0; host_out[1] = *c1; *host_tmo = *tmo; }
}
// up to 10 device buffers copied by ONE launch: the state a persistent erm_run saves before it starts (and restores if the launch times out)
struct CopySegs { const void* src[10]; void* dst[10]; unsigned long long bytes[10]; int n; };
__global__ void __launch_bounds__(256) copy_segments_kernel(CopySegs S)
{
    const size_t gt = (size_t)blockIdx.x * blockDim.x + threadIdx.x, gn = (size_t)gridDim.x * blockDim.x;
    for (int k = 0; k < S.n; ++k) {
        const size_t nb = (size_t)S.bytes[k], nw = nb / 16;                       // hipMalloc'd buffers: 256-byte aligned
        const uint4* s = reinterpret_cast<const uint4*>(S.src[k]);
        uint4* d = reinterpret_cast<uint4*>(S.dst[k]);
        for (size_t i = gt; i < nw; i += gn) d[i] = s[i];
        for (size_t i = nw * 16 + gt; i < nb; i += gn) reinterpret_cast<unsigned char*>(S.dst[k])[i] = reinterpret_cast<const unsigned char*>(S.src[k])[i];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// DIC on the device.  getDic (src/GibbsRtIrt.pl.jl:432-472, src/GibbsRtIrtCross.pl.jl:330-353, src/GibbsRtIrtLatent.pl.jl:342-365):
//   Dhat = -2 logLik(Post.mean), Dbar = -2 mean(Post.logLike) over all iterations, pD = Dbar - Dhat, DIC = Dbar + pD.
// Post.mean never leaves the device: the post-burn-in SUMS of the subject-level draws are resident (sum_theta / sum_zeta / sum_nu), the item-level
// ones are summed from the resident item trace (item_sum_kernel, row order), and loglik_kernel evaluates the model's log-likelihood
//   getLogLikelihoodMlIrt / RtIrt / RtIrtNull (src/GibbsRtIrt.pl.jl:195-204, 262-272, 351-362), ...Cross / CrossQr (src/GibbsRtIrtCross.pl.jl:158-170, 240-258),
//   ...Latent / LatentQr (src/GibbsRtIrtLatent.pl.jl:151-162, 243-264)
// at sums * inv over the resident data set.  Plain fp64 with libm's log1p / exp / log (this runs once per sample!, not per sweep); every thread adds its
// terms in a fixed order, a workgroup's threads are summed by a fixed tree, the host adds the workgroups' partial sums in order: reproducible bit for bit.
// `sum` layout (the chain farm's summary vector): [item-level trace columns: a b lambda sig2t | small part of qr][theta N][zeta N, response-time models][nu N or N*J].
// ---------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) item_sum_kernel(const double* tr_item, long long wi, long long row0, long long row1, double* out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= wi) return;
    double t = 0.0;
    for (long long r = row0; r < row1; ++r) t += tr_item[r * wi + k];
    out[k] += t;
}
// sum of the first n entries of the log-likelihood trace: thread t adds entries t, t + 256, ... in order, then a fixed tree
__global__ void __launch_bounds__(256) ll_trace_sum_kernel(const double* tr_ll, long long n, double* out)
{
    __shared__ double sh[256];
    double t = 0.0;
    for (long long r = threadIdx.x; r < n; r += 256) t += tr_ll[r];
    sh[threadIdx.x] = t;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w]; __syncthreads(); }
    if (threadIdx.x == 0) *out = sh[0];
}
struct LogLikArgs {
    const uint8_t* Y; const void* C; const void* X;      // resident data set: Y u8 [N][J], centred logT and X in the engine's cell type, row-major
    const double* cm;                                     // column means of logT [J]
    const double* sum; double inv;                        // Post.mean = sum * inv
    long long N; int J, F, model;                         // F = covariate columns the kernels see
    long long off_theta, off_zeta, off_nu;                // offsets into `sum` (off_zeta / off_nu < 0: absent)
    double k1, k2;
    long long rows_per_block;
    double* part;                                         // [gridDim.x]
};
__device__ inline double ll_log1pexp(double x) { return x > 0.0 ? x + log1p(exp(-x)) : log1p(exp(x)); }
template <typename real>
__global__ void __launch_bounds__(256) loglik_kernel(LogLikArgs D)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    double* sa = reinterpret_cast<double*>(smem);        // a b lambda sig2t rho [5][J] | Sigp [4] | beta [2 PMAX]
    const int J = D.J, F = D.F, p = F + 1, M = D.model, tid = (int)threadIdx.x;
    double* sb = sa + J, *sl = sa + 2 * J, *sg = sa + 3 * J, *sr = sa + 4 * J, *sS = sa + 5 * J, *sbeta = sS + 4;
    __shared__ double red[256];
    for (int e = tid; e < 4 * J; e += 256) sa[e] = D.sum[e] * D.inv;
    const double* q = D.sum + 4 * J;                      // the small part of qr (tiny_publish's order)
    for (int j = tid; j < J; j += 256) sr[j] = fam_cq(M) ? q[j] * D.inv : 0.0;
    if (tid < 4) sS[tid] = (M == MLIRT) ? (tid == 0 || tid == 3 ? 1.0 : 0.0) : q[(fam_cq(M) ? J : (fam_rt(M) ? 2 * p : p + 1)) + tid] * D.inv;
    if (tid < 2 * PMAX) {
        double v = 0.0;
        if (M == MLIRT) { if (tid < p) v = q[tid] * D.inv; }
        else if (M == RTIRT) { if (tid < p) v = q[tid] * D.inv; else if (tid >= PMAX && tid < PMAX + p) v = q[p + tid - PMAX] * D.inv; }      // [theta column | zeta column at PMAX]
        else if (fam_lq(M)) { if (tid < p + 1) v = q[tid] * D.inv; }
        sbeta[tid] = v;
    }
    __syncthreads();
    const real* C = reinterpret_cast<const real*>(D.C);
    const real* X = reinterpret_cast<const real*>(D.X);
    const long long r0 = (long long)blockIdx.x * D.rows_per_block, r1 = (r0 + D.rows_per_block < D.N) ? r0 + D.rows_per_block : D.N;
    const long long ncell = (r1 > r0 ? r1 - r0 : 0) * J;
    double ll = 0.0;
    const bool qw = M == CROSSQR;                         // per-cell quantile weights
    for (long long c = tid; c < ncell; c += 256) {
        const long long i = r0 + c / J;
        const int j = (int)(c % J);
        const size_t e = (size_t)i * J + j;
        const double th = D.sum[D.off_theta + i] * D.inv;
        const double eta = sa[j] * (th - sb[j]);
        ll += (D.Y[e] ? eta : 0.0) - ll_log1pexp(eta);
        if (M != MLIRT) {
            const double ze = D.sum[D.off_zeta + i] * D.inv;
            const double lt = (double)C[e] + D.cm[j];
            double mu = sl[j] - ze, var = sg[j];
            if (fam_cq(M)) {
                const double nu = qw ? D.sum[D.off_nu + (long long)e] * D.inv : 1.0;
                mu += -th * sr[j] + D.k1 * nu;
                var *= D.k2 * nu;
            }
            const double er = lt - mu;
            ll += -0.5 * LOG_2PI - 0.5 * log(var) - 0.5 * er * er / var;
        }
    }
    const double det = sS[0] * sS[3] - sS[1] * sS[2];
    for (long long i = r0 + tid; i < r1; i += 256) {
        const double th = D.sum[D.off_theta + i] * D.inv;
        double xb0 = 0.0, xb1 = 0.0;
        if (M == MLIRT || M == RTIRT || fam_lq(M)) {
            xb0 = sbeta[0]; xb1 = sbeta[PMAX];
            for (int f = 0; f < F; ++f) { const double x = (double)X[(size_t)i * F + f]; xb0 += x * sbeta[1 + f]; xb1 += x * sbeta[PMAX + 1 + f]; }
        }
        if (M == MLIRT) { const double e0 = th - xb0; ll += -0.5 * LOG_2PI - 0.5 * e0 * e0; continue; }
        const double ze = D.sum[D.off_zeta + i] * D.inv;
        if (fam_lq(M)) {
            const double nu = (M == LATENTQR) ? D.sum[D.off_nu + i] * D.inv : 1.0;
            const double mu = xb0 + th * sbeta[p] + D.k1 * nu, var = sS[3] * D.k2 * nu, er = ze - mu;
            ll += -0.5 * LOG_2PI - 0.5 * log(var) - 0.5 * er * er / var;
        } else {
            const double e0 = th - (M == RTIRT ? xb0 : 0.0), e1 = ze - (M == RTIRT ? xb1 : 0.0);
            const double quad = (sS[3] * e0 * e0 - (sS[1] + sS[2]) * e0 * e1 + sS[0] * e1 * e1) / det;
            ll += -LOG_2PI - 0.5 * log(det) - 0.5 * quad;
        }
    }
    red[tid] = ll;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) { if (tid < w) red[tid] += red[tid + w]; __syncthreads(); }
    if (tid == 0) D.part[blockIdx.x] = red[0];
}

// checkConvergence's counts (src/SimTools.jl:427-437) from the device arrays of ess / rhat: c[0] columns with a defined ESS, c[1] of them with ESS > ess_min,
// c[2] columns with a defined R-hat, c[3] of them with R-hat < rhat_max (integer atomics: order-independent)
__global__ void __launch_bounds__(256) diag_count_kernel(const double* ess, const double* rhat, long long n, double ess_min, double rhat_max, unsigned long long* c)
{
    unsigned long long t[4] = {0ull, 0ull, 0ull, 0ull};
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        const double e = ess[k], r = rhat[k];
        if (e == e) { ++t[0]; if (e > ess_min) ++t[1]; }
        if (r == r) { ++t[2]; if (r < rhat_max) ++t[3]; }
    }
    for (int q = 0; q < 4; ++q) if (t[q]) atomicAdd(c + q, t[q]);
}

// n draws of the structural step's 2 x 2 inverse Wishart (erm_debug_invwishart): stream (seed, SIGP, i = k, sweep)
__global__ void __launch_bounds__(256) invwishart_batch_kernel(uint64_t seed, uint32_t sweep, long long n, double df, double p0, double p1, double p2, double p3, double* out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Stream ss(seed, 0u, SITE_SIGP, (uint32_t)k, 0u, sweep);
    double v3[3], S[4];
    const double Psi[4] = { p0, p1, p2, p3 };
    bartlett2_variates(ss, df, v3);
    invwishart2(Psi, v3, S);
    for (int e = 0; e < 4; ++e) out[4 * k + e] = S[e];
}

// ---------------------------------------------------------------------------------------------------------------------
// unit kernels for parity tests of the device samplers against the oracle
// ---------------------------------------------------------------------------------------------------------------------
template <typename real>
__global__ void sample_batch_kernel(int which, uint64_t seed, uint32_t site, uint32_t sweep, long long n,
                                    const double* par0, const double* par1, double* out, const double* pgtab)
{
    __shared__ double2 sh_logtab[128];
    fm::fill_log_table(sh_logtab, (int)threadIdx.x, (int)blockDim.x);
    __syncthreads();
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Stream st(seed, 0u, site, (uint32_t)k, 0u, sweep);
    double v = 0.0;
    switch (which) {
    case 15: v = fm::log(par0[k], sh_logtab); break;           // the table form of the cell path's logarithm
    case 16: v = fm::cos2pi(par0[k]); break;
    case 18: v = fm::exp_neg_ll(par0[k]); break;               // the cell log-likelihood's form of e^{-a}
    case 19: v = fm::log_word((uint32_t)par0[k], sh_logtab); break;      // log((w + 1/2) 2^-32) of the word par0 stands for (the PG attempt's -log u1)
    case 0: v = (double)uniform<real>(st); break;
    case 1: v = (double)normal<real>(st); break;
    case 2: v = (double)expo<real>(st); break;
    case 3: v = (double)pg1<real>(st, (real)par0[k], pgtab); break;
    case 4: v = (double)invgauss(st, (real)par0[k], (real)par1[k]); break;
    case 5: v = truncnorm0(st, par0[k], par1[k]); break;
    case 6: v = gamma_mt(st, par0[k]); break;
    case 7: v = pg_tail_weight(par0[k], pgtab); break;
    case 8: v = (double)qr_weight<real>(st, (real)par0[k], (real)par1[k]); break;
    case 9: v = (double)ndtri((real)par0[k]); break;
    case 17: v = qr_weight_q(st, par0[k], par1[k], par1[k] * par1[k], sh_logtab); break;      // the fp64 cell path's form of the quantile weight (parA = par0, parB = par1 at unit scale)
    case 11: v = fm::log(par0[k]); break;           // the cell path's fp64 elementary functions (erm_rng.hpp, namespace fm)
    case 12: v = fm::exp_neg(par0[k]); break;
    case 13: v = fm::sqrt(par0[k]); break;
    case 14: v = fm::div(par0[k], par1[k]); break;
    case 10: {     // PG(1, par0) through the reference form of the attempt (every statement in fp64)
        const double z = 0.5 * fabs(par0[k]);
        double o = 0.0;
        for (int tries = 0; tries < MAX_TRIES; ++tries) {
            const uint32_t w0 = st.next(), w1 = st.next(), w2 = st.next(), w3 = st.next();
            if (pg1_attempt_ref(z, w0, w1, w2, w3, pgtab, o)) break;
        }
        v = o;
    } break;
    }
    out[k] = v;
}

__global__ void gig_batch_kernel(uint64_t seed, uint32_t site, uint32_t sweep, long long n, double p, double a, double b, double* out)
{
    const long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    Stream st(seed, 0u, site, (uint32_t)k, 0u, sweep);
    out[k] = gig(st, p, a, b);
}

}  // namespace erm
