/*
 * ertirt.h -- C ABI of libertirt.so, the MI355X (gfx950) Gibbs engine behind the `sample!` path of
 * ExtendedRtIrtModeling.jl.
 *
 * The reference has no FFI: `sample!` is an ordinary Julia method whose loop body calls the draw functions of
 * src/Draw.pl.jl.  This boundary is therefore created at the `sample!` method level; each entry point names the
 * reference interface it replaces (paths relative to /root/reference).  The Julia-side binding (ccall) is in
 * INTEGRATION.md and extendedrtirtmodeling.jl_amd/julia/.
 *
 * Conventions: every function returns 0 on success or a negative ERM_ERR_* code; the message is available from
 * erm_last_error() (thread-local).  All host arrays are owned by the caller, are read/written only during the call,
 * and use Julia's column-major layout.  The library owns all device memory behind the opaque handle.
 * A handle is not re-entrant; distinct handles are independent.
 */
#ifndef ERTIRT_H
#define ERTIRT_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct erm_engine* erm_handle;

enum {
    ERM_MODEL_MLIRT = 0,    /* GibbsMlIrt          src/GibbsRtIrt.pl.jl:76-106, sample! :210-257 */
    ERM_MODEL_RTIRT = 1,    /* GibbsRtIrt          src/GibbsRtIrt.pl.jl:114-146, sample! :278-346 */
    ERM_MODEL_CROSSQR = 2,  /* GibbsRtIrtCrossQr   src/GibbsRtIrtCross.pl.jl:115-147, sample! :265-325 */
    ERM_MODEL_LATENTQR = 3, /* GibbsRtIrtLatentQr  src/GibbsRtIrtLatent.pl.jl:105-137, sample! :271-337 */
    /* the non-quantile variants */
    ERM_MODEL_NULL = 4,     /* GibbsRtIrtNull      src/GibbsRtIrt.pl.jl:151-183, sample! :367-426 (no regression; Data.X ignored) */
    ERM_MODEL_CROSS = 5,    /* GibbsRtIrtCross     src/GibbsRtIrtCross.pl.jl:77-110, sample! :176-235 */
    ERM_MODEL_LATENT = 6    /* GibbsRtIrtLatent    src/GibbsRtIrtLatent.pl.jl:70-102, sample! :168-233 */
};
enum { ERM_OK = 0, ERM_ERR_ARG = -1, ERM_ERR_HIP = -2, ERM_ERR_STATE = -3, ERM_ERR_NONFINITE = -4, ERM_ERR_NOTRACE = -5, ERM_ERR_NOMEM = -6 };
enum { ERM_PREC_F32 = 0, ERM_PREC_F64 = 1 };
enum { ERM_TRACE_SUMMARY = 0, ERM_TRACE_FULL = 1 };
enum { ERM_TRACE_RA = 0, ERM_TRACE_RT = 1, ERM_TRACE_QR = 2, ERM_TRACE_LOGLIKE = 3 };

/* Mirrors SimConditions / setCond (src/Base.pl.jl:45-62) plus the sample! kwargs
 * (intercept, itemtype, cov2one: src/GibbsRtIrt.pl.jl:210,278; src/GibbsRtIrtCross.pl.jl:265; src/GibbsRtIrtLatent.pl.jl:271). */
typedef struct {
    int32_t model;          /* ERM_MODEL_* */
    int32_t n_item;         /* Cond.nItem; 1 .. 896 (the fused sweep kernel keeps per-wave item accumulators in LDS; beyond ~400 items the engine takes the two-kernel schedule) */
    int64_t n_subj;         /* Cond.nSubj */
    int32_t n_feat;         /* Cond.nFeat (columns of Data.X; ignored by the Cross family and Null); 0 .. 14 (the structural draws hold the <= 16-column design [1 X theta] in LDS) */
    int32_t n_iter;         /* Cond.nIter */
    int32_t n_chain;        /* Cond.nChain: sweep (m,l) of the reference's interleaved loop is trace row m*nChain+l */
    int32_t n_burnin;       /* Cond.nBurnin (setCond forces round(nIter/2); the host shim does the same) */
    int32_t intercept;      /* sample!(...; intercept=false) */
    int32_t one_pl;         /* itemtype == "1pl" */
    int32_t cov2one;        /* sample!(...; cov2one) */
    int32_t sigp_mode;      /* LatentQr Sigma_p scale: 0 = reference expression src/Draw.pl.jl:594 (the N x N `/` in closed form), 1 = the evidently intended sum r_i^2/(2 k2 nu_i) */
    int32_t chain_id;       /* selects an independent random stream (one chain per GPU farms use the rank); 0 .. 255 (eight bits of the Philox counter) */
    double  q_rt;           /* Cond.qRt */
    uint64_t seed;
    int32_t device;         /* HIP device ordinal */
    int32_t precision;      /* ERM_PREC_F32: fp32 cell arithmetic + fp64 accumulation; ERM_PREC_F64: all fp64 */
    int32_t trace_mode;     /* ERM_TRACE_FULL keeps theta/zeta(/nu) per sweep (Post.ra/rt/qr); SUMMARY keeps item-level traces + running means */
    int32_t lanes_per_row;  /* 0 = auto; power of two in [1,64]: lanes that share one subject */
    int32_t block_threads;  /* 0 = auto */
    int32_t grid_blocks;    /* 0 = auto */
    int32_t profile;        /* 1 = bracket sweep-kernel launches with HIP events for erm_get_timing: two single sweeps before every replayed 32-sweep block of a
                             * long run; every sweep (in replayed 16- / 4-sweep graphs) of a run shorter than 68 sweeps */
    int32_t flags;          /* ERM_FLAG_* (diagnostics; a schedule flag keeps the launch geometry, so none changes a result); bits outside ERM_FLAG_ALL are refused */
    double  nu_trace_max_gb; /* GibbsRtIrtCrossQr, ERM_TRACE_FULL: budget in GiB for the per-sweep vec(nu) block of Post.qr (0 = the default, 16; negative or NaN is refused) */
} erm_config;
enum {
    ERM_FLAG_NO_FUSE = 1,          /* two kernels per sweep (stand-alone tiny step + row pass) instead of the fused sweep kernel */
    ERM_FLAG_NO_GRAPH = 2,         /* enqueue every sweep instead of replaying the captured 32-sweep hipGraph */
    ERM_FLAG_FARM_FORCE_RCCL = 4,  /* erm_farm_get_mean reduces over RCCL even when all chains share one device (one-rank communicator; tests) */
    ERM_FLAG_NO_PERSIST = 8,       /* small data sets: one launch per sweep instead of ONE persistent launch per erm_run (the statistics cross between its sweeps as tagged packets) */
    ERM_FLAG_TEST_PERSIST_TIMEOUT = 16, /* tests: the SECOND persistent erm_run of the engine loses one workgroup's statistics packet and waits 2 ms instead of 1 s, so that
                                        * the time-out -> restore -> per-sweep replay path of erm_run is exercised (the chain is still the per-sweep chain bit for bit) */
    ERM_FLAG_ALL = 31
};

/* Mirrors InputPara (src/Base.pl.jl:100-115).  NULL members are skipped.  Shapes:
 * theta,zeta [nSubj]; a,b,lambda,sig2t,rho [nItem]; sigp [4] = vec(Sigma_p);
 * beta: MlIrt [nFeat+1], RtIrt / Null [(nFeat+1)*2] = vec(beta) (Null: always zero), LatentQr / Latent [nFeat+2];
 * nu: LatentQr [nSubj], CrossQr [nSubj*nItem] column-major. */
typedef struct {
    double *theta, *a, *b, *zeta, *lambda, *sig2t, *beta, *sigp, *rho, *nu;
} erm_state;

typedef struct {
    double run_ms;          /* device time of the last erm_run (HIP events on the engine's stream) */
    double pass_ms_total;   /* sum of row-pass kernel durations in the last erm_run (profile=1; event-pair overhead subtracted), else 0 */
    double event_overhead_ms; /* mean duration of an empty HIP event pair on the engine's stream, measured in the same run */
    int64_t pass_launches;  /* number of row-pass launches timed */
    int64_t sweeps;         /* sweeps in the last erm_run */
    int32_t lanes_per_row, block_threads, grid_blocks, lds_bytes;
    int32_t cu_count;
    int32_t persistent;     /* 1 = this engine runs erm_run as ONE persistent launch (small data sets; ERM_FLAG_NO_PERSIST turns it off) */
    int32_t persist_fallbacks; /* persistent launches that timed out waiting for a workgroup that never became resident (another process holds compute units): the
                             * erm_run restored the state it had saved, replayed the call on the per-sweep schedule and the engine stays there (persistent = 0) */
    int32_t reserved_;
} erm_timing;

/* Replaces the Gibbs* constructors' allocation of Post and Para (src/GibbsRtIrt.pl.jl:94-104,134-144). */
int erm_create(const erm_config* cfg, erm_handle* out);
void erm_destroy(erm_handle h);

/* Replaces InputData (src/Base.pl.jl:67-78): Y 0/1 bytes [nSubj x nItem], logT = log.(T), X [nSubj x nFeat]; column-major.
 * logT may be NULL for MlIrt; X may be NULL when n_feat == 0 or for CrossQr. */
int erm_set_data(erm_handle h, const uint8_t* Y, const double* logT, const double* X);

/* Simulation studies.  Replaces setDataRtIrt / setDataRtIrtNull / setDataMlIrt / setDataRtIrtCross / setDataRtIrtLatent
 * (src/SimTools.jl:117-368): draws covariates, subject parameters and responses ON the device from the true item / structural
 * parameters in `truth` (a, b, lambda, sig2t, rho, sigp as in erm_state; beta WITHOUT intercept row: RtIrt [nFeat][2] column-major, MlIrt
 * [nFeat], Latent(Qr) [nFeat+1]; theta / zeta / nu ignored) and installs them as the resident data set, exactly as erm_set_data would.
 * noise: 0 "norm", 1 "tail" (t5), 2 "skew" (Gamma(1/2,1) - 1) for the Cross / Latent generators.  erm_get_truth returns the generated
 * theta, zeta; erm_get_data the resident data set (either source) in the caller's column-major layout (any pointer may be NULL). */
int erm_simulate_data(erm_handle h, const erm_state* truth, uint64_t seed, int noise);
int erm_get_truth(erm_handle h, double* theta, double* zeta);
int erm_get_data(erm_handle h, uint8_t* Y, double* logT, double* X);

/* Para in / out (setInitialValues: src/GibbsRtIrt.pl.jl:84-93,122-133; Cross :123-134; Latent :113-124). */
int erm_set_state(erm_handle h, const erm_state* st);
int erm_get_state(erm_handle h, erm_state* st);

/* The body of `for m in 1:nIter, l in 1:nChain` (src/GibbsRtIrt.pl.jl:221-246, 289-324; Cross :276-302; Latent :282-314):
 * runs `nsweeps` sweeps continuing from the current state; sweeps fill trace rows in order. */
int erm_run(erm_handle h, int64_t nsweeps);
/* A new seed for the chain's random streams (a simulation study re-uses ONE engine for all replications of a condition: src/SimTools.jl:457-495 constructs a
 * fresh sampler per replication, which draws from Julia's global stream).  Takes effect with the next erm_run, which then draws the chain a freshly created
 * engine with this seed would draw from the same data and state (the sweep counter that addresses the streams starts over). */
int erm_set_seed(erm_handle h, uint64_t seed);
int64_t erm_rows_done(erm_handle h);
int erm_reset_trace(erm_handle h);   /* forget recorded rows and running means (state is kept) */

/* Post.ra / Post.rt / Post.qr / Post.logLike (src/GibbsRtIrt.pl.jl:35-71; Cross :55-69; Latent :50-64) in Julia layout
 * [nIter][width][nChain], nIter fastest.  Needs ERM_TRACE_FULL for RA/RT/QR.  GibbsRtIrtCrossQr's qr carries vec(nu) (nSubj*nItem values per
 * sweep); it is recorded when nIter*nChain*nSubj*nItem values fit erm_config.nu_trace_max_gb (default 16 GiB), else ERM_ERR_NOTRACE. */
int64_t erm_trace_width(erm_handle h, int which);
int erm_get_trace(erm_handle h, int which, double* out);
/* item-level trace, always kept: out[row][4*nItem + nq] = a, b, lambda, sig2t, small part of qr (row-major) */
int64_t erm_item_trace_width(erm_handle h);
int erm_get_item_trace(erm_handle h, double* out);

/* Post.mean (src/GibbsRtIrt.pl.jl:249-254, 327-343): mean over rows >= nBurnin*nChain of everything in erm_state. */
int erm_get_mean(erm_handle h, erm_state* out);
int64_t erm_post_count(erm_handle h);   /* number of rows that entered the means */

/* checkConvergence's inputs (src/SimTools.jl:419-443, MCMCChains' ess_rhat on Post.ra / rt / qr after burn-in), computed on the device
 * from the resident traces: ess[k], rhat[k] for every column k of trace `which` (width erm_trace_width); split-R-hat and the effective
 * sample size by Geyer's initial monotone sequence over the 2*nChain split chains (the non-rank-normalised estimator); NaN for a column
 * that never moves.  Needs ERM_TRACE_FULL and a completed run. */
int erm_get_diagnostics(erm_handle h, int which, double* ess, double* rhat);
/* checkConvergence's summary itself (src/SimTools.jl:427-437) without the N-wide vectors: counts4 = { columns with a defined ESS, of those ESS > 400, columns with a
 * defined R-hat, of those R-hat < 1.1 } for trace `which`, counted on the device. */
int erm_get_convergence(erm_handle h, int which, int64_t* counts4);

/* getDic (src/GibbsRtIrt.pl.jl:432-472, src/GibbsRtIrtCross.pl.jl:330-353, src/GibbsRtIrtLatent.pl.jl:342-365) from device-resident state:
 * out = { Dbar, Dhat, pD, DIC } with Dbar = -2 mean(Post.logLike) over ALL recorded rows (burn-in included, as the reference does), Dhat = -2 logLik(Post.mean) from
 * ONE evaluation pass over the resident data set at the running means (theta, zeta, nu sums and the post-burn-in item-level trace rows; nothing N-wide crosses
 * the boundary), pD = Dbar - Dhat, DIC = Dbar + pD.  Needs a run with post-burn-in rows. */
int erm_get_dic(erm_handle h, double* out4);

int erm_get_timing(erm_handle h, erm_timing* out);
/* Subject sharding of ONE chain over several devices (second bullet).  The reference has no counterpart: its
 * conditionals (src/Draw.pl.jl:36-606) make subjects independent given the item / structural parameters, so each device keeps
 * n_subj of the n_subj_total subjects (local row 0 is subject row_base), every device repeats the tiny step on identical inputs, and
 * the only exchange is an all-gather of one row of sufficient statistics per row pass (stat width x 8 bytes per device) plus two
 * all-gathers of column sums inside erm_set_data.  Random streams are addressed by the global subject index: a sharded chain equals
 * the unsharded one up to the summation order of the statistics.
 * `exchange` is the caller's all-gather: it must place device r's `bytes_per_rank` bytes from `dev_send` at
 * dev_recv + r * bytes_per_rank on every device, return 0 when dev_recv is complete, non-zero on failure (erm_run then returns
 * ERM_ERR_STATE).  Both pointers are device memory of this engine; the engine's stream is idle during the call.  The library itself
 * stays free of any communication dependency: the host side plugs in RCCL (torch.distributed) or whatever moves the bytes.
 * Call after erm_create and before erm_set_data, on every device with the same count / n_subj_total and disjoint row ranges.
 * theta / zeta / nu in erm_state, the subject blocks of the traces and erm_get_mean cover the LOCAL subjects; item and structural
 * entries are identical on all devices.  erm_simulate_data is not available on a shard. */
typedef int (*erm_exchange_fn)(void* user, const void* dev_send, void* dev_recv, size_t bytes_per_rank);
int erm_set_shard(erm_handle h, int rank, int count, int64_t n_subj_total, int64_t row_base, erm_exchange_fn exchange, void* user);
/* The same, with the all-gather enqueued by the library itself on the engine's stream over RCCL (xGMI): no host synchronisation per
 * pass, the one-launch-per-sweep schedule and hipGraph replay stay in force.  RCCL is bound at run time (dlopen; ERM_RCCL_LIB
 * overrides the name), so nothing changes for callers that never shard.  erm_rccl_unique_id fills the 128-byte ncclUniqueId on ONE
 * process; the caller hands it to all ranks (any transport), then every rank calls erm_set_shard_rccl (collective: ncclCommInitRank). */
int erm_rccl_unique_id(void* out128);
int erm_set_shard_rccl(erm_handle h, int rank, int count, int64_t n_subj_total, int64_t row_base, const void* unique_id128);
/* hipMemcpy(dst, src, bytes, hipMemcpyDefault): lets a host-side exchange stage the buffers above without binding HIP itself. */
int erm_copy(void* dst, const void* src, size_t bytes);

/* Chain farm: the reference's nChain chains as INDEPENDENT chains, one per GPU (north_star).  Replaces the chain index of
 * `for m in 1:nIter, l in 1:nChain` (src/GibbsRtIrt.pl.jl:289) and the joint mean over iterations and chains (:327-343).  Chain l runs on
 * HIP device devices[l] (devices may repeat) with random stream chain_id = l and a full copy of the data; cfg->device, cfg->chain_id and
 * cfg->n_chain are ignored (every chain records cfg->n_iter rows).  One host thread per chain inside the library drives that chain's
 * stream, so the chains sample concurrently and never communicate.  erm_farm_get_mean is the only collective: the post-burn-in sums
 * of the chains on one device are added on that device, the devices' vectors are summed by ONE ncclAllReduce (RCCL over xGMI, bound at
 * run time as for erm_set_shard_rccl; skipped when all chains share one device unless ERM_FLAG_FARM_FORCE_RCCL is set) and divided by the
 * total number of post-burn-in rows.  Deliberate deviation from the reference: its nChain > 1 is ONE chain whose sweeps are dealt
 * round-robin to nChain trace slabs (erm_create with n_chain > 1 reproduces that); independent chains leave Post.mean unchanged in
 * expectation and make R-hat meaningful.
 * erm_farm_get_trace fills Post.ra / rt / qr / logLike [nIter][width][nChain] with chain l in slab l.  erm_farm_engine lends chain l's
 * engine (owned by the farm) for erm_get_item_trace / erm_get_diagnostics / erm_get_timing. */
typedef struct erm_farm* erm_farm_handle;
int erm_farm_create(const erm_config* cfg, const int32_t* devices, int32_t n_chains, erm_farm_handle* out);
void erm_farm_destroy(erm_farm_handle f);
int32_t erm_farm_chains(erm_farm_handle f);
erm_handle erm_farm_engine(erm_farm_handle f, int32_t chain);
int erm_farm_set_data(erm_farm_handle f, const uint8_t* Y, const double* logT, const double* X);   /* as erm_set_data, to every chain */
int erm_farm_set_state(erm_farm_handle f, int32_t chain, const erm_state* st);                     /* each chain's own setInitialValues */
int erm_farm_get_state(erm_farm_handle f, int32_t chain, erm_state* st);
int erm_farm_run(erm_farm_handle f, int64_t nsweeps);                                              /* nsweeps sweeps of EVERY chain, concurrently */
int erm_farm_reset_trace(erm_farm_handle f);
int erm_farm_get_trace(erm_farm_handle f, int which, double* out);
int erm_farm_get_mean(erm_farm_handle f, erm_state* out);
int erm_farm_get_dic(erm_farm_handle f, double* out4);     /* as erm_get_dic: Dbar over the rows of all chains, Dhat at the joint Post.mean (the same reduction as erm_farm_get_mean, evaluated on the first device) */
int erm_farm_set_seed(erm_farm_handle f, uint64_t seed);
int64_t erm_farm_post_count(erm_farm_handle f);
int erm_farm_used_rccl(erm_farm_handle f);                  /* 1 if the last erm_farm_get_mean reduced over RCCL */
/* What a multi-GPU benchmark of the farm reports: wall-clock of the last erm_farm_run (all chains, host side), the device time of each chain's
 * last erm_run (run_ms[n_chains], may be NULL), the wall-clock of the last erm_farm_get_mean (without the one-off communicator creation) and of its all-reduce alone, and the number of ranks
 * of the library's own RCCL communicator (ncclCommCount; 0 if no communicator exists). */
typedef struct {
    double run_wall_ms, gather_ms, allreduce_ms;
    double comm_init_ms;    /* ncclCommInitAll, paid by the first erm_farm_get_mean over more than one device (not part of gather_ms) */
    int32_t rccl_ranks, n_devices;
} erm_farm_timing;
int erm_farm_get_timing(erm_farm_handle f, erm_farm_timing* out, double* run_ms);

const char* erm_last_error(void);
const char* erm_version(void);
/* Layout version of the structs above (erm_config, erm_timing, ...): a binder compares it with the ERM_ABI_VERSION it was written against before the first
 * erm_create.  4: erm_timing grew by persist_fallbacks; erm_config.flags refuses unknown bits. */
#define ERM_ABI_VERSION 4
int erm_abi_version(void);

/* Diagnostics: run a device sampler on n independent streams (stream k = (seed, site, i=k, sweep)); used by the parity
 * tests to compare the device restatement of each sampler with the oracle.  which: 0 uniform, 1 normal, 2 expo,
 * 3 PG(1, par0), 4 IG(par0, par1), 5 TN(par0, par1; 0, inf), 6 Gamma(par0), 7 PG mixture weight(par0), 8 QR weight(par0, par1),
 * 9 normal quantile(par0), 10 PG(1, par0) through the reference form of the attempt (fp64: every decision in fp64),
 * 11-14 the fp64 cell path's log(par0), exp(-par0), sqrt(par0), par0 / par1; 15 its table-driven log(par0), 16 cos(2 pi par0), 17 its form of the QR weight (par0 = |residual|, par1 = parB at unit scale: the same variate as 8),
 * 18 the cell log-likelihood's exp(-par0), 19 log((w + 1/2) 2^-32) of the 32-bit word w = par0 (the PG attempt's logarithm of a uniform). */
int erm_debug_sample(int device, int precision, int which, uint64_t seed, uint32_t site, uint32_t sweep, int64_t n,
                     const double* par0, const double* par1, double* out);

/* n draws of the 2 x 2 InverseWishart(nu, Psi) of drawSubjCovariance (src/Draw.pl.jl:499-515) through the device code the structural step runs (Bartlett factor of
 * the Wishart on Psi^-1, then the inverse): out[4 k .. 4 k + 3] = vec(Sigma) of draw k, which uses stream (seed, site SIGP, i = k, sweep).  psi = vec(Psi). */
int erm_debug_invwishart(int device, uint64_t seed, uint32_t sweep, int64_t n, double nu, const double* psi4, double* out);

/* n draws of the generalized inverse Gaussian GIG(p, a, b) (density ~ x^(p-1) exp(-(a x + b/x)/2); the distribution type of
 * src/GenInvGaussian.jl:17-30, whose sampler :76-106 is dead code in the reference) by Devroye's (2014) sampler, fp64; element k uses
 * stream (seed, site, i = k, sweep).  The live quantile weights are the p = -1/2 case and use the inverse-Gaussian sampler. */
int erm_sample_gig(int device, uint64_t seed, uint32_t site, uint32_t sweep, int64_t n, double p, double a, double b, double* out);

#ifdef __cplusplus
}
#endif
#endif
