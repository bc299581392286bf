// ertirt.hip -- C-ABI host implementation of libertirt.so (declared in include/ertirt.h).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -I include ertirt.hip -o libertirt.so
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>      // types only: the library is bound at run time (dlopen) and only by subject-sharded chains
#include <dlfcn.h>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>
#include "ertirt.h"
#include "erm_kernels.hpp"
#include "erm_geometry.hpp"

using namespace erm;

namespace {

thread_local std::string g_err;
int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HIPCHK(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t e__ = (expr);                                                                         \
        if (e__ != hipSuccess) {                                                                         \
            return fail(ERM_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e__));                \
        }                                                                                                \
    } while (0)

// RCCL, bound lazily: libertirt.so has no link-time communication dependency.  A process that already loaded RCCL (torch does)
// gets that same copy back from dlopen by soname.
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::mutex mu;
    int load() {
        std::lock_guard<std::mutex> lock(mu);      // engines of several host threads may shard at the same time
        if (lib) return 0;
        const char* env = getenv("ERM_RCCL_LIB");
        const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char* n : names) { if (n && (lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break; }
        if (!lib) return fail(ERM_ERR_STATE, std::string("cannot load RCCL: ") + dlerror());
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(dlsym(lib, "ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(dlsym(lib, "ncclCommInitRank"));
        AllGather = reinterpret_cast<decltype(AllGather)>(dlsym(lib, "ncclAllGather"));
        AllReduce = reinterpret_cast<decltype(AllReduce)>(dlsym(lib, "ncclAllReduce"));
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(dlsym(lib, "ncclCommInitAll"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(dlsym(lib, "ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(dlsym(lib, "ncclGroupEnd"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(dlsym(lib, "ncclCommDestroy"));
        CommCount = reinterpret_cast<decltype(CommCount)>(dlsym(lib, "ncclCommCount"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(dlsym(lib, "ncclGetErrorString"));
        if (!GetUniqueId || !CommInitRank || !AllGather || !CommDestroy || !CommCount || !GetErrorString || !AllReduce || !CommInitAll || !GroupStart || !GroupEnd) { lib = nullptr; return fail(ERM_ERR_STATE, "RCCL library lacks an expected symbol"); }
        return 0;
    }
};
Rccl g_rccl;
#define RCCLCHK(expr)                                                                                    \
    do {                                                                                                 \
        ncclResult_t r__ = (expr);                                                                       \
        if (r__ != ncclSuccess) return fail(ERM_ERR_STATE, std::string(#expr) + ": " + g_rccl.GetErrorString(r__)); \
    } while (0)

// host -> device, complete on return for EVERY stream: the copy runs on the NULL stream, with which the engines' non-blocking streams are not ordered
#define H2D(dst, src, n) do { HIPCHK(hipMemcpy((dst), (src), (n), hipMemcpyHostToDevice)); HIPCHK(hipStreamSynchronize(nullptr)); } while (0)
struct DevBuf {
    void* p = nullptr; size_t bytes = 0;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t n) {
        if (p) { (void)hipFree(p); p = nullptr; }
        bytes = n;
        if (n == 0) return 0;
        HIPCHK(hipMalloc(&p, n));
        // hipMemset on device memory is asynchronous with respect to the host and runs on the NULL stream, with which the engines' non-blocking
        // streams are not ordered: a kernel enqueued next on an engine stream could write the buffer before (and be wiped by) the fill.  Seen as
        // a farm chain whose data constants were off about once in 500 runs, with three host threads in erm_set_data at once.
        HIPCHK(hipMemset(p, 0, n));
        HIPCHK(hipStreamSynchronize(nullptr));
        return 0;
    }
    template <typename T> T* as() const { return reinterpret_cast<T*>(p); }
};

// A persistent launch needs all its workgroups resident at once.  Two persistent launches of ONE process on one device (a farm's chains sharing a
// device, several engines) could each hold some compute units and wait for the other's for ever: they take turns.
std::mutex g_persist_mu[64];

struct EngineBase {
    erm_config cfg{};
    virtual ~EngineBase() {}
    virtual int init() = 0;
    virtual int set_data(const uint8_t*, const double*, const double*) = 0;
    virtual int set_state(const erm_state*) = 0;
    virtual int get_state(erm_state*) = 0;
    virtual int run(int64_t) = 0;
    virtual int get_trace(int, double*) = 0;
    virtual int get_item_trace(double*) = 0;
    virtual int get_mean(erm_state*) = 0;
    virtual int get_diagnostics(int, double*, double*) = 0;
    virtual int get_convergence(int, int64_t*) = 0;
    virtual int get_dic(double*) = 0;
    virtual int set_seed(uint64_t) = 0;
    // DIC pieces for the chain farm: the log-likelihood at sum * inv (sum: a device vector in the summary layout) and the sum of the recorded logLike rows
    virtual int loglik_at(const double* dsum, double inv, double* ll) = 0;
    virtual int ll_trace_sum(double* out) = 0;
    virtual int simulate_data(const erm_state*, uint64_t, int) = 0;
    virtual int get_data(uint8_t*, double*, double*) = 0;
    virtual int get_truth(double*, double*) = 0;
    virtual int reset_trace() = 0;
    virtual int set_shard(int, int, int64_t, int64_t, erm_exchange_fn, void*, const void*) = 0;
    // chain farms (erm_farm_*): this chain's post-burn-in SUMS [item-level trace columns | theta | zeta | nu] added into `acc` (device memory
    // of this engine's device, summary_len() doubles), and the inverse step from a vector of means to an erm_state
    virtual int64_t summary_len() const = 0;
    virtual int summary_add(double* acc) = 0;
    virtual int summary_unpack(const double* mean, erm_state* out) const = 0;
    int64_t rows_done = 0;
    int64_t post_rows = 0;
    erm_timing timing{};
    int64_t trace_width(int which) const {
        const int64_t N = cfg.n_subj, J = cfg.n_item, F = cfg.n_feat;
        switch (which) {
        case ERM_TRACE_RA: return N + 2 * J;                                              // src/GibbsRtIrt.pl.jl:44,65
        case ERM_TRACE_RT: return cfg.model == ERM_MODEL_MLIRT ? 0 : N + 2 * J;           // :66 (MlIrt's rt stays [])
        case ERM_TRACE_QR:
            switch (cfg.model) {
            case ERM_MODEL_MLIRT: return F + 1;                                           // :45
            case ERM_MODEL_RTIRT: return 2 * (F + 1) + 4;                                 // :67
            case ERM_MODEL_CROSSQR: return J + 4 + N * J;                                 // src/GibbsRtIrtCross.pl.jl:65
            case ERM_MODEL_LATENTQR: return F + 2 + 4 + N;                                // src/GibbsRtIrtLatent.pl.jl:60
            case ERM_MODEL_NULL: return 2 * (F + 1) + 4;                                  // OutputPost, src/GibbsRtIrt.pl.jl:67
            case ERM_MODEL_CROSS: return J + 4;                                           // OutputPostCross, src/GibbsRtIrtCross.pl.jl:46
            case ERM_MODEL_LATENT: return F + 2 + 4;                                      // OutputPostRtIrtLatent, src/GibbsRtIrtLatent.pl.jl:43
            }
            return 0;
        case ERM_TRACE_LOGLIKE: return 1;
        }
        return 0;
    }
    int nq() const {
        switch (cfg.model) {
        case ERM_MODEL_MLIRT: return cfg.n_feat + 1;
        case ERM_MODEL_RTIRT: return 2 * (cfg.n_feat + 1) + 4;
        case ERM_MODEL_CROSSQR: case ERM_MODEL_CROSS: return cfg.n_item + 4;
        case ERM_MODEL_NULL: return 2 + 4;            // the kernels see no covariates: [beta_theta0, beta_zeta0] = 0, then Sigp
        default: return cfg.n_feat + 2 + 4;
        }
    }
    int64_t item_trace_width() const { return 4 * (int64_t)cfg.n_item + nq(); }
};

template <typename real> struct Engine : EngineBase {
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::vector<hipEvent_t> pass_ev;
    int cu_count = 256;
    int64_t N = 0; int J = 0, F = 0, Fk = 0;   // Fk = covariate columns the kernels see (0 for CrossQr)
    int64_t rows_cap = 0;
    bool has_data = false;
    int W = 8, logW = 3, IPL = 1, block_threads = 1024, grid_blocks = 256;
    int64_t rows_per_block = 0; int rows_per_wave = 0;
    size_t lds_pass[2] = {0, 0};
    int ns[2] = {0, 0};
    uint32_t sweeps_total = 0;

    DevBuf dY, dC, dOmega, dNu, dX, dTheta, dZeta, dCst, dSlab0, dSlab1, dGslab1, dGcnt;
    // double-buffered: a fused sweep kernel reads [cur] and writes [1 - cur] (parameter block, counters, group-reduced statistics)
    DevBuf dParB[2], dCtlB[2], dGslab0B[2];
    DevBuf dDbgTs;                                   // ERM_TIMELINE diagnostics
    DevBuf dXbuf;                                    // persistent launches: packet rows of the statistics exchange
    DevBuf dSnap;                                    // persistent launches: the state saved before an erm_run (restored if the launch times out: run())
    bool snap_stats_valid = false;
    int persist_fault_countdown = 0;                 // ERM_FLAG_TEST_PERSIST_TIMEOUT: the engine's SECOND persistent erm_run loses a statistics row (the first leaves rows and sums for the restore to keep)
    uint32_t xtag = 0;                               // last packet tag handed out (tags only grow; the buffer is cleared before they wrap)
    int cur = 0;
    int n_groups = 1;
    // subject sharding (erm_set_shard): this device holds subjects [row_base, row_base + N) of n_total
    int shard_rank = 0, shard_count = 1;
    int64_t n_total = 0, row_base = 0;
    erm_exchange_fn exch = nullptr; void* exch_user = nullptr;     // the caller's all-gather (host-synchronous) ...
    ncclComm_t comm = nullptr;                                       // ... or RCCL enqueued on the engine's stream
    DevBuf dShardSend, dShardRecv[2];
    bool sharded() const { return exch != nullptr || comm != nullptr; }
    bool persist = false;                             // small data sets: ONE launch per erm_run (pass_kernel<..., PERSIST>: the statistics rows cross between its sweeps as tagged packets)
    bool fuse_ok = true;                              // false when the fused kernel's LDS layout cannot fit (very long tests): two kernels per sweep then
    bool fused() const { return !m_cq() && fuse_ok; }  // single-pass models run the tiny step inside the row-pass kernel
    DevBuf dSumTheta, dSumZeta, dSumNu, dTrTheta, dTrZeta, dTrNu, dTrItem, dTrLl;

    ~Engine() override {
        drop_graphs();
        if (comm) (void)g_rccl.CommDestroy(comm);
        for (auto e : pass_ev) (void)hipEventDestroy(e);
        if (host_ctl) (void)hipHostFree(host_ctl);
        if (host_run) (void)hipHostFree(host_run);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }

    bool is_rt() const { return cfg.model != ERM_MODEL_MLIRT; }
    bool m_cq() const { return fam_cq(cfg.model); }
    bool m_nu() const { return has_nu(cfg.model); }
    int p() const { return Fk + 1; }
    // calls f(std::integral_constant<int, MODEL>) for the configured model
    template <typename Fn> int dispatch(Fn&& f) {
        switch (cfg.model) {
        case ERM_MODEL_MLIRT: return f(std::integral_constant<int, MLIRT>{});
        case ERM_MODEL_RTIRT: return f(std::integral_constant<int, RTIRT>{});
        case ERM_MODEL_CROSSQR: return f(std::integral_constant<int, CROSSQR>{});
        case ERM_MODEL_LATENTQR: return f(std::integral_constant<int, LATENTQR>{});
        case ERM_MODEL_NULL: return f(std::integral_constant<int, NULLM>{});
        case ERM_MODEL_CROSS: return f(std::integral_constant<int, CROSS>{});
        case ERM_MODEL_LATENT: return f(std::integral_constant<int, LATENT>{});
        }
        return fail(ERM_ERR_ARG, "unknown model");
    }
    // LatentQr with sigp_mode 1 also accumulates the 1/nu-weighted Gram entries of [1 X theta | u]
    int ngx() const {
        if (cfg.model != ERM_MODEL_LATENTQR || cfg.sigp_mode != 1) return 0;
        const int q = p() + 1;
        return q * (q + 1) / 2 + q + 1;
    }
    Geom G;                                          // launch geometry and LDS layout: decided by plan_geometry (erm_geometry.hpp) and nowhere else
    DevBuf dPgTab;                                   // the Polya-Gamma proposal table [PG_NBIN][4] (erm_rng.hpp, pg_bin)

    int init() override {
        N = cfg.n_subj; J = cfg.n_item; F = cfg.n_feat;
        Fk = (m_cq() || cfg.model == ERM_MODEL_NULL) ? 0 : F;     // the Cross family and Null never touch Data.X
        if (N <= 0 || J <= 0 || F < 0) return fail(ERM_ERR_ARG, "n_subj, n_item must be positive and n_feat non-negative");
        if (N >= (1LL << 32)) return fail(ERM_ERR_ARG, "n_subj must fit 32 bits");
        if (cfg.model < 0 || cfg.model > ERM_MODEL_LATENT) return fail(ERM_ERR_ARG, "unknown model");
        if (Fk + 2 > PMAX) return fail(ERM_ERR_ARG, "n_feat too large (max " + std::to_string(PMAX - 2) + ")");
        if (J > 896) return fail(ERM_ERR_ARG, "n_item too large (max 896)");
        if (m_nu() && !(cfg.q_rt > 0.0 && cfg.q_rt < 1.0))
            return fail(ERM_ERR_ARG, "qRt must be between 0 and 1");   // @assert at src/Draw.pl.jl:476
        if (cfg.n_iter < 0 || cfg.n_chain < 1 || cfg.n_burnin < 0) return fail(ERM_ERR_ARG, "bad n_iter / n_chain / n_burnin");
        if (cfg.sigp_mode != 0 && cfg.sigp_mode != 1) return fail(ERM_ERR_ARG, "sigp_mode must be 0 or 1");
        if (cfg.chain_id < 0 || cfg.chain_id > 255) return fail(ERM_ERR_ARG, "chain_id must be in [0, 255] (the random streams carry eight bits of it: chain 256 would replay chain 0)");
        if (cfg.flags & ~(int32_t)ERM_FLAG_ALL) return fail(ERM_ERR_ARG, "unknown bits in erm_config.flags (built against another version of ertirt.h? erm_abi_version() = " + std::to_string(ERM_ABI_VERSION) + ")");
        if (!(cfg.nu_trace_max_gb >= 0.0) || !std::isfinite(cfg.nu_trace_max_gb)) return fail(ERM_ERR_ARG, "nu_trace_max_gb must be finite and non-negative (0 = the default)");
        if (cfg.precision != ERM_PREC_F32 && cfg.precision != ERM_PREC_F64) return fail(ERM_ERR_ARG, "unknown precision");
        if (cfg.trace_mode != ERM_TRACE_SUMMARY && cfg.trace_mode != ERM_TRACE_FULL) return fail(ERM_ERR_ARG, "unknown trace_mode");
        HIPCHK(hipSetDevice(cfg.device));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, cfg.device));
        cu_count = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        HIPCHK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreate(&ev0));
        HIPCHK(hipEventCreate(&ev1));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&host_ctl), 4 * sizeof(Ctl), hipHostMallocDefault));      // [1], [2]: the two device copies of the counters, [3]: a persistent launch's time-out word -- written by run_end_kernel
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_ctl_dev), host_ctl, 0));
        HIPCHK(hipHostMalloc(reinterpret_cast<void**>(&host_run), sizeof(RunParams), hipHostMallocDefault));      // an erm_run's parameters, read by run_begin_kernel
        HIPCHK(hipHostGetDevicePointer(reinterpret_cast<void**>(&host_run_dev), host_run, 0));

        // ---- geometry (pure host function, CPU-tested: erm_geometry.hpp)
        {
            GeomIn gi;
            gi.model = cfg.model; gi.f64 = sizeof(real) == 8; gi.N = N; gi.J = J; gi.Fk = Fk; gi.ngx = ngx();
            gi.lanes_per_row = cfg.lanes_per_row; gi.block_threads = cfg.block_threads; gi.grid_blocks = cfg.grid_blocks;
            gi.cu_count = cu_count; gi.no_fuse = (cfg.flags & ERM_FLAG_NO_FUSE) != 0;
            std::string msg;
            // small data sets (the reference's own sizes): few, full workgroups and ONE persistent launch per erm_run -- decided by the planner
            gi.no_persist = (cfg.flags & ERM_FLAG_NO_PERSIST) != 0;
#ifdef ERM_DIAG_BUILD
            gi.no_persist = true;                       // the stage-timing early exits would strand the other workgroups polling for a row that never comes
#endif
            if (plan_geometry(gi, G, msg) != 0) return fail(ERM_ERR_ARG, msg);
            persist = G.persist;
            W = G.W; logW = G.logW; IPL = G.IPL; block_threads = G.block_threads; grid_blocks = G.grid_blocks; n_groups = G.n_groups;
            rows_per_block = G.rows_per_block; rows_per_wave = G.rows_per_wave; fuse_ok = G.fused;
            for (int ph = 0; ph < 2; ++ph) { lds_pass[ph] = G.lds_pass[ph]; ns[ph] = G.ns[ph]; }
        }

        // ---- device memory
        rows_cap = (int64_t)cfg.n_iter * cfg.n_chain;
        const size_t NJ = (size_t)N * J;
        int rc = 0;
        rc |= dY.alloc(NJ);
        rc |= dOmega.alloc(NJ * sizeof(real));
        if (is_rt()) rc |= dC.alloc(NJ * sizeof(real));
        if (cfg.model == ERM_MODEL_CROSSQR) rc |= dNu.alloc(NJ * sizeof(real));
        if (cfg.model == ERM_MODEL_LATENTQR) rc |= dNu.alloc((size_t)N * sizeof(real));
        if (Fk > 0) rc |= dX.alloc((size_t)N * Fk * sizeof(real));
        rc |= dTheta.alloc((size_t)N * sizeof(real));
        rc |= dZeta.alloc((size_t)N * sizeof(real));
        for (int k = 0; k < 2; ++k) rc |= dParB[k].alloc((size_t)par_size(J) * sizeof(double));
        rc |= dCst.alloc((size_t)cst_size(J) * sizeof(double));
        rc |= dSlab0.alloc((size_t)grid_blocks * ns[0] * sizeof(double));
        // (at least GROUP rows, zero-filled: the fused head requests its first GROUP group rows unconditionally and masks those beyond n_groups)
        for (int k = 0; k < 2; ++k) rc |= dGslab0B[k].alloc((size_t)std::max(n_groups, GROUP) * ns[0] * sizeof(double));
        rc |= dGcnt.alloc(((size_t)2 * n_groups + 4) * sizeof(unsigned int));       // group tickets | a persistent launch's time-out flag, wait bound (ticks), test hook, pad
        if (persist) {      // packet rows of the persistent launch's statistics exchange: [parity][workgroup][2 * ns] 64-bit packets, tags start at 1
            rc |= dXbuf.alloc((size_t)2 * grid_blocks * 2 * ns[0] * sizeof(unsigned long long));
            if (!rc) HIPCHK(hipMemset(dXbuf.p, 0, dXbuf.bytes));
        }
        if (persist) rc |= dSnap.alloc(snap_bytes());
        if (m_cq()) { rc |= dSlab1.alloc((size_t)grid_blocks * ns[1] * sizeof(double)); rc |= dGslab1.alloc((size_t)std::max(n_groups, GROUP) * ns[1] * sizeof(double)); }      // >= GROUP rows: see dGslab0B
        for (int k = 0; k < 2; ++k) rc |= dCtlB[k].alloc(sizeof(Ctl));
        rc |= dSumTheta.alloc((size_t)N * sizeof(double));
        rc |= dSumZeta.alloc((size_t)N * sizeof(double));
        if (cfg.model == ERM_MODEL_CROSSQR) rc |= dSumNu.alloc(NJ * sizeof(double));
        if (cfg.model == ERM_MODEL_LATENTQR) rc |= dSumNu.alloc((size_t)N * sizeof(double));
        rc |= dTrItem.alloc((size_t)std::max<int64_t>(rows_cap, 1) * item_trace_width() * sizeof(double));
        rc |= dTrLl.alloc((size_t)std::max<int64_t>(rows_cap, 1) * sizeof(double));
        if (cfg.trace_mode == ERM_TRACE_FULL && rows_cap > 0) {
            rc |= dTrTheta.alloc((size_t)rows_cap * N * sizeof(real));
            if (is_rt()) rc |= dTrZeta.alloc((size_t)rows_cap * N * sizeof(real));
            if (cfg.model == ERM_MODEL_LATENTQR) rc |= dTrNu.alloc((size_t)rows_cap * N * sizeof(real));
            if (cfg.model == ERM_MODEL_CROSSQR) {
                // Post.qr of GibbsRtIrtCrossQr carries vec(nu) (N*J values) per sweep (src/GibbsRtIrtCross.pl.jl:65,296): kept on the device
                // when it fits the budget (erm_config.nu_trace_max_gb, default 16 GiB), otherwise only nu's running mean is available
                const double cap_gb = cfg.nu_trace_max_gb > 0.0 ? cfg.nu_trace_max_gb : 16.0;
                const double need_gb = (double)rows_cap * (double)NJ * sizeof(real) / 1073741824.0;
                if (need_gb <= cap_gb) rc |= dTrNu.alloc((size_t)rows_cap * NJ * sizeof(real));
            }
        }
#ifdef ERM_TIMELINE_BUILD
        if (getenv("ERM_TIMELINE")) rc |= dDbgTs.alloc(2 * 16 * 16 * sizeof(unsigned long long));
#endif
        if (rc) return rc;
        if (cfg.profile) {
            pass_ev.resize(2 * 4096);
            for (auto& e : pass_ev) HIPCHK(hipEventCreate(&e));
        }

        // ---- default state (constructors' deterministic part: a = 1, b = 0, lambda = 0, sig2t = 1, Sigp = I)
        std::vector<double> par(par_size(J), 0.0);
        for (int j = 0; j < J; ++j) { par[j] = 1.0; par[3 * J + j] = 1.0; }
        par[par_off_sigp(J) + 0] = 1.0; par[par_off_sigp(J) + 3] = 1.0;
        par[par_off_derived(J)] = (double)J;
        for (int k = 0; k < 2; ++k) H2D(dParB[k].p, par.data(), par.size() * sizeof(double));
        if (dNu.p) {
            std::vector<real> ones(dNu.bytes / sizeof(real), real(1));
            H2D(dNu.p, ones.data(), dNu.bytes);
        }
        {   // the Polya-Gamma proposal table, computed here in fp64 (the oracle restates the same closed form)
            std::vector<double> tab((size_t)PG_NBIN * 4);
            for (int k = 0; k < PG_NBIN; ++k) pg_bin(k, &tab[(size_t)4 * k]);
            if (int rc2 = dPgTab.alloc(tab.size() * sizeof(double))) return rc2;
            H2D(dPgTab.p, tab.data(), tab.size() * sizeof(double));
        }
        persist_fault_countdown = (cfg.flags & ERM_FLAG_TEST_PERSIST_TIMEOUT) ? 2 : 0;
        timing.lanes_per_row = W; timing.block_threads = block_threads; timing.grid_blocks = grid_blocks;
        timing.lds_bytes = (int32_t)std::max(lds_pass[0], lds_pass[1]); timing.cu_count = cu_count; timing.persistent = persist ? 1 : 0;
        return configure_kernels();
    }

    // -------------------------------------------------------------------------------------------- kernels
    template <int MODEL, int PHASE, bool FUSED> int set_lds_attr(size_t bytes) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<MODEL, real, PHASE, FUSED>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
        return 0;
    }
    // dynamic LDS limits of every kernel this engine launches; the planner's figure for the kernels' STATIC LDS is checked against the compiler's
    size_t fused_lds() const { return G.lds_fused; }
    size_t tiny_lds() const { return G.lds_tiny; }
    template <int MODEL, int PHASE, bool FUSED> int check_static_lds() {
        hipFuncAttributes fa;
        HIPCHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&pass_kernel<MODEL, real, PHASE, FUSED>)));
        if (fa.sharedSizeBytes > G.lds_static[PHASE])
            return fail(ERM_ERR_STATE, "internal: pass_kernel declares " + std::to_string(fa.sharedSizeBytes) + " B of static LDS, the planner assumes " + std::to_string(G.lds_static[PHASE]));
        return 0;
    }
    int configure_kernels() {
        const int tl = (int)tiny_lds();
        return dispatch([&](auto m) -> int {
            constexpr int M = decltype(m)::value;
            if (int rc = check_static_lds<M, 0, false>()) return rc;
            if (int rc = set_lds_attr<M, 0, false>(lds_pass[0])) return rc;
            HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tiny_kernel<M, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, tl));
            if constexpr (!fam_cq(M)) {
                if (fused()) { if (int rc = check_static_lds<M, 0, true>()) return rc; if (int rc = set_lds_attr<M, 0, true>(fused_lds())) return rc; }
                if (persist) {
                    hipFuncAttributes fa;
                    HIPCHK(hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(&pass_kernel<M, real, 0, true, true>)));
                    if (fa.sharedSizeBytes > G.lds_static[0]) return fail(ERM_ERR_STATE, "internal: the persistent kernel's static LDS exceeds the planner's figure");
                    HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&pass_kernel<M, real, 0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fused_lds()));
                    // every workgroup of a persistent launch must be resident at once: ask the runtime how many fit a compute unit with this block size and LDS
                    int per_cu = 0;
                    HIPCHK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&pass_kernel<M, real, 0, true, true>), block_threads, fused_lds()));
                    if ((long long)per_cu * cu_count < grid_blocks) { persist = false; timing.persistent = 0; }
                }
            }
            if constexpr (fam_cq(M)) {
                if (int rc = check_static_lds<M, 1, false>()) return rc;
                if (int rc = set_lds_attr<M, 1, false>(lds_pass[1])) return rc;
                HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void*>(&tiny_kernel<M, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, tl));
            }
            return 0;
        });
    }

    // stage-timing knobs: read only by a -DERM_DIAG_BUILD library (the kernels of the shipped one contain no early return at all)
    static int diag_stop(const char* name) {
#ifdef ERM_DIAG_BUILD
        const char* e = getenv(name);
        return e ? atoi(e) : 0;
#else
        (void)name;
        return 0;
#endif
    }
    PassArgs<real> pass_args(int phase, int mode, bool fz = false) const {
        PassArgs<real> a{};
        a.Y = dY.as<uint8_t>(); a.C = dC.as<real>(); a.omega = dOmega.as<real>(); a.nu = dNu.as<real>(); a.X = dX.as<real>();
        a.theta = dTheta.as<real>(); a.zeta = dZeta.as<real>();
        a.par = dParB[cur].template as<double>(); a.cst = dCst.as<double>(); a.pgtab = dPgTab.as<double>();
        a.slab = phase == 0 ? dSlab0.as<double>() : dSlab1.as<double>();
        a.gslab = phase == 0 ? dGslab0B[fz ? 1 - cur : cur].template as<double>() : dGslab1.as<double>();
        a.gcnt = dGcnt.as<unsigned int>() + (phase == 0 ? 0 : n_groups);
        a.ctl = dCtlB[cur].template as<Ctl>();
        a.sum_theta = dSumTheta.as<double>(); a.sum_zeta = dSumZeta.as<double>(); a.sum_nu = dSumNu.as<double>();
        a.tr_theta = dTrTheta.as<real>(); a.tr_zeta = dTrZeta.as<real>(); a.tr_nu = dTrNu.as<real>();
        a.N = N; a.rows_per_block = rows_per_block; a.rows_per_wave = rows_per_wave; a.J = J; a.nFeat = Fk; a.W = W; a.logW = logW; a.IPL = IPL; a.mode = mode; a.ngx = phase == 0 ? ngx() : 0;
        a.chain = (uint32_t)cfg.chain_id; a.seed = cfg.seed;
        const double q = cfg.q_rt;
        a.k1 = (1.0 - 2.0 * q) / (q * (1.0 - q)); a.k2 = 2.0 / (q * (1.0 - q));   // src/Draw.pl.jl:163-164
        if (!m_nu()) { a.k1 = 0.0; a.k2 = 1.0; }                                   // no quantile weights: nu == 1, k1 = 0, k2 = 1
        a.dbg_stop = diag_stop("ERM_PASS_STOP"); a.dbg_sweep = (uint32_t)diag_stop("ERM_STOP_SWEEP");
        a.dbg_ts = dDbgTs.as<unsigned long long>();
        a.row_base = (uint32_t)row_base;
        a.acc_off = fz ? G.acc_off_fused : G.acc_off[phase];     // the accumulators close the launch's dynamic LDS
        for (int k = 0; k < 2; ++k) { a.parB[k] = dParB[k].template as<double>(); a.ctlB[k] = dCtlB[k].template as<Ctl>(); a.gslabB[k] = dGslab0B[k].template as<double>(); }
        a.nsweeps = 1u; a.cur0 = (uint32_t)cur;
        a.xbuf = dXbuf.as<unsigned long long>(); a.tag0 = 0u; a.tmo = dGcnt.as<unsigned int>() + 2 * n_groups;
        return a;
    }
    TinyArgs tiny_args(int mode, bool fz = false) const {
        TinyArgs t{};
        const int o = fz ? 1 - cur : cur;
        t.par = dParB[cur].template as<double>(); t.par_out = dParB[o].template as<double>();
        t.cst = dCst.as<double>(); t.slab0 = dGslab0B[cur].template as<double>(); t.slab1 = dGslab1.as<double>();
        t.ctl = dCtlB[cur].template as<Ctl>(); t.ctl_out = dCtlB[o].template as<Ctl>(); t.ctl_err = dCtlB[0].template as<Ctl>();
        t.tr_item = dTrItem.as<double>(); t.tr_ll = dTrLl.as<double>();
        t.N = N; t.J = J; t.nFeat = Fk; t.nb0 = n_groups; t.nb1 = n_groups; t.mode = mode;
        t.intercept = cfg.intercept; t.onepl = cfg.one_pl; t.cov2one = cfg.cov2one; t.sigp_mode = cfg.sigp_mode;
        t.chain = (uint32_t)cfg.chain_id; t.seed = cfg.seed;
        const double q = cfg.q_rt;
        t.k1 = (1.0 - 2.0 * q) / (q * (1.0 - q)); t.k2 = 2.0 / (q * (1.0 - q));
        if (!m_nu()) { t.k1 = 0.0; t.k2 = 1.0; }
        t.nq = nq(); t.ngx = ngx();
        if (sharded()) {     // the statistics rows of all devices, gathered after every row pass; N = the whole data set
            t.slab0 = dShardRecv[0].as<double>(); t.slab1 = dShardRecv[1].as<double>(); t.nb0 = shard_count; t.nb1 = shard_count; t.N = n_total;
        }
        t.dbg_stop = diag_stop("ERM_TINY_STOP"); t.dbg_sweep = (uint32_t)diag_stop("ERM_STOP_SWEEP");
        return t;
    }

    int64_t n_pass_timed = 0;                        // sweep-kernel launches inside event brackets in the current run
    int64_t n_brackets = 0;                          // event pairs used
    std::vector<int> bracket_launches;               // launches inside each bracket
    template <int MODEL, int PHASE> int launch_pass(int mode, bool timed) {
        PassArgs<real> a = pass_args(PHASE, mode);
        TinyArgs t{};
        const bool ev = timed && cfg.profile && (size_t)(2 * n_brackets + 1) + 64 < pass_ev.size();
        if (ev) HIPCHK(hipEventRecord(pass_ev[2 * n_brackets], stream));
        hipLaunchKernelGGL((pass_kernel<MODEL, real, PHASE, false>), dim3(grid_blocks), dim3(block_threads), lds_pass[PHASE], stream, a, t);
        if (ev) { HIPCHK(hipEventRecord(pass_ev[2 * n_brackets + 1], stream)); ++n_brackets; ++n_pass_timed; bracket_launches.push_back(1); }
        if (sharded()) return shard_exchange(PHASE, a.gslab);
        return 0;
    }
    // Subject-sharded chains: this device's statistics of the pass just enqueued -> one row -> all-gather over the devices
    int shard_exchange(int phase, const double* gslab) {
        hipLaunchKernelGGL(shard_pack_kernel, dim3(1), dim3(256), 0, stream, gslab, n_groups, ns[phase], dShardSend.as<double>());
        HIPCHK(hipGetLastError());
        return gather_row(phase, (size_t)ns[phase]);
    }
    // all-gather of n doubles of dShardSend into dShardRecv[k]: in stream order over RCCL, or through the caller's callback
    int gather_row(int k, size_t n) {
        if (comm) { RCCLCHK(g_rccl.AllGather(dShardSend.p, dShardRecv[k].p, n, ncclDouble, comm, stream)); return 0; }
        HIPCHK(hipStreamSynchronize(stream));
        if (exch(exch_user, dShardSend.p, dShardRecv[k].p, n * sizeof(double)) != 0) return fail(ERM_ERR_STATE, "the shard exchange callback failed");
        return 0;
    }
    // element-wise sum over the devices of a small host vector, in rank order on every device (data constants in erm_set_data)
    int shard_allsum(std::vector<double>& v) {
        if (!sharded()) return 0;
        const size_t nb = v.size() * sizeof(double);
        if (nb > dShardSend.bytes) return fail(ERM_ERR_STATE, "shard scratch too small");
        HIPCHK(hipMemcpyAsync(dShardSend.p, v.data(), nb, hipMemcpyHostToDevice, stream));
        if (int rc = gather_row(0, v.size())) return rc;
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<double> all(v.size() * (size_t)shard_count);
        HIPCHK(hipMemcpy(all.data(), dShardRecv[0].p, nb * (size_t)shard_count, hipMemcpyDeviceToHost));
        for (size_t e = 0; e < v.size(); ++e) { double t = 0.0; for (int r = 0; r < shard_count; ++r) t += all[(size_t)r * v.size() + e]; v[e] = t; }
        return 0;
    }
    int set_shard(int rank, int count, int64_t ntot, int64_t base, erm_exchange_fn fn, void* user, const void* rccl_id) override {
        if (has_data || rows_done > 0 || sharded()) return fail(ERM_ERR_STATE, "erm_set_shard must precede erm_set_data and be called once");
        if (count < 1 || rank < 0 || rank >= count) return fail(ERM_ERR_ARG, "bad shard rank / count");
        if (!fn && !rccl_id) return fail(ERM_ERR_ARG, "exchange callback / RCCL id is NULL");
        if (base < 0 || ntot < N || base + N > ntot) return fail(ERM_ERR_ARG, "local subjects must lie inside [0, n_subj_total)");
        if (ntot >= (1LL << 32)) return fail(ERM_ERR_ARG, "n_subj_total must fit 32 bits");
        HIPCHK(hipSetDevice(cfg.device));
        const size_t width = (size_t)std::max(std::max(ns[0], ns[1]), 3 * J + PMAX * PMAX + 8);
        if (int rc = dShardSend.alloc(width * sizeof(double))) return rc;
        for (int k = 0; k < (m_cq() ? 2 : 1); ++k) { if (int rc = dShardRecv[k].alloc(width * (size_t)std::max(count, GROUP) * sizeof(double))) return rc; }   // >= GROUP rows: see dGslab0B
        if (rccl_id) {
            if (int rc = g_rccl.load()) return rc;
            ncclUniqueId id;
            std::memcpy(&id, rccl_id, sizeof(id));
            RCCLCHK(g_rccl.CommInitRank(&comm, count, id, rank));
        }
        shard_rank = rank; shard_count = count; n_total = ntot; row_base = base; exch = fn; exch_user = user;
        persist = false; timing.persistent = 0;          // a sharded sweep exchanges its statistics rows on the host side of every launch
        return 0;
    }
    // one whole sweep of a single-pass model: tiny step + row pass in one launch; reads buffers [cur], writes [1 - cur]
    template <int MODEL> int launch_fused(bool timed) {
        PassArgs<real> a = pass_args(0, 1, true);
        TinyArgs t = tiny_args(0, true);
        const bool ev = timed && cfg.profile && (size_t)(2 * n_brackets + 1) + 64 < pass_ev.size();
        if (ev) HIPCHK(hipEventRecord(pass_ev[2 * n_brackets], stream));
        hipLaunchKernelGGL((pass_kernel<MODEL, real, 0, true>), dim3(grid_blocks), dim3(block_threads), fused_lds(), stream, a, t);
        if (ev) { HIPCHK(hipEventRecord(pass_ev[2 * n_brackets + 1], stream)); ++n_brackets; ++n_pass_timed; bracket_launches.push_back(1); }
        cur ^= 1;
        if (sharded()) return shard_exchange(0, a.gslab);     // a.gslab: the group rows this launch wrote
        return 0;
    }
    // nsweeps whole sweeps in ONE launch (small data sets): reads buffers [cur] first, alternates inside the launch, leaves cur where nsweeps single launches would
    template <int MODEL> int launch_persist(int64_t nsweeps) {
        PassArgs<real> a = pass_args(0, 1, true);
        TinyArgs t = tiny_args(0, true);
        a.nsweeps = (uint32_t)nsweeps; a.cur0 = (uint32_t)cur;
        if (xtag > 0x7fffffffu - (uint32_t)nsweeps) { HIPCHK(hipMemsetAsync(dXbuf.p, 0, dXbuf.bytes, stream)); xtag = 0; }
        a.tag0 = xtag; xtag += (uint32_t)nsweeps;
        const bool ev = cfg.profile && (size_t)(2 * n_brackets + 1) + 64 < pass_ev.size();
        if (ev) HIPCHK(hipEventRecord(pass_ev[2 * n_brackets], stream));
        hipLaunchKernelGGL((pass_kernel<MODEL, real, 0, true, true>), dim3(grid_blocks), dim3(block_threads), fused_lds(), stream, a, t);
        if (ev) { HIPCHK(hipEventRecord(pass_ev[2 * n_brackets + 1], stream)); ++n_brackets; n_pass_timed += nsweeps; bracket_launches.push_back((int)nsweeps); }
        cur = (int)((cur + nsweeps) & 1);
        return 0;
    }
    template <int MODEL, int STEP> int launch_tiny(int mode) {
        TinyArgs t = tiny_args(mode);
        hipLaunchKernelGGL((tiny_kernel<MODEL, STEP>), dim3(1), dim3(TINY_THREADS), tiny_lds(), stream, t);
        return 0;
    }

    // One sweep = tiny step + row pass (CrossQr: two of each).  Kernel arguments never change between sweeps -- the sweep / trace-row counters and the
    // "first sweep of this call" flag live in device memory (Ctl) -- so EVERY sweep of a run is the same launch sequence: blocks of 32 / 16 / 4 / 2
    // sweeps are captured once into hipGraphs and replayed, at most one sweep per run is enqueued singly.  That removes the per-launch host overhead
    // and the idle gaps between singly launched kernels (a 20-sweep erm_run: two graph launches instead of one sweep + four graphs + three sweeps).
#ifndef ERM_GRAPH_SWEEPS
#define ERM_GRAPH_SWEEPS 32
#endif
    static constexpr int GRAPH_SWEEPS = ERM_GRAPH_SWEEPS;
    static constexpr int PROFILE_STRIDE = 8;
    static constexpr int NGRAPH = 4;
    // (every count is even: a fused sweep flips the double buffers, and a graph must be replayed with the buffer parity it was captured with -- every
    // run starts from buffer 0 and replays its graphs BEFORE its one single sweep)
    const int graph_sweeps[NGRAPH] = {GRAPH_SWEEPS, 16, 4, 2};
    hipGraphExec_t graphs[NGRAPH] = {nullptr, nullptr, nullptr, nullptr};
    // A WHOLE call as graphs (single-pass and Cross samplers on the per-sweep schedule, statistics resident): calls of up to GRAPH_SWEEPS sweeps are ONE graph --
    // run_begin_kernel, the sweeps, the closing tiny step and run_end_kernel (full[k]) --, longer ones end in tail[r] = r sweeps + tiny step + run_end_kernel behind
    // their blocks of GRAPH_SWEEPS.  Between a graph and an ordinary launch
    // the device idles 10-14 us (measured: tools/run_timeline.py), inside a graph 0: a 20-sweep call 1 435 -> 1 400 us of device time.  Built by the first call of each length.
    hipGraphExec_t graphs_full[GRAPH_SWEEPS + 1] = {}, graphs_tail[GRAPH_SWEEPS + 1] = {};
    void drop_graphs() {
        for (auto& g : graphs) { if (g) (void)hipGraphExecDestroy(g); g = nullptr; }
        for (auto* arr : {graphs_full, graphs_tail}) for (int k = 0; k <= GRAPH_SWEEPS; ++k) { if (arr[k]) (void)hipGraphExecDestroy(arr[k]); arr[k] = nullptr; }
    }
    bool ev_calibrated = false; double ev_null_ms = 0.0;
    template <int MODEL> int enqueue_sweep(bool timed) {
        if constexpr (!fam_cq(MODEL)) { if (fused()) return launch_fused<MODEL>(timed); }
        if (int rc = launch_tiny<MODEL, 0>(0)) return rc;
        if (int rc = launch_pass<MODEL, 0>(1, timed)) return rc;
        if constexpr (fam_cq(MODEL)) {
            if (int rc = launch_tiny<MODEL, 1>(0)) return rc;
            if (int rc = launch_pass<MODEL, 1>(1, timed)) return rc;
        }
        return 0;
    }
    void launch_run_begin() {
        hipLaunchKernelGGL(run_begin_kernel, dim3(1), dim3(256), 0, stream, dCtlB[0].template as<Ctl>(), dCtlB[1].template as<Ctl>(), host_run_dev, dGcnt.as<unsigned int>(), 2 * n_groups + 4);
    }
    void launch_run_end() {
        hipLaunchKernelGGL(run_end_kernel, dim3(1), dim3(64), 0, stream, dCtlB[0].template as<Ctl>(), dCtlB[1].template as<Ctl>(), dGcnt.as<unsigned int>() + 2 * n_groups,
                           host_ctl_dev + 1, reinterpret_cast<unsigned int*>(host_ctl_dev + 3));
    }
    // nsw sweeps; `begin`: run_begin_kernel first; `end`: the closing tiny step and run_end_kernel last.
    // The capture starts at buffer parity 0 (every call does) and -- a graph is only ever replayed at that parity -- restores it afterwards.
    template <int MODEL> int build_graph(int nsw, hipGraphExec_t* out, bool begin = false, bool end = false) {
        hipGraph_t g = nullptr;
        const int cur0 = cur;                        // a fused sweep flips the double buffers while it is being captured: a failed capture -- and a whole-call
        HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));       // graph, whose sweeps the caller does not count -- must not leave the parity changed
        int rc = 0;
        if (begin) launch_run_begin();
        for (int k = 0; k < nsw && !rc; ++k) rc = enqueue_sweep<MODEL>(false);
        if (end && !rc) { rc = launch_tiny<MODEL, 0>(1); launch_run_end(); }
        const hipError_t e = hipStreamEndCapture(stream, &g);      // always ends the capture, also after a failed enqueue
        if (!rc && e != hipSuccess) rc = fail(ERM_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
        if (!rc) {
            const hipError_t ei = hipGraphInstantiate(out, g, nullptr, nullptr, 0);
            if (ei != hipSuccess) { *out = nullptr; rc = fail(ERM_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(ei)); }
        }
        if (g) (void)hipGraphDestroy(g);             // on every exit
        if (rc || end) cur = cur0;                   // (a whole-call graph: run_model moves the parity when it replays it)
        return rc;
    }
    // whole: 0 = the caller launches run_begin_kernel / run_end_kernel around this; 1 = the call is ONE graph (built by the caller: graphs_full[_ev][nsweeps]);
    // 2 = run_begin_kernel is enqueued, the call's last graph (graphs_tail) carries the closing tiny step and run_end_kernel
    template <int MODEL> int run_model(int64_t nsweeps, int whole = 0) {
        // prologue: omega_{t+1} (and nu_{t+1}) and the statistics of the current state.  A run that CONTINUES the previous one finds both
        // in place -- the last pass drew omega_{t+1} from the same addressed streams and left the same statistics -- and skips it.
        if (!stats_valid) {
            if (int rc = launch_pass<MODEL, 0>(0, false)) return rc;
            if constexpr (fam_cq(MODEL)) { if (int rc = launch_pass<MODEL, 1>(0, false)) return rc; }
        }
        int64_t k = 0;
        if constexpr (!fam_cq(MODEL)) {
            if (persist && !sharded() && nsweeps > 0) {
                // blocks of 2^20 sweeps (packet tags are 32 bits and only grow)
                for (int64_t done = 0; done < nsweeps; ) {
                    const int64_t nb = std::min<int64_t>(nsweeps - done, 1 << 20);
                    if (int rc = launch_persist<MODEL>(nb)) return rc;
                    done += nb;
                }
                if (int rc = launch_tiny<MODEL, 0>(1)) return rc;
                return 0;
            }
        }
        // (a callback exchange synchronises with the host once per pass and cannot be captured; RCCL's all-gather is a stream operation)
        const bool use_graph = exch == nullptr && (cfg.flags & ERM_FLAG_NO_GRAPH) == 0;
        // profile mode (erm_get_timing: the live kernel time of bench.py's roofline).  Single-pass models: every replayed graph holds launches of the sweep
        // kernel and nothing else, so the event pairs go around (up to four consecutive) graph launches and every sweep of the run is inside a bracket while
        // the run still proceeds at graph-replay speed.  The Cross family's sweeps hold tiny kernels too, a sharded sweep its pack kernel and all-gather:
        // those are bracketed launch by launch on singly enqueued sweeps -- all sweeps of a short run, two (the first one timed; two keep the buffer
        // parity) before every replayed block of a long one.
        const bool graph_timing = cfg.profile && !fam_cq(MODEL) && fused() && !sharded();
        const bool single_timing = cfg.profile && !graph_timing;
        const bool flips = !fam_cq(MODEL) && fused();      // a fused sweep flips the double buffers (launch_fused); the two-kernel schedules do not
        if (whole == 1) {
            // (profile mode: the bracket's events go around the graph on the stream -- event records captured INTO a graph do not time its kernels on this runtime:
            // tried, 13.9 us per sweep -- so it also holds the three small kernels, ~12 us per call; as two launches, run_begin_kernel and a tail graph, it held the
            // 10.7 us gap between them instead of the 3.5 us kernel)
            const bool ev = graph_timing && (size_t)(2 * n_brackets + 1) + 64 < pass_ev.size();
            if (ev) HIPCHK(hipEventRecord(pass_ev[2 * n_brackets], stream));
            HIPCHK(hipGraphLaunch(graphs_full[nsweeps], stream));
            if (ev) { HIPCHK(hipEventRecord(pass_ev[2 * n_brackets + 1], stream)); ++n_brackets; n_pass_timed += nsweeps; bracket_launches.push_back((int)nsweeps); }
            if (flips) cur = (int)((cur + nsweeps) & 1);
            return 0;
        }
        if (whole == 2) {
            const int r = (nsweeps % GRAPH_SWEEPS) == 0 ? GRAPH_SWEEPS : (int)(nsweeps % GRAPH_SWEEPS);
            if (nsweeps > r && !graphs[0]) { if (int rc = build_graph<MODEL>(GRAPH_SWEEPS, &graphs[0])) return rc; }
            if (!graphs_tail[r]) { if (int rc = build_graph<MODEL>(r, &graphs_tail[r], false, true)) return rc; }
            int in_bracket = 0, launches = 0;
            bool open = false;
            auto close = [&]() -> int {
                if (!open) return 0;
                HIPCHK(hipEventRecord(pass_ev[2 * n_brackets + 1], stream));
                ++n_brackets; n_pass_timed += launches; bracket_launches.push_back(launches);
                open = false; in_bracket = 0; launches = 0;
                return 0;
            };
            auto replay = [&](hipGraphExec_t g, int nsw) -> int {
                if (graph_timing && !open && (size_t)(2 * n_brackets + 1) + 64 < pass_ev.size()) { HIPCHK(hipEventRecord(pass_ev[2 * n_brackets], stream)); open = true; }
                HIPCHK(hipGraphLaunch(g, stream));
                k += nsw; launches += nsw; ++in_bracket;
                if (in_bracket >= 4) return close();
                return 0;
            };
            while (nsweeps - k > r) { if (int rc = replay(graphs[0], GRAPH_SWEEPS)) return rc; }
            if (int rc = replay(graphs_tail[r], r)) return rc;      // (its bracket also holds the closing tiny step and run_end_kernel: ~8 us once per call)
            if (flips) cur = (int)((cur + r) & 1);
            return close();
        }
        if (use_graph && !single_timing) {
            bool open = false;
            int in_bracket = 0, launches = 0;
            auto close = [&]() -> int {
                if (!open) return 0;
                HIPCHK(hipEventRecord(pass_ev[2 * n_brackets + 1], stream));
                ++n_brackets; n_pass_timed += launches; bracket_launches.push_back(launches);
                open = false; in_bracket = 0; launches = 0;
                return 0;
            };
            for (int gi = 0; gi < NGRAPH; ++gi) {
                const int nsw = graph_sweeps[gi];
                while (nsweeps - k >= nsw) {
                    if (!graphs[gi]) {       // built by the first run that needs it (a benchmark's warm-up), never inside an event bracket
                        if (int rc = close()) return rc;
                        if (int rc = build_graph<MODEL>(nsw, &graphs[gi])) return rc;
                    }
                    if (graph_timing && !open && (size_t)(2 * n_brackets + 1) + 64 < pass_ev.size()) { HIPCHK(hipEventRecord(pass_ev[2 * n_brackets], stream)); open = true; }
                    HIPCHK(hipGraphLaunch(graphs[gi], stream));
                    k += nsw; launches += nsw; ++in_bracket;
                    if (in_bracket >= 4) { if (int rc = close()) return rc; }
                }
            }
            if (int rc = close()) return rc;
        } else if (use_graph && nsweeps >= 2 * (GRAPH_SWEEPS + 2)) {
            for (; nsweeps - k >= GRAPH_SWEEPS + 2; k += GRAPH_SWEEPS + 2) {
                if (int rc = enqueue_sweep<MODEL>(true)) return rc;
                if (int rc = enqueue_sweep<MODEL>(false)) return rc;
                if (!graphs[0]) { if (int rc = build_graph<MODEL>(GRAPH_SWEEPS, &graphs[0])) return rc; }
                HIPCHK(hipGraphLaunch(graphs[0], stream));
            }
        }
        const bool short_run = nsweeps < 2 * (GRAPH_SWEEPS + 2);
        for (int64_t r = 0; k < nsweeps; ++k, ++r) { if (int rc = enqueue_sweep<MODEL>(graph_timing || (single_timing && (short_run || (r % PROFILE_STRIDE) == 0)))) return rc; }
        if (int rc = launch_tiny<MODEL, 0>(1)) return rc;      // the log-likelihood of the last sweep (a call without sweeps: nothing to reduce, the step returns at once)
        return 0;
    }

    // ---- the state an erm_run changes in place (persistent launches save it first: a launch that times out is replayed per sweep from the copy).
    // Counters and tickets are re-initialised by every run, trace rows >= rows_done are simply written again, and the host's own counters move only
    // after a run has succeeded; what is left: theta, zeta, omega, LatentQr's nu, the parameter block and statistics of buffer 0 (every run starts
    // there), and the post-burn-in sums.
    template <typename Fn> void snap_each(Fn&& f) const {
        f(dTheta); if (is_rt()) f(dZeta);
        f(dOmega); if (cfg.model == ERM_MODEL_LATENTQR) f(dNu);
        f(dParB[0]); f(dGslab0B[0]);
        f(dSumTheta); if (is_rt()) f(dSumZeta); if (dSumNu.p) f(dSumNu);
    }
    size_t snap_bytes() const {
        // (called before the buffers exist: sizes from the configuration)
        const size_t NJ = (size_t)N * J, r = sizeof(real);
        size_t b = (size_t)N * r + (is_rt() ? (size_t)N * r : 0) + NJ * r + (cfg.model == ERM_MODEL_LATENTQR ? (size_t)N * r : 0);
        b += (size_t)par_size(J) * 8 + (size_t)std::max(n_groups, GROUP) * ns[0] * 8;
        b += (size_t)N * 8 + (is_rt() ? (size_t)N * 8 : 0) + (cfg.model == ERM_MODEL_LATENTQR ? (size_t)N * 8 : 0);
        return b + 10 * 256;                         // every segment starts on a 256-byte boundary
    }
    int snap_copy(bool restore) {
        CopySegs S{};
        size_t off = 0;
        bool ok = true;
        snap_each([&](const DevBuf& d) {
            if (S.n >= 10 || off + d.bytes > dSnap.bytes) { ok = false; return; }
            char* sp = dSnap.as<char>() + off;
            S.src[S.n] = restore ? (const void*)sp : (const void*)d.p; S.dst[S.n] = restore ? d.p : (void*)sp; S.bytes[S.n] = d.bytes; ++S.n;
            off = (off + d.bytes + 255) & ~(size_t)255;
        });
        if (!ok) return fail(ERM_ERR_STATE, "internal: the persistent launch's snapshot buffer is too small");
        hipLaunchKernelGGL(copy_segments_kernel, dim3(256), dim3(256), 0, stream, S);
        HIPCHK(hipGetLastError());
        return 0;
    }

    // A run that fails after it has started enqueueing leaves the double buffers, the counters and the traces in an unknown state: the
    // stream is drained, the buffer parity reset, and the engine refuses to continue until the caller installs a state again.
    bool poisoned = false;
    bool stats_valid = false;                        // omega_{t+1} (nu_{t+1}) and the statistics of the CURRENT state are resident (set by a completed run)
    Ctl* host_ctl = nullptr;                         // pinned: [1], [2] the two device copies of the counters, [3] the time-out word (run_end_kernel stores them)
    Ctl* host_ctl_dev = nullptr;                     // the same memory as the device addresses it
    RunParams* host_run = nullptr;                   // pinned: the parameters of the erm_run being enqueued (run_begin_kernel reads them)
    RunParams* host_run_dev = nullptr;
    bool has_stats_state() const { return host_ctl != nullptr; }
    int run(int64_t nsweeps) override {
        if (!has_data) return fail(ERM_ERR_STATE, "erm_set_data has not been called");
        if (poisoned) return fail(ERM_ERR_STATE, "a previous erm_run failed part-way: call erm_set_state (and erm_reset_trace) before running again");
        if (nsweeps < 0) return fail(ERM_ERR_ARG, "nsweeps must be non-negative");
        if (rows_done + nsweeps > rows_cap) return fail(ERM_ERR_ARG, "trace capacity exceeded: n_iter*n_chain rows were allocated");
        int rc = run_checked(nsweeps);
        if (rc == ERM_PERSIST_TIMEOUT) {
            // the persistent launch never had all its workgroups resident (another process holds compute units): every workgroup has left the launch;
            // put back what the call found, leave the persistent schedule for good and run the call again, one launch per sweep at the same geometry
            // (bit for bit the chain the persistent launch would have produced)
            persist = false; timing.persistent = 0; ++timing.persist_fallbacks;
            rc = snap_copy(true);
            if (rc == 0) { cur = 0; stats_valid = snap_stats_valid; rc = run_checked(nsweeps); }
            if (rc == ERM_PERSIST_TIMEOUT) rc = fail(ERM_ERR_STATE, "internal: persistent time-out reported by a per-sweep run");
        }
        if (rc != 0 && rc != ERM_ERR_NONFINITE) {
            const std::string msg = g_err;            // keep the first error's message
            (void)hipStreamSynchronize(stream);
            (void)hipGetLastError();
            cur = 0;
            poisoned = true;
            stats_valid = false;
            g_err = msg;
        }
        return rc;
    }
    static constexpr int ERM_PERSIST_TIMEOUT = -1000;      // internal: run_checked -> run
    int run_checked(int64_t nsweeps) {
        HIPCHK(hipSetDevice(cfg.device));
        if (sharded() || !has_stats_state()) stats_valid = false;
        Ctl c{};
        c.sweep = sweeps_total; c.row = (uint32_t)rows_done; c.burn_rows = (uint32_t)((int64_t)cfg.n_burnin * cfg.n_chain); c.err = 0;
        c.first = 1u;                                // no sweep of this call has been drawn yet: the first one writes trace row rows_done itself
        if (cur == 1) {      // every run starts from buffer 0 so that a captured graph always replays with the buffer parity it was built with
            HIPCHK(hipMemcpyAsync(dParB[0].p, dParB[1].p, dParB[0].bytes, hipMemcpyDeviceToDevice, stream));
            if (stats_valid) HIPCHK(hipMemcpyAsync(dGslab0B[0].p, dGslab0B[1].p, dGslab0B[0].bytes, hipMemcpyDeviceToDevice, stream));
            cur = 0;
        }
        const bool persistent_run = persist && !sharded() && nsweeps > 0 && !m_cq();
        // counters, tickets, the persistent launch's wait bound (1 s of the 100 MHz wall clock; 2 ms under the test hook) in ONE small launch, its parameters in pinned memory
        const bool fault = persistent_run && persist_fault_countdown > 0 && --persist_fault_countdown == 0;
        host_run->v = c; host_run->tmo_ticks = fault ? 200000u : 100000000u; host_run->tmo_fault = fault ? 1u : 0u;
        volatile unsigned int* h_tmo = reinterpret_cast<volatile unsigned int*>(&host_ctl[3]);
        *h_tmo = 0u;
        n_pass_timed = 0; n_brackets = 0; bracket_launches.clear();
        const bool calibrate = cfg.profile && pass_ev.size() >= 64 && !ev_calibrated;
        // the whole call as graphs (see graphs_full): per-sweep schedule, statistics resident, nothing that has to be bracketed launch by launch
        const bool graph_timing = cfg.profile && !m_cq() && fused() && !sharded();
        int whole = 0;
        if (exch == nullptr && (cfg.flags & ERM_FLAG_NO_GRAPH) == 0 && !sharded() && !persistent_run && stats_valid && !calibrate && nsweeps >= 1 && !(cfg.profile && !graph_timing))
            whole = nsweeps <= GRAPH_SWEEPS ? 1 : 2;
        if (whole == 1 && !graphs_full[nsweeps]) {
            if (int rcb = dispatch([&](auto m) -> int { return build_graph<decltype(m)::value>((int)nsweeps, &graphs_full[nsweeps], true, true); })) return rcb;
        }
        if (whole != 1) { launch_run_begin(); HIPCHK(hipGetLastError()); }
        if (persistent_run) { snap_stats_valid = stats_valid; if (int rc = snap_copy(false)) return rc; }
        if (calibrate) {   // empty event pairs, once per engine: the bracketing overhead that is subtracted from every timed launch
            for (int k = 0; k < 16; ++k) { HIPCHK(hipEventRecord(pass_ev[pass_ev.size() - 2 - 2 * k], stream)); HIPCHK(hipEventRecord(pass_ev[pass_ev.size() - 1 - 2 * k], stream)); }
        }
        // persistent launches of one process take turns on a device (see g_persist_mu); held until the stream has drained
        std::unique_lock<std::mutex> turn;
        if (persistent_run) turn = std::unique_lock<std::mutex>(g_persist_mu[(unsigned)cfg.device % 64u]);
        // (a one-graph call in profile mode: its bracket IS the call -- two event records ahead of the graph's launch instead of one keep the device idle 4 us longer)
        const bool own_events = !(whole == 1 && graph_timing && pass_ev.size() >= 66);
        if (own_events) HIPCHK(hipEventRecord(ev0, stream));
        if (int rc = dispatch([&](auto m) -> int { return run_model<decltype(m)::value>(nsweeps, whole); })) return rc;
        if (own_events) HIPCHK(hipEventRecord(ev1, stream));
        HIPCHK(hipGetLastError());
        // the counters of both buffers and the time-out word, stored into pinned host memory by one small launch (no copy operations; inside the call's last graph when it has one)
        if (!whole) { launch_run_end(); HIPCHK(hipGetLastError()); }
        HIPCHK(hipStreamSynchronize(stream));
        if (turn.owns_lock()) turn.unlock();
        if (*h_tmo != 0u) {
            g_err = "the persistent sweep kernel timed out waiting for another workgroup's statistics (its workgroups were not all resident)";
            return persistent_run ? ERM_PERSIST_TIMEOUT : fail(ERM_ERR_STATE, "internal: " + g_err);
        }
        float ms = 0.f;
        if (own_events) HIPCHK(hipEventElapsedTime(&ms, ev0, ev1)); else HIPCHK(hipEventElapsedTime(&ms, pass_ev[0], pass_ev[1]));
        timing.run_ms = ms; timing.sweeps = nsweeps; timing.pass_ms_total = 0.0; timing.pass_launches = n_pass_timed;
        if (calibrate) {
            double t16 = 0.0;
            for (int k = 0; k < 16; ++k) { float t = 0.f; HIPCHK(hipEventElapsedTime(&t, pass_ev[pass_ev.size() - 2 - 2 * k], pass_ev[pass_ev.size() - 1 - 2 * k])); t16 += t; }
            ev_null_ms = t16 / 16.0; ev_calibrated = true;
        }
        const double null_ms = ev_null_ms;
        timing.event_overhead_ms = null_ms;
        for (int64_t k = 0; k < n_brackets; ++k) {   // a bracket holds 1 launch, or the TAIL_SWEEPS launches of one replayed graph
            float t = 0.f;
            HIPCHK(hipEventElapsedTime(&t, pass_ev[2 * k], pass_ev[2 * k + 1]));
            timing.pass_ms_total += std::max(0.0, (double)t - null_ms);
        }
        Ctl back = host_ctl[1 + cur];
        back.err = host_ctl[1].err;                  // the sticky flag lives in buffer 0
        {   // the diagnostic counters (ERM_PASS_STOP=9 of a diagnostic build) accumulate in whichever buffer a launch read: add the other one
            const Ctl& b1 = host_ctl[1 + (1 - cur)];
            back.dbg_attempts += b1.dbg_attempts; back.dbg_trips += b1.dbg_trips; back.dbg_cells += b1.dbg_cells;
        }
        const int64_t burn = (int64_t)cfg.n_burnin * cfg.n_chain;
        const int64_t lo = std::max<int64_t>(rows_done, burn), hi = rows_done + nsweeps;
        if (hi > lo) post_rows += hi - lo;
        rows_done += nsweeps;
        sweeps_total += (uint32_t)nsweeps;
        if (diag_stop("ERM_PASS_STOP") == 9)
            fprintf(stderr, "[erm dbg] attempts %llu cells %llu wave-trips %llu -> attempts/cell %.4f, lane efficiency %.4f\n", back.dbg_attempts, back.dbg_cells, back.dbg_trips,
                    (double)back.dbg_attempts / (double)back.dbg_cells, (double)back.dbg_attempts / (64.0 * (double)back.dbg_trips));
        if (dDbgTs.p) {      // per-wave phase timeline of the LAST launch (us since the workgroup's first stamp)
            std::vector<unsigned long long> ts(2 * 16 * 16);
            HIPCHK(hipMemcpy(ts.data(), dDbgTs.p, ts.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            static const char* names[16] = {"start", "B1", "B3", "struct", "sums", "ready", "draws", "PG", "barrier", "gstats", "phase2", "slab", "ticket", "end", "hd-str0", "hd-item"};
            for (int b = 0; b < 2; ++b) {
                unsigned long long t0 = ~0ull;
                for (int w = 0; w < 16; ++w) if (ts[(b * 16 + w) * 16] && ts[(b * 16 + w) * 16] < t0) t0 = ts[(b * 16 + w) * 16];
                fprintf(stderr, "[erm timeline] workgroup %s\n  wave", b == 0 ? "0" : "grid/2");
                for (int k = 0; k < 16; ++k) fprintf(stderr, " %7s", names[k]);
                fprintf(stderr, "\n");
                for (int w = 0; w < 16; ++w) {
                    fprintf(stderr, "  %4d", w);
                    for (int k = 0; k < 16; ++k) { const unsigned long long v = ts[(b * 16 + w) * 16 + k]; fprintf(stderr, " %7.2f", v ? (double)(v - t0) * 0.01 : -1.0); }
                    fprintf(stderr, "\n");
                }
            }
        }
        stats_valid = back.err == 0;
        if (back.err) {
            const int e = (int)back.err - 1;
            const char* names[] = {"a", "b", "lambda", "sig2t", "rho"};
            std::string what = e < 5 * J ? std::string(names[e / J]) + "[" + std::to_string(e % J) + "]"
                             : (e < 5 * J + 4 ? "Sigp[" + std::to_string(e - 5 * J) + "]" : "beta[" + std::to_string(e - 5 * J - 4) + "]");
            return fail(ERM_ERR_NONFINITE, "non-finite parameter " + what + " at sweep " + std::to_string(back.sweep));
        }
        return 0;
    }

    int reset_trace() override {
        rows_done = 0; post_rows = 0;
        // on the engine's own stream: ordered after the run that filled the sums and before the next one (a NULL-stream fill is ordered with neither)
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipMemsetAsync(dSumTheta.p, 0, dSumTheta.bytes, stream));
        HIPCHK(hipMemsetAsync(dSumZeta.p, 0, dSumZeta.bytes, stream));
        if (dSumNu.p) HIPCHK(hipMemsetAsync(dSumNu.p, 0, dSumNu.bytes, stream));
        return 0;
    }

    // -------------------------------------------------------------------------------------------- data
    // InputData (src/Base.pl.jl:67-78) -> the resident data set.  The column-major host arrays are uploaded as they are and turned into the
    // engine's row-major buffers ON the device (transposition, column sums, centring of logT by its column means, centred sums of squares);
    // the host only adds up per-workgroup partial sums in a fixed order and forms x'x (N p^2 flops).
    int set_data(const uint8_t* Y, const double* logT, const double* X) override {
        if (!Y) return fail(ERM_ERR_ARG, "Y is NULL");
        if (is_rt() && !logT) return fail(ERM_ERR_ARG, "logT is required for response-time models");
        if (Fk > 0 && !X) return fail(ERM_ERR_ARG, "X is required when n_feat > 0");
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamSynchronize(stream));
        // the resident buffers are overwritten from here on: until this call succeeds the engine holds NO data set (a rejected Y / logT must
        // not leave the previous set's flags -- and its statistics -- standing over the new bytes)
        has_data = false; stats_valid = false;
        const size_t NJ = (size_t)N * J;
        const double Ntot = sharded() ? (double)n_total : (double)N;      // a shard's column sums are completed over the devices
        const int pp = p();
        std::vector<double> cst(cst_size(J), 0.0);
        std::vector<double> g1((size_t)2 * J + PMAX * PMAX, 0.0);         // K0 | column sums of logT | x'x : summed over the devices
        DevBuf dYc, dLc, dStat, dFlag;
        if (int rc = dYc.alloc(NJ)) return rc;
        H2D(dYc.p, Y, NJ);
        if (is_rt()) { if (int rc = dLc.alloc(NJ * sizeof(double))) return rc; H2D(dLc.p, logT, NJ * sizeof(double)); }
        if (int rc = dStat.alloc((size_t)2 * J * sizeof(double))) return rc;
        if (int rc = dFlag.alloc(sizeof(unsigned int))) return rc;
        hipLaunchKernelGGL(colstats_cm_kernel, dim3(J), dim3(256), 0, stream, dYc.as<uint8_t>(), dLc.as<double>(), (long long)N, is_rt() ? 1 : 0, dStat.as<double>(), J, dFlag.as<unsigned int>());
        const dim3 tgrid((unsigned)((N + 31) / 32), (unsigned)((J + 31) / 32));
        hipLaunchKernelGGL((to_rows_kernel<uint8_t, uint8_t>), tgrid, dim3(256), 0, stream, dYc.as<uint8_t>(), (long long)N, J, (const double*)nullptr, dY.as<uint8_t>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        {
            unsigned int flags = 0;
            HIPCHK(hipMemcpy(&flags, dFlag.p, sizeof(flags), hipMemcpyDeviceToHost));
            if (flags & 1u) return fail(ERM_ERR_ARG, "Y must be 0/1");
            if (flags & 2u) return fail(ERM_ERR_ARG, "logT must be finite");
            HIPCHK(hipMemcpy(g1.data(), dStat.p, (size_t)2 * J * sizeof(double), hipMemcpyDeviceToHost));      // K0 | column sums of logT
        }
        for (int u = 0; u < pp; ++u) for (int v = u; v < pp; ++v) {     // x'x with x = [1 X]
            double t = 0.0;
            if (u == 0 && v == 0) t = (double)N;
            else if (u == 0) { const double* xv = X + (size_t)(v - 1) * N; for (int64_t i = 0; i < N; ++i) t += xv[i]; }
            else { const double* xu = X + (size_t)(u - 1) * N; const double* xv = X + (size_t)(v - 1) * N; for (int64_t i = 0; i < N; ++i) t += xu[i] * xv[i]; }
            g1[(size_t)2 * J + u + v * PMAX] = t; g1[(size_t)2 * J + v + u * PMAX] = t;
        }
        if (int rc = shard_allsum(g1)) return rc;
        for (int j = 0; j < J; ++j) cst[cst_off_k0(J) + j] = g1[j];
        if (is_rt()) {
            // column-centred logT; mean/std(Data.logT) are the kwargs of drawItemIntensity (src/Draw.pl.jl:215)
            std::vector<double> mean(J), g2(J, 0.0);
            double tot = 0.0;
            for (int j = 0; j < J; ++j) { tot += g1[(size_t)J + j]; mean[j] = g1[(size_t)J + j] / Ntot; cst[cst_off_m(J) + j] = mean[j]; }
            DevBuf dMean, dPart;
            const int NB = 256;
            if (int rc = dMean.alloc((size_t)J * sizeof(double))) return rc;
            if (int rc = dPart.alloc((size_t)NB * J * sizeof(double))) return rc;
            H2D(dMean.p, mean.data(), (size_t)J * sizeof(double));
            hipLaunchKernelGGL((to_rows_kernel<double, real>), tgrid, dim3(256), 0, stream, dLc.as<double>(), (long long)N, J, dMean.as<double>(), dC.as<real>());
            hipLaunchKernelGGL((colsq_kernel<real>), dim3(NB), dim3(128), 0, stream, dC.as<real>(), (long long)N, J, dPart.as<double>());
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));
            std::vector<double> part((size_t)NB * J);
            HIPCHK(hipMemcpy(part.data(), dPart.p, part.size() * sizeof(double), hipMemcpyDeviceToHost));
            for (int j = 0; j < J; ++j) { double sq = 0.0; for (int b = 0; b < NB; ++b) sq += part[(size_t)b * J + j]; g2[j] = sq; }
            if (int rc = shard_allsum(g2)) return rc;
            for (int j = 0; j < J; ++j) cst[cst_off_csq(J) + j] = g2[j];
            const double mu = tot / (Ntot * (double)J);
            double ss = 0.0;
            for (int j = 0; j < J; ++j) { const double dm = cst[cst_off_m(J) + j] - mu; ss += cst[cst_off_csq(J) + j] + Ntot * dm * dm; }
            cst[cst_off_mu(J)] = mu; cst[cst_off_mu(J) + 1] = std::sqrt(ss / (Ntot * (double)J - 1.0));
        }
        {   // X -> row-major
            for (int u = 0; u < pp; ++u) for (int v = 0; v < pp; ++v) cst[cst_off_xtx(J) + u + v * PMAX] = g1[(size_t)2 * J + u + v * PMAX];
            if (int rc = invert_xtx(cst)) return rc;
            if (Fk > 0) {
                DevBuf dXc;
                if (int rc = dXc.alloc((size_t)N * Fk * sizeof(double))) return rc;
                H2D(dXc.p, X, (size_t)N * Fk * sizeof(double));
                hipLaunchKernelGGL((to_rows_kernel<double, real>), dim3((unsigned)((N + 31) / 32), (unsigned)((Fk + 31) / 32)), dim3(256), 0, stream, dXc.as<double>(), (long long)N, Fk,
                                   (const double*)nullptr, dX.as<real>());
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(stream));
            }
        }
        H2D(dCst.p, cst.data(), cst.size() * sizeof(double));
        has_data = true;
        stats_valid = false;
        return 0;
    }

    // (x'x)^-1 by Gauss-Jordan with partial pivoting (fp64, p <= PMAX); x'x is in cst at cst_off_xtx with leading dimension PMAX
    int invert_xtx(std::vector<double>& cst) {
        const int pp = p();
        std::vector<double> M((size_t)pp * 2 * pp, 0.0);
        for (int i = 0; i < pp; ++i) { for (int jx = 0; jx < pp; ++jx) M[(size_t)i * 2 * pp + jx] = cst[cst_off_xtx(J) + i + jx * PMAX]; M[(size_t)i * 2 * pp + pp + i] = 1.0; }
        for (int c = 0; c < pp; ++c) {
            int piv = c;
            for (int r = c + 1; r < pp; ++r) if (std::fabs(M[(size_t)r * 2 * pp + c]) > std::fabs(M[(size_t)piv * 2 * pp + c])) piv = r;
            if (!(std::fabs(M[(size_t)piv * 2 * pp + c]) > 0.0)) return fail(ERM_ERR_ARG, "x'x is singular (collinear covariates)");
            if (piv != c) for (int jx = 0; jx < 2 * pp; ++jx) std::swap(M[(size_t)c * 2 * pp + jx], M[(size_t)piv * 2 * pp + jx]);
            const double d = M[(size_t)c * 2 * pp + c];
            for (int jx = 0; jx < 2 * pp; ++jx) M[(size_t)c * 2 * pp + jx] /= d;
            for (int r = 0; r < pp; ++r) if (r != c) {
                const double f = M[(size_t)r * 2 * pp + c];
                if (f != 0.0) for (int jx = 0; jx < 2 * pp; ++jx) M[(size_t)r * 2 * pp + jx] -= f * M[(size_t)c * 2 * pp + jx];
            }
        }
        for (int i = 0; i < pp; ++i) for (int jx = 0; jx < pp; ++jx) cst[cst_off_xinv(J) + i + jx * PMAX] = M[(size_t)i * 2 * pp + pp + jx];
        return 0;
    }

    // -------------------------------------------------------------------------------------------- synthetic data on the device
    DevBuf dTruthTheta, dTruthZeta;
    std::vector<double> col_mean;      // column means of logT (kept for erm_get_data)
    int simulate_data(const erm_state* tr, uint64_t seed, int noise) override {
        if (sharded()) return fail(ERM_ERR_STATE, "erm_simulate_data is not available on a shard");
        if (!tr || !tr->a || !tr->b) return fail(ERM_ERR_ARG, "truth needs a and b");
        if (is_rt() && (!tr->lambda || !tr->sig2t)) return fail(ERM_ERR_ARG, "truth needs lambda and sig2t for response-time models");
        if (noise < 0 || noise > 2) return fail(ERM_ERR_ARG, "noise must be 0 (norm), 1 (tail) or 2 (skew)");
        int gen;
        switch (cfg.model) {
        case ERM_MODEL_MLIRT: gen = 0; break;
        case ERM_MODEL_RTIRT: gen = 1; break;
        case ERM_MODEL_NULL: gen = 2; break;
        case ERM_MODEL_CROSS: case ERM_MODEL_CROSSQR: gen = 3; break;
        default: gen = 4;
        }
        if ((gen == 0 || gen == 1 || gen == 4) && Fk > 0 && !tr->beta) return fail(ERM_ERR_ARG, "truth needs beta");
        if (gen == 3 && !tr->rho) return fail(ERM_ERR_ARG, "truth needs rho");
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamSynchronize(stream));
        has_data = false; stats_valid = false;
        // truth vector: a b lambda sig2t rho | chol(Sigp) | beta
        std::vector<double> tv((size_t)5 * J + 3 + 2 * PMAX + 2, 0.0);
        for (int j = 0; j < J; ++j) {
            tv[j] = tr->a[j]; tv[J + j] = tr->b[j];
            tv[2 * J + j] = tr->lambda ? tr->lambda[j] : 0.0; tv[3 * J + j] = tr->sig2t ? tr->sig2t[j] : 1.0; tv[4 * J + j] = tr->rho ? tr->rho[j] : 0.0;
            if (is_rt() && !(tv[3 * J + j] > 0.0)) return fail(ERM_ERR_ARG, "sig2t must be positive");
        }
        double S00 = 1.0, S10 = 0.0, S11 = 1.0;
        if (tr->sigp) { S00 = tr->sigp[0]; S10 = tr->sigp[1]; S11 = tr->sigp[3]; }
        if (!(S00 > 0.0) || !(S11 - S10 * S10 / S00 > 0.0)) return fail(ERM_ERR_ARG, "Sigp must be positive definite");
        tv[5 * J] = std::sqrt(S00); tv[5 * J + 1] = S10 / tv[5 * J]; tv[5 * J + 2] = std::sqrt(S11 - tv[5 * J + 1] * tv[5 * J + 1]);
        // beta: the generators' truth has no intercept row (src/SimTools.jl:86,107,283): RtIrt [nFeat][2] column-major -> (f, c) at 2f + c
        double* bt = &tv[5 * J + 3];
        if (tr->beta) {
            if (gen == 1) for (int f = 0; f < Fk; ++f) { bt[2 * f] = tr->beta[f]; bt[2 * f + 1] = tr->beta[Fk + f]; }
            else if (gen == 0) for (int f = 0; f < Fk; ++f) bt[f] = tr->beta[f];
            else if (gen == 4) for (int f = 0; f <= Fk; ++f) bt[f] = tr->beta[f];
        }
        DevBuf dT;
        if (int rc = dT.alloc(tv.size() * sizeof(double))) return rc;
        H2D(dT.p, tv.data(), tv.size() * sizeof(double));
        if (!dTruthTheta.p) { if (int rc = dTruthTheta.alloc((size_t)N * sizeof(double))) return rc; if (int rc = dTruthZeta.alloc((size_t)N * sizeof(double))) return rc; }
        GenArgs g{};
        g.Y = dY.as<uint8_t>(); g.C = dC.p; g.X = dX.p; g.theta = dTruthTheta.as<double>(); g.zeta = dTruthZeta.as<double>();
        g.truth = dT.as<double>(); g.N = N; g.J = J; g.F = Fk; g.gen = gen; g.noise = noise; g.seed = seed;
        hipLaunchKernelGGL((gen_kernel<real>), dim3((unsigned)((N + 127) / 128)), dim3(128), 0, stream, g);
        // ---- constants of the data set (the same as erm_set_data's): K0, column means / centred squares of logT, x'x and its inverse
        const int NB = 256, pp = p(), PW = 3 * J + pp * pp;
        DevBuf dPart, dMean, dPart2;
        if (int rc = dPart.alloc((size_t)NB * PW * sizeof(double))) return rc;
        hipLaunchKernelGGL((colsum_kernel<real>), dim3(NB), dim3(128), 0, stream, dY.as<uint8_t>(), dC.as<real>(), dX.as<real>(), (long long)N, J, Fk,
                           is_rt() ? 1 : 0, dPart.as<double>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<double> part((size_t)NB * PW), cst(cst_size(J), 0.0);
        HIPCHK(hipMemcpy(part.data(), dPart.p, part.size() * sizeof(double), hipMemcpyDeviceToHost));
        col_mean.assign(J, 0.0);
        for (int j = 0; j < J; ++j) {
            double sk = 0.0, s1 = 0.0;
            for (int b = 0; b < NB; ++b) { sk += part[(size_t)b * PW + j]; s1 += part[(size_t)b * PW + J + j]; }
            cst[cst_off_k0(J) + j] = sk; col_mean[j] = s1 / (double)N; cst[cst_off_m(J) + j] = col_mean[j];
        }
        for (int e = 0; e < pp * pp; ++e) { double t = 0.0; for (int b = 0; b < NB; ++b) t += part[(size_t)b * PW + 3 * J + e]; cst[cst_off_xtx(J) + (e % pp) + (e / pp) * PMAX] = t; }
        if (int rc = invert_xtx(cst)) return rc;
        if (is_rt()) {
            if (int rc = dMean.alloc((size_t)J * sizeof(double))) return rc;
            if (int rc = dPart2.alloc((size_t)NB * J * sizeof(double))) return rc;
            H2D(dMean.p, col_mean.data(), (size_t)J * sizeof(double));
            hipLaunchKernelGGL((center_kernel<real>), dim3(NB), dim3(128), 0, stream, dC.as<real>(), (long long)N, J, dMean.as<double>(), dPart2.as<double>());
            HIPCHK(hipGetLastError());
            HIPCHK(hipStreamSynchronize(stream));
            std::vector<double> p2((size_t)NB * J);
            HIPCHK(hipMemcpy(p2.data(), dPart2.p, p2.size() * sizeof(double), hipMemcpyDeviceToHost));
            double tot = 0.0;
            for (int j = 0; j < J; ++j) { double sq = 0.0; for (int b = 0; b < NB; ++b) sq += p2[(size_t)b * J + j]; cst[cst_off_csq(J) + j] = sq; tot += col_mean[j]; }
            const double mu = tot / (double)J;              // equal column lengths: the grand mean is the mean of the column means
            double ss = 0.0;
            for (int j = 0; j < J; ++j) { const double dm = col_mean[j] - mu; ss += cst[cst_off_csq(J) + j] + (double)N * dm * dm; }
            cst[cst_off_mu(J)] = mu; cst[cst_off_mu(J) + 1] = std::sqrt(ss / ((double)N * J - 1.0));
        }
        H2D(dCst.p, cst.data(), cst.size() * sizeof(double));
        has_data = true;
        stats_valid = false;
        return 0;
    }
    // the resident data set back in the caller's (column-major) layout: Y bytes, logT = centred value + column mean, X
    int get_data(uint8_t* Y, double* logT, double* X) override {
        if (!has_data) return fail(ERM_ERR_STATE, "no data set is resident");
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamSynchronize(stream));
        const size_t NJ = (size_t)N * J;
        if (Y) { std::vector<uint8_t> t(NJ); HIPCHK(hipMemcpy(t.data(), dY.p, NJ, hipMemcpyDeviceToHost)); for (int j = 0; j < J; ++j) for (int64_t i = 0; i < N; ++i) Y[(size_t)j * N + i] = t[(size_t)i * J + j]; }
        if (logT && is_rt()) {
            std::vector<double> cst(cst_size(J));
            HIPCHK(hipMemcpy(cst.data(), dCst.p, cst.size() * sizeof(double), hipMemcpyDeviceToHost));
            std::vector<real> t(NJ);
            HIPCHK(hipMemcpy(t.data(), dC.p, NJ * sizeof(real), hipMemcpyDeviceToHost));
            for (int j = 0; j < J; ++j) for (int64_t i = 0; i < N; ++i) logT[(size_t)j * N + i] = (double)t[(size_t)i * J + j] + cst[cst_off_m(J) + j];
        }
        if (X && Fk > 0) { std::vector<real> t((size_t)N * Fk); HIPCHK(hipMemcpy(t.data(), dX.p, t.size() * sizeof(real), hipMemcpyDeviceToHost)); for (int f = 0; f < Fk; ++f) for (int64_t i = 0; i < N; ++i) X[(size_t)f * N + i] = (double)t[(size_t)i * Fk + f]; }
        return 0;
    }
    int get_truth(double* theta, double* zeta) override {
        if (!dTruthTheta.p) return fail(ERM_ERR_STATE, "erm_simulate_data has not been called");
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamSynchronize(stream));
        if (theta) HIPCHK(hipMemcpy(theta, dTruthTheta.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
        if (zeta) HIPCHK(hipMemcpy(zeta, dTruthZeta.p, (size_t)N * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    }

    int nbeta() const {
        switch (cfg.model) {
        case ERM_MODEL_MLIRT: return F + 1;
        case ERM_MODEL_RTIRT: return 2 * (F + 1);
        case ERM_MODEL_LATENTQR: case ERM_MODEL_LATENT: return F + 2;
        case ERM_MODEL_NULL: return 2 * (F + 1);          // always zero (src/GibbsRtIrt.pl.jl:380)
        default: return 0;
        }
    }

    int up_real(DevBuf& d, const double* src, size_t n) {
        std::vector<real> t(n);
        for (size_t k = 0; k < n; ++k) t[k] = (real)src[k];
        H2D(d.p, t.data(), n * sizeof(real));
        return 0;
    }
    int down_real(const DevBuf& d, double* dst, size_t n) {
        std::vector<real> t(n);
        HIPCHK(hipMemcpy(t.data(), d.p, n * sizeof(real), hipMemcpyDeviceToHost));
        for (size_t k = 0; k < n; ++k) dst[k] = (double)t[k];
        return 0;
    }

    int set_state(const erm_state* st) override {
        if (!st) return fail(ERM_ERR_ARG, "state is NULL");
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamSynchronize(stream));
        poisoned = false;
        stats_valid = false;
        std::vector<double> par(par_size(J));
        HIPCHK(hipMemcpy(par.data(), dParB[cur].p, par.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (st->a) memcpy(&par[0], st->a, J * sizeof(double));
        if (st->b) memcpy(&par[J], st->b, J * sizeof(double));
        if (st->lambda) memcpy(&par[2 * J], st->lambda, J * sizeof(double));
        if (st->sig2t) { for (int j = 0; j < J; ++j) if (!(st->sig2t[j] > 0.0)) return fail(ERM_ERR_ARG, "sig2t must be positive"); memcpy(&par[3 * J], st->sig2t, J * sizeof(double)); }
        if (st->rho) memcpy(&par[4 * J], st->rho, J * sizeof(double));
        if (st->sigp) memcpy(&par[par_off_sigp(J)], st->sigp, 4 * sizeof(double));
        if (st->beta) {
            double* b = &par[par_off_beta(J)];
            const int pp = F + 1;
            if (cfg.model == ERM_MODEL_RTIRT) { for (int u = 0; u < pp; ++u) { b[u] = st->beta[u]; b[PMAX + u] = st->beta[pp + u]; } }
            else if (cfg.model != ERM_MODEL_NULL) for (int u = 0; u < nbeta(); ++u) b[u] = st->beta[u];
        }
        { double t = 0.0; for (int j = 0; j < J; ++j) t += 1.0 / par[3 * J + j]; par[par_off_derived(J)] = t; }
        H2D(dParB[cur].p, par.data(), par.size() * sizeof(double));
        if (st->theta) if (int rc = up_real(dTheta, st->theta, N)) return rc;
        if (st->zeta) if (int rc = up_real(dZeta, st->zeta, N)) return rc;
        if (st->nu && dNu.p) {
            if (cfg.model == ERM_MODEL_LATENTQR) { if (int rc = up_real(dNu, st->nu, N)) return rc; }
            else {
                std::vector<real> t((size_t)N * J);
                for (int j = 0; j < J; ++j) for (int64_t i = 0; i < N; ++i) {
                    const double v = st->nu[(size_t)j * N + i];
                    if (!(v > 0.0)) return fail(ERM_ERR_ARG, "nu must be positive");   // @assert src/Draw.pl.jl:477
                    t[(size_t)i * J + j] = (real)v;
                }
                H2D(dNu.p, t.data(), t.size() * sizeof(real));
            }
        }
        return 0;
    }

    int get_state(erm_state* st) override {
        if (!st) return fail(ERM_ERR_ARG, "state is NULL");
        if (poisoned) return fail(ERM_ERR_STATE, "a previous erm_run failed part-way: the state is undefined until erm_set_state");
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<double> par(par_size(J));
        HIPCHK(hipMemcpy(par.data(), dParB[cur].p, par.size() * sizeof(double), hipMemcpyDeviceToHost));
        if (st->a) memcpy(st->a, &par[0], J * sizeof(double));
        if (st->b) memcpy(st->b, &par[J], J * sizeof(double));
        if (st->lambda) memcpy(st->lambda, &par[2 * J], J * sizeof(double));
        if (st->sig2t) memcpy(st->sig2t, &par[3 * J], J * sizeof(double));
        if (st->rho) memcpy(st->rho, &par[4 * J], J * sizeof(double));
        if (st->sigp) memcpy(st->sigp, &par[par_off_sigp(J)], 4 * sizeof(double));
        if (st->beta) {
            const double* b = &par[par_off_beta(J)];
            const int pp = F + 1;
            if (cfg.model == ERM_MODEL_RTIRT) { for (int u = 0; u < pp; ++u) { st->beta[u] = b[u]; st->beta[pp + u] = b[PMAX + u]; } }
            else if (cfg.model == ERM_MODEL_NULL) for (int u = 0; u < nbeta(); ++u) st->beta[u] = 0.0;
            else for (int u = 0; u < nbeta(); ++u) st->beta[u] = b[u];
        }
        if (st->theta) if (int rc = down_real(dTheta, st->theta, N)) return rc;
        if (st->zeta) if (int rc = down_real(dZeta, st->zeta, N)) return rc;
        if (st->nu && dNu.p) {
            if (cfg.model == ERM_MODEL_LATENTQR) { if (int rc = down_real(dNu, st->nu, N)) return rc; }
            else {
                std::vector<real> t((size_t)N * J);
                HIPCHK(hipMemcpy(t.data(), dNu.p, t.size() * sizeof(real), hipMemcpyDeviceToHost));
                for (int j = 0; j < J; ++j) for (int64_t i = 0; i < N; ++i) st->nu[(size_t)j * N + i] = (double)t[(size_t)i * J + j];
            }
        }
        return 0;
    }

    // -------------------------------------------------------------------------------------------- traces
    int fetch_item_trace(std::vector<double>& it) {
        it.resize((size_t)std::max<int64_t>(rows_done, 0) * item_trace_width());
        if (!it.empty()) HIPCHK(hipMemcpy(it.data(), dTrItem.p, it.size() * sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    }
    int get_item_trace(double* out) override {
        HIPCHK(hipSetDevice(cfg.device));
        std::vector<double> it;
        if (int rc = fetch_item_trace(it)) return rc;
        if (!it.empty()) memcpy(out, it.data(), it.size() * sizeof(double));
        return 0;
    }

    // Julia layout [nIter][width][nChain], nIter fastest; trace row r = m*nChain + l
    int get_trace(int which, double* out) override {
        HIPCHK(hipSetDevice(cfg.device));
        const int64_t nIter = cfg.n_iter, nChain = cfg.n_chain, wd = trace_width(which);
        if (wd <= 0) return fail(ERM_ERR_ARG, "this model has no such trace");
        if (rows_done != rows_cap) return fail(ERM_ERR_STATE, "trace incomplete: run n_iter*n_chain sweeps first");
        auto at = [&](int64_t r, int64_t k) -> double& { const int64_t m = r / nChain, l = r % nChain; return out[m + nIter * (k + wd * l)]; };
        if (which == ERM_TRACE_LOGLIKE) {
            std::vector<double> ll(rows_cap);
            HIPCHK(hipMemcpy(ll.data(), dTrLl.p, rows_cap * sizeof(double), hipMemcpyDeviceToHost));
            for (int64_t r = 0; r < rows_cap; ++r) at(r, 0) = ll[r];
            return 0;
        }
        if (cfg.trace_mode != ERM_TRACE_FULL) return fail(ERM_ERR_NOTRACE, "subject-level traces need trace_mode = ERM_TRACE_FULL");
        if (which == ERM_TRACE_QR && cfg.model == ERM_MODEL_CROSSQR && !dTrNu.p)
            return fail(ERM_ERR_NOTRACE, "the per-sweep nu trace (N*J values per sweep) exceeds erm_config.nu_trace_max_gb; use erm_get_item_trace (rho, Sigp) + erm_get_mean (nu)");
        std::vector<double> it;
        if (int rc = fetch_item_trace(it)) return rc;
        const int64_t wi = item_trace_width();
        // subject block of a trace: transposed to Julia's iteration-fastest layout ON the device (one chain at a time into a scratch buffer of
        // nSubj x nIter doubles), then ONE contiguous copy per chain; if the scratch buffer cannot be had, row by row through the host
        auto subj = [&](const DevBuf& d, int64_t k0) -> int {
            double* tmp = nullptr;
            const size_t need = (size_t)N * (size_t)nIter * sizeof(double);
            if (hipMalloc(reinterpret_cast<void**>(&tmp), need) == hipSuccess) {
                int rc = 0;
                for (int64_t l = 0; l < nChain && !rc; ++l) {
                    hipLaunchKernelGGL((trace_transpose_kernel<real>), dim3((unsigned)((N + 31) / 32), (unsigned)((nIter + 31) / 32)), dim3(256), 0, stream,
                                       d.as<real>(), (long long)N, (long long)N, (int)nIter, (int)nChain, (int)l, tmp);
                    hipError_t e = hipGetLastError();
                    if (e == hipSuccess) e = hipStreamSynchronize(stream);
                    if (e == hipSuccess) e = hipMemcpy(out + nIter * (k0 + wd * l), tmp, need, hipMemcpyDeviceToHost);
                    if (e != hipSuccess) rc = fail(ERM_ERR_HIP, std::string("trace transpose: ") + hipGetErrorString(e));
                }
                (void)hipFree(tmp);
                return rc;
            }
            (void)hipGetLastError();
            std::vector<real> rowbuf(N);
            for (int64_t r = 0; r < rows_cap; ++r) {
                HIPCHK(hipMemcpy(rowbuf.data(), d.as<real>() + (size_t)r * N, N * sizeof(real), hipMemcpyDeviceToHost));
                for (int64_t i = 0; i < N; ++i) at(r, k0 + i) = (double)rowbuf[i];
            }
            return 0;
        };
        if (which == ERM_TRACE_RA) {            // [theta; a; b]  src/GibbsRtIrt.pl.jl:242,320
            if (int rc = subj(dTrTheta, 0)) return rc;
            for (int64_t r = 0; r < rows_cap; ++r) for (int j = 0; j < J; ++j) { at(r, N + j) = it[r * wi + j]; at(r, N + J + j) = it[r * wi + J + j]; }
        } else if (which == ERM_TRACE_RT) {     // [zeta; lambda; sig2t]  :321
            if (int rc = subj(dTrZeta, 0)) return rc;
            for (int64_t r = 0; r < rows_cap; ++r) for (int j = 0; j < J; ++j) { at(r, N + j) = it[r * wi + 2 * J + j]; at(r, N + J + j) = it[r * wi + 3 * J + j]; }
        } else {                                 // qr  :241,319 ; Latent :308
            const int q = nq();
            if (cfg.model == ERM_MODEL_NULL) {          // [vec(beta) = 0 (2(nFeat+1)); vec(Sigp)]  src/GibbsRtIrt.pl.jl:398
                const int nb = 2 * ((int)F + 1);
                for (int64_t r = 0; r < rows_cap; ++r) { for (int k = 0; k < nb; ++k) at(r, k) = 0.0; for (int k = 0; k < 4; ++k) at(r, nb + k) = it[r * wi + 4 * J + 2 + k]; }
                return 0;
            }
            for (int64_t r = 0; r < rows_cap; ++r) for (int k = 0; k < q; ++k) at(r, k) = it[r * wi + 4 * J + k];
            if (cfg.model == ERM_MODEL_LATENTQR) if (int rc = subj(dTrNu, q)) return rc;
            if (cfg.model == ERM_MODEL_CROSSQR) {       // vec(nu): column-major N x J after [rho; vec(Sigp)]
                std::vector<real> cells((size_t)N * J);
                for (int64_t r = 0; r < rows_cap; ++r) {
                    HIPCHK(hipMemcpy(cells.data(), dTrNu.as<real>() + (size_t)r * N * J, cells.size() * sizeof(real), hipMemcpyDeviceToHost));
                    for (int j = 0; j < J; ++j) for (int64_t i = 0; i < N; ++i) at(r, q + i + N * j) = (double)cells[(size_t)i * J + j];
                }
            }
        }
        return 0;
    }

    // ess / rhat of every column of Post.ra / rt / qr, computed on the device from the resident traces (diag_kernel)
    // ess / rhat of every column of trace `which` into device arrays of trace_width(which) doubles.  GibbsRtIrtCrossQr's vec(nu) block of Post.qr is
    // column-major N x J in Julia's layout and row-major on the device: it is diagnosed in DEVICE order (entry q + i * J + j), *nu_block says so.
    int diag_device(int which, DevBuf& dE, DevBuf& dR, bool* nu_block) {
        *nu_block = false;
        HIPCHK(hipSetDevice(cfg.device));
        const int64_t wd = trace_width(which);
        if (which == ERM_TRACE_LOGLIKE || wd <= 0) return fail(ERM_ERR_ARG, "diagnostics exist for the ra / rt / qr traces");
        if (rows_done != rows_cap) return fail(ERM_ERR_STATE, "trace incomplete: run n_iter*n_chain sweeps first");
        if (cfg.trace_mode != ERM_TRACE_FULL) return fail(ERM_ERR_NOTRACE, "subject-level traces need trace_mode = ERM_TRACE_FULL");
        const int Tn = cfg.n_iter - cfg.n_burnin;
        if (Tn / 2 < 4) return fail(ERM_ERR_ARG, "too few post-burn-in iterations for split-chain diagnostics (need >= 8)");
        if (2 * cfg.n_chain > DIAG_MAXSEQ) return fail(ERM_ERR_ARG, "too many chains for the diagnostics kernel");
        HIPCHK(hipStreamSynchronize(stream));
        if (int rc = dE.alloc((size_t)wd * sizeof(double))) return rc;
        if (int rc = dR.alloc((size_t)wd * sizeof(double))) return rc;
        auto launch_real = [&](const DevBuf& tr, int64_t ncol, int64_t off) -> int {
            if (!tr.p) return fail(ERM_ERR_NOTRACE, "this trace was not recorded");
            hipLaunchKernelGGL((diag_kernel<real>), dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, stream, tr.as<real>(), (long long)ncol, (long long)ncol,
                               cfg.n_iter, cfg.n_chain, cfg.n_burnin, dE.as<double>() + off, dR.as<double>() + off);
            return 0;
        };
        auto launch_item = [&](int64_t col0, int64_t ncol, int64_t off) -> int {
            hipLaunchKernelGGL((diag_kernel<double>), dim3((unsigned)((ncol + 255) / 256)), dim3(256), 0, stream, dTrItem.as<double>() + col0, (long long)ncol,
                               (long long)item_trace_width(), cfg.n_iter, cfg.n_chain, cfg.n_burnin, dE.as<double>() + off, dR.as<double>() + off);
            return 0;
        };
        const int q = nq();
        if (which == ERM_TRACE_RA) {            // [theta; a; b]
            if (int rc = launch_real(dTrTheta, N, 0)) return rc;
            if (int rc = launch_item(0, 2 * J, N)) return rc;
        } else if (which == ERM_TRACE_RT) {     // [zeta; lambda; sig2t]
            if (int rc = launch_real(dTrZeta, N, 0)) return rc;
            if (int rc = launch_item(2 * J, 2 * J, N)) return rc;
        } else if (cfg.model == ERM_MODEL_NULL) {   // [vec(beta) = 0; vec(Sigp)]: the zeros are constant -> NaN
            const int nb = 2 * ((int)F + 1);
            std::vector<double> nanv(nb, std::nan(""));
            HIPCHK(hipMemcpyAsync(dE.p, nanv.data(), nb * sizeof(double), hipMemcpyHostToDevice, stream));
            HIPCHK(hipMemcpyAsync(dR.p, nanv.data(), nb * sizeof(double), hipMemcpyHostToDevice, stream));
            HIPCHK(hipStreamSynchronize(stream));
            if (int rc = launch_item(4 * J + 2, 4, nb)) return rc;
        } else {
            if (int rc = launch_item(4 * J, q, 0)) return rc;
            if (cfg.model == ERM_MODEL_LATENTQR) { if (int rc = launch_real(dTrNu, N, q)) return rc; }
            if (cfg.model == ERM_MODEL_CROSSQR) {
                if (!dTrNu.p) return fail(ERM_ERR_NOTRACE, "the per-sweep nu trace was not recorded (erm_config.nu_trace_max_gb)");
                if (int rc = launch_real(dTrNu, (int64_t)N * J, q)) return rc;
                *nu_block = true;
            }
        }
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }
    // ess / rhat of every column of Post.ra / rt / qr, computed on the device from the resident traces (diag_kernel)
    int get_diagnostics(int which, double* ess, double* rhat) override {
        DevBuf dE, dR;
        bool nu_block = false;
        if (int rc = diag_device(which, dE, dR, &nu_block)) return rc;
        const int64_t wd = trace_width(which);
        HIPCHK(hipMemcpy(ess, dE.p, (size_t)wd * sizeof(double), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(rhat, dR.p, (size_t)wd * sizeof(double), hipMemcpyDeviceToHost));
        if (nu_block) {      // device order (row-major N x J) -> Julia's vec(nu) (column-major)
            const int q = nq();
            std::vector<double> he(ess + q, ess + wd), hr(rhat + q, rhat + wd);
            for (int j = 0; j < J; ++j) for (int64_t i = 0; i < N; ++i) { ess[q + i + N * j] = he[(size_t)i * J + j]; rhat[q + i + N * j] = hr[(size_t)i * J + j]; }
        }
        return 0;
    }
    // checkConvergence's summary (src/SimTools.jl:419-443) without the N-wide vectors: counts of the columns with a defined ESS / R-hat and of those with
    // ESS > 400 / R-hat < 1.1 (the reference's thresholds), counted on the device
    int get_convergence(int which, int64_t* c4) override {
        DevBuf dE, dR, dC;
        bool nu_block = false;
        if (int rc = diag_device(which, dE, dR, &nu_block)) return rc;
        if (int rc = dC.alloc(4 * sizeof(unsigned long long))) return rc;
        const int64_t wd = trace_width(which);
        hipLaunchKernelGGL(diag_count_kernel, dim3((unsigned)std::min<int64_t>((wd + 255) / 256, 1024)), dim3(256), 0, stream, dE.as<double>(), dR.as<double>(), (long long)wd, 400.0, 1.1,
                           dC.as<unsigned long long>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        unsigned long long h[4];
        HIPCHK(hipMemcpy(h, dC.p, sizeof(h), hipMemcpyDeviceToHost));
        for (int k = 0; k < 4; ++k) c4[k] = (int64_t)h[k];
        return 0;
    }

    // sums over the post-burn-in rows of the item-level trace columns (a, b, lambda, sig2t, small part of qr)
    int item_sums(std::vector<double>& m) {
        std::vector<double> it;
        if (int rc = fetch_item_trace(it)) return rc;
        const int64_t wi = item_trace_width(), burn = (int64_t)cfg.n_burnin * cfg.n_chain;
        m.assign(wi, 0.0);
        for (int64_t r = burn; r < rows_done; ++r) for (int64_t k = 0; k < wi; ++k) m[k] += it[r * wi + k];
        return 0;
    }
    // item-level means -> the fields of Post.mean (src/GibbsRtIrt.pl.jl:249-254, 327-343; Cross :304-318; Latent :316-330)
    void unpack_items(const double* m, erm_state* out) const {
        if (out->a) memcpy(out->a, &m[0], J * sizeof(double));
        if (out->b) memcpy(out->b, &m[J], J * sizeof(double));
        if (out->lambda) memcpy(out->lambda, &m[2 * J], J * sizeof(double));
        if (out->sig2t) memcpy(out->sig2t, &m[3 * J], J * sizeof(double));
        const double* q = &m[4 * J];
        switch (cfg.model) {
        case ERM_MODEL_MLIRT: if (out->beta) memcpy(out->beta, q, (F + 1) * sizeof(double)); break;
        case ERM_MODEL_RTIRT: if (out->beta) memcpy(out->beta, q, 2 * (F + 1) * sizeof(double)); if (out->sigp) memcpy(out->sigp, q + 2 * (F + 1), 4 * sizeof(double)); break;
        case ERM_MODEL_CROSSQR: case ERM_MODEL_CROSS: if (out->rho) memcpy(out->rho, q, J * sizeof(double)); if (out->sigp) memcpy(out->sigp, q + J, 4 * sizeof(double)); break;
        case ERM_MODEL_NULL: if (out->beta) memset(out->beta, 0, 2 * (F + 1) * sizeof(double)); if (out->sigp) memcpy(out->sigp, q + 2, 4 * sizeof(double)); break;
        default: if (out->beta) memcpy(out->beta, q, (F + 2) * sizeof(double)); if (out->sigp) memcpy(out->sigp, q + F + 2, 4 * sizeof(double));
        }
    }
    int get_mean(erm_state* out) override {
        if (!out) return fail(ERM_ERR_ARG, "state is NULL");
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamSynchronize(stream));
        if (post_rows <= 0) return fail(ERM_ERR_STATE, "no post-burn-in sweeps recorded");
        const double inv = 1.0 / (double)post_rows;
        auto subj = [&](const DevBuf& d, double* dst, size_t n, bool transpose) -> int {
            std::vector<double> t(n);
            HIPCHK(hipMemcpy(t.data(), d.p, n * sizeof(double), hipMemcpyDeviceToHost));
            if (!transpose) for (size_t k = 0; k < n; ++k) dst[k] = t[k] * inv;
            else for (int j = 0; j < J; ++j) for (int64_t i = 0; i < N; ++i) dst[(size_t)j * N + i] = t[(size_t)i * J + j] * inv;
            return 0;
        };
        if (out->theta) if (int rc = subj(dSumTheta, out->theta, N, false)) return rc;
        if (out->zeta && is_rt()) if (int rc = subj(dSumZeta, out->zeta, N, false)) return rc;
        if (out->nu && dSumNu.p) {
            if (cfg.model == ERM_MODEL_LATENTQR) { if (int rc = subj(dSumNu, out->nu, N, false)) return rc; }
            else if (int rc = subj(dSumNu, out->nu, (size_t)N * J, true)) return rc;
        }
        std::vector<double> m;
        if (int rc = item_sums(m)) return rc;
        for (auto& v : m) v *= inv;
        unpack_items(m.data(), out);
        return 0;
    }

    // ---- chain farms: [item sums (wi) | sum theta (N) | sum zeta (N, response-time models) | sum nu (N or N*J, quantile models)]
    int64_t nu_len() const { return cfg.model == ERM_MODEL_LATENTQR ? N : (cfg.model == ERM_MODEL_CROSSQR ? N * (int64_t)J : 0); }
    int64_t summary_len() const override { return item_trace_width() + N + (is_rt() ? N : 0) + nu_len(); }
    int summary_add(double* acc) override {
        HIPCHK(hipSetDevice(cfg.device));
        auto add = [&](double* dst, const double* src, int64_t n) {
            hipLaunchKernelGGL(acc_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, dst, src, (long long)n);
        };
        // item-level columns: summed over the post-burn-in rows of the resident item trace ON the device, in row order (the order erm_get_mean's host sum uses)
        const int64_t wi = item_trace_width(), burn = (int64_t)cfg.n_burnin * cfg.n_chain;
        hipLaunchKernelGGL(item_sum_kernel, dim3((unsigned)((wi + 255) / 256)), dim3(256), 0, stream, dTrItem.as<double>(), (long long)wi, (long long)std::min(burn, rows_done), (long long)rows_done, acc);
        int64_t o = wi;
        add(acc + o, dSumTheta.as<double>(), N); o += N;
        if (is_rt()) { add(acc + o, dSumZeta.as<double>(), N); o += N; }
        if (nu_len() > 0) add(acc + o, dSumNu.as<double>(), nu_len());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        return 0;
    }

    // ---- DIC from device-resident state (include/ertirt.h, erm_get_dic)
    int loglik_at(const double* dsum, double inv, double* ll) override {
        if (!has_data) return fail(ERM_ERR_STATE, "no data set is resident");
        HIPCHK(hipSetDevice(cfg.device));
        const int nb = (int)std::min<int64_t>(1024, (N + 63) / 64);
        DevBuf dPart;
        if (int rc = dPart.alloc((size_t)nb * sizeof(double))) return rc;
        LogLikArgs D{};
        D.Y = dY.as<uint8_t>(); D.C = dC.p; D.X = dX.p; D.cm = dCst.as<double>() + cst_off_m(J);
        D.sum = dsum; D.inv = inv; D.N = N; D.J = J; D.F = Fk; D.model = cfg.model;
        const int64_t wi = item_trace_width();
        D.off_theta = wi; D.off_zeta = is_rt() ? wi + N : -1; D.off_nu = nu_len() > 0 ? wi + N + (is_rt() ? N : 0) : -1;
        const double q = cfg.q_rt;
        D.k1 = m_nu() ? (1.0 - 2.0 * q) / (q * (1.0 - q)) : 0.0; D.k2 = m_nu() ? 2.0 / (q * (1.0 - q)) : 1.0;
        D.rows_per_block = (N + nb - 1) / nb; D.part = dPart.as<double>();
        const size_t lds = ((size_t)5 * J + 4 + 2 * PMAX) * sizeof(double);
        hipLaunchKernelGGL((loglik_kernel<real>), dim3(nb), dim3(256), lds, stream, D);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        std::vector<double> part(nb);
        HIPCHK(hipMemcpy(part.data(), dPart.p, part.size() * sizeof(double), hipMemcpyDeviceToHost));
        double t = 0.0;
        for (double v : part) t += v;
        *ll = t;
        return 0;
    }
    int ll_trace_sum(double* out) override {
        HIPCHK(hipSetDevice(cfg.device));
        DevBuf d;
        if (int rc = d.alloc(sizeof(double))) return rc;
        hipLaunchKernelGGL(ll_trace_sum_kernel, dim3(1), dim3(256), 0, stream, dTrLl.as<double>(), (long long)rows_done, d.as<double>());
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(stream));
        HIPCHK(hipMemcpy(out, d.p, sizeof(double), hipMemcpyDeviceToHost));
        return 0;
    }
    int get_dic(double* out4) override {
        if (poisoned) return fail(ERM_ERR_STATE, "a previous erm_run failed part-way: the state is undefined until erm_set_state");
        if (sharded()) return fail(ERM_ERR_STATE, "erm_get_dic is not available on a shard (the log-likelihood at Post.mean needs every subject)");
        if (rows_done <= 0 || post_rows <= 0) return fail(ERM_ERR_STATE, "no post-burn-in sweeps recorded");
        HIPCHK(hipSetDevice(cfg.device));
        DevBuf acc;
        if (int rc = acc.alloc((size_t)summary_len() * sizeof(double))) return rc;
        if (int rc = summary_add(acc.as<double>())) return rc;
        double ll_hat = 0.0, ll_sum = 0.0;
        if (int rc = loglik_at(acc.as<double>(), 1.0 / (double)post_rows, &ll_hat)) return rc;
        if (int rc = ll_trace_sum(&ll_sum)) return rc;
        const double Dhat = -2.0 * ll_hat, Dbar = -2.0 * ll_sum / (double)rows_done;
        out4[0] = Dbar; out4[1] = Dhat; out4[2] = Dbar - Dhat; out4[3] = Dbar + (Dbar - Dhat);
        return 0;
    }
    int set_seed(uint64_t seed) override {
        HIPCHK(hipSetDevice(cfg.device));
        HIPCHK(hipStreamSynchronize(stream));
        cfg.seed = seed;
        sweeps_total = 0;            // a new seed starts a new chain: its streams are addressed from sweep 1, as a freshly created engine's are
        drop_graphs();               // the captured launches carry the seed as a kernel argument
        stats_valid = false;         // the resident omega_{t+1} / nu_{t+1} were drawn from the old streams: the next run draws them again
        return 0;
    }
    int summary_unpack(const double* mean, erm_state* out) const override {
        unpack_items(mean, out);
        int64_t o = item_trace_width();
        if (out->theta) memcpy(out->theta, mean + o, N * sizeof(double));
        o += N;
        if (is_rt()) { if (out->zeta) memcpy(out->zeta, mean + o, N * sizeof(double)); o += N; }
        if (out->nu && nu_len() > 0) {
            if (cfg.model == ERM_MODEL_LATENTQR) memcpy(out->nu, mean + o, N * sizeof(double));
            else for (int j = 0; j < J; ++j) for (int64_t i = 0; i < N; ++i) out->nu[(size_t)j * N + i] = mean[o + (size_t)i * J + j];     // device row-major -> column-major
        }
        return 0;
    }
};

}  // namespace

struct erm_engine { std::unique_ptr<EngineBase> e; };

// Chain farm (first bullet; north_star): nChain INDEPENDENT chains, chain l on device devices[l] with random stream
// chain_id = l and its own copy of the data; one host thread per chain drives its engine's stream; no communication while sampling.
// Post.mean (src/GibbsRtIrt.pl.jl:327-343: mean over iterations AND chains jointly) = (sum over chains of their post-burn-in sums) /
// (total rows): the sums of the chains that share a device are added on that device, the devices' vectors are summed by ONE
// ncclAllReduce over RCCL (xGMI), and device 0's copy is divided by the count.
struct erm_farm {
    erm_config cfg{};
    std::vector<std::unique_ptr<erm_engine>> eng;      // eng[l]: chain l
    std::vector<int> dev;                                // dev[l]: its device
    std::vector<int> udev;                               // distinct devices, in order of first use
    std::vector<ncclComm_t> comms;                       // one communicator rank per distinct device (empty until the first reduction over > 1 device)
    std::vector<hipStream_t> rstream;                    // one stream per distinct device for the reduction (never the NULL stream: the engines' streams are non-blocking)
    bool used_rccl = false;
    erm_farm_timing tm{};
    ~erm_farm() {
        stop_workers();
        for (auto c : comms) if (c) (void)g_rccl.CommDestroy(c);
        for (size_t d = 0; d < rstream.size(); ++d) if (rstream[d]) { (void)hipSetDevice(udev[d]); (void)hipStreamDestroy(rstream[d]); }
    }
    // One PERSISTENT host thread per chain (created with the farm, joined by its destructor): erm_farm_run / set_data / get_trace hand each chain's
    // call to its thread and wait.  (Creating the threads per call cost ~25 us per chain -- an eighth of a 20-sweep erm_farm_run on eight GPUs.)
    struct Worker {
        std::thread th;
        std::mutex mu;
        std::condition_variable cv;
        std::function<int()> job;
        bool has_job = false, done = false, quit = false;
        int rc = 0;
        std::string msg;
    };
    std::vector<std::unique_ptr<Worker>> workers;
    void start_workers() {
        for (size_t l = workers.size(); l < eng.size(); ++l) {
            workers.emplace_back(new Worker());
            Worker* w = workers.back().get();
            w->th = std::thread([w] {
                std::unique_lock<std::mutex> lk(w->mu);
                for (;;) {
                    w->cv.wait(lk, [w] { return w->has_job || w->quit; });
                    if (w->quit) return;
                    w->has_job = false;
                    lk.unlock();
                    const int rc = w->job();
                    const std::string m = rc ? g_err : std::string();      // g_err is thread-local: the chain's message lives in THIS thread
                    lk.lock();
                    w->rc = rc; w->msg = m; w->done = true;
                    w->cv.notify_all();
                }
            });
        }
    }
    void stop_workers() {
        for (auto& w : workers) {
            { std::lock_guard<std::mutex> lk(w->mu); w->quit = true; }
            w->cv.notify_all();
            if (w->th.joinable()) w->th.join();
        }
        workers.clear();
    }
    // runs f(l) for every chain on the chain's thread; returns the first non-zero code (its message becomes the calling thread's last error)
    template <typename Fn> int parallel(Fn&& f) {
        const int n = (int)eng.size();
        start_workers();
        for (int l = 0; l < n; ++l) {
            Worker* w = workers[l].get();
            { std::lock_guard<std::mutex> lk(w->mu); w->job = [&f, l] { return f(l); }; w->done = false; w->has_job = true; }
            w->cv.notify_all();
        }
        int first = -1;
        for (int l = 0; l < n; ++l) {
            Worker* w = workers[l].get();
            std::unique_lock<std::mutex> lk(w->mu);
            w->cv.wait(lk, [w] { return w->done; });
            if (w->rc && first < 0) first = l;
        }
        if (first >= 0) return fail(workers[first]->rc, "chain " + std::to_string(first) + ": " + workers[first]->msg);
        return 0;
    }
};

extern "C" {

int erm_create(const erm_config* cfg, erm_handle* out)
{
    if (!cfg || !out) return fail(ERM_ERR_ARG, "cfg/out is NULL");
    *out = nullptr;
    std::unique_ptr<EngineBase> e;
    if (cfg->precision == ERM_PREC_F32) e.reset(new Engine<float>());
    else if (cfg->precision == ERM_PREC_F64) e.reset(new Engine<double>());
    else return fail(ERM_ERR_ARG, "unknown precision");
    e->cfg = *cfg;
    if (int rc = e->init()) return rc;
    *out = new erm_engine{std::move(e)};
    return ERM_OK;
}
void erm_destroy(erm_handle h) { delete h; }
#define CHK_H if (!h) return fail(ERM_ERR_ARG, "handle is NULL")
int erm_set_data(erm_handle h, const uint8_t* Y, const double* logT, const double* X) { CHK_H; return h->e->set_data(Y, logT, X); }
int erm_set_state(erm_handle h, const erm_state* st) { CHK_H; return h->e->set_state(st); }
int erm_get_state(erm_handle h, erm_state* st) { CHK_H; return h->e->get_state(st); }
int erm_run(erm_handle h, int64_t nsweeps) { CHK_H; return h->e->run(nsweeps); }
int64_t erm_rows_done(erm_handle h) { return h ? h->e->rows_done : -1; }
int erm_reset_trace(erm_handle h) { CHK_H; return h->e->reset_trace(); }
int64_t erm_trace_width(erm_handle h, int which) { return h ? h->e->trace_width(which) : -1; }
int erm_get_trace(erm_handle h, int which, double* out) { CHK_H; if (!out) return fail(ERM_ERR_ARG, "out is NULL"); return h->e->get_trace(which, out); }
int64_t erm_item_trace_width(erm_handle h) { return h ? h->e->item_trace_width() : -1; }
int erm_get_item_trace(erm_handle h, double* out) { CHK_H; if (!out) return fail(ERM_ERR_ARG, "out is NULL"); return h->e->get_item_trace(out); }
int erm_get_mean(erm_handle h, erm_state* out) { CHK_H; return h->e->get_mean(out); }
int64_t erm_post_count(erm_handle h) { return h ? h->e->post_rows : -1; }
int erm_simulate_data(erm_handle h, const erm_state* truth, uint64_t seed, int noise) { CHK_H; return h->e->simulate_data(truth, seed, noise); }
int erm_get_data(erm_handle h, uint8_t* Y, double* logT, double* X) { CHK_H; return h->e->get_data(Y, logT, X); }
int erm_get_truth(erm_handle h, double* theta, double* zeta) { CHK_H; return h->e->get_truth(theta, zeta); }
int erm_get_diagnostics(erm_handle h, int which, double* ess, double* rhat) { CHK_H; if (!ess || !rhat) return fail(ERM_ERR_ARG, "out is NULL"); return h->e->get_diagnostics(which, ess, rhat); }
int erm_get_convergence(erm_handle h, int which, int64_t* counts4) { CHK_H; if (!counts4) return fail(ERM_ERR_ARG, "out is NULL"); return h->e->get_convergence(which, counts4); }
int erm_get_dic(erm_handle h, double* out4) { CHK_H; if (!out4) return fail(ERM_ERR_ARG, "out is NULL"); return h->e->get_dic(out4); }
int erm_set_seed(erm_handle h, uint64_t seed) { CHK_H; return h->e->set_seed(seed); }
int erm_get_timing(erm_handle h, erm_timing* out) { CHK_H; if (!out) return fail(ERM_ERR_ARG, "out is NULL"); *out = h->e->timing; return 0; }
int erm_set_shard(erm_handle h, int rank, int count, int64_t n_subj_total, int64_t row_base, erm_exchange_fn exchange, void* user)
{
    CHK_H;
    if (!exchange) return fail(ERM_ERR_ARG, "exchange callback is NULL");
    return h->e->set_shard(rank, count, n_subj_total, row_base, exchange, user, nullptr);
}
int erm_rccl_unique_id(void* out128)
{
    if (!out128) return fail(ERM_ERR_ARG, "out is NULL");
    if (int rc = g_rccl.load()) return rc;
    ncclUniqueId id;
    RCCLCHK(g_rccl.GetUniqueId(&id));
    std::memcpy(out128, &id, sizeof(id));
    return 0;
}
int erm_set_shard_rccl(erm_handle h, int rank, int count, int64_t n_subj_total, int64_t row_base, const void* unique_id128)
{
    CHK_H;
    if (!unique_id128) return fail(ERM_ERR_ARG, "unique id is NULL");
    return h->e->set_shard(rank, count, n_subj_total, row_base, nullptr, nullptr, unique_id128);
}
int erm_copy(void* dst, const void* src, size_t bytes)
{
    if (bytes == 0) return 0;
    if (!dst || !src) return fail(ERM_ERR_ARG, "dst/src is NULL");
    HIPCHK(hipMemcpy(dst, src, bytes, hipMemcpyDefault));
    return 0;
}

/* ---- chain farm ---- */
int erm_farm_create(const erm_config* cfg, const int32_t* devices, int32_t n_chains, erm_farm_handle* out)
{
    if (!cfg || !out || !devices) return fail(ERM_ERR_ARG, "cfg/devices/out is NULL");
    *out = nullptr;
    if (n_chains < 1 || n_chains > 255) return fail(ERM_ERR_ARG, "n_chains must be in [1, 255]");
    int ndev = 0;
    HIPCHK(hipGetDeviceCount(&ndev));
    std::unique_ptr<erm_farm> f(new erm_farm());
    f->cfg = *cfg;
    for (int l = 0; l < n_chains; ++l) {
        if (devices[l] < 0 || devices[l] >= ndev) return fail(ERM_ERR_ARG, "chain " + std::to_string(l) + ": no such device " + std::to_string(devices[l]));
        erm_config c = *cfg;
        c.device = devices[l]; c.chain_id = l; c.n_chain = 1;      // every chain records n_iter rows of its own
        erm_handle h = nullptr;
        if (int rc = erm_create(&c, &h)) return fail(rc, "chain " + std::to_string(l) + ": " + g_err);
        f->eng.emplace_back(h);
        f->dev.push_back(devices[l]);
        bool seen = false;
        for (int d : f->udev) seen = seen || d == devices[l];
        if (!seen) f->udev.push_back(devices[l]);
    }
    *out = f.release();
    return ERM_OK;
}
void erm_farm_destroy(erm_farm_handle f) { delete f; }
#define CHK_F if (!f) return fail(ERM_ERR_ARG, "farm handle is NULL")
int32_t erm_farm_chains(erm_farm_handle f) { return f ? (int32_t)f->eng.size() : -1; }
erm_handle erm_farm_engine(erm_farm_handle f, int32_t chain) { return (f && chain >= 0 && chain < (int32_t)f->eng.size()) ? f->eng[chain].get() : nullptr; }
int erm_farm_set_data(erm_farm_handle f, const uint8_t* Y, const double* logT, const double* X)
{
    CHK_F;
    return f->parallel([&](int l) { return f->eng[l]->e->set_data(Y, logT, X); });
}
int erm_farm_set_state(erm_farm_handle f, int32_t chain, const erm_state* st)
{
    CHK_F;
    if (chain < 0 || chain >= (int32_t)f->eng.size()) return fail(ERM_ERR_ARG, "no such chain");
    return f->eng[chain]->e->set_state(st);
}
int erm_farm_get_state(erm_farm_handle f, int32_t chain, erm_state* st)
{
    CHK_F;
    if (chain < 0 || chain >= (int32_t)f->eng.size()) return fail(ERM_ERR_ARG, "no such chain");
    return f->eng[chain]->e->get_state(st);
}
int erm_farm_run(erm_farm_handle f, int64_t nsweeps)
{
    CHK_F;
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = f->parallel([&](int l) { return f->eng[l]->e->run(nsweeps); });
    f->tm.run_wall_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}
int erm_farm_get_timing(erm_farm_handle f, erm_farm_timing* out, double* run_ms)
{
    CHK_F;
    if (!out) return fail(ERM_ERR_ARG, "out is NULL");
    *out = f->tm;
    out->n_devices = (int32_t)f->udev.size();
    out->rccl_ranks = 0;
    if (!f->comms.empty() && f->comms[0]) { int n = 0; RCCLCHK(g_rccl.CommCount(f->comms[0], &n)); out->rccl_ranks = n; }
    if (run_ms) for (size_t l = 0; l < f->eng.size(); ++l) run_ms[l] = f->eng[l]->e->timing.run_ms;
    return 0;
}
int erm_farm_reset_trace(erm_farm_handle f)
{
    CHK_F;
    for (auto& e : f->eng) if (int rc = e->e->reset_trace()) return rc;
    return 0;
}
int64_t erm_farm_post_count(erm_farm_handle f)
{
    if (!f) return -1;
    int64_t n = 0;
    for (auto& e : f->eng) n += e->e->post_rows;
    return n;
}
int erm_farm_used_rccl(erm_farm_handle f) { return f ? (f->used_rccl ? 1 : 0) : -1; }
int erm_farm_get_trace(erm_farm_handle f, int which, double* out)
{
    CHK_F;
    if (!out) return fail(ERM_ERR_ARG, "out is NULL");
    // Julia layout [nIter][width][nChain], nIter fastest: chain l's (nIter x width x 1) block is contiguous at offset l * nIter * width
    const int64_t wd = f->eng[0]->e->trace_width(which);
    if (wd <= 0) return fail(ERM_ERR_ARG, "this model has no such trace");
    const int64_t blk = (int64_t)f->cfg.n_iter * wd;
    return f->parallel([&](int l) { return f->eng[l]->e->get_trace(which, out + (size_t)l * blk); });
}
// The farm's one collective: every chain's post-burn-in sums added into its device's accumulator (chains that share a device add in chain order), the
// devices' vectors summed by ONE ncclAllReduce (RCCL over xGMI), each rank on a stream of its own.  Leaves the total on every device (acc[d]) and returns the
// number of post-burn-in rows.  ERM_FLAG_FARM_FORCE_RCCL takes the RCCL path with a one-device communicator too (tests).
static int farm_reduce(erm_farm_handle f, std::vector<DevBuf>& acc, int64_t* total_out, double* init_ms_out)
{
    const int64_t total = erm_farm_post_count(f);
    if (total <= 0) return fail(ERM_ERR_STATE, "no post-burn-in sweeps recorded");
    const int64_t len = f->eng[0]->e->summary_len();
    const int nd = (int)f->udev.size();
    acc.clear(); acc.resize(nd);
    for (int d = 0; d < nd; ++d) {
        HIPCHK(hipSetDevice(f->udev[d]));
        if (int rc = acc[d].alloc((size_t)len * sizeof(double))) return rc;      // zeroed, complete on return
    }
    for (size_t l = 0; l < f->eng.size(); ++l) {
        int d = 0;
        while (f->udev[d] != f->dev[l]) ++d;
        if (int rc = f->eng[l]->e->summary_add(acc[d].as<double>())) return fail(rc, "chain " + std::to_string(l) + ": " + g_err);      // synchronises its stream
    }
    const bool force = (f->cfg.flags & ERM_FLAG_FARM_FORCE_RCCL) != 0;
    f->used_rccl = false;
    f->tm.allreduce_ms = 0.0;
    *init_ms_out = 0.0;                       // communicator creation inside THIS call (first reduction only): reported apart from gather_ms
    if (nd > 1 || force) {
        if (int rc = g_rccl.load()) return rc;
        if (f->rstream.empty()) {
            f->rstream.assign(nd, nullptr);
            for (int d = 0; d < nd; ++d) { HIPCHK(hipSetDevice(f->udev[d])); HIPCHK(hipStreamCreateWithFlags(&f->rstream[d], hipStreamNonBlocking)); }
        }
        if (f->comms.empty()) {
            const auto c0 = std::chrono::steady_clock::now();
            f->comms.assign(nd, nullptr);
            const ncclResult_t r = g_rccl.CommInitAll(f->comms.data(), nd, f->udev.data());
            if (r != ncclSuccess) { f->comms.clear(); return fail(ERM_ERR_STATE, std::string("ncclCommInitAll: ") + g_rccl.GetErrorString(r)); }
            f->tm.comm_init_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - c0).count();
            *init_ms_out = f->tm.comm_init_ms;
        }
        const auto a0 = std::chrono::steady_clock::now();
        // a group that has been started is always ended, whatever happens inside it
        ncclResult_t r = g_rccl.GroupStart();
        if (r != ncclSuccess) return fail(ERM_ERR_STATE, std::string("ncclGroupStart: ") + g_rccl.GetErrorString(r));
        std::string what;
        for (int d = 0; d < nd && r == ncclSuccess; ++d) {
            if (hipSetDevice(f->udev[d]) != hipSuccess) { what = "hipSetDevice"; r = ncclUnhandledCudaError; break; }
            r = g_rccl.AllReduce(acc[d].p, acc[d].p, (size_t)len, ncclDouble, ncclSum, f->comms[d], f->rstream[d]);
            if (r != ncclSuccess) what = "ncclAllReduce";
        }
        const ncclResult_t re = g_rccl.GroupEnd();
        if (r != ncclSuccess) return fail(ERM_ERR_STATE, what + ": " + g_rccl.GetErrorString(r));
        if (re != ncclSuccess) return fail(ERM_ERR_STATE, std::string("ncclGroupEnd: ") + g_rccl.GetErrorString(re));
        for (int d = 0; d < nd; ++d) { HIPCHK(hipSetDevice(f->udev[d])); HIPCHK(hipStreamSynchronize(f->rstream[d])); }
        f->tm.allreduce_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count();
        f->used_rccl = true;
    }
    *total_out = total;
    return 0;
}
int erm_farm_get_mean(erm_farm_handle f, erm_state* out)
{
    CHK_F;
    if (!out) return fail(ERM_ERR_ARG, "state is NULL");
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<DevBuf> acc;
    int64_t total = 0;
    double init_ms = 0.0;
    if (int rc = farm_reduce(f, acc, &total, &init_ms)) return rc;
    const int64_t len = f->eng[0]->e->summary_len();
    std::vector<double> m((size_t)len);
    HIPCHK(hipSetDevice(f->udev[0]));
    HIPCHK(hipMemcpy(m.data(), acc[0].p, (size_t)len * sizeof(double), hipMemcpyDeviceToHost));
    const double inv = 1.0 / (double)total;
    for (auto& v : m) v *= inv;
    const int rc = f->eng[0]->e->summary_unpack(m.data(), out);
    f->tm.gather_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() - init_ms;
    return rc;
}
// getDic of the farm: Dbar over the logLike rows of ALL chains, Dhat at the joint Post.mean -- the vector erm_farm_get_mean reduces, evaluated where it lies
// (the first chain's device and data copy; every chain holds the same data set); only the four numbers cross the boundary
int erm_farm_get_dic(erm_farm_handle f, double* out4)
{
    CHK_F;
    if (!out4) return fail(ERM_ERR_ARG, "out is NULL");
    std::vector<DevBuf> acc;
    int64_t total = 0;
    double init_ms = 0.0;
    if (int rc = farm_reduce(f, acc, &total, &init_ms)) return rc;
    int d0 = 0;
    while (f->udev[d0] != f->dev[0]) ++d0;
    double ll_hat = 0.0;
    if (int rc = f->eng[0]->e->loglik_at(acc[d0].as<double>(), 1.0 / (double)total, &ll_hat)) return rc;
    double ll_sum = 0.0; int64_t rows = 0;
    for (auto& e : f->eng) { double t = 0.0; if (int rc = e->e->ll_trace_sum(&t)) return rc; ll_sum += t; rows += e->e->rows_done; }
    if (rows <= 0) return fail(ERM_ERR_STATE, "no sweeps recorded");
    const double Dhat = -2.0 * ll_hat, Dbar = -2.0 * ll_sum / (double)rows;
    out4[0] = Dbar; out4[1] = Dhat; out4[2] = Dbar - Dhat; out4[3] = Dbar + (Dbar - Dhat);
    return 0;
}
int erm_farm_set_seed(erm_farm_handle f, uint64_t seed)
{
    CHK_F;
    for (auto& e : f->eng) if (int rc = e->e->set_seed(seed)) return rc;
    return 0;
}

const char* erm_last_error(void) { return g_err.c_str(); }
const char* erm_version(void) { return "ertirt-amd 0.4.0 (gfx950)"; }
int erm_abi_version(void) { return ERM_ABI_VERSION; }

int erm_debug_invwishart(int device, uint64_t seed, uint32_t sweep, int64_t n, double nu, const double* psi4, double* out)
{
    if (n <= 0 || !out || !psi4) return fail(ERM_ERR_ARG, "bad n / psi / out");
    if (!(nu > 1.0) || !(psi4[0] > 0.0) || !(psi4[0] * psi4[3] - psi4[1] * psi4[2] > 0.0)) return fail(ERM_ERR_ARG, "InverseWishart(nu, Psi) needs nu > 1 and a positive definite Psi");
    HIPCHK(hipSetDevice(device));
    DevBuf dout;
    if (int rc = dout.alloc((size_t)n * 4 * sizeof(double))) return rc;
    hipLaunchKernelGGL(invwishart_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, seed, sweep, (long long)n, nu, psi4[0], psi4[1], psi4[2], psi4[3], dout.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, (size_t)n * 4 * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int erm_debug_sample(int device, int precision, int which, uint64_t seed, uint32_t site, uint32_t sweep, int64_t n,
                     const double* par0, const double* par1, double* out)
{
    if (n <= 0 || !out) return fail(ERM_ERR_ARG, "bad n/out");
    HIPCHK(hipSetDevice(device));
    DevBuf d0, d1, dout;
    if (int rc = dout.alloc(n * sizeof(double))) return rc;
    if (par0) { if (int rc = d0.alloc(n * sizeof(double))) return rc; H2D(d0.p, par0, n * sizeof(double)); }
    if (par1) { if (int rc = d1.alloc(n * sizeof(double))) return rc; H2D(d1.p, par1, n * sizeof(double)); }
    DevBuf dtab;
    {
        std::vector<double> tab((size_t)PG_NBIN * 4);
        for (int k = 0; k < PG_NBIN; ++k) pg_bin(k, &tab[(size_t)4 * k]);
        if (int rc = dtab.alloc(tab.size() * sizeof(double))) return rc;
        H2D(dtab.p, tab.data(), tab.size() * sizeof(double));
    }
    const int bt = 256; const int gb = (int)((n + bt - 1) / bt);
    if (precision == ERM_PREC_F32)
        hipLaunchKernelGGL((sample_batch_kernel<float>), dim3(gb), dim3(bt), 0, 0, which, seed, site, sweep, (long long)n, d0.as<double>(), d1.as<double>(), dout.as<double>(), dtab.as<double>());
    else
        hipLaunchKernelGGL((sample_batch_kernel<double>), dim3(gb), dim3(bt), 0, 0, which, seed, site, sweep, (long long)n, d0.as<double>(), d1.as<double>(), dout.as<double>(), dtab.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

int erm_sample_gig(int device, uint64_t seed, uint32_t site, uint32_t sweep, int64_t n, double p, double a, double b, double* out)
{
    if (n <= 0 || !out) return fail(ERM_ERR_ARG, "bad n/out");
    if (!(a > 0.0) || !(b > 0.0) || !std::isfinite(p) || p == 0.0) return fail(ERM_ERR_ARG, "GIG(p, a, b) needs a > 0, b > 0 and a finite p != 0");
    HIPCHK(hipSetDevice(device));
    DevBuf dout;
    if (int rc = dout.alloc(n * sizeof(double))) return rc;
    hipLaunchKernelGGL(gig_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, seed, site, sweep, (long long)n, p, a, b, dout.as<double>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out, dout.p, n * sizeof(double), hipMemcpyDeviceToHost));
    return 0;
}

}  // extern "C"
