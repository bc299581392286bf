// erm_geometry.hpp -- launch geometry and LDS layout of the sweep kernels as a PURE host function.
//
// Engine::init (ertirt.hip) calls plan_geometry() and nothing else decides a workgroup size, a grid, a schedule or an LDS offset; the same
// header is compiled by g++ with -fsanitize=undefined -ftrapv into tests/geometry_check and swept over sizes, models, precisions and the
// caller's overrides (tests/test_geometry_planner.py) -- an arithmetic slip here would otherwise kill the host process (a Julia session,
// through ccall) and could only be found with a GPU.  No HIP, no allocation, no environment.
#pragma once
#include <algorithm>
#include <cstdint>
#include <string>
#include "erm_layout.hpp"

namespace erm {

struct GeomIn {
    int model = RTIRT;          // erm::Model
    bool f64 = true;            // fp64 engine (cell type double) or the fp32 fast mode
    long long N = 0;            // subjects resident on this device
    int J = 0;                  // items
    int Fk = 0;                 // covariate columns the kernels see (0 for the Cross family and Null)
    int ngx = 0;                // extra global statistics (LatentQr, sigp_mode 1)
    int lanes_per_row = 0, block_threads = 0, grid_blocks = 0;      // the caller's overrides, 0 = automatic
    int cu_count = 256;
    bool no_fuse = false;       // diagnostics: keep the two-kernel schedule
    bool no_persist = false;    // per-sweep launches also for small data sets (ERM_FLAG_NO_PERSIST; sharded chains; the stage-timing build)
};

struct Geom {
    int W = 8, logW = 3, IPL = 1;               // lanes per subject in the row-sum phase, items per lane
    int block_threads = 1024, grid_blocks = 256, n_groups = 1;
    long long rows_per_block = 0;
    int rows_per_wave = 0;                      // capacity of a wave's per-subject LDS caches
    bool fused = false;                         // tiny step inside the row-pass kernel (single-pass models when it fits)
    int ns[2] = {0, 0};                         // doubles per statistics row, per pass
    size_t lds_pass[2] = {0, 0};                // dynamic LDS of the stand-alone row pass, per pass
    size_t lds_fused = 0;                       // dynamic LDS of the fused sweep kernel (when fused)
    size_t lds_tiny = 0;                        // dynamic LDS of the stand-alone tiny kernel
    size_t lds_static[2] = {0, 0};              // static LDS of pass_kernel, per pass
    int acc_off[2] = {0, 0}, acc_off_fused = 0; // byte offset of the per-wave item accumulators (the last region of a launch's dynamic LDS)
    int rounds = 1;                             // ceil(grid / resident workgroups)
    bool persist = false;                       // small data sets: ONE launch per erm_run (pass_kernel<..., PERSIST>), every workgroup resident
};

// Small data sets -- up to this many cells, of at most this many items (every workgroup polls every workgroup's statistics row: ~5 J doubles each) -- run all
// sweeps of an erm_run in one persistent launch of at most PERSIST_MAX_GRID workgroups (never more than one per CU) of at most PERSIST_THREADS threads
#ifndef ERM_PERSIST_MAX_CELLS
#define ERM_PERSIST_MAX_CELLS (1 << 17)
#endif
constexpr long long PERSIST_MAX_CELLS = ERM_PERSIST_MAX_CELLS;
constexpr int PERSIST_MAX_ITEMS = 128;
constexpr long long PERSIST_MAX_SUBJ = 6000;      // beyond it the subject phases of <= 64 workgroups outweigh the boundary saved (8 000 x 16: 27.6 against 26.3 us per sweep)
constexpr int PERSIST_MAX_GRID = 64;

namespace geom_detail {
inline int nstat(const GeomIn& g, int phase) { return nstat_of(g.model, phase); }
inline int ng(const GeomIn& g, int phase) { return ng_of(g.model, phase, g.Fk + 1) + (phase == 0 ? g.ngx : 0); }
inline int stat_size(const GeomIn& g, int phase) { return nstat(g, phase) * g.J + ng_of(g.model, phase, g.Fk + 1) + g.ngx; }
inline size_t real_size(const GeomIn& g) { return g.f64 ? 8 : 4; }
// the per-wave item accumulators [nWaves][NSTAT][J] close a launch's dynamic LDS
inline size_t tail_lds(const GeomIn& g, int phase, int nWaves) { return (size_t)nWaves * (size_t)nstat(g, phase) * (size_t)g.J * sizeof(double); }
// dynamic LDS of a row pass = fixed part | per-wave and per-subject caches | (fused: the tiny step's scratch) | accumulators
inline size_t pass_lds(const GeomIn& g, int phase, int nWaves, long long rows_per_block, int rows_per_wave)
{
    const int ngl = stat_size(g, phase) - nstat(g, phase) * g.J;
    const size_t d = (size_t)(8 + 2 * PMAX) + (((size_t)nWaves * (size_t)ngl + 1) & ~(size_t)1);      // (an even count: the item arrays behind it are 16-byte aligned)
    return d * sizeof(double) + ((size_t)NITEMARR * item_stride(g.J) + (size_t)nWaves * 4 * (size_t)rows_per_wave + (size_t)rows_per_block * (size_t)nv_of(g.model, g.Fk)) * real_size(g) + 8 + tail_lds(g, phase, nWaves);
}
inline size_t fused_extra(const GeomIn& g) { return 8 + (size_t)(2 * stat_size(g, 0) + TINY_WORK + 2 * PMAX * PMAX + par_size(g.J) + 3 * g.J + 2) * sizeof(double); }
}  // namespace geom_detail

// Returns 0 and fills `out`, or -1 with a message (the caller maps it to ERM_ERR_ARG).
inline int plan_geometry_core(const GeomIn& g, Geom& out, std::string& err)
{
    using namespace geom_detail;
    const long long N = g.N;
    const int J = g.J;
    if (N <= 0 || J <= 0 || g.Fk < 0) { err = "n_subj, n_item must be positive and n_feat non-negative"; return -1; }
    if (N >= (1LL << 32)) { err = "n_subj must fit 32 bits"; return -1; }
    if (g.model < MLIRT || g.model > LATENT) { err = "unknown model"; return -1; }
    if (g.Fk + 2 > PMAX) { err = "n_feat too large (max " + std::to_string(PMAX - 2) + ")"; return -1; }
    if (J > MAX_ITEMS) { err = "n_item too large (max " + std::to_string(MAX_ITEMS) + ")"; return -1; }
    if (g.cu_count < 1) { err = "bad compute-unit count"; return -1; }
    const bool cq = fam_cq(g.model);
    Geom o;
    for (int ph = 0; ph < 2; ++ph) { o.ns[ph] = stat_size(g, ph); o.lds_static[ph] = pass_static_lds(g.f64, ph); }
    const size_t stat_max = std::max(o.lds_static[0], cq ? o.lds_static[1] : (size_t)0);

    // ---- W lanes per subject.  W only shapes the row-sum phase (the PG phase walks flattened cells): few items per lane keeps the
    // dependent load batches short, many subjects per wave-iteration keeps the number of iterations low; W = 8 balances both for nItem
    // around 50 (measured); never more lanes than items (rounded up to a power of two).
    if (g.lanes_per_row > 0) {
        o.W = g.lanes_per_row;
        if (o.W > 64 || (o.W & (o.W - 1))) { err = "lanes_per_row must be a power of two <= 64"; return -1; }
    } else {
        o.W = 8;
        while (o.W > 1 && o.W / 2 >= J) o.W /= 2;
    }
    o.logW = 0; while ((1 << o.logW) < o.W) ++o.logW;
    o.IPL = (J + o.W - 1) / o.W;

    // ---- threads per workgroup
    const int max_threads = max_block_threads(g.model, g.f64);         // = the kernels' launch bounds
    if (g.block_threads < 0 || g.grid_blocks < 0) { err = "block_threads / grid_blocks must be non-negative"; return -1; }
    int bt = g.block_threads > 0 ? g.block_threads : max_threads;
    if (bt % 64 || bt > max_threads) { err = "block_threads must be a multiple of 64, <= " + std::to_string(max_threads) + " for this model and precision"; return -1; }
    if (g.block_threads == 0) {
        // the per-wave item accumulators (nWaves x NSTAT x nItem doubles) dominate LDS for long tests: fewer waves per workgroup then
        while (bt > 64) {
            const size_t acc = (size_t)(bt / 64) * nstat(g, 0) * J * sizeof(double);
            const size_t fixed = (size_t)NITEMARR * item_stride(J) * real_size(g) + (size_t)(o.ns[0] + 5 * J + 2 * J + TINY_WORK + 2 * PMAX * PMAX + 64) * sizeof(double);
            if (acc + fixed + stat_max <= 120 * 1024) break;
            bt = std::max(64, bt / 2 / 64 * 64);
        }
    }
    // small data sets: a workgroup whose threads would get fewer than two cells each in the PG phase is halved (down to 256 threads) -- its
    // head, its barriers and its reductions are paid per wave (fp64, 50 items: 1 000 subjects 37.2 -> 32.0 us per sweep, 10 000: 42.8 -> 40.4)
    if (g.block_threads == 0 && g.grid_blocks == 0) {
        while (bt > 256) {
            const int nw = bt / 64;
            const int pc = g.f64 ? 1 : std::max(1, 16 / nw);
            const long long gb = std::max<long long>(1, std::min<long long>((N + nw - 1) / nw, (long long)g.cu_count * pc));
            const long long rows = (N + gb - 1) / gb;
            if (rows * J >= 2 * (long long)bt) break;
            bt = std::max(256, bt / 2 / 64 * 64);
        }
    }
    o.block_threads = bt;
    const int nWaves = bt / 64;                                         // >= 1
    const long long need = (N + nWaves - 1) / nWaves;                   // at least one subject per wave
    const int per_cu = g.f64 ? 1 : std::max(1, 16 / nWaves);            // resident workgroups per CU (the fp64 kernel's registers admit one)
    const long long slots = (long long)g.cu_count * per_cu;
    long long gb = g.grid_blocks > 0 ? g.grid_blocks : std::min<long long>(need, slots);
    if (gb < 1) gb = 1;
    if (gb > N) gb = N;                                                 // never an empty workgroup
    const bool single_pass = !cq;

    // ---- each workgroup owns a contiguous range of subjects, split evenly over its waves; the per-subject LDS caches grow with the rows
    // a workgroup owns, so very long data sets get more workgroups than the chip holds at once.  Those run in rounds, and a partial round
    // costs as much as a full one: past one round the count grows by whole rounds -- the smallest number of rounds whose workgroups fit
    // their subjects into LDS.  Only the row pass has to fit (fp64, 200 000 x 50: one round 195 us, two fused rounds 212; 500 000 x 100:
    // 2 rounds 857 us, 4 rounds 884); when the tiny step's scratch no longer fits beside the larger slices the sweep takes the two-kernel
    // schedule (tiny step in a kernel of its own instead of every workgroup's head).
    auto layout = [&](long long rpb, bool fuse, int& rpw, size_t lds[2]) {
        rpw = (int)((rpb + nWaves - 1) / nWaves);
        // fused sweeps take subjects off wave 0 (it runs the tiny step's structural chain first): the other waves' slices grow
        if (fuse && nWaves > 1) rpw = (int)((rpb + nWaves - 2) / (nWaves - 1)) + 1;
        for (int ph = 0; ph < 2; ++ph) lds[ph] = pass_lds(g, ph, nWaves, rpb, rpw);
    };
    const size_t budget = LDS_LIMIT - 2048;                             // dynamic + static below 158 KB
    bool fuse = single_pass && !g.no_fuse;
    long long rpb = (N + gb - 1) / gb;                                  // >= 1
    int rpw = 0; size_t lds[2] = {0, 0};
    for (;;) {
        layout(rpb, fuse, rpw, lds);
        gb = (N + rpb - 1) / rpb;                                       // the smallest count with that many subjects per workgroup (767, not 768)
        const size_t need_lds = std::max(lds[0] + o.lds_static[0], cq ? lds[1] + o.lds_static[1] : (size_t)0);
        const bool cells_ok = rpb * J < (1LL << 22);                    // the PG phase's cell indices (erm_kernels.hpp, `locate`: exact below 2^22)
        if ((need_lds <= budget && cells_ok) || g.grid_blocks > 0 || rpb <= nWaves) break;
        long long g2 = gb >= slots ? ((gb + slots - 1) / slots + 1) * slots : std::min(slots, gb + std::max<long long>(1, gb / 4));
        if (g2 > N) g2 = N;
        long long r2 = (N + g2 - 1) / g2;
        if (r2 >= rpb) r2 = rpb - 1;                                    // progress is monotone (rpb > nWaves >= 1 here): the loop ends
        rpb = r2;
    }
    if (gb > 0x7FFFFFFFLL / GROUP) { err = "grid too large"; return -1; }
    // fused head: only if the tiny step's scratch fits beside the pass layout (static LDS included)
    if (fuse && lds[0] + fused_extra(g) + o.lds_static[0] > LDS_LIMIT) {
        fuse = false;
        layout(rpb, false, rpw, lds);
    }
    for (int ph = 0; ph < (cq ? 2 : 1); ++ph) {
        if (lds[ph] + o.lds_static[ph] > LDS_LIMIT) { err = "LDS footprint too large; lower block_threads or raise grid_blocks"; return -1; }
    }
    if (rpb * J >= (1LL << 22)) { err = "a workgroup would own 2^22 cells or more; raise grid_blocks"; return -1; }
    o.grid_blocks = (int)gb;
    o.rows_per_block = rpb;
    o.rows_per_wave = rpw;
    o.fused = fuse;
    o.lds_pass[0] = lds[0]; o.lds_pass[1] = lds[1];
    o.lds_fused = fuse ? lds[0] + fused_extra(g) : 0;
    o.lds_tiny = (size_t)tiny_lds_doubles(o.ns[0], cq ? o.ns[1] : 0, J) * sizeof(double);
    if (o.lds_tiny > LDS_LIMIT) { err = "tiny-step LDS footprint too large"; return -1; }
    o.n_groups = (o.grid_blocks + GROUP - 1) / GROUP;
    if (o.n_groups > TINY_THREADS) { err = "grid too large"; return -1; }
    for (int ph = 0; ph < 2; ++ph) o.acc_off[ph] = (int)((o.lds_pass[ph] - tail_lds(g, ph, nWaves)) & ~(size_t)7);
    o.acc_off_fused = fuse ? (int)((o.lds_fused - tail_lds(g, 0, nWaves)) & ~(size_t)7) : 0;
    o.rounds = (int)((gb + slots - 1) / slots);
    out = o;
    return 0;
}

// The plan the engine uses: the persistent schedule for small data sets of the single-pass models (its own geometry unless the caller gave one),
// otherwise the per-sweep plan.  `no_persist` (ERM_FLAG_NO_PERSIST, sharded chains, the fallback after a persistent launch timed out) changes the
// SCHEDULE only: the geometry stays the persistent plan's, so the statistics are summed in the same association and the chain is the same bit for bit.
inline int plan_geometry(const GeomIn& g, Geom& out, std::string& err)
{
    const bool cq = g.model == CROSSQR || g.model == CROSS;
    const bool shape = !cq && !g.no_fuse && g.N > 0 && g.J > 0 && g.N <= PERSIST_MAX_SUBJ && g.N * (long long)g.J <= PERSIST_MAX_CELLS && g.J <= PERSIST_MAX_ITEMS;
    const int max_grid = std::min(g.cu_count, PERSIST_MAX_GRID);
    auto fits = [&](const Geom& p) { return p.fused && p.rounds == 1 && p.grid_blocks <= max_grid && p.block_threads <= PERSIST_THREADS; };
    if (shape && g.block_threads == 0 && g.grid_blocks == 0) {
        GeomIn gp = g;
        gp.block_threads = std::min(PERSIST_THREADS, max_block_threads(g.model, g.f64));
        // measured (1 000 x 15 ... 8 000 x 16, fp64): 32 workgroups up to ~1 500 subjects (16.8 / 16.9 / 17.8 us at 250 / 500 / 1 000 x 15 against 18.4 / 18.7 / 18.8
        // with 64), 64 beyond (2 000 x 15: 19.8 against 20.2; 4 000 x 30: 24.7 against 29.6; 6 000 x 20: 25.4 against 33.2)
        gp.grid_blocks = g.N <= 1500 ? (int)std::max<long long>(1, std::min<long long>(32, (g.N + 7) / 8)) : std::min(64, max_grid);
        Geom p;
        std::string e2;
        if (plan_geometry_core(gp, p, e2) == 0 && fits(p)) { p.persist = !g.no_persist; out = p; return 0; }
    }
    if (int rc = plan_geometry_core(g, out, err)) return rc;
    out.persist = shape && !g.no_persist && fits(out);
    return 0;
}

}  // namespace erm
