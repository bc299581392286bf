// tests/geometry_check.cpp -- CPU sweep of the launch-geometry planner (extendedrtirtmodeling.jl_amd/csrc/erm_geometry.hpp).
// Built by tests/test_geometry_planner.py with g++ -fsanitize=undefined -fno-sanitize-recover -ftrapv: any division by zero, signed overflow or
// out-of-range shift aborts the run.  For every accepted plan it asserts the invariants the kernels rely on; prints a summary and, with
// `case <model> <f64> <N> <J> <Fk> <bt> <gb> <W> [cus] [nofuse] [nopersist]`, one plan as key=value pairs.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "erm_geometry.hpp"

using namespace erm;

#include <map>
static long long n_ok = 0, n_rej = 0, n_fused = 0, n_auto_rej = 0, n_persist = 0;
static std::map<std::string, long long> reasons;
static int fails = 0;
#define REQUIRE(cond, ...) do { if (!(cond)) { if (fails++ < 20) { fprintf(stderr, "FAIL %s: ", #cond); fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } } while (0)

static void describe(const GeomIn& g, char* buf, size_t n)
{
    snprintf(buf, n, "model=%d f64=%d N=%lld J=%d Fk=%d ngx=%d W=%d bt=%d gb=%d cus=%d nofuse=%d", g.model, (int)g.f64, g.N, g.J, g.Fk, g.ngx, g.lanes_per_row, g.block_threads,
             g.grid_blocks, g.cu_count, (int)g.no_fuse);
}

static bool check(const GeomIn& g)
{
    Geom o;
    std::string err;
    char d[256];
    describe(g, d, sizeof d);
    const int rc = plan_geometry(g, o, err);
    if (rc != 0) {
        ++n_rej; ++reasons[err];
        REQUIRE(!err.empty(), "%s: rejected without a message", d);
        // with nothing overridden, every size inside the documented limits (up to 2^22 subjects on one device) has a plan
        if (g.block_threads == 0 && g.grid_blocks == 0 && g.lanes_per_row == 0 && g.N > 0 && g.N < (1LL << 32) && g.J >= 1 && g.J <= MAX_ITEMS && g.Fk >= 0 && g.Fk + 2 <= PMAX && g.cu_count >= 1 &&
            g.model >= 0 && g.model <= 6 && g.N <= 4194304) { ++n_auto_rej; REQUIRE(false, "%s: no automatic plan: %s", d, err.c_str()); }
        return false;
    }
    ++n_ok;
    const int nWaves = o.block_threads / 64;
    const bool cq = fam_cq(g.model);
    REQUIRE(o.block_threads >= 64 && o.block_threads % 64 == 0 && o.block_threads <= max_block_threads(g.model, g.f64), "%s: bt %d", d, o.block_threads);
    REQUIRE(o.grid_blocks >= 1 && (long long)o.grid_blocks <= g.N, "%s: grid %d", d, o.grid_blocks);
    REQUIRE(o.rows_per_block >= 1 && o.rows_per_block * (long long)o.grid_blocks >= g.N, "%s: rows_per_block %lld grid %d do not cover N", d, o.rows_per_block, o.grid_blocks);
    REQUIRE((o.rows_per_block - 1) * (long long)o.grid_blocks < g.N || o.grid_blocks == 1 || g.grid_blocks > 0, "%s: grid %d is not the smallest for %lld rows", d, o.grid_blocks, o.rows_per_block);
    REQUIRE(((long long)o.grid_blocks - 1) * o.rows_per_block < g.N, "%s: the last workgroup would be empty", d);
    REQUIRE(o.rows_per_block * g.J < (1LL << 22), "%s: %lld cells per workgroup", d, o.rows_per_block * g.J);
    // a wave's LDS caches hold its slice: balanced split, or (fused) wave 0 relieved of up to rows_per_block / nWaves subjects
    const long long per_wave = (o.rows_per_block + nWaves - 1) / nWaves;
    REQUIRE(o.rows_per_wave >= per_wave, "%s: rows_per_wave %d < %lld", d, o.rows_per_wave, per_wave);
    if (o.fused && nWaves > 1) REQUIRE((long long)o.rows_per_wave * (nWaves - 1) >= o.rows_per_block, "%s: fused slices do not hold the workgroup's rows", d);
    for (int ph = 0; ph < (cq ? 2 : 1); ++ph) {
        REQUIRE(o.lds_pass[ph] + o.lds_static[ph] <= LDS_LIMIT, "%s: pass %d LDS %zu + %zu", d, ph, o.lds_pass[ph], o.lds_static[ph]);
        const size_t tail = (size_t)nWaves * nstat_of(g.model, ph) * g.J * 8;
        REQUIRE(o.acc_off[ph] >= 0 && o.acc_off[ph] % 8 == 0 && (size_t)o.acc_off[ph] + tail <= o.lds_pass[ph], "%s: acc_off[%d] %d + %zu > %zu", d, ph, o.acc_off[ph], tail, o.lds_pass[ph]);
    }
    if (o.fused) {
        ++n_fused;
        REQUIRE(!cq && !g.no_fuse, "%s: fused where it must not be", d);
        REQUIRE(o.lds_fused + o.lds_static[0] <= LDS_LIMIT, "%s: fused LDS %zu + %zu", d, o.lds_fused, o.lds_static[0]);
        const size_t tail = (size_t)nWaves * nstat_of(g.model, 0) * g.J * 8;
        REQUIRE(o.acc_off_fused % 8 == 0 && (size_t)o.acc_off_fused + tail <= o.lds_fused, "%s: acc_off_fused", d);
        REQUIRE((size_t)o.acc_off_fused >= o.lds_pass[0] - tail, "%s: the tiny step's scratch overlaps the pass layout", d);
    }
    if (o.persist) {
        // one persistent launch per erm_run: every workgroup resident at once (at most one per CU), 512-thread launch bounds, the fused single-pass sweep
        ++n_persist;
        REQUIRE(o.fused && !cq && !g.no_persist && !g.no_fuse, "%s: persistent where it must not be", d);
        REQUIRE(o.rounds == 1 && o.grid_blocks <= g.cu_count && o.grid_blocks <= PERSIST_MAX_GRID, "%s: persistent grid %d (rounds %d)", d, o.grid_blocks, o.rounds);
        REQUIRE(o.block_threads <= PERSIST_THREADS, "%s: persistent block of %d threads", d, o.block_threads);
        REQUIRE(g.N * (long long)g.J <= PERSIST_MAX_CELLS && g.J <= PERSIST_MAX_ITEMS && g.N <= PERSIST_MAX_SUBJ, "%s: persistent beyond its size limits", d);
    }
    REQUIRE(o.lds_tiny <= LDS_LIMIT, "%s: tiny LDS %zu", d, o.lds_tiny);
    REQUIRE(o.n_groups == (o.grid_blocks + GROUP - 1) / GROUP && o.n_groups <= TINY_THREADS, "%s: n_groups %d", d, o.n_groups);
    REQUIRE(o.W >= 1 && o.W <= 64 && (o.W & (o.W - 1)) == 0 && (1 << o.logW) == o.W && o.IPL * o.W >= g.J, "%s: W %d IPL %d", d, o.W, o.IPL);
    return true;
}

int main(int argc, char** argv)
{
    if (argc >= 10 && !strcmp(argv[1], "case")) {
        GeomIn g;
        g.model = atoi(argv[2]); g.f64 = atoi(argv[3]) != 0; g.N = atoll(argv[4]); g.J = atoi(argv[5]); g.Fk = atoi(argv[6]);
        g.block_threads = atoi(argv[7]); g.grid_blocks = atoi(argv[8]); g.lanes_per_row = atoi(argv[9]);
        if (argc > 10) g.cu_count = atoi(argv[10]);
        if (argc > 11) g.no_fuse = atoi(argv[11]) != 0;
        if (argc > 12) g.no_persist = atoi(argv[12]) != 0;
        Geom o; std::string err;
        if (plan_geometry(g, o, err) != 0) { printf("error=%s\n", err.c_str()); return 0; }
        check(g);
        printf("W=%d block_threads=%d grid_blocks=%d rows_per_block=%lld rows_per_wave=%d fused=%d lds0=%zu lds1=%zu lds_fused=%zu lds_static=%zu rounds=%d n_groups=%d persist=%d\n", o.W, o.block_threads,
               o.grid_blocks, o.rows_per_block, o.rows_per_wave, (int)o.fused, o.lds_pass[0], o.lds_pass[1], o.lds_fused, o.lds_static[0], o.rounds, o.n_groups, (int)o.persist);
        return fails ? 1 : 0;
    }
    // ---- the sweep
    static const long long Ns[] = {1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 300, 1000, 4096, 5000, 6000, 10000, 30000, 65536, 100000, 173000, 175000, 177000, 200000, 250000, 500000, 1000000, 2000000,
                                   4194304, 50000000, 4294967295LL};
    static const int Js[] = {1, 2, 3, 5, 7, 8, 15, 16, 31, 50, 64, 65, 100, 128, 200, 255, 260, 300, 400, 512, 640, 895, 896};
    static const int Fs[] = {0, 1, 3, 7, 14};
    static const int BTs[] = {0, 64, 128, 192, 256, 512, 768, 1024};
    static const int GBs[] = {0, 1, 7, 256, 300, 1500, 100000};
    static const int CUs[] = {256, 304, 8, 1};
    for (int model = 0; model <= 6; ++model)
        for (int f64 = 0; f64 <= 1; ++f64)
            for (long long N : Ns)
                for (int J : Js)
                    for (int F : Fs) {
                        GeomIn g;
                        g.model = model; g.f64 = f64 != 0; g.N = N; g.J = J;
                        g.Fk = (fam_cq(model) || model == NULLM) ? 0 : F;
                        if (g.Fk != F && F != 0) continue;
                        for (int ngx = 0; ngx <= (model == LATENTQR ? 1 : 0); ++ngx) {
                            g.ngx = ngx ? (g.Fk + 2) * (g.Fk + 3) / 2 + g.Fk + 2 + 1 : 0;
                            for (int bt : BTs) for (int gb : GBs) {
                                if ((bt || gb) && (J % 7 != 1 && J != 50 && J != 300) ) continue;       // overrides on a subset of lengths (keeps the sweep at seconds)
                                for (int cu : CUs) {
                                    if (cu != 256 && (bt || gb || F != 3)) continue;
                                    for (int nofuse = 0; nofuse <= 1; ++nofuse) {
                                        if (nofuse && (bt || gb || cu != 256)) continue;
                                        g.block_threads = bt; g.grid_blocks = gb; g.cu_count = cu; g.no_fuse = nofuse != 0; g.lanes_per_row = 0;
                                        g.no_persist = false; check(g);
                                        if (N * J <= PERSIST_MAX_CELLS) { g.no_persist = true; check(g); }
                                    }
                                }
                            }
                        }
                    }
    // lanes_per_row overrides, bad arguments
    for (int W : {1, 2, 4, 8, 16, 32, 64, 3, 128, -1}) { GeomIn g; g.N = 1000; g.J = 50; g.Fk = 3; g.lanes_per_row = W; check(g); }
    { GeomIn g; g.N = 0; g.J = 5; check(g); g.N = 5; g.J = 0; check(g); g.J = 897; check(g); g.J = 5; g.Fk = 15; check(g); g.Fk = 1; g.block_threads = 100; check(g); g.block_threads = -64; check(g);
      g.block_threads = 0; g.grid_blocks = -1; check(g); g.grid_blocks = 0; g.cu_count = 0; check(g); g.cu_count = 256; g.model = 9; check(g); }
    printf("plans accepted %lld (fused %lld, persistent %lld), rejected %lld (automatic geometry: %lld), invariant failures %d\n", n_ok, n_fused, n_persist, n_rej, n_auto_rej, fails);
    for (auto& kv : reasons) printf("  rejected %8lld: %s\n", kv.second, kv.first.c_str());
    return fails ? 1 : 0;
}
