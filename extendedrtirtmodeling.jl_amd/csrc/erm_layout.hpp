// erm_layout.hpp -- constants and layouts shared by the kernels (erm_kernels.hpp), the host side (ertirt.hip) and the geometry planner
// (erm_geometry.hpp).  Plain C++: also compiled by g++ for the planner's CPU test (tests/test_geometry_planner.py).
#pragma once
#include <cstddef>
#include <cstdint>

#if defined(__HIPCC__)
#define ERM_HD __host__ __device__
#else
#define ERM_HD
#endif

namespace erm {

enum Model : int { MLIRT = 0, RTIRT = 1, CROSSQR = 2, LATENTQR = 3, NULLM = 4, CROSS = 5, LATENT = 6 };
// Model families.  The non-quantile variants (GibbsRtIrtNull / Cross / Latent, /root/reference/src/GibbsRtIrt.pl.jl:367-426,
// src/GibbsRtIrtCross.pl.jl:176-235, src/GibbsRtIrtLatent.pl.jl:168-233) run their quantile sibling's kernels with nu == 1, k1 = 0,
// k2 = 1 (the host passes those; x*1, x/1 and x+0 are exact) and without any nu traffic; the draws that differ are spelled out.
ERM_HD constexpr bool fam_rt(int M) { return M == RTIRT || M == NULLM; }       // bivariate (theta, zeta) structure, one pass
ERM_HD constexpr bool fam_lq(int M) { return M == LATENTQR || M == LATENT; }   // zeta regressed on [1 X theta], one pass
ERM_HD constexpr bool fam_cq(int M) { return M == CROSSQR || M == CROSS; }     // cross-relation rho, two passes
ERM_HD constexpr bool has_nu(int M) { return M == CROSSQR || M == LATENTQR; }

constexpr int PMAX = 16;            // max columns of the latent-regression design ([1 X theta])
// per-subject values parked in LDS for the global statistics (pass_kernel, sh_val): the X row, theta, zeta -- and, where the latent regression of zeta on
// [1 X theta] needs them (Latent family), u = zeta - k1 nu and nu.  Round 4: the two extra slots only where they are read (16 B per subject in fp64: a
// 500 000 x 100 GibbsRtIrt chain fits two rounds of workgroups instead of three)
ERM_HD constexpr int nv_of(int M, int nFeat) { return nFeat + (fam_lq(M) ? 4 : 2); }
constexpr int NITEMARR = 8;         // per-item arrays staged in LDS
// ... at a stride the row-sum loop knows at compile time for the usual test lengths: its five arrays are then read at immediate offsets from ONE address per item pair
// (one vector add per array and pair otherwise: a sixth of that loop's instructions)
#ifndef ERM_ITEM_STRIDE
#define ERM_ITEM_STRIDE 128
#endif
constexpr int ITEM_STRIDE = ERM_ITEM_STRIDE;
ERM_HD constexpr int item_stride(int J) { return (ITEM_STRIDE > 0 && J <= ITEM_STRIDE) ? ITEM_STRIDE : J; }
constexpr int MAX_ITEMS = 896;      // n_item limit of the engine

// parameter block written by the tiny step (fp64): a, b, lambda, sig2t, rho : 5 x J, then Sigp(4), beta(2*PMAX),
// then derived scalars: [0] sum_j 1/sig2t_j
ERM_HD inline int par_off_sigp(int J) { return 5 * J; }
ERM_HD inline int par_off_beta(int J) { return 5 * J + 4; }
ERM_HD inline int par_off_derived(int J) { return 5 * J + 4 + 2 * PMAX; }
ERM_HD inline int par_size(int J) { return 5 * J + 4 + 2 * PMAX + 4; }

// data constants (fp64): K0[J], m[J] (column means of logT), csq[J] (sum of squared centred logT), muLam, sdLam,
// XtX[PMAX*PMAX] = x'x with x = [1 X] (iteration-invariant, src/Draw.pl.jl:383-386 recomputes it every sweep), XtXinv[PMAX*PMAX]
ERM_HD inline int cst_off_k0(int) { return 0; }
ERM_HD inline int cst_off_m(int J) { return J; }
ERM_HD inline int cst_off_csq(int J) { return 2 * J; }
ERM_HD inline int cst_off_mu(int J) { return 3 * J; }
ERM_HD inline int cst_off_xtx(int J) { return 3 * J + 2; }
ERM_HD inline int cst_off_xinv(int J) { return 3 * J + 2 + PMAX * PMAX; }
ERM_HD inline int cst_size(int J) { return 3 * J + 2 + 2 * PMAX * PMAX; }

// statistics layout of one slab row: NSTAT item statistics x J, then NG globals
//   MlIrt        : S0 S1 S2 K1             | x'theta, LL
//   RtIrt family : S0 S1 S2 K1 G           | x'theta, x'zeta, tt, tz, zz, LL
//   Latent family: S0 S1 S2 K1 G           | x'theta, tt, x'u, tu, uu, snu, snu2, sz, zz, LL      (u = zeta - k1 nu)
//   Cross family : S0 S1 S2 K1 W0 W1 W2 V  | LL_A        (pass A)       R0 R1 | zz, LL_B   (pass B)
ERM_HD constexpr int nstat_of(int MODEL, int PHASE) { return (MODEL == MLIRT) ? 4 : (fam_cq(MODEL) ? (PHASE == 0 ? 8 : 2) : 5); }
ERM_HD constexpr int ng_of(int MODEL, int PHASE, int p) { return (MODEL == MLIRT) ? p + 1 : fam_rt(MODEL) ? 2 * p + 4 : fam_lq(MODEL) ? 2 * p + 8 : (PHASE == 0 ? 1 : 2); }
template <int MODEL, int PHASE> struct Stats {
    static constexpr int NSTAT = nstat_of(MODEL, PHASE);
    ERM_HD static int ng(int p) { return ng_of(MODEL, PHASE, p); }
};

#ifndef ERM_F32_THREADS
#define ERM_F32_THREADS 1024     // threads per workgroup of the fp32 engine = the register budget the row-pass kernel is compiled for (128 VGPRs)
#endif
#ifndef ERM_F64_THREADS
#define ERM_F64_THREADS 1024     // fp64 engine: 16 waves per CU (4 per SIMD), 128 VGPRs each -- ONE workgroup per CU, like the fp32 engine
#endif
#ifndef ERM_F64_THREADS_LATENTQR
#define ERM_F64_THREADS_LATENTQR 768      // LatentQr (fp64 inverse-Gaussian weights in the subject draws): 58 spilled registers at 128 VGPRs, none at 168
#endif
constexpr int PERSIST_THREADS = 512;           // launch bound of the persistent (small data set) sweep kernel: 256 VGPRs, no spills around its sweep loop
ERM_HD constexpr int max_block_threads(int model, bool f64) { return f64 ? (model == LATENTQR ? ERM_F64_THREADS_LATENTQR : ERM_F64_THREADS) : ERM_F32_THREADS; }

constexpr int GROUP = 16;           // workgroups whose slab rows are summed by the last of them to finish
constexpr int TINY_THREADS = 1024;
constexpr int TINY_WORK = 2 * (2 * PMAX) * (2 * PMAX) + 12 * PMAX + 16;   // LDS scratch doubles for the structural wave
inline int tiny_lds_doubles(int NS0, int NS1, int J) { return NS0 + NS1 + J + TINY_WORK + 2 * PMAX * PMAX + par_size(J); }

constexpr int PG_NBIN = 128;        // z-bins of the Polya-Gamma proposal table (erm_rng.hpp)
// STATIC LDS of pass_kernel<MODEL, real, PHASE, *> (declared inside the kernel, on top of its dynamic LDS): the fp64 engine's logarithm table
// (128 x double2), the Polya-Gamma proposal table (PG_NBIN x float4; phase 0) and its fp64 1/lam column (fp64 engine).  The planner adds it to
// every LDS limit; Engine::init compares it with hipFuncGetAttributes().sharedSizeBytes.
ERM_HD constexpr size_t pass_static_lds(bool f64, int phase) { return (f64 ? 128 * 16 : 0) + (phase == 0 ? PG_NBIN * 16 + (f64 ? PG_NBIN * 8 : 0) : 0); }
constexpr size_t LDS_LIMIT = 160 * 1024;    // per workgroup on gfx950

}  // namespace erm
