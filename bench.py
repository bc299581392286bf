#!/usr/bin/env python3
"""bench.py -- Gibbs sweeps/s of the MI355X engine on BASELINE.json's workload.

A "step" is ONE Gibbs sweep (every full conditional of one `for m, l` iteration of sample!) over one synthetic data set
that is already resident in HBM.  Default workload: GibbsRtIrt, nSubj=100000, nItem=50, nFeat=3 (BASELINE.json configs[2],
the configuration the north-star target is quoted on); data per setDataRtIrt's distributions, fixed seed.
N GPUs = N independent chains (one process per GPU, chain_id = rank, no data-path collective): "scaling": "weak".
After the timed region the ranks all-reduce their posterior summaries over RCCL (timed separately, reported as gather_ms).

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (the fused row pass): algorithmic bytes per launch / mean launch duration measured live with
                  HIP events on the engine's stream, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (kind "port": fp64 C restatement of the reference's un-fused schedule, 1 thread) timed on this
                  host for a bounded number of sweeps of the SAME workload.  The reference itself is Julia and cannot run here.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # the oracle's OpenMP threads must not spin on a shared host
import __graft_entry__ as ge  # noqa: E402

ALGO_BYTES = {"mlirt": 9, "rtirt": 13, "latentqr": 13, "crossqr": 29,          # fp32 matrices, Y as 1 byte
              "null": 13, "latent": 13, "cross": 17}                        # variants: as their families; Cross: omega r/w, Y, logT twice, no nu
FAMILY = {"null": "rtirt", "latent": "latentqr", "cross": "crossqr"}
HBM_PEAK_GBS = 8000.0                                                     # MI355X_MICROARCH.md: 8.0 TB/s spec


def make_data(pkg, model, N, J, F, seed):
    Cond = pkg.setCond(nSubj=N, nItem=J, nFeat=F, nIter=10, nChain=1, qRt=0.85)
    g = np.random.default_rng(seed)
    if model == "mlirt":
        tp = pkg.setTrueParaMlIrt(Cond, seed=g); D = pkg.setDataMlIrt(Cond, tp, seed=g)
        return D.Y, None, D.X
    model = FAMILY.get(model, model)
    if model == "rtirt":
        tp = pkg.setTrueParaRtIrt(Cond, seed=g); D = pkg.setDataRtIrt(Cond, tp, seed=g)
        return D.Y, D.logT, D.X
    if model == "crossqr":
        tp = pkg.setTrueParaRtIrtCross(Cond, seed=g); D = pkg.setDataRtIrtCross(Cond, tp, seed=g)
        return D.Y, D.logT, None
    tp = pkg.setTrueParaRtIrtLatent(Cond, seed=g); D = pkg.setDataRtIrtLatent(Cond, tp, seed=g)
    return D.Y, D.logT, D.X


def init_state(model, N, J, F, rank):
    g = np.random.default_rng([99, rank])
    st = dict(theta=g.standard_normal(N))
    if model != "mlirt":
        st.update(zeta=g.standard_normal(N), sigp=np.eye(2))
    if model == "mlirt":
        st["beta"] = g.standard_normal(F + 1)
    elif model == "rtirt":
        st["beta"] = g.standard_normal((F + 1, 2))
    elif model in ("latentqr", "latent"):
        st["beta"] = g.standard_normal(F + 2)
    elif model == "null":
        pass
    else:
        st["rho"] = g.standard_normal(J)
    return st


def host_cores():
    """Threads this process may really use: the affinity mask capped by the cgroup CPU quota (a GPU box shows every core of the host
    in the mask but grants a share of them) and by 16, the documented share of a one-GPU box."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(model, Y, logT, X, st, sweeps, threads=1, budget_s=30.0):
    """Oracle (kind 'port') on the same workload.  threads=1 mirrors the reference's own execution model (single Julia thread);
    threads>1 is the OpenMP run SURVEY.md 8(d) asks for as the stronger bar.  Bounded twice: at most `sweeps` sweeps and at most
    ~budget_s seconds (checked after every sweep).  Returns (seconds per sweep, sweeps timed)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity_util as pu
    pu.oracle().orc_set_threads(int(threads))
    try:
        op = pu.OracleProblem(model, Y, logT, X, st, qRt=0.85, cov2one=(model not in ("latentqr", "latent")))
        t0 = time.perf_counter()
        op.run(1)                       # warm-up sweep (page in, first omega)
        first = time.perf_counter() - t0
        if first > budget_s:
            return first, 1
        n, t0 = 0, time.perf_counter()
        while n < sweeps and time.perf_counter() - t0 < budget_s:
            op.run(1)
            n += 1
        return (time.perf_counter() - t0) / max(n, 1), n
    finally:
        pu.oracle().orc_set_threads(1)


def traffic_bytes(model, N, J, args):
    """HBM-side bytes per pass_kernel launch from the PMC counters.  Counters cannot be read from inside this process; they are
    collected by tools/collect_profiles.sh in separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this same command and
    stored, with the gfx950 FETCH_SIZE x2 correction calibrated on a known byte count, in profiles/traffic.json.  Returned only
    when that file describes exactly this workload; otherwise null."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    try:
        t = json.load(open(path))
    except Exception:
        return None
    key = f"{model}:{N}x{J}:{args.precision}"
    e = t.get(key)
    return None if e is None else e.get("traffic_bytes_per_launch")


def valu_profile(model, N, J, args):
    """VALU-side counters of the same offline profile (profiles/traffic.json), or None: the kernel's real limiter."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"{model}:{N}x{J}:{args.precision}")
        return None if e is None else e.get("valu")
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--model", default="rtirt", choices=list(ALGO_BYTES))
    ap.add_argument("--nsubj", type=int, default=100000)
    ap.add_argument("--nitem", type=int, default=50)
    ap.add_argument("--nfeat", type=int, default=3)
    ap.add_argument("--precision", default="f32", choices=["f32", "f64"])
    ap.add_argument("--trace", default="full", choices=["full", "summary"])
    ap.add_argument("--lanes-per-row", type=int, default=0)
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--grid-blocks", type=int, default=0)
    ap.add_argument("--cpu-sweeps", type=int, default=-1, help="oracle sweeps for cpu_baseline (-1 = auto ~15 s, 0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket row-pass launches with HIP events")
    ap.add_argument("--shard-exchange", default="rccl", choices=["rccl", "callback"], help="--shard: in-stream RCCL all-gather (default) or the "
                                                        "host callback over torch.distributed")
    ap.add_argument("--shard", action="store_true", help="NOT the headline: ONE chain of --nsubj subjects sharded over the ranks (strong scaling; "
                                                        "one all-gather of a statistics row per row pass, SURVEY.md 8(e))")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rehearse = os.environ.get("ERM_BENCH_REHEARSE") == "1"
    coll_dev = "cpu" if rehearse else None
    import torch
    dist = None
    if world > 1 or os.environ.get("ERM_BENCH_FORCE_DIST") == "1":     # FORCE_DIST: exercise the RCCL calls with one rank on a one-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:     # ERM_BENCH_REHEARSE=1: the N>1 code path on a ONE-GPU box -- every rank on cuda:0, collectives over gloo on the CPU
            local_rank = 0
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    ge.build_hip()
    pkg = ge.load_package()
    L = pkg._lib
    model, N, J, F = args.model, args.nsubj, args.nitem, args.nfeat
    Y, logT, X = make_data(pkg, model, N, J, F, seed=1234)
    shard = args.shard and dist is not None
    st = init_state(model, N, J, F, 0 if shard else rank)
    rows = args.warmup + args.steps
    lo, n_loc = (0, N)
    if shard:       # every rank builds the same data set and keeps its rows
        lo, n_loc = pkg.parallel.shard_rows(N, world)[rank]
        Y, logT, X = Y[lo:lo + n_loc], (None if logT is None else logT[lo:lo + n_loc]), (None if X is None else X[lo:lo + n_loc])
        st = dict(st, **{k: st[k][lo:lo + n_loc] for k in ("theta", "zeta") if k in st})
    eng = L.Engine(model=getattr(L, "MODEL_" + model.upper()), n_item=J, n_subj=n_loc, n_feat=0 if X is None else F, n_iter=rows, n_chain=1,
                   n_burnin=args.warmup, cov2one=int(model not in ("latentqr", "latent")), q_rt=0.85, seed=1234, chain_id=0 if shard else rank, device=local_rank,
                   precision=L.PREC_F32 if args.precision == "f32" else L.PREC_F64,
                   trace_mode=L.TRACE_FULL if args.trace == "full" else L.TRACE_SUMMARY, lanes_per_row=args.lanes_per_row,
                   block_threads=args.block_threads, grid_blocks=args.grid_blocks, profile=0 if args.no_profile else 1)
    if shard and (rehearse or args.shard_exchange == "callback"):
        eng.set_shard(rank, world, N, lo, pkg.parallel.TorchExchange(L.load(), device=None if rehearse else f"cuda:{local_rank}"))
    elif shard:     # the library's own in-stream RCCL all-gather; the 128-byte id travels over the process group
        box = [L.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.set_shard_rccl(rank, world, N, lo, box[0])
    eng.set_data(Y, logT, X)       # inputs resident in HBM from here on
    eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in st.items()})

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        eng.run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    eng.run(args.steps)            # erm_run returns after hipStreamSynchronize on the engine's stream
    barrier()
    dt = time.perf_counter() - t0
    tm = eng.timing()
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=coll_dev or f"cuda:{local_rank}")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # posterior-summary gather over RCCL (outside the timed region; this is the only collective of the path)
    gather_ms = None
    if dist is not None and not shard:
        mean = eng.get_mean()
        P = pkg.InputPara(theta=mean["theta"], a=mean["a"], b=mean["b"])
        for k_src, k_dst in (("zeta", "zeta"), ("lambda_", "lam"), ("sig2t", "sig2t"), ("beta", "beta"), ("sigp", "Sigp"), ("rho", "rho")):
            if mean.get(k_src) is not None:
                setattr(P, k_dst, mean[k_src])
        torch.cuda.synchronize()
        g0 = time.perf_counter()
        pkg.parallel.gather_posterior_summaries(P, eng.post_count, device=None if rehearse else f"cuda:{local_rank}")
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - g0) * 1e3

    if rank == 0:
        cells = float(N) * J
        value = cells * args.steps * (1 if shard else world) / dt
        out = {
            "metric": "Gibbs cell-updates/s (nSubj x nItem x sweeps/s)", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
            "config": {"workload": f"Gibbs{ {'mlirt': 'MlIrt', 'rtirt': 'RtIrt', 'latentqr': 'RtIrtLatentQr', 'crossqr': 'RtIrtCrossQr', 'null': 'RtIrtNull', 'cross': 'RtIrtCross', 'latent': 'RtIrtLatent'}[model] } "
                                   f"nSubj={N} nItem={J} nFeat={F} " + (f"ONE chain, subjects sharded over {world} devices" if shard else "nChain=1 per GPU (BASELINE.json configs[2])"),
                       "chains": 1 if shard else world, "subject_shards": world if shard else 1, "shard_exchange": (("callback" if rehearse else args.shard_exchange) if shard else None), "trace": args.trace, "lanes_per_row": tm["lanes_per_row"], "block_threads": tm["block_threads"],
                       "grid_blocks": tm["grid_blocks"], "lds_bytes": tm["lds_bytes"]},
            "sweeps_per_s": args.steps * (1 if shard else world) / dt, "device_ms_per_step": tm["run_ms"] / args.steps,
        }
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
        if tm["pass_launches"] > 0:
            per_launch_s = tm["pass_ms_total"] / tm["pass_launches"] * 1e-3
            launches_per_sweep = 2 if model in ("crossqr", "cross") else 1
            algo = ALGO_BYTES[model] * float(n_loc) * J / launches_per_sweep        # a launch covers this rank's subjects
            ach = algo / per_launch_s / 1e9
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                               "traffic": traffic_bytes(model, N, J, args), "kernel": "pass_kernel (one launch per sweep: tiny step + fused row pass; CrossQr: one of its two row passes)",
                               "frac_of_achievable": ach / 6300.0, "achievable_peak": 6300.0, "launch_us": per_launch_s * 1e6, "event_overhead_us": tm["event_overhead_ms"] * 1e3,
                               "algorithmic_bytes_per_launch": algo, "launches_timed": int(tm["pass_launches"]),
                               "valu": valu_profile(model, N, J, args)}
        ncpu = args.cpu_sweeps
        if ncpu != 0 and world == 1:          # the CPU baseline is reported at N=1 only
            if ncpu < 0:
                ncpu = max(2, int(round(15.0 / (cells * 2.6e-7))))      # ~0.26 us per cell-update on one host core
            sec, nrun = cpu_baseline(model, Y, logT, X, st, ncpu, budget_s=25.0)
            out["cpu_baseline"] = {"value": cells / sec, "unit": "cell-updates/s", "cores": 1, "kind": "port",
                                   "sample": f"{nrun} sweeps of the same workload after 1 warm-up sweep; oracle/erm_oracle.c (fp64, "
                                             f"reference's un-fused schedule); proxy for Julia sample! (Julia unavailable)",
                                   "s_per_sweep": sec}
            out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
            ncores = host_cores()
            if ncores > 1:
                sec_mt, nmt = cpu_baseline(model, Y, logT, X, st, 8 * ncpu, threads=ncores, budget_s=8.0)
                out["cpu_baseline_all_cores"] = {"value": cells / sec_mt, "unit": "cell-updates/s", "cores": ncores, "kind": "port",
                                                 "sample": f"{nmt} sweeps, same oracle with OpenMP over subjects/items ({ncores} threads)",
                                                 "s_per_sweep": sec_mt}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
