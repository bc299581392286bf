#!/usr/bin/env python3
"""bench.py -- Gibbs sweeps/s of the MI355X engine on BASELINE.json's workload.

A "step" is ONE Gibbs sweep (every full conditional of one `for m, l` iteration of sample!) over one synthetic data set
that is already resident in HBM.  Default workload: GibbsRtIrt, nSubj=100000, nItem=50, nFeat=3 (BASELINE.json configs[2],
the configuration the north-star target is quoted on); data per setDataRtIrt's distributions, fixed seed.

Headline = the fp64 engine (`dtype: f64`): the reference's arithmetic is Float64 throughout (src/Base.pl.jl:67-78, the Post arrays).
The fp32 fast mode (fp32 cell arithmetic, fp64 accumulation) is measured in the same invocation on the same resident inputs and
carried in the same JSON line as the nested object `fp32`.

N GPUs = N independent chains, one per GPU, no data-path collective: "scaling": "weak".  What is measured at N > 1 is the path a Julia
`sample!(MCMC; devices = 0:N-1)` binds: the library's own chain farm behind the C ABI (erm_farm_create / set_data / run / get_mean: ONE process, one
host thread per chain inside the library, the posterior summaries reduced by the library's own RCCL communicator -- ncclCommInitAll over the N
devices, one ncclAllReduce; `rccl_ranks` is that communicator's ncclCommCount, `gather_ms` the erm_farm_get_mean time).  `python bench.py --gpus N`
runs it directly; under `torch.distributed.run --nproc-per-node N` (the driver's launch) rank 0 drives the farm over the N devices and the other ranks
only take part in the barriers and the MAX-reduction of the time (torch.distributed over gloo: they never touch a GPU).  `--multiprocess` keeps the
older layout (one process per GPU, torch.distributed "nccl", chain_id = rank; `python bench.py --gpus N --multiprocess` spawns its ranks itself).
At N > 1 the line also carries BASELINE.json configs[4] (500000 x 100, nChain = N, one chain per GPU) as `configs4_value` / `configs4_ms_per_step`
and the nested object `configs4`.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline     -- dominant kernel (the fused sweep kernel): algorithmic bytes per launch / mean launch duration measured live with
                  HIP events on the engine's stream, against the 8 TB/s HBM peak;
  cpu_baseline -- the CPU oracle (kind "port": fp64 C restatement of the reference's un-fused schedule, 1 thread) timed on this
                  host for a bounded number of sweeps of the SAME workload.  The reference itself is Julia and cannot run here.
"""
from __future__ import annotations

import argparse
import datetime
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))

# algorithmic bytes per cell-update, Y as 1 byte; matrices in the engine's cell type (fp32: 4 B, fp64: 8 B)
#   MlIrt: Y + omega read + omega write; RtIrt / LatentQr (+ variants): + logT; CrossQr: two passes (omega r, Y, logT, nu r | logT, nu r, nu w, omega w);
#   Cross: omega r/w, Y, logT twice, no nu
ALGO_BYTES = {"f32": {"mlirt": 9, "rtirt": 13, "latentqr": 13, "crossqr": 29, "null": 13, "latent": 13, "cross": 17},
              "f64": {"mlirt": 17, "rtirt": 25, "latentqr": 25, "crossqr": 57, "null": 25, "latent": 25, "cross": 33}}
FAMILY = {"null": "rtirt", "latent": "latentqr", "cross": "crossqr"}
HBM_PEAK_GBS = 8000.0                                                     # MI355X_MICROARCH.md: 8.0 TB/s spec
NAMES = {'mlirt': 'MlIrt', 'rtirt': 'RtIrt', 'latentqr': 'RtIrtLatentQr', 'crossqr': 'RtIrtCrossQr', 'null': 'RtIrtNull', 'cross': 'RtIrtCross', 'latent': 'RtIrtLatent'}


def baseline_config(model, N, J, chains=1):
    """Which entry of BASELINE.json's `configs` a workload is (the label in `config.workload`): '' for a workload BASELINE.json does not list."""
    key = (model, N, J)
    if key == ("mlirt", 1000, 15):
        return "BASELINE.json configs[0]"
    if key == ("mlirt", 100000, 50):
        return "BASELINE.json configs[1]"
    if key == ("rtirt", 100000, 50):
        return "BASELINE.json configs[2]" + (" per GPU" if chains > 1 else "")
    if key == ("latentqr", 100000, 50):
        return "BASELINE.json configs[3]"
    if key == ("rtirt", 500000, 100):
        return "BASELINE.json configs[4]" + ("" if chains == 8 else f": its per-GPU load, {chains} of 8 chains")
    return "not a BASELINE.json configuration"


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--model", default="rtirt", choices=list(ALGO_BYTES["f32"]))
    ap.add_argument("--nsubj", type=int, default=100000)
    ap.add_argument("--nitem", type=int, default=50)
    ap.add_argument("--nfeat", type=int, default=3)
    ap.add_argument("--precision", default="f64", choices=["f32", "f64"], help="engine of the headline value (default: the reference's Float64)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the nested measurements (the fp32 fast mode; at N = 1 also the two chains sharing the GPU): ONE engine runs, "
                                                             "which is what the profiling tools under tools/ want")
    ap.add_argument("--no-configs4", action="store_true", help="at N > 1: skip the nested measurement on configs[4]'s per-GPU load (500000 x 100)")
    ap.add_argument("--trace", default="full", choices=["full", "summary"])
    ap.add_argument("--lanes-per-row", type=int, default=0)
    ap.add_argument("--block-threads", type=int, default=0)
    ap.add_argument("--grid-blocks", type=int, default=0)
    ap.add_argument("--cpu-sweeps", type=int, default=-1, help="oracle sweeps for cpu_baseline (-1 = auto ~15 s, 0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="do not bracket sweep-kernel launches with HIP events")
    ap.add_argument("--clock-warmup-ms", type=float, default=300.0, help="after the W warm-up steps, keep the GPU busy with further UNTIMED sweeps of the same engine for "
                                                                         "this long before the timed region (a few ms of work do not bring an idle MI355X to its sustained clock; 0 = off)")
    ap.add_argument("--shard-exchange", default="rccl", choices=["rccl", "callback"], help="--shard: in-stream RCCL all-gather (default) or the "
                                                        "host callback over torch.distributed")
    ap.add_argument("--shard", action="store_true", help="NOT the headline: ONE chain of --nsubj subjects sharded over the ranks (strong scaling; "
                                                        "one all-gather of a statistics row per row pass, SURVEY.md 8(e))")
    ap.add_argument("--multiprocess", action="store_true", help="N > 1: one process per GPU over torch.distributed instead of the library's in-process chain farm")
    ap.add_argument("--no-self-check", action="store_true", help="N > 1: skip the farm's self-validation (communicator size, farm mean = separately run engines, per-chain time)")
    ap.add_argument("--no-two-chains", action="store_true", help="N = 1: skip the nested measurement of TWO independent chains sharing the one GPU (the chain farm on devices [0, 0])")
    ap.add_argument("--no-cold", action="store_true", help="skip the extra cold measurement (the same W + K steps from an idle device, before the clock warm-up)")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher: `python bench.py --gpus N` without torch.distributed.run.  Runs in a process that has imported neither torch nor the HIP
# library (a process that has initialised the GPU must never exec or be replaced; this one only waits for its children).
# ---------------------------------------------------------------------------------------------------------------------
def spawn_ranks(n, argv):
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ERM_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # rank 0's output is collected by a reader thread while every child is polled: a rank that dies before or at the rendezvous must not leave
    # the others (and this parent) waiting for torch's own time-out
    import threading
    import time as _time
    chunks = []
    th = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    th.start()
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        _time.sleep(0.1)
    for p in procs:
        try:
            p.wait(timeout=30)
        except subprocess.TimeoutExpired:
            p.kill()
    th.join(timeout=30)
    out = "".join(c or "" for c in chunks)
    rcs = [p.returncode for p in procs]
    line = None
    for ln in (out or "").splitlines():
        if ln.startswith('{"metric"'):
            line = ln
        else:
            print(ln, file=sys.stderr)
    if any(rcs) or line is None:
        print(f"bench.py: ranks exited with {rcs}" + ("" if line else "; rank 0 printed no result line"), file=sys.stderr)
        return max([abs(rc) for rc in rcs] + [1])
    print(line, flush=True)
    return 0


class _OneLineStdout:
    """Everything this process -- or a library inside it: RCCL prints a version banner to stdout when its first communicator is made -- writes to
    stdout goes to stderr instead; emit() writes the ONE JSON line to the real stdout."""

    def __init__(self):
        sys.stdout.flush()
        self.real = os.dup(1)
        os.dup2(2, 1)

    def emit(self, line):
        sys.stdout.flush()
        os.write(self.real, (line + "\n").encode())


def make_data(pkg, model, N, J, F, seed):
    import numpy as np
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
    import numpy as np
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


def offline_counters(model, N, J, precision):
    """PMC counters cannot be read from inside this process: they are collected by tools/collect_profiles.sh in separate rocprofv3
    --pmc passes of this same command and kept, with the gfx950 FETCH_SIZE correction calibrated on a known byte count, in
    profiles/traffic.json.  Returned (labelled as offline) only when that file describes exactly this workload and precision."""
    try:
        e = json.load(open(os.path.join(ROOT, "profiles", "traffic.json"))).get(f"{model}:{N}x{J}:{precision}")
    except Exception:
        return None
    return e


def measure(pkg, ge_mod, torch, dist, args, model, N, J, F, precision, data, st, rank, world, local_rank, rehearse, shard, trace, truth=None):
    """warm-up + timed region of one engine on resident inputs; returns (dt seconds [max over ranks], timing dict, engine, n_loc)."""
    L = pkg._lib
    Y, logT, X = data if data is not None else (None, None, True)
    rows = args.warmup + args.steps
    lo, n_loc = (0, N)
    if shard:       # every rank builds the same data set and keeps its rows
        lo, n_loc = pkg.parallel.shard_rows(N, world)[rank]
        Y, logT, X = Y[lo:lo + n_loc], (None if logT is None else logT[lo:lo + n_loc]), (None if X is None else X[lo:lo + n_loc])
        st = dict(st, **{k: st[k][lo:lo + n_loc] for k in ("theta", "zeta") if k in st})
    eng = L.Engine(model=getattr(L, "MODEL_" + model.upper()), n_item=J, n_subj=n_loc, n_feat=0 if X is None else F, n_iter=rows, n_chain=1,
                   n_burnin=args.warmup, cov2one=int(model not in ("latentqr", "latent")), q_rt=0.85, seed=1234, chain_id=0 if shard else pkg.parallel.rank_chain_id(rank), device=local_rank,
                   precision=L.PREC_F32 if precision == "f32" else L.PREC_F64,
                   trace_mode=L.TRACE_FULL if trace == "full" else L.TRACE_SUMMARY, lanes_per_row=args.lanes_per_row,
                   block_threads=args.block_threads, grid_blocks=args.grid_blocks, profile=0 if args.no_profile else 1)
    coll_dev = "cpu" if rehearse else f"cuda:{local_rank}"
    if shard and (rehearse or args.shard_exchange == "callback"):
        eng.set_shard(rank, world, N, lo, pkg.parallel.TorchExchange(L.load(), device=None if rehearse else f"cuda:{local_rank}"))
    elif shard:     # the library's own in-stream RCCL all-gather; the 128-byte id travels over the process group
        box = [L.rccl_unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        eng.set_shard_rccl(rank, world, N, lo, box[0])
    if truth is not None:          # the data set is generated ON the device (erm_simulate_data: setDataRtIrt's distributions, data seed 1234)
        eng.simulate_data(seed=1234, **truth)
    else:
        eng.set_data(Y, logT, X)   # inputs resident in HBM from here on
    eng.set_state(**{("lambda_" if k == "lam" else k): v for k, v in st.items()})

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # 1. cold: the W warm-up steps and K timed steps from an idle device -- what `--warmup W` alone produces (reported as ms_per_step_cold)
    dt_cold = None
    if not args.no_cold and args.clock_warmup_ms > 0:
        if args.warmup > 0:
            eng.run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        eng.run(args.steps)
        barrier()
        dt_cold = time.perf_counter() - t0
        eng.reset_trace()
    # 2. untimed clock warm-up: the chain simply continues (its trace rows are recycled) until the device has been busy for --clock-warmup-ms
    spun = 0
    if args.clock_warmup_ms > 0:
        t_end = time.perf_counter() + args.clock_warmup_ms * 1e-3
        chunk = args.steps if args.steps <= min(rows, 64) else min(rows, 64)      # (calls of the timed length where that is short: the library replays a whole short call from ONE graph it builds on the first call of each length)
        while time.perf_counter() < t_end:
            eng.reset_trace()
            eng.run(chunk)
            spun += chunk
        eng.reset_trace()
    # 3. the W warm-up steps (= the burn-in rows), then EXACTLY K timed steps, all of them post-burn-in
    if args.warmup > 0:
        eng.run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    eng.run(args.steps)            # erm_run returns after hipStreamSynchronize on the engine's stream
    barrier()
    dt = time.perf_counter() - t0
    tm = eng.timing()
    tm["clock_warmup_sweeps"] = spun
    if dist is not None:
        t = torch.tensor([dt, dt_cold if dt_cold is not None else 0.0], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        dt_cold = float(t[1].item()) if dt_cold is not None else None
    tm["dt_cold"] = dt_cold
    return dt, tm, eng, n_loc


def measure_farm(pkg, torch, dist, args, model, N, J, F, precision, data, n_dev, devices, trace, flags=0, truth=None, rank=0):
    """The N > 1 measurement: the library's chain farm (erm_farm_*), driven by THIS process -- chain l on device devices[l], one host thread per
    chain inside the library.  Ranks other than 0 (under torch.distributed.run) only join the barriers.  Returns (dt, dt_cold, farm timing dict,
    chain 0's engine timing, farm) on the driving rank, (dt, dt_cold, None, None, None) elsewhere."""
    rows = args.warmup + args.steps
    driver = rank == 0
    L = pkg._lib if driver else None

    def barrier():
        if dist is not None:
            dist.barrier()
        if driver and torch.cuda.is_available():      # the other ranks have no GPU work (and never create a context)
            for d in sorted(set(devices)):
                torch.cuda.synchronize(d)

    farm = None
    if driver:
        farm = L.Farm(devices, model=getattr(L, "MODEL_" + model.upper()), n_item=J, n_subj=N, n_feat=F if model not in ("crossqr", "cross") else 0, n_iter=rows, n_chain=1,
                      n_burnin=args.warmup, cov2one=int(model not in ("latentqr", "latent")), q_rt=0.85, seed=1234, precision=L.PREC_F32 if precision == "f32" else L.PREC_F64,
                      trace_mode=L.TRACE_FULL if trace == "full" else L.TRACE_SUMMARY, lanes_per_row=args.lanes_per_row, block_threads=args.block_threads,
                      grid_blocks=args.grid_blocks, profile=0 if args.no_profile else 1, flags=flags)
        if truth is not None:      # generated ON each device from the same data seed: every chain sees the same data set
            for l in range(n_dev):
                farm.engine(l).simulate_data(seed=1234, **truth)
        else:
            farm.set_data(*data)   # inputs resident in HBM from here on (every chain its own copy, uploaded by the chain's host thread)
        for l in range(n_dev):     # every chain its own setInitialValues
            farm.set_state(l, **{("lambda_" if k == "lam" else k): v for k, v in init_state(model, N, J, F, l).items()})
    dt_cold = None
    if not args.no_cold and args.clock_warmup_ms > 0:
        if driver and args.warmup > 0:
            farm.run(args.warmup)
        barrier()
        t0 = time.perf_counter()
        if driver:
            farm.run(args.steps)
        barrier()
        dt_cold = time.perf_counter() - t0
        if driver:
            farm.reset_trace()
    spun = 0
    if driver and args.clock_warmup_ms > 0:
        t_end = time.perf_counter() + args.clock_warmup_ms * 1e-3
        chunk = args.steps if args.steps <= min(rows, 64) else min(rows, 64)
        while time.perf_counter() < t_end:
            farm.reset_trace()
            farm.run(chunk)
            spun += chunk
        farm.reset_trace()
    if driver and args.warmup > 0:
        farm.run(args.warmup)
    barrier()
    t0 = time.perf_counter()
    if driver:
        farm.run(args.steps)       # erm_farm_run returns when every chain's stream has been synchronised
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt, dt_cold if dt_cold is not None else 0.0], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0].item())
        dt_cold = float(t[1].item()) if dt_cold is not None else None
    if not driver:
        return dt, dt_cold, None, None, None
    ftm = farm.timing()
    ftm["clock_warmup_sweeps"] = spun
    etm = farm.engine(0).timing()
    return dt, dt_cold, ftm, etm, farm


def roofline(model, N, J, n_loc, precision, tm):
    if tm["pass_launches"] <= 0:
        return None
    per_launch_s = tm["pass_ms_total"] / tm["pass_launches"] * 1e-3
    launches_per_sweep = 2 if model in ("crossqr", "cross") else 1
    algo = ALGO_BYTES[precision][model] * float(n_loc) * J / launches_per_sweep        # a launch covers this rank's subjects
    ach = algo / per_launch_s / 1e9
    off = offline_counters(model, N, J, precision)
    out = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
           "traffic": None if off is None else off.get("traffic_bytes_per_launch"),
           "traffic_source": None if off is None else f"offline rocprofv3 --pmc passes of this command ({off.get('source', 'profiles/traffic.json')}); not measured in this run",
           "kernel": f"pass_kernel<{model}, {'float' if precision == 'f32' else 'double'}> (one launch per sweep: tiny step + fused row pass; Cross family: one of its two row passes)",
           "frac_of_achievable": ach / 6300.0, "achievable_peak": 6300.0, "launch_us": per_launch_s * 1e6, "event_overhead_us": tm["event_overhead_ms"] * 1e3,
           "algorithmic_bytes_per_cell_update": ALGO_BYTES[precision][model], "algorithmic_bytes_per_launch": algo, "launches_timed": int(tm["pass_launches"]),
           "launch_us_source": "HIP events on the engine's stream around sweep-kernel launches of the timed region (live)"}
    # what the same launch time is worth against other yardsticks: the fp32 layout SURVEY.md 8(d) quotes first (13 B per cell-update for GibbsRtIrt) when the engine
    # stores fp64 (25 B: the representation this engine chose, not a property of the algorithm), and where the bytes come from at this size
    ws = float(n_loc) * J * {"mlirt": 2, "rtirt": 3, "latentqr": 3, "null": 3, "latent": 3, "cross": 3, "crossqr": 4}[model] * (8 if precision == "f64" else 4)
    out["working_set_bytes"] = ws
    out["infinity_cache_resident"] = bool(ws <= 256 * 2 ** 20)
    if precision == "f64":
        out["frac_fp32_layout"] = ALGO_BYTES["f32"][model] * float(n_loc) * J / launches_per_sweep / per_launch_s / 1e9 / HBM_PEAK_GBS
    out["yardstick_note"] = ("frac prices the engine's own representation (fp64 engine: 25 B per cell-update; frac_fp32_layout prices the same launch at SURVEY.md 8(d)'s fp32 figure). "
                             + ("The matrices of this workload fit the 256 MiB Infinity Cache, so most of these bytes never reach HBM: the fraction of the HBM peak is nominal."
                                if ws <= 256 * 2 ** 20 else "The matrices of this workload exceed the 256 MiB Infinity Cache: the bytes come from HBM."))
    if off is not None and off.get("valu") is not None:
        out["valu"] = dict(off["valu"], source="offline rocprofv3 SQ counters (profiles/), not measured in this run")
        # what actually limits the kernel: VALU issue in its Polya-Gamma and column phases (profiles/round3_budget_*.md itemises it stage by stage)
        out["bound_actual"] = "valu"
        out["valu_issue_frac"] = off["valu"].get("valu_busy_frac")
        if off["valu"].get("valu_busy_frac_weighted") is not None:      # the same busy fraction with every instruction class at its MEASURED issue cost (profiles/round4_valu_issue_rates.txt)
            out["valu_issue_frac_class_weighted"] = off["valu"]["valu_busy_frac_weighted"]
        out["bound_note"] = ("nominally HBM-bound (SURVEY.md 8(d)); measured: the VALUs are busy for valu_issue_frac of the kernel's cycles (offline SQ counters), the rest is the "
                             "latency of its serial head / tail; HBM-side traffic is far from the 8 TB/s peak")
    return out


def farm_self_check(pkg, L, devices, rehearse, prec, headline_per_chain_ms, args, model, N, J, F, data):
    """Makes a multi-GPU run validate itself: (1) the library's RCCL communicator spans the farm's distinct devices; (2) the farm's
    Post.mean is the count-weighted mean of the same chains run as SEPARATE engines (one per device, chain_id = l) to 1e-12 -- the comparison
    tests/test_gpu_farm.py makes on one device; (3) every chain's device time per sweep is within 10 % of a single engine's on device 0 (no chain is slowed down
    by its neighbours: the farm's chains never communicate while sampling).  Returns a dict with `ok`."""
    import numpy as np
    n_dev = len(devices)
    Nv, Jv, Fv, rows, burn = 20000, 50, F, 24, 8
    Yv, lv, Xv = make_data(pkg, model, Nv, Jv, Fv, seed=4321)
    kw = dict(model=getattr(L, "MODEL_" + model.upper()), n_item=Jv, n_subj=Nv, n_feat=Fv if Xv is not None else 0, n_iter=rows, n_chain=1, n_burnin=burn,
              cov2one=int(model not in ("latentqr", "latent")), q_rt=0.85, seed=1234, precision=L.PREC_F32 if prec == "f32" else L.PREC_F64, trace_mode=L.TRACE_SUMMARY)
    farm = L.Farm(devices, flags=L.FLAG_FARM_FORCE_RCCL if rehearse else 0, **kw)
    farm.set_data(Yv, lv, Xv)
    states = [{("lambda_" if k == "lam" else k): v for k, v in init_state(model, Nv, Jv, Fv, l).items()} for l in range(n_dev)]
    for l in range(n_dev):
        farm.set_state(l, **states[l])
    farm.run(rows)
    fm = farm.get_mean()
    ft = farm.timing()
    tot = farm.post_count
    del farm
    acc, cnt = None, 0
    for l in range(n_dev):
        eng = L.Engine(chain_id=l, device=devices[l], **kw)
        eng.set_data(Yv, lv, Xv)
        eng.set_state(**states[l])
        eng.run(rows)
        m, c = eng.get_mean(), eng.post_count
        vec = np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for k, v in sorted(m.items()) if v is not None])
        acc = vec * c if acc is None else acc + vec * c
        cnt += c
        del eng
    want = acc / cnt
    got = np.concatenate([np.asarray(v, dtype=np.float64).reshape(-1) for k, v in sorted(fm.items()) if v is not None])
    err = float(np.max(np.abs(got - want) / (1.0 + np.abs(want))))
    chk = {"rccl_ranks": int(ft["rccl_ranks"]), "rccl_ranks_expected": 1 if rehearse else len(set(devices)), "farm_mean_vs_separate_engines_max_rel_err": err, "post_rows": int(tot),
           "post_rows_expected": int(cnt)}
    ok = chk["rccl_ranks"] == chk["rccl_ranks_expected"] and err <= 1e-12 and tot == cnt
    # (3) a single engine on device 0 on the headline workload, same steps and warm-up, device time per sweep
    if not rehearse and headline_per_chain_ms:
        a1 = argparse.Namespace(**vars(args))
        a1.no_cold = True                    # the same untimed clock warm-up as the farm's chains had
        import torch
        singles, worst = [], None
        for _ in range(2):                   # one repeat before a timing verdict: a 20-step device time moves by a few per cent with the clock
            dt1, tm1, eng1, _ = measure(pkg, None, torch, None, a1, model, N, J, F, prec, data, init_state(model, N, J, F, 0), 0, 1, devices[0], False, False, args.trace)
            del eng1
            singles.append(tm1["run_ms"] / args.steps)
            w = max(abs(v / singles[-1] - 1.0) for v in headline_per_chain_ms)
            worst = w if worst is None else min(worst, w)
            if worst <= 0.10:
                break
        single = singles[-1]
        chk.update(single_engine_device_ms_per_step=single, single_engine_runs=singles, per_chain_device_ms_per_step=[float(v) for v in headline_per_chain_ms],
                   worst_relative_deviation=worst)
        ok = ok and worst <= 0.10
    chk["ok"] = bool(ok)
    return chk


def main_farm(args, world_env):
    """`--gpus N` (N > 1): the library's chain farm.  One driving process (rank 0, or the only process); see the module docstring."""
    import numpy as np
    out_line = _OneLineStdout()
    sys.path.insert(0, ROOT)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    rank = int(os.environ.get("RANK", "0"))
    world = int(world_env or "1")
    n_dev = args.gpus
    if world_env is not None and world != n_dev:
        raise SystemExit(f"bench.py: --gpus {n_dev} but WORLD_SIZE={world}")
    rehearse = os.environ.get("ERM_BENCH_REHEARSE") == "1"      # the N > 1 path on a ONE-GPU box: every chain on device 0, the reduction over a one-rank RCCL communicator
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=600))       # barriers and one MAX only: the data path has no torch collective
    if os.environ.get("ERM_BENCH_LAUNCH_ONLY") == "1":
        tot = None
        if dist is not None:
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t)
            tot = float(t.item())
        if rank == 0:
            out_line.emit(json.dumps({"metric": "launcher self-test", "mode": "farm", "n_gpus": n_dev, "ranks": world, "rank_sum": tot, "backend": "gloo" if dist is not None else None,
                                      "steps": args.steps}))
        if dist is not None:
            dist.destroy_process_group()
        return
    driver = rank == 0
    rc = 0
    pkg = L = None
    data = None
    model, N, J, F = args.model, args.nsubj, args.nitem, args.nfeat
    prec = args.precision
    devices = [0] * n_dev if rehearse else list(range(n_dev))
    flags = 0
    if driver:
        import __graft_entry__ as ge
        ge.build_hip()
        pkg = ge.load_package()
        L = pkg._lib
        data = make_data(pkg, model, N, J, F, seed=1234)
        flags = L.FLAG_FARM_FORCE_RCCL if rehearse else 0
    common = dict(pkg=pkg, torch=torch, dist=dist, args=args, n_dev=n_dev, devices=devices, flags=flags, rank=rank)
    dt, dt_cold, ftm, etm, farm = measure_farm(model=model, N=N, J=J, F=F, precision=prec, data=data, trace=args.trace, **common)
    gather = None
    if driver:      # Post.mean over iterations and chains: the ONE collective of the path (the library's RCCL all-reduce)
        farm.get_mean()
        g = farm.timing()
        gather = {"gather_ms": g["gather_ms"], "allreduce_ms": g["allreduce_ms"], "comm_init_ms": g["comm_init_ms"], "rccl_ranks": g["rccl_ranks"], "n_devices": g["n_devices"], "used_rccl": farm.used_rccl,
                  "post_count": farm.post_count}
        del farm
    fp32 = None
    if prec == "f64" and not args.no_fp32:
        dt32, _, _, etm32, farm32 = measure_farm(model=model, N=N, J=J, F=F, precision="f32", data=data, trace=args.trace, **common)
        if driver:
            del farm32
            fp32 = {"value": float(N) * J * args.steps * n_dev / dt32, "unit": "cell-updates/s", "ms_per_step": dt32 / args.steps * 1e3, "dtype": "f32",
                    "note": "fp32 cell arithmetic, fp64 accumulation and item-level draws; same workload, steps and warm-up", "roofline": roofline(model, N, J, N, "f32", etm32)}
    # BASELINE.json configs[4]: GibbsRtIrt 500000 x 100, nChain = N, one chain per GPU
    cfg4 = None
    if not args.no_configs4 and model == "rtirt" and (N, J) != (500000, 100):
        N4, J4 = (20000, 100) if rehearse else (500000, 100)
        a4 = argparse.Namespace(**vars(args))
        a4.steps, a4.warmup = min(args.steps, 100), min(args.warmup, 10)
        truth4 = None
        if driver:  # 5e7 cells per chain: generated on the device from setTrueParaRtIrt's item / structural truth
            tp4 = pkg.setTrueParaRtIrt(pkg.setCond(nSubj=N4, nItem=J4, nFeat=F, nIter=10, nChain=1), seed=np.random.default_rng(1234))
            truth4 = dict(a=tp4.a, b=tp4.b, lambda_=tp4.lam, sig2t=tp4.sig2t, sigp=np.asarray(tp4.Sigp, dtype=np.float64).reshape(-1, order="F"),
                          beta=np.asarray(tp4.beta, dtype=np.float64).reshape(-1, order="F"))
        dt4, dt4c, ftm4, etm4, farm4 = measure_farm(model=model, N=N4, J=J4, F=F, precision=prec, data=None, trace="summary", truth=truth4, **dict(common, args=a4))
        if driver:
            farm4.get_mean()
            g4 = farm4.timing()
            del farm4
            cfg4 = {"workload": f"GibbsRtIrt nSubj={N4} nItem={J4} nFeat={F} nChain={n_dev}, one chain per GPU (BASELINE.json configs[4]), summary traces",
                    "data": "synthetic, generated on the device (erm_simulate_data)", "value": float(N4) * J4 * a4.steps * n_dev / dt4, "unit": "cell-updates/s",
                    "ms_per_step": dt4 / a4.steps * 1e3, "ms_per_step_cold": None if dt4c is None else dt4c / a4.steps * 1e3, "steps": a4.steps, "warmup": a4.warmup, "dtype": prec,
                    "per_chain_device_ms_per_step": [float(v) / a4.steps for v in ftm4["run_ms"]], "gather_ms": g4["gather_ms"], "allreduce_ms": g4["allreduce_ms"],
                    "roofline": roofline(model, N4, J4, N4, prec, etm4)}
    if driver:
        cells = float(N) * J
        out = {
            "metric": "Gibbs cell-updates/s (nSubj x nItem x sweeps/s)", "value": cells * args.steps * n_dev / dt, "unit": "cell-updates/s",
            "n_gpus": n_dev, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": prec, "data": "synthetic",
            "config": {"workload": f"Gibbs{NAMES[model]} nSubj={N} nItem={J} nFeat={F} nChain={n_dev}, one chain per GPU ({baseline_config(model, N, J, n_dev)})",
                       "path": "erm_farm_create / set_data / set_state / run / get_mean (the C-ABI chain farm: one process, one host thread per chain inside the library)",
                       "chains": n_dev, "devices": devices, "subject_shards": 1, "trace": args.trace, "lanes_per_row": etm["lanes_per_row"], "block_threads": etm["block_threads"],
                       "grid_blocks": etm["grid_blocks"], "lds_bytes": etm["lds_bytes"], "ranks": world,
                       "collective_backend": "RCCL inside libertirt.so (ncclCommInitAll over the farm's devices, one ncclAllReduce in erm_farm_get_mean)"
                                             + ("; torch.distributed gloo for the launcher's barriers only" if dist is not None else ""),
                       "rccl_ranks": gather["rccl_ranks"]},
            "sweeps_per_s": args.steps * n_dev / dt, "farm_run_wall_ms_per_step": ftm["run_wall_ms"] / args.steps,
            "per_chain_device_ms_per_step": [float(v) / args.steps for v in ftm["run_ms"]],
            "gather_ms": gather["gather_ms"], "allreduce_ms": gather["allreduce_ms"], "gather": gather,
            "untimed_clock_warmup": {"ms": args.clock_warmup_ms, "sweeps": ftm["clock_warmup_sweeps"],
                                     "note": "untimed sweeps of the same chains BEFORE the W warm-up steps; ms_per_step_cold is the same W + K steps without them"},
        }
        if dt_cold is not None:
            out["ms_per_step_cold"] = dt_cold / args.steps * 1e3
        rf = roofline(model, N, J, N, prec, etm)
        if rf is not None:
            out["roofline"] = dict(rf, note="chain 0's sweep kernel (every chain runs the same kernel on its own device)")
        if fp32 is not None:
            out["fp32"] = fp32
        if cfg4 is not None:
            out["configs4_value"], out["configs4_ms_per_step"], out["configs4"] = cfg4["value"], cfg4["ms_per_step"], cfg4
        if not args.no_self_check:
            out["self_check"] = farm_self_check(pkg, L, devices, rehearse, prec, out["per_chain_device_ms_per_step"], args, model, N, J, F, data)
            rc = 0 if out["self_check"]["ok"] else 3
        out_line.emit(json.dumps(out))
    if dist is not None:
        # every rank leaves with the driver's verdict (a failed self-check must fail the whole launch)
        t = torch.tensor([float(rc)], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        rc = int(t.item())
        dist.destroy_process_group()
    if rc:
        print("bench.py: the multi-GPU self-check failed (see self_check in the JSON line)", file=sys.stderr)
        sys.exit(rc)


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and not args.multiprocess and not args.shard:
        return main_farm(args, world_env)
    if args.gpus > 1 and world_env is None:
        # not under torch.distributed.run: make the ranks ourselves (nothing GPU-related has been imported in this process)
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))

    import numpy as np  # noqa: F401
    out_line = _OneLineStdout()
    sys.path.insert(0, ROOT)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")     # the oracle's OpenMP threads must not spin on a shared host
    import __graft_entry__ as ge

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    if os.environ.get("ERM_BENCH_DIE_RANK") == str(rank):      # launcher self-test: this rank dies before the rendezvous
        raise SystemExit(3)
    rehearse = os.environ.get("ERM_BENCH_REHEARSE") == "1"
    import torch
    dist = None
    backend = None
    if world > 1 or os.environ.get("ERM_BENCH_FORCE_DIST") == "1":     # FORCE_DIST: exercise the RCCL calls with one rank on a one-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearse:     # ERM_BENCH_REHEARSE=1: the N>1 code path on a ONE-GPU box -- every rank on cuda:0, collectives over gloo on the CPU
            local_rank = 0
            backend = "gloo"
            dist.init_process_group("gloo", rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
        else:
            backend = "nccl"
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=300))
        assert dist.get_world_size() == world

    if os.environ.get("ERM_BENCH_LAUNCH_ONLY") == "1":
        # launcher self-test (tests/test_bench_launcher.py, runs without a GPU): the ranks exist, rendezvous and reduce; no engine is created
        tot = None
        if dist is not None:
            t = torch.tensor([float(rank + 1)], dtype=torch.float64)
            dist.all_reduce(t)
            tot = float(t.item())
        if rank == 0:
            out_line.emit(json.dumps({"metric": "launcher self-test", "n_gpus": world, "ranks": world, "rank_sum": tot, "backend": backend, "steps": args.steps}))
        if dist is not None:
            dist.destroy_process_group()
        return

    ge.build_hip()
    pkg = ge.load_package()
    model, N, J, F = args.model, args.nsubj, args.nitem, args.nfeat
    data = make_data(pkg, model, N, J, F, seed=1234)
    shard = args.shard and dist is not None
    st = init_state(model, N, J, F, 0 if shard else rank)

    common = dict(pkg=pkg, ge_mod=ge, torch=torch, dist=dist, args=args, rank=rank, world=world, local_rank=local_rank, rehearse=rehearse, shard=shard)
    prec = args.precision
    dt, tm, eng, n_loc = measure(model=model, N=N, J=J, F=F, precision=prec, data=data, st=st, trace=args.trace, **common)

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
    del eng

    # the fp32 fast mode on the same resident-input protocol (every rank takes part: the barriers are collective)
    fp32 = None
    if prec == "f64" and not args.no_fp32 and not shard:
        dt32, tm32, eng32, _ = measure(model=model, N=N, J=J, F=F, precision="f32", data=data, st=st, trace=args.trace, **common)
        del eng32
        fp32 = {"value": float(N) * J * args.steps * world / dt32, "unit": "cell-updates/s", "ms_per_step": dt32 / args.steps * 1e3, "dtype": "f32",
                "note": "fp32 cell arithmetic, fp64 accumulation and item-level draws; same workload, steps and warm-up",
                "roofline": roofline(model, N, J, N, "f32", tm32)}

    # nChain = 2 on the ONE GPU: two independent chains of the C-ABI farm on devices [0, 0] (a host thread and a stream each).  One chain leaves the SIMDs idle
    # in its serial head, its subject draws and its tail; a second chain's launches fill them.  NOT the headline (`value` is one chain): the reference's default
    # is nChain = 4 (src/Base.pl.jl:59), so this is what a user's sample!(MCMC; devices=[0, 0]) gets per GPU.
    two = None
    if world == 1 and dist is None and not args.no_two_chains and not args.no_fp32 and not shard:
        a2 = argparse.Namespace(**vars(args))
        a2.no_cold = True
        dt2, _, ftm2, etm2, farm2 = measure_farm(pkg=pkg, torch=torch, dist=None, args=a2, n_dev=2, devices=[local_rank, local_rank], flags=0, rank=0, model=model, N=N, J=J, F=F,
                                                 precision=prec, data=data, trace=args.trace)
        del farm2
        two = {"workload": f"Gibbs{NAMES[model]} nSubj={N} nItem={J} nFeat={F} nChain=2, both chains on this GPU (erm_farm_create with devices [0, 0])",
               "value": float(N) * J * args.steps * 2 / dt2, "unit": "cell-updates/s", "ms_per_step": dt2 / args.steps * 1e3, "us_per_chain_sweep": dt2 / args.steps * 1e6 / 2,
               "dtype": prec, "steps": args.steps, "warmup": args.warmup, "per_chain_device_ms_per_step": [float(v) / args.steps for v in ftm2["run_ms"]],
               "block_threads": etm2["block_threads"], "grid_blocks": etm2["grid_blocks"],
               "note": "a step = one sweep of EACH chain; same kernels, same geometry, the two chains' launches overlap on the device"}

    # configs[4]'s per-GPU load, one chain per GPU (BASELINE.json: GibbsRtIrt 500000 x 100, nChain = 8 over 8 GPUs)
    cfg4 = None
    if world > 1 and not shard and not args.no_configs4 and model == "rtirt" and (N, J) != (500000, 100):
        N4, J4 = (20000, 100) if rehearse else (500000, 100)
        a4 = argparse.Namespace(**vars(args))
        a4.steps, a4.warmup = min(args.steps, 100), min(args.warmup, 10)
        # 5e7 cells: generated on the device from setTrueParaRtIrt's item / structural truth (host generation would take half a minute per rank)
        import numpy as np
        tp4 = pkg.setTrueParaRtIrt(pkg.setCond(nSubj=N4, nItem=J4, nFeat=F, nIter=10, nChain=1), seed=np.random.default_rng(1234))
        truth4 = dict(a=tp4.a, b=tp4.b, lambda_=tp4.lam, sig2t=tp4.sig2t, sigp=np.asarray(tp4.Sigp, dtype=np.float64).reshape(-1, order="F"),
                      beta=np.asarray(tp4.beta, dtype=np.float64).reshape(-1, order="F"))
        st4 = init_state(model, N4, J4, F, rank)
        dt4, tm4, eng4, _ = measure(model=model, N=N4, J=J4, F=F, precision=prec, data=None, st=st4, trace="summary", truth=truth4, **dict(common, args=a4))
        del eng4
        cfg4 = {"workload": f"GibbsRtIrt nSubj={N4} nItem={J4} nFeat={F}, one chain per GPU (BASELINE.json configs[4] per-GPU load), summary traces",
                "data": "synthetic, generated on the device (erm_simulate_data)",
                "value": float(N4) * J4 * a4.steps * world / dt4, "unit": "cell-updates/s", "ms_per_step": dt4 / a4.steps * 1e3, "steps": a4.steps, "warmup": a4.warmup,
                "dtype": prec, "roofline": roofline(model, N4, J4, N4, prec, tm4)}

    if rank == 0:
        cells = float(N) * J
        value = cells * args.steps * (1 if shard else world) / dt
        out = {
            "metric": "Gibbs cell-updates/s (nSubj x nItem x sweeps/s)", "value": value, "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "strong" if shard else "weak", "vs_baseline": None, "dtype": prec, "data": "synthetic",
            "config": {"workload": f"Gibbs{NAMES[model]} nSubj={N} nItem={J} nFeat={F} " + (f"ONE chain, subjects sharded over {world} devices (not a BASELINE.json configuration)" if shard else f"nChain=1 per GPU ({baseline_config(model, N, J, world)})"),
                       "chains": 1 if shard else world, "subject_shards": world if shard else 1, "shard_exchange": (("callback" if rehearse else args.shard_exchange) if shard else None), "trace": args.trace, "lanes_per_row": tm["lanes_per_row"], "block_threads": tm["block_threads"],
                       "grid_blocks": tm["grid_blocks"], "lds_bytes": tm["lds_bytes"], "ranks": world, "collective_backend": backend,
                       "rccl_ranks": world if backend == "nccl" else 0},
            "sweeps_per_s": args.steps * (1 if shard else world) / dt, "device_ms_per_step": tm["run_ms"] / args.steps,
            "untimed_clock_warmup": {"ms": args.clock_warmup_ms, "sweeps": tm["clock_warmup_sweeps"],
                                     "note": "untimed sweeps of the same chain BEFORE the W warm-up steps (a few ms of work do not bring an idle MI355X to its sustained clock); "
                                             "ms_per_step_cold is the same W + K steps without them"},
        }
        if tm.get("dt_cold") is not None:
            out["ms_per_step_cold"] = tm["dt_cold"] / args.steps * 1e3
        if gather_ms is not None:
            out["gather_ms"] = gather_ms
        rf = roofline(model, N, J, n_loc, prec, tm)
        if rf is not None:
            out["roofline"] = rf
        if fp32 is not None:
            out["fp32"] = fp32
        if cfg4 is not None:
            out["configs4"] = cfg4
        if two is not None:
            out["two_chains_one_gpu"] = two
        ncpu = args.cpu_sweeps
        if ncpu != 0 and world == 1:          # the CPU baseline is reported at N=1 only
            Y, logT, X = data
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
        out_line.emit(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
