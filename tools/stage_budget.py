#!/usr/bin/env python3
"""Itemised budget of ONE sweep-kernel launch (GPU box): VALU / SALU / LDS instructions and time per stage, from a -DERM_DIAG_BUILD library whose
launch for sweep 10 returns after stage k while every other launch runs in full (ERM_PASS_STOP=k, ERM_STOP_SWEEP=10: the truncated launch works on a
valid chain state).  Stage k's share = (counters, duration of the launch truncated after k) - (the same after the stage before).
Counters: rocprofv3 --pmc (per dispatch); durations: a separate rocprofv3 --kernel-trace pass (counter collection serialises dispatches).
usage (repo root, GPU box): python3 tools/stage_budget.py [--precisions f64 f32] [--model rtirt] [--out gpurun_out/budget]  ->  <out>/budget_<prec>.json + .md"""
import argparse, csv, glob, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGES = [(30, "head: inputs + reduction of the previous statistics (to the first barrier)"), (31, "head: item draws + staging (to the staging barrier)"),
          (32, "stand-alone staging (no-op in the fused kernel)"), (1, "wave 0: structural chain (beta, Sigma_p); slice set-up"), (5, "row sums over omega, Y, logT"),
          (2, "subject draws theta, zeta (+ traces, structural log-likelihood)"), (3, "Polya-Gamma phase: omega_{t+1} for every cell"),
          (4, "barrier, global statistics, column phase (item statistics + cell log-likelihood)"), (0, "slab row, ticket, group reduction (to the end)")]
PMC = ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY"]


def run(cmd, env, log):
    with open(log, "w") as f:
        subprocess.run(cmd, env=env, stdout=f, stderr=subprocess.STDOUT, check=False, cwd="/tmp")


def fused_rows(path, name_key):
    rows = [r for r in csv.DictReader(open(path)) if ", true, false>" in r[name_key] and "pass_kernel" in r[name_key]]
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--precisions", nargs="+", default=["f64", "f32"]); ap.add_argument("--model", default="rtirt")
    ap.add_argument("--nsubj", type=int, default=100000); ap.add_argument("--nitem", type=int, default=50)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "budget")); ap.add_argument("--sweep", type=int, default=10)
    a = ap.parse_args()
    a.out = os.path.abspath(a.out)
    os.makedirs(a.out, exist_ok=True)
    diag = os.path.join(a.out, "libertirt_diag.so")
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "extendedrtirtmodeling.jl_amd", "csrc", "ertirt.hip"), "-DERM_DIAG_BUILD", "-o", diag], check=True)
    for prec in a.precisions:
        res = []
        for stop, label in STAGES:
            env = dict(os.environ, ERM_LIB_PATH=diag, ERM_PASS_STOP=str(stop), ERM_STOP_SWEEP=str(a.sweep), TMPDIR="/tmp")
            prog = ["python3", os.path.join(ROOT, "tools", "one_chain.py"), "--model", a.model, "--precision", prec, "--nsubj", str(a.nsubj), "--nitem", str(a.nitem), "--sweeps", str(a.sweep + 2)]
            d1, d2 = os.path.join(a.out, f"{prec}_pmc_{stop}"), os.path.join(a.out, f"{prec}_trace_{stop}")
            for d in (d1, d2):
                subprocess.run(["rm", "-rf", d])
            run(["rocprofv3", "--kernel-trace", "--pmc"] + PMC + ["--output-format", "csv", "-d", d1, "--"] + prog, env, d1 + ".log")
            run(["rocprofv3", "--kernel-trace", "--output-format", "csv", "-d", d2, "--"] + prog, env, d2 + ".log")
            e = {"stop": stop, "stage": label}
            f = sorted(glob.glob(d1 + "/*/*counter_collection.csv"))
            if f:
                by = {}
                for r in fused_rows(f[-1], "Kernel_Name"):
                    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
                ids = sorted(by)
                if len(ids) >= a.sweep:
                    e.update(by[ids[a.sweep - 1]])
                    e["full_launch_valu"] = by[ids[a.sweep - 3]].get("SQ_INSTS_VALU")
            f = sorted(glob.glob(d2 + "/*/*kernel_trace.csv"))
            if f:
                rows = sorted(fused_rows(f[-1], "Kernel_Name"), key=lambda r: int(r["Dispatch_Id"]))
                if len(rows) >= a.sweep:
                    r = rows[a.sweep - 1]
                    e["us"] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
                    full = rows[a.sweep - 3]
                    e["full_launch_us"] = (int(full["End_Timestamp"]) - int(full["Start_Timestamp"])) * 1e-3
            res.append(e)
            print(prec, stop, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items() if k not in ("stage",)}, flush=True)
        cells = a.nsubj * a.nitem
        out = {"workload": f"Gibbs {a.model} {a.nsubj} x {a.nitem}, {prec}", "sweep": a.sweep, "stages": []}
        prev = {}
        lines = [f"# Itemised budget of one sweep-kernel launch -- {out['workload']} (sweep {a.sweep} of a chain from bench.py's initial state)", "",
                 "| stage | VALU wave-instr | per cell-update (lane-instr) | SALU | LDS instr | cumulative us | stage us |", "|---|---|---|---|---|---|---|"]
        for e in res:
            row = {"stage": e["stage"], "stop": e["stop"]}
            for k in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM", "us"):
                if k in e:
                    row[k] = e[k]; row["d_" + k] = e[k] - prev.get(k, 0.0); prev[k] = e[k]
            out["stages"].append(row)
            if "SQ_INSTS_VALU" in row:
                lines.append(f"| {e['stage']} | {row['d_SQ_INSTS_VALU']:.0f} | {row['d_SQ_INSTS_VALU'] * 64 / cells:.1f} | {row.get('d_SQ_INSTS_SALU', 0):.0f} | {row.get('d_SQ_INSTS_LDS', 0):.0f} | "
                             f"{row.get('us', float('nan')):.1f} | {row.get('d_us', float('nan')):.1f} |")
        last = res[-1]
        if "SQ_INSTS_VALU" in last:
            lines += ["", f"Whole launch: {last['SQ_INSTS_VALU']:.0f} VALU wave-instructions = {last['SQ_INSTS_VALU'] * 64 / cells:.0f} lane-instructions per cell-update; "
                          f"{last.get('us', float('nan')):.1f} us under rocprofv3 (single launches, no graph).  VALU issue fraction = instr x 4 cycles / (1024 SIMDs x launch cycles at 2.4 GHz) = "
                          f"{last['SQ_INSTS_VALU'] * 4 / (1024 * last.get('us', float('nan')) * 2400):.2f}."]
        json.dump(out, open(os.path.join(a.out, f"budget_{prec}.json"), "w"), indent=1)
        open(os.path.join(a.out, f"budget_{prec}.md"), "w").write("\n".join(lines) + "\n")
        print("\n".join(lines))


if __name__ == "__main__":
    main()
