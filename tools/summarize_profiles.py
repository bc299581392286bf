#!/usr/bin/env python3
"""Turns gpurun_out/<tag>/ (tools/collect_profiles.sh) into profiles/<tag>_*.csv, profiles/<tag>_summary.md and profiles/traffic.json."""
import collections, csv, glob, json, os, shutil, sys

tag = sys.argv[1] if len(sys.argv) > 1 else "round2"
prec = sys.argv[2] if len(sys.argv) > 2 else "f64"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out", f"{tag}_{prec}"), os.path.join(root, "profiles")
tag = f"{tag}_{prec}"
BYTES = {"f32": 13.0, "f64": 25.0}[prec]            # GibbsRtIrt algorithmic bytes per cell-update
CELL = {"f32": 4.0, "f64": 8.0}[prec]
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = sorted(glob.glob(os.path.join(src, pattern)), key=os.path.getmtime)      # gpurun merges runs: take the newest
    return f[-1] if f else None


def counters(path, kernel=", true, false>"):      # the fused sweep kernel pass_kernel<MODEL, real, 0, true, false> (the two non-fused launches are the prologue)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if kernel in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v[2:]) / max(1, len(v[2:])) for k, v in agg.items()}


stats = one("stats/*/*_kernel_stats.csv")
shutil.copy(stats, os.path.join(dst, f"{tag}_kernel_stats.csv"))
rows = list(csv.DictReader(open(stats)))
for name, pat in (("fetch", "fetch/*/*counter_collection.csv"), ("write", "write/*/*counter_collection.csv"), ("fetch_cal", "fetch_cal/*/*counter_collection.csv"), ("sq", "sq/*/*counter_collection.csv")):
    f = one(pat)
    if f:
        c = counters(f)
        json.dump(c, open(os.path.join(dst, f"{tag}_{name}_counters.json"), "w"), indent=1)
bench = None
for line in open(os.path.join(src, "stats.log")):
    if line.startswith('{"metric"'):
        bench = json.loads(line)
fetch = json.load(open(os.path.join(dst, f"{tag}_fetch_counters.json")))["FETCH_SIZE"]
write = json.load(open(os.path.join(dst, f"{tag}_write_counters.json")))["WRITE_SIZE"]
cal = json.load(open(os.path.join(dst, f"{tag}_fetch_cal_counters.json")))["FETCH_SIZE"]
wl = bench["config"]["workload"]
N = int(wl.split("nSubj=")[1].split()[0]); J = int(wl.split("nItem=")[1].split()[0])
known = N * J * (2.0 * CELL + 1.0) / 1024.0        # KB read by the row-sum phase alone: omega + Y (1 byte) + logT per cell
corr = known / cal                                 # gfx950: FETCH_SIZE under-reports coalesced streaming reads (MI355X_MICROARCH.md, HBM)
traffic = (fetch * corr + write) * 1024.0
key = f"rtirt:{N}x{J}:{bench['dtype']}"
tj_path = os.path.join(dst, "traffic.json")
tj = json.load(open(tj_path)) if os.path.exists(tj_path) else {}
sq_path = os.path.join(dst, f"{tag}_sq_counters.json")
valu = None
if os.path.exists(sq_path):
    sq = json.load(open(sq_path))
    # SQ_ACTIVE_INST_VALU and SQ_WAVE_CYCLES count quad-cycles; SQ_BUSY_CYCLES sums the 32 shader engines' busy cycles; 1024 SIMDs
    if sq.get("SQ_BUSY_CYCLES") and sq.get("SQ_ACTIVE_INST_VALU"):
        kernel_cycles = sq["SQ_BUSY_CYCLES"] / 32.0
        valu = {"valu_busy_frac": 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (1024.0 * kernel_cycles), "valu_insts_per_launch": sq.get("SQ_INSTS_VALU"),
                "valu_insts_per_cell_update": sq.get("SQ_INSTS_VALU", 0.0) * 64.0 / (N * J), "kernel_cycles": kernel_cycles}
tj[key] = {"valu": valu, "traffic_bytes_per_launch": traffic, "fetch_size_kb_raw": fetch, "write_size_kb": write, "fetch_correction": corr,
           "calibration": f"row-sum phase alone (ERM_PASS_STOP=5) reads {known:.0f} KB and reports FETCH_SIZE {cal:.0f} KB", "source": f"profiles/{tag}_*"}
json.dump(tj, open(tj_path, "w"), indent=1)
with open(os.path.join(dst, f"{tag}_summary.md"), "w") as f:
    f.write(f"# rocprofv3 summary ({tag})\n\ncommand: `python3 bench.py --precision {prec} --no-fp32 --cpu-sweeps 0` (the default workload and step counts; counter passes: `--steps 200 --warmup 20 --no-profile`)\n\n")
    f.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
    for r in rows[:4]:
        f.write(f"| `{r['Name'][:70]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} | {float(r['MinNs'])/1e3:.2f} | {float(r['MaxNs'])/1e3:.2f} | {r['Percentage']} |\n")
    rf = bench.get("roofline") or {}
    f.write(f"\nbench line of the PROFILED run: ms_per_step {bench['ms_per_step']:.4f}, live launch_us {rf.get('launch_us', float('nan')):.2f} (HIP events around replayed graphs of the sweep "
            f"kernel: under rocprofv3's kernel tracing the launches of a graph run several microseconds apart and the brackets include those gaps; the kernel's own average "
            f"duration is the table's, and the unprofiled line -- profiles/{tag.rsplit('_', 1)[0]}_bench_line.json -- is the one whose live figure agrees with it)\n\n")
    f.write(f"pass_kernel HBM-side traffic per launch: FETCH_SIZE {fetch:.0f} KB x {corr:.2f} (calibrated) + WRITE_SIZE {write:.0f} KB = {traffic/1e6:.1f} MB "
            f"(algorithmic {BYTES*N*J/1e6:.1f} MB)\n")
    if valu:
        f.write(f"\nVALU: {valu['valu_insts_per_launch']:.3g} wave-instructions per launch = {valu['valu_insts_per_cell_update']:.0f} per cell-update; VALU busy "
                f"{100*valu['valu_busy_frac']:.0f} % of the kernel's cycles (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x SQ_BUSY_CYCLES / 32))\n")
# the other samplers' kernel statistics (tools/collect_model_stats.sh), when collected
rows_m = []
for m in ("mlirt", "latentqr", "crossqr"):
    fm = os.path.join(src, f"kernel_stats_{m}.csv")
    if os.path.exists(fm):
        shutil.copy(fm, os.path.join(dst, f"{tag}_kernel_stats_{m}.csv"))
    fm = os.path.join(dst, f"{tag}_kernel_stats_{m}.csv")
    if os.path.exists(fm):
        rows_m += [(m, r) for r in list(csv.DictReader(open(fm)))[:4] if "erm::" in r["Name"]]
if rows_m:
    with open(os.path.join(dst, f"{tag}_summary.md"), "a") as f:
        f.write("\n## Other samplers (same command with `--model`, `tools/collect_model_stats.sh`)\n\n| model | kernel | calls | avg us |\n|---|---|---|---|\n")
        for m, r in rows_m:
            f.write(f"| {m} | `{r['Name'][:60]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.2f} |\n")
print(open(os.path.join(dst, f"{tag}_summary.md")).read())
