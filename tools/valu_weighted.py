#!/usr/bin/env python3
"""Class-weighted VALU-busy fraction of the sweep kernel.

`valu_busy_frac` in profiles/traffic.json is SQ_ACTIVE_INST_VALU x 4 / kernel cycles: it prices every VALU wave-instruction at 4 cycles.  The measured issue
costs differ by class (tools/ubench/valu_rate.hip on the same box: cycles per wave-instruction per SIMD at 4 waves per SIMD), and the kernel's dynamic mix by
class comes from the SQ_INSTS_VALU_* counters (tools/valu_classes.sh).  This script multiplies the two:

    weighted busy = sum_class n_class x cost_class / (1024 SIMDs x kernel cycles),        kernel cycles = SQ_BUSY_CYCLES / 32 (as in summarize_profiles.py)

with the instructions no class counter claims (moves, compares, selects, lane permutes) at the full-rate integer cost.
usage: python3 tools/valu_weighted.py <tag> <f64|f32>   (reads gpurun_out/<tag>_valu_<prec>/, writes profiles/<tag>_valu_classes_<prec>.json, profiles/<tag>_valu_issue_rates.txt,
       and adds valu_busy_frac_weighted to profiles/traffic.json)"""
import json, os, re, shutil, sys

tag, prec = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"{tag}_valu_{prec}")
cls = json.load(open(os.path.join(src, "classes.json")))
rates = {}
for ln in open(os.path.join(src, "valu_rate.txt")):
    m = re.match(r"(.+?)\s+[\d.]+ ms\s+-> ([\d.]+) cycles per wave-instruction.*\((\d+) instr", ln)
    if m:
        rates[m.group(1).strip()] = (float(m.group(2)), int(m.group(3)))
full = rates["xor-shift-add (3 int ops)"][0]                      # full-rate integer / logic
r = lambda k: rates[k][0]
# a measured pair / triple's companions (add, xor, and / cmp / add) are full-rate: the named instruction's own cost = n x average - (n - 1) x full
alone = lambda k: rates[k][0] * rates[k][1] - (rates[k][1] - 1) * full
cost = {
    "ADD_F32": r("v_add_f32"), "MUL_F32": r("v_fma_f32"), "FMA_F32": r("v_fma_f32"), "TRANS_F32": 0.5 * (r("v_exp_f32") + r("v_rcp_f32")),
    "ADD_F64": r("v_add_f64"), "MUL_F64": r("v_mul_f64"), "FMA_F64": r("v_fma_f64"), "TRANS_F64": r("v_rcp_f64"),
    "INT32": 0.5 * (full + r("v_bitop3_b32")), "INT64": r("v_mad_u64_u32"), "CVT": alone("v_cvt_f64_u32 + hi + add"),
}
total = cls["SQ_INSTS_VALU"]
named = {k: cls.get("SQ_INSTS_VALU_" + k, 0.0) for k in cost}
other = max(0.0, total - sum(named.values()))
cycles = cls["SQ_BUSY_CYCLES"] / 32.0
busy4 = 4.0 * cls["SQ_ACTIVE_INST_VALU"] / (1024.0 * cycles)
weighted = (sum(named[k] * cost[k] for k in cost) + other * full) / (1024.0 * cycles)
out = {"workload": f"GibbsRtIrt 100000 x 50, {prec}, fused sweep kernel", "valu_wave_instructions_per_launch": total, "by_class": named, "unclassified": other,
       "issue_cost_cycles": dict(cost, unclassified=full), "kernel_cycles": cycles, "valu_busy_frac_4_cycles": busy4, "valu_busy_frac_weighted": weighted,
       "mean_cycles_per_valu_instruction_weighted": (sum(named[k] * cost[k] for k in cost) + other * full) / total,
       "mean_simd_cycles_available_per_valu_instruction": 1024.0 * cycles / total}
json.dump(out, open(os.path.join(root, "profiles", f"{tag}_valu_classes_{prec}.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "valu_rate.txt"), os.path.join(root, "profiles", f"{tag}_valu_issue_rates.txt"))
tj_path = os.path.join(root, "profiles", "traffic.json")
tj = json.load(open(tj_path))
for k, v in tj.items():
    if k.endswith(":" + prec) and v.get("valu"):
        v["valu"]["valu_busy_frac_weighted"] = weighted
        v["valu"]["weighted_source"] = f"profiles/{tag}_valu_classes_{prec}.json (SQ_INSTS_VALU_* x measured issue costs, profiles/{tag}_valu_issue_rates.txt)"
json.dump(tj, open(tj_path, "w"), indent=1)
print(json.dumps(out, indent=1))
