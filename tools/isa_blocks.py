#!/usr/bin/env python3
"""Instruction census of one kernel's basic blocks from hipcc's assembly (diagnostics).
usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -I include extendedrtirtmodeling.jl_amd/csrc/ertirt.hip -o all.s
       python tools/isa_blocks.py all.s 'pass_kernelILi1EdLi0ELb1' [first_line last_line]
Prints, per label: line, VALU count, of which fp64, transcendental (quarter-rate), SALU, LDS, VMEM, and the loop annotation."""
import re
import sys

path, pat = sys.argv[1], sys.argv[2]
lo = int(sys.argv[3]) if len(sys.argv) > 3 else 0
hi = int(sys.argv[4]) if len(sys.argv) > 4 else 1 << 30
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + pat + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
blk, rows = None, []
tot = dict(v=0, f=0, t=0, s=0, d=0, m=0)
for i in range(start, end + 1):
    l = lines[i]
    n = i - start + 1
    if n < lo or n > hi:
        continue
    m = re.match(r"^(\.LBB\w+):(.*)", l)
    if m:
        blk = dict(name=m.group(1), line=n, note=m.group(2).strip(" ;"), v=0, f=0, t=0, s=0, d=0, m=0)
        rows.append(blk)
        continue
    if blk is None:
        blk = dict(name="(entry)", line=n, note="", v=0, f=0, t=0, s=0, d=0, m=0)
        rows.append(blk)
    op = l.strip().split(" ")[0] if l.startswith("\t") else ""
    if op.startswith("v_"):
        blk["v"] += 1
        if "f64" in op:
            blk["f"] += 1
        if re.match(r"v_(exp|log|rcp|rsq|sqrt|cos|sin)_", op):
            blk["t"] += 1
    elif op.startswith("s_"):
        blk["s"] += 1
    elif op.startswith("ds_"):
        blk["d"] += 1
    elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        blk["m"] += 1
print(f"{'label':14s} {'line':>6s} {'valu':>5s} {'f64':>5s} {'trans':>5s} {'salu':>5s} {'lds':>4s} {'vmem':>4s}  note")
for b in rows:
    if b["v"] + b["s"] + b["d"] + b["m"] == 0:
        continue
    print(f"{b['name']:14s} {b['line']:6d} {b['v']:5d} {b['f']:5d} {b['t']:5d} {b['s']:5d} {b['d']:4d} {b['m']:4d}  {b['note'][:60]}")
    for k in tot:
        tot[k] += b[k]
print("total", tot)
