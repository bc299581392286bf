#!/usr/bin/env python3
"""Per-kernel register / spill / LDS table of the library's kernels from hipcc's -Rpass-analysis=kernel-resource-usage remarks.
usage: python tools/kernel_resources.py [extra hipcc flags ...]   (compiles to a throw-away file; prints one line per kernel)"""
import re, subprocess, sys, os, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(tempfile.gettempdir(), "libertirt_res.so")
cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC", "-I", os.path.join(ROOT, "include"),
       os.path.join(ROOT, "extendedrtirtmodeling.jl_amd", "csrc", "ertirt.hip"), "-o", out, "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for ln in txt.splitlines():
    m = re.search(r"remark: Function Name: (\S+)", ln)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(.*", "", cur).replace("erm::", "").replace("void ", "")
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([\w /\[\]]+): (\d+)", ln)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
    if "error" in ln:
        print(ln)
for k, v in rows.items():
    if not v:
        continue
    print(f"{k:60s} VGPR {v.get('VGPRs', 0):4d} AGPR {v.get('AGPRs', 0):3d} spillV {v.get('VGPRs Spill', 0):3d} spillS {v.get('SGPRs Spill', 0):3d} scratch {v.get('ScratchSize [bytes/lane]', 0):4d} LDS {v.get('LDS Size [bytes/block]', 0):5d} occ {v.get('Occupancy [waves/SIMD]', 0)}")
