#!/usr/bin/env python3
"""Per-kernel VGPR / scratch / occupancy table of one HIP translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kres.py pointcloudpdf_amd/csrc/rowlin2.hip [extra hipcc flags]"""
import re, subprocess, sys
src, extra = sys.argv[1], sys.argv[2:]
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics", "-Wno-unused-function", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if "error" in line:
        print(line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()}
    else:
        cur[k.split(" ")[0]] = v
        if k.startswith("LDS"):
            print(f"vgpr {cur.get('VGPRs'):>4} agpr {cur.get('AGPRs'):>4} scratch {cur.get('ScratchSize'):>5} occ {cur.get('Occupancy'):>2} lds {cur.get('LDS'):>6}  {cur['name'][:110]}")
