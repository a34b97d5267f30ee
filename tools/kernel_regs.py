#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    tools/kernel_regs.py crdmodel_amd/csrc/crd_fused.hip [extra hipcc flags]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(root, "include"), "-I/opt/rocm/include",
       "-I" + os.path.join(root, "crdmodel_amd", "csrc"), "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: \s*(Function Name|VGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"\(anonymous namespace\)::|crd::|void ", "", cur).split("(")[0]
        rows[cur] = {}
    elif cur:
        rows[cur][k.split(" ")[0]] = v
print("%-64s %6s %6s %8s %5s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occ", "LDS"))
for k, r in rows.items():
    print("%-64s %6s %6s %8s %5s %6s" % (k[:64], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy"), r.get("LDS")))
