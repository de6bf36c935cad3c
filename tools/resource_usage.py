"""usage: python tools/resource_usage.py [out]   (no GPU needed: hipcc cross-compiles)
`make -C uav_motion_planning_amd/csrc resource-usage` (hipcc -Rpass-analysis=kernel-resource-usage) as ONE table, a row per kernel:
registers, scratch, spills, occupancy, static LDS -> profiles/r06_resource_usage.txt."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r06_resource_usage.txt")
txt = subprocess.run(["make", "-C", os.path.join(ROOT, "uav_motion_planning_amd", "csrc"), "resource-usage"], capture_output=True, text=True).stderr
rows, cur = [], None
for line in txt.split("\n"):
    m = re.search(r"remark: Function Name: (\S+)", line)
    if m:
        cur = {"name": m.group(1)}
        rows.append(cur)
        continue
    m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\S+)", line)
    if m and cur is not None:
        cur[m.group(1).strip()] = m.group(2)
names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True).stdout.split("\n")
cols = ("VGPRs", "AGPRs", "TotalSGPRs", "ScratchSize [bytes/lane]", "VGPRs Spill", "SGPRs Spill", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")
with open(out_path, "w") as f:
    f.write("# hipcc -O3 --offload-arch=gfx950 -Rpass-analysis=kernel-resource-usage of every translation unit of uav_motion_planning_amd/csrc (uavqp.hip, k_*.hip; tools/resource_usage.py);"
            " library source hash: the MANIFEST of the same round\n")
    f.write("kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | VGPR spills | SGPR spills | waves/SIMD | LDS B/block (static)\n")
    for r, n in zip(rows, names):
        f.write(" | ".join([n[:120]] + [r.get(c, "?") for c in cols]) + "\n")
print(len(rows), "kernels ->", out_path)
