"""Registers, scratch, LDS and occupancy of every kernel (hipcc -Rpass-analysis=kernel-resource-usage on the three kernel sources, gfx950):
  python tools/kernel_resources.py > profiles/rNN/kernel_resources.txt        (no GPU needed; a few minutes)"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "fountain_amd", "csrc")
FLAGS = "--offload-arch=gfx950 -std=c++17 -O3 -fPIC -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -Rpass-analysis=kernel-resource-usage".split()
print("Registers, scratch, LDS and occupancy of every kernel of the final build (hipcc -Rpass-analysis=kernel-resource-usage, gfx950;\nthe dynamic LDS of the traversal kernels is set at launch and not in these figures)\n")
for f in ("ftn_trace4.hip", "ftn_wavefront.hip", "ftn_kernels.hip"):
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", os.path.join(SRC, f), "-o", "/dev/null"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: (?:\S+: )?\s*(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*)$", line)
        if not m: continue
        k, v = m.group(1), m.group(2).replace("[-Rpass-analysis=kernel-resource-usage]", "").strip()
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k] = v
            if k.startswith("LDS Size"):
                name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void ", "")
                print("%-64s TotalSGPRs: %s  VGPRs: %s  AGPRs: %s  ScratchSize [bytes/lane]: %s  Occupancy [waves/SIMD]: %s  LDS Size [bytes/block]: %s" % (
                    name, cur.get("TotalSGPRs"), cur.get("VGPRs"), cur.get("AGPRs"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), v))
