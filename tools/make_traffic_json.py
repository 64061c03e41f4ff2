"""Builds profiles/<round>/traffic.json -- the PMC side of bench.py's roofline block -- from rocprofv3 counter passes over bench.py itself.

Run ON THE GPU BOX (from the repository root):
    python tools/make_traffic_json.py profiles/r03 [--spp-per-gpu 16]
It runs `bench.py --steps 1 --warmup 0 --no-cpu-baseline` once per counter group under
`rocprofv3 --pmc <counters> --kernel-trace --output-format csv` (separate passes: the TCC block has four slots and FETCH_SIZE /
WRITE_SIZE do not fit together, MI355X_MICROARCH.md "rocprofv3 PMC slots"), sums every counter per kernel group and per timed step,
and stores the sums with a hash of the kernel sources.  bench.py reports the figures only while that hash matches the sources it runs.

Byte accounting (checked in profiles/r02/fetch_size_calibration.md): on gfx950 FETCH_SIZE (KiB) x 1024 x 2 == TCC_EA0_RDREQ_128B x 128 B for
these gather kernels as well (the reads leave L2 as 128-byte requests and FETCH_SIZE tallies them at 64 B) -- the microarchitecture
guide's x2 correction; WRITE_SIZE (KiB) is taken as is.  The passes run bench.py with --no-count-step, so that every dispatch of the run belongs to the one timed step (the
counting builds of the traversal kernels are kept apart by name anyway)."""
import collections
import csv
import glob
import hashlib
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SOURCES = ["fountain_amd/csrc/ftn_trace4.hip", "fountain_amd/csrc/ftn_wavefront.hip", "fountain_amd/csrc/ftn_wf_common.h", "fountain_amd/csrc/ftn_device.h",
           "fountain_amd/csrc/ftn_kernels.hip", "fountain_amd/csrc/ftn_kernels.h", "fountain_amd/csrc/ftn_wavefront.h", "fountain_amd/csrc/ftn_host.cpp", "fountain_amd/csrc/ftn_math.h", "fountain_amd/csrc/detmath.h", "fountain_amd/csrc/ftn_texture.h"]
PASSES = [("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE"]), ("tcc", ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_EA0_RDREQ_64B_sum"]),
          ("tcc2", ["TCC_REQ_sum", "TCC_READ_sum", "TCC_WRITE_sum", "TCC_ATOMIC_sum"]), ("tcc3", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum"]),
          ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_BUSY_CYCLES", "SQ_WAIT_INST_ANY"]),
          ("sq2", ["SQ_THREAD_CYCLES_VALU", "SQ_INSTS_LDS"]),
          ("grbm", ["GRBM_GUI_ACTIVE"])]


def source_hash():
    h = hashlib.sha256()
    for rel in SOURCES:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


def group_of(name):
    """kernel name -> (group, counting build?)"""
    n = name
    counting = ("k_wf_trace4<true" in n) or ("k_wf_trace4_any<true" in n) or ("k_wf_trace8_any<true" in n) or ("k_wf_trace<false, true" in n) or ("k_wf_trace<true, true" in n)
    if "k_wf_trace8_any" in n or "k_wf_trace4_any" in n or "k_wf_trace_any2" in n or "k_wf_trace<true" in n:
        return "any_hit", counting
    if "k_wf_trace4" in n or "k_wf_trace<false" in n:
        return "closest", counting
    if "k_wf_shade" in n or "k_wf_classify" in n:
        return "shade", False
    if "rocprim" in n or "hipcub" in n or "k_wf_ray_keys" in n or "k_wf_hit_keys" in n:
        return "sort", False
    if "ftn::" in n:
        return "other", False
    return None, False


def main():
    out_dir = sys.argv[1] if len(sys.argv) > 1 else "profiles/r03"
    extra = sys.argv[2:]
    out_dir = os.path.join(ROOT, out_dir) if not os.path.isabs(out_dir) else out_dir
    os.makedirs(out_dir, exist_ok=True)
    scratch = os.path.join(ROOT, "gpurun_out", "traffic_passes")
    os.makedirs(scratch, exist_ok=True)
    sums = collections.defaultdict(lambda: collections.defaultdict(float))       # group -> counter -> sum over the timed step
    launches = collections.defaultdict(int)
    dominant = collections.defaultdict(float)
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))   # short kernel name -> counter -> sum (production instantiations only)
    bench_line = None
    env = dict(os.environ, TMPDIR="/tmp")
    for name, ctrs in PASSES:
        d = os.path.join(scratch, name)
        subprocess.run(["rm", "-rf", d])
        cmd = ["rocprofv3", "--pmc"] + ctrs + ["--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable, os.path.join(ROOT, "bench.py"),
               "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-count-step"] + extra
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=900)
        if r.returncode != 0:
            print("pass %s failed (rc %d): %s" % (name, r.returncode, r.stderr[-500:]), file=sys.stderr)
            continue
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                bench_line = json.loads(line)
        seen = set()
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            for row in csv.DictReader(open(f)):
                g, counting = group_of(row["Kernel_Name"])
                if g is None or counting:
                    continue
                sums[g][row["Counter_Name"]] += float(row["Counter_Value"])
                short = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("ftn::", "")
                per_kernel[short][row["Counter_Name"]] += float(row["Counter_Value"])
                if g == "closest" and "k_wf_trace4" in row["Kernel_Name"]:
                    dominant[row["Counter_Name"]] += float(row["Counter_Value"])
                if name == "fetch" and (row["Dispatch_Id"], g) not in seen:
                    seen.add((row["Dispatch_Id"], g))
                    launches[g] += 1
                    per_kernel[short]["launches"] += 1
                    if g == "closest" and "k_wf_trace4" in row["Kernel_Name"]:
                        launches["closest_dominant"] += 1
        subprocess.run(["rm", "-rf", d])

    def hbm_bytes(c):
        return c.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 + c.get("WRITE_SIZE", 0.0) * 1024.0

    groups = {}
    for g, c in sums.items():
        groups[g] = {"hbm_bytes_per_step": hbm_bytes(c), "read_bytes_per_step": c.get("FETCH_SIZE", 0.0) * 2048.0, "write_bytes_per_step": c.get("WRITE_SIZE", 0.0) * 1024.0,
                     "read_bytes_from_rdreq_128B": c.get("TCC_EA0_RDREQ_128B_sum", 0.0) * 128.0 + c.get("TCC_EA0_RDREQ_64B_sum", 0.0) * 64.0,
                     "l2_hit_rate": (c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0)),
                     "valu_wave_instructions": c.get("SQ_INSTS_VALU", 0.0), "salu_instructions": c.get("SQ_INSTS_SALU", 0.0), "vmem_read_instructions": c.get("SQ_INSTS_VMEM_RD", 0.0),
                     # SQ_ACTIVE_INST_VALU counts quad-cycles; 1024 SIMDs; GRBM_GUI_ACTIVE is summed over the 8 XCDs
                     "valu_busy": (c.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * c["GRBM_GUI_ACTIVE"] / 8.0)) if c.get("GRBM_GUI_ACTIVE") else None,
                     "wave_cycles_waiting_on_memory": (c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"]) if c.get("SQ_WAVE_CYCLES") else None,
                     "lanes_per_valu_inst": (c["SQ_THREAD_CYCLES_VALU"] / c["SQ_ACTIVE_INST_VALU"]) if c.get("SQ_THREAD_CYCLES_VALU") and c.get("SQ_ACTIVE_INST_VALU") else None,
                     "gpu_cycles": c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0, "launches_per_step": launches.get(g, 0)}
    cfg = (bench_line or {}).get("config", {})
    out = {"source_hash": source_hash(), "sources": SOURCES, "workload": cfg.get("workload"), "per_gpu_workload": cfg.get("per_gpu_workload"), "bench_args": extra,
           "dominant_kernel": {"name": "k_wf_trace4<closest>", "launches_per_step": launches.get("closest_dominant", 0),
                               "hbm_bytes_per_launch": hbm_bytes(dominant) / max(launches.get("closest_dominant", 0), 1),
                               "l2_hit_rate": dominant.get("TCC_HIT_sum", 0.0) / max(dominant.get("TCC_HIT_sum", 0.0) + dominant.get("TCC_MISS_sum", 0.0), 1.0),
                               "valu_busy": (dominant.get("SQ_ACTIVE_INST_VALU", 0.0) * 4.0 / (1024.0 * dominant["GRBM_GUI_ACTIVE"] / 8.0)) if dominant.get("GRBM_GUI_ACTIVE") else None,
                               "wave_cycles_waiting_on_memory": (dominant.get("SQ_WAIT_ANY", 0.0) / dominant["SQ_WAVE_CYCLES"]) if dominant.get("SQ_WAVE_CYCLES") else None,
                               "lanes_per_valu_inst": (dominant["SQ_THREAD_CYCLES_VALU"] / dominant["SQ_ACTIVE_INST_VALU"]) if dominant.get("SQ_THREAD_CYCLES_VALU") and dominant.get("SQ_ACTIVE_INST_VALU") else None},
           "groups": groups,
           "kernels": {k: {"launches_per_step": int(c.get("launches", 0)), "hbm_bytes_per_step": hbm_bytes(c), "read_bytes_per_step": c.get("FETCH_SIZE", 0.0) * 2048.0,
                           "write_bytes_per_step": c.get("WRITE_SIZE", 0.0) * 1024.0,
                           "l2_hit_rate": c.get("TCC_HIT_sum", 0.0) / max(c.get("TCC_HIT_sum", 0.0) + c.get("TCC_MISS_sum", 0.0), 1.0),
                           "valu_wave_instructions": c.get("SQ_INSTS_VALU", 0.0), "gpu_cycles": c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0, "counters": dict(c)} for k, c in sorted(per_kernel.items())},
           "method": "rocprofv3 --pmc (one counter group per pass: " + "; ".join("+".join(c) for _, c in PASSES) + ") --kernel-trace over `python bench.py --steps 1 --warmup 0 "
                     "--no-cpu-baseline --no-count-step`; production kernel instantiations only (the bench's counting step runs other instantiations); HBM-side bytes = FETCH_SIZE KiB x 1024 x 2 + "
                     "WRITE_SIZE KiB x 1024 (gfx950: 128-byte read requests are tallied at 64 B; cross-checked against TCC_EA0_RDREQ_128B x 128 B in the same file)"}
    json.dump(out, open(os.path.join(out_dir, "traffic.json"), "w"), indent=1)
    print(json.dumps({k: (v if not isinstance(v, dict) else "...") for k, v in out.items()}))
    for g, v in groups.items():
        print(g, json.dumps(v))
    for k, v in out["kernels"].items():
        print(k, json.dumps(v))


if __name__ == "__main__":
    main()
