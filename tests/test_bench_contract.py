"""bench.py's one JSON line: the fields the driver's contract names are present, typed and mutually consistent -- checked on the line of
the last profiled run (profiles/r03/bench.json, written by `python bench.py --steps 10 --warmup 2` on the GPU box) and on the helper
functions the line is built from.  CPU only; the line itself is produced on a GPU (tests/test_multigpu_gloo.py runs bench.py end to end
there)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_line():
    path = os.path.join(ROOT, "profiles", "r03", "bench.json")
    lines = [l for l in open(path).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1, "stdout of bench.py carries the result line alone"
    return json.loads(lines[0])


def test_contract_fields():
    b = load_line()
    for k, t in (("metric", str), ("value", float), ("unit", str), ("n_gpus", int), ("steps", int), ("warmup", int), ("ms_per_step", float),
                 ("higher_is_better", bool), ("scaling", str), ("dtype", str), ("data", str), ("config", dict), ("roofline", dict), ("cpu_baseline", dict)):
        assert isinstance(b[k], t), k
    assert b["vs_baseline"] is None                     # BASELINE.md holds no published number for this metric
    assert b["unit"] == "Mrays/s" and b["higher_is_better"] is True and b["scaling"] in ("weak", "strong") and b["dtype"] == "f32" and b["data"] == "synthetic"
    assert "workload" in b["config"] and "model" not in b["config"] and "10002588 triangles" in b["config"]["workload"]
    # value = rays of all ranks / wall time of the timed steps
    assert abs(b["config"]["rays_per_step"] / (b["ms_per_step"] * 1e-3) / 1e6 - b["value"]) / b["value"] < 1e-3
    assert b["config"]["device_ms_per_step"] <= b["ms_per_step"] * 1.001
    # (2.98 GB of scene arrays + 134 MB of environment cell records)
    assert sum(v for k, v in b["config"]["device_bytes"].items() if k != "total_bytes") == b["config"]["device_bytes"]["total_bytes"] < 3.2e9
    assert b["config"]["device_bytes"]["total_bytes"] - b["config"]["device_bytes"]["lights_bytes"] < 3.0e9
    for k in ("device_ms_per_step", "render_wall_ms", "merge_ms"):
        assert 0 <= b["ranks"][k]["min"] <= b["ranks"][k]["max"]


def test_roofline_block():
    r = load_line()["roofline"]
    assert r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0 < r["frac"] <= 1.0
    assert abs(r["achieved"] / r["peak"] - r["frac"]) < 1e-3
    assert abs(r["traffic"] / (r["avg_launch_ms"] * 1e-3) / 1e9 - r["achieved"]) / r["achieved"] < 1e-3          # counter bytes per launch / live launch time
    assert r["bound"] in ("memory_latency", "valu_issue", "hbm")
    i = r["issue"]
    assert abs(i["valu_busy"] * i["lanes_per_valu_inst"] / i["of_lanes"] - i["frac"]) < 1e-3 and 0 < i["frac"] <= 1.0
    assert r["hbm"]["frac"] == r["frac"] and r["hbm"]["traffic"] == r["traffic"]
    g = r["groups"]
    assert set(g) >= {"closest", "any_hit", "shade", "other"} and all(v["ms_per_step"] is not None and v["ms_per_step"] >= 0 for v in g.values())
    assert abs(sum(v["ms_per_step"] for v in g.values()) - r["whole_step"]["device_ms"]) < 0.05 * r["whole_step"]["device_ms"]
    assert 0 < r["whole_step"]["hbm_frac"] <= 1.0 and g["shade"]["hbm_bytes_per_event"] > 0
    # algorithmic bytes are reported, but never as a fraction of the HBM roof
    assert "GB_per_s" in r["algorithmic"] and "frac" not in r["algorithmic"]


def test_cpu_baseline_block():
    c = load_line()["cpu_baseline"]
    assert c["kind"] == "port" and c["unit"] == "Mrays/s" and c["cores"] >= 1 and c["value"] > 0 and "tiles" in c["sample"]


def test_step_and_workload_helpers():
    import argparse
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    a = argparse.Namespace(copies=2309, res=4096, spp_per_gpu=16, scaling="weak", gpus=8)
    assert bench.per_gpu_workload(a) == "copies=2309,res=4096,spp_per_gpu=16"      # weak: one GPU's share does not depend on N
    a.scaling = "strong"
    assert bench.per_gpu_workload(a) == "copies=2309,res=4096,spp=16,tiles=1/8"
    a.gpus = 1
    assert bench.per_gpu_workload(a) == "copies=2309,res=4096,spp_per_gpu=16"
    assert bench.PROFILE_ROUND == "r03" and os.path.exists(os.path.join(ROOT, "profiles", bench.PROFILE_ROUND, "traffic.json"))
    tj = json.load(open(os.path.join(ROOT, "profiles", bench.PROFILE_ROUND, "traffic.json")))
    if tj["source_hash"] != bench.source_hash():          # (bench.py then leaves the HBM-side figures out of its line and says so: no silent staleness)
        pytest.skip("profiles/r03/traffic.json was taken on other kernel sources: rerun tools/make_traffic_json.py on the GPU box")
