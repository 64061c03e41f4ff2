"""The N > 1 path on CPU: world_size-2 (and 3) gloo jobs shard the film tiles exactly as bench.py / the multi-GPU driver do
(fountain_amd.distributed) and merge with the single end-of-frame reduce; the merged film must equal the unsharded one."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_render_equals_whole(tmp_path, world):
    out = str(tmp_path / "merged.npz")
    port = 29500 + (os.getpid() + world) % 500
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_gloo_worker.py"), out]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    d = np.load(out)
    assert int(d["world"]) == world
    merged, whole = d["merged"], d["whole"]
    # every pixel has one home tile; a pixel that also got a spill sample from another rank's tile sums the same two
    # addends as the single-process merge (a + b is commutative), so the films are identical
    assert np.array_equal(merged.view(np.uint32), whole.view(np.uint32))
    assert merged[..., 3].min() >= 4.0
    # bench.py's per-rank fields (fountain_amd.distributed.rank_spread): min / max over the ranks of what each rank measured
    import json
    spread = json.load(open(out + ".spread.json"))
    assert spread["device_ms_per_step"] == {"min": 10.0, "max": 10.0 + world - 1} and spread["merge_ms"] == {"min": 1.0, "max": 1.0 + 0.5 * (world - 1)}


def test_step_samples_weak_and_strong():
    from fountain_amd.distributed import rank_spread, step_samples
    assert [step_samples("weak", n, 16) for n in (1, 2, 4, 8)] == [16, 32, 64, 128]          # every GPU renders 16 spp of its tiles
    assert [step_samples("strong", n, 16) for n in (1, 2, 4, 8)] == [16, 16, 16, 16]         # the job renders 16 spp, 1/N of the tiles each
    with pytest.raises(ValueError):
        step_samples("medium", 2, 16)
    assert rank_spread({"a": 3.0}) == {"a": {"min": 3.0, "max": 3.0}}                         # no process group: this rank alone


def test_tile_shard_partition():
    from fountain_amd.distributed import tile_shard
    n = 8160                                                   # 1920x1080 -> 120 x 68 tiles
    seen = np.zeros(n, int)
    for r in range(8):
        first, stride, _ = tile_shard(r, 8)
        seen[first::stride] += 1
    assert (seen == 1).all() and max(len(range(r, n, 8)) for r in range(8)) == 1020


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gpu_render_processes(tmp_path, world):
    """The product itself under torch.distributed: `world` processes (sharing this box's one GPU) render interleaved tile shards with
    ftn_render_device into device films and merge them with the frame's single reduce; the result equals the unsharded render."""
    out = str(tmp_path / "merged_gpu.npz")
    port = 29700 + (os.getpid() + world) % 200
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_gpu_shard_worker.py"), out]
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = np.load(out)
    assert int(d["world"]) == world and np.array_equal(d["rays"], d["rays_whole"])
    merged, whole = d["merged"], d["whole"]
    diff = (merged.view(np.uint32) != whole.view(np.uint32)).any(axis=-1)
    assert int(diff.sum()) <= 4 * int(d["spill"])              # a spill pixel's two addends may arrive in the other order
    assert np.allclose(merged, whole, rtol=2e-6, atol=1e-7) and merged[..., 3].min() >= 2.0


@pytest.mark.gpu
def test_render_cli_under_torchrun(tmp_path):
    """`python -m fountain_amd.render` started as a 2-rank job (both ranks on this box's GPU, films merged over gloo) writes the
    same image as the single-process run."""
    from fountain_amd import default_backend, read_exr
    golden = os.path.join(ROOT, "tests", "golden", "cornell.pbrt")
    one, two = str(tmp_path / "one.exr"), str(tmp_path / "two.exr")
    env = dict(os.environ, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "fountain_amd.render", golden, "-o", one, "--samples", "4"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    port = 29900 + os.getpid() % 90
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
                        "-m", "fountain_amd.render", golden, "-o", two, "--samples", "4", "--dist-backend", "gloo"], env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    be = default_backend()
    a, b = read_exr(one, be), read_exr(two, be)
    assert a.shape == (64, 64, 3) and np.allclose(a, b, rtol=2e-6, atol=1e-7) and (a != b).mean() < 0.01 and a.mean() > 0.05


def _run(cmd, env=None, timeout=300):
    return subprocess.run(cmd, env=dict(os.environ, OMP_NUM_THREADS="1", **(env or {})), capture_output=True, text=True, timeout=timeout, cwd=ROOT)


@pytest.mark.parametrize("n", [2, 8])
def test_bench_starts_its_own_ranks(n):
    """`python bench.py --gpus N` started plainly (no launcher environment) must start N ranks itself and report n_gpus = N (8: the
    round-end driver's largest job).  --launch-check stops after the process group has formed (no GPU here); the same code path
    carries the real run."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--launch-check"], env=dict(env, OMP_NUM_THREADS="1"),
                       capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout                      # stdout carries the result line alone
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["world_size_env"] == n and out["rank_sum"] == n * (n - 1) // 2


def test_bench_refuses_a_launcher_with_another_world_size():
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--launch-check"], env=dict(WORLD_SIZE="3", RANK="0", LOCAL_RANK="0"))
    assert r.returncode == 2 and "WORLD_SIZE 3" in r.stderr


def test_bench_rank_failure_is_an_error():
    """a rank that dies makes the whole job exit non-zero (here: every rank fails at argument parsing)"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from fountain_amd.launch import spawn_ranks; "
                        "sys.exit(spawn_ranks(2, ['--no-such-flag'], script=%r))" % (ROOT, os.path.join(ROOT, "bench.py"))],
                       env=env, capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert r.returncode != 0


def test_spawn_ranks_deadline_stops_hung_ranks(tmp_path):
    """a rank that hangs with its stdout open: the deadline ends the job (exit code 124) and no rank process is left behind"""
    hang = tmp_path / "hang.py"
    hang.write_text("import os, sys, time\nopen(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))\nprint('started', flush=True)\ntime.sleep(600)\n" % str(tmp_path))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", "import sys; sys.path.insert(0, %r); from fountain_amd.launch import spawn_ranks; "
                        "sys.exit(spawn_ranks(2, [], script=%r, timeout=20))" % (ROOT, str(hang))], env=env, capture_output=True, text=True, timeout=200, cwd=ROOT)
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    import time
    time.sleep(1.0)
    pids = [int(open(str(tmp_path / ("pid%d" % k))).read()) for k in range(2)]
    for pid in pids:
        alive = os.path.exists("/proc/%d" % pid) and "hang.py" in open("/proc/%d/cmdline" % pid).read().replace("\0", " ")
        assert not alive, "rank process %d survived the deadline" % pid


def test_bench_share_gpu_needs_gloo():
    """--share-gpu puts every rank on GPU 0, where RCCL would fail with a duplicate-device error: asking for nccl with it is refused"""
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-gpu", "--dist-backend", "nccl", "--launch-check"],
             env=dict(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999"))
    assert r.returncode == 2 and "gloo" in r.stderr


@pytest.mark.gpu
def test_bench_two_ranks_on_this_box(tmp_path):
    """the whole of `bench.py --gpus 2`, started plainly: two ranks (sharing this box's one GPU, films merged over gloo) render their
    tile shards of a small scene; the line reports n_gpus 2 and twice the per-GPU camera samples"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--copies", "8", "--res", "256", "--spp-per-gpu", "2", "--steps", "2",
                        "--warmup", "1", "--dist-backend", "gloo", "--share-gpu", "--no-cpu-baseline"], env=dict(env, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["camera_samples_per_step"] == 256 * 256 * 4 and out["value"] > 0 and out["scaling"] == "weak"
    for k in ("device_ms_per_step", "render_wall_ms", "merge_ms"):
        assert 0 <= out["ranks"][k]["min"] <= out["ranks"][k]["max"], out["ranks"]
    # the same job with the work fixed (strong scaling): 2 spp of the whole film, each rank half of the tiles
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--copies", "8", "--res", "256", "--spp-per-gpu", "2", "--steps", "2", "--scaling", "strong",
                        "--warmup", "1", "--share-gpu", "--no-cpu-baseline"], env=dict(env, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0"),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["config"]["camera_samples_per_step"] == 256 * 256 * 2
