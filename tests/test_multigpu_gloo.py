"""The N > 1 path on CPU: world_size-2 (and 3) gloo jobs shard the film tiles exactly as bench.py / the multi-GPU driver do
(fountain_amd.distributed) and merge with the single end-of-frame reduce; the merged film must equal the unsharded one."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
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
