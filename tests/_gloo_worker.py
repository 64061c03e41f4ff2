"""Worker for tests/test_multigpu_gloo.py: one rank of a world_size-N gloo job that renders its interleaved tile shard with the
CPU oracle (stand-in for a GPU) and merges the films with fountain_amd.distributed's single reduce."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, scenes  # noqa: E402
from fountain_amd.distributed import rank_spread, render_sharded  # noqa: E402
from oracle_loader import oracle_backend  # noqa: E402

out_path = sys.argv[1]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
orc = oracle_backend(det=True)
b, cam, res = scenes.cornell(orc, res=80)
scene = b.create_scene()
si = SamplerIntegrator(cam, PathIntegrator.new(4, 1.0))
smp = RandomSampler(4, 0, indexed=True)


def render_fn(tiles, out):
    film = Film(orc, res)
    si.render_parallel(scene, film, smp, tiles=tiles, n_threads=2)
    out.copy_(torch.from_numpy(film.pixels))


merged = render_sharded(render_fn, (res[1], res[0], 4), rank, world, lambda shape: torch.zeros(shape, dtype=torch.float32))
spread = rank_spread({"device_ms_per_step": 10.0 + rank, "merge_ms": 1.0 + 0.5 * rank})      # bench.py's per-rank fields
if rank == 0:
    import json
    json.dump(spread, open(out_path + ".spread.json", "w"))
    whole = Film(orc, res)
    si.render_parallel(scene, whole, smp, n_threads=2)
    np.savez(out_path, merged=merged.numpy(), whole=whole.pixels, world=world)
dist.barrier()
dist.destroy_process_group()
