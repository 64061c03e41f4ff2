"""Worker for tests/test_multigpu_gloo.py::test_sharded_gpu_render_*: one rank of a world_size-N job in which every rank renders its
interleaved tile shard with the HIP library into a device-resident film (ftn_render_device, as bench.py does) -- all ranks share
the box's single GPU, so the films are merged over gloo on the host instead of RCCL (the reduce itself is the same call)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, default_backend, scenes, _abi as A  # noqa: E402
from fountain_amd.distributed import merge_film, tile_shard  # noqa: E402

out_path = sys.argv[1]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
gpu = default_backend()
b, cam, res = scenes.instanced_cubes(gpu, n_copies=8, res=(192, 160), env_n=32)
scene = b.create_scene(device=0)
si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
smp = lambda: RandomSampler(64, 0, indexed=True, first_sample=3, sample_count=2)
dev = torch.zeros((res[1], res[0], 4), dtype=torch.float32, device="cuda:0")
st = si.render_device(scene, Film(gpu, res), smp(), dev.data_ptr(), torch.cuda.current_stream().cuda_stream, tiles=tile_shard(rank, world),
                      pipeline=A.FTN_PIPELINE_WAVEFRONT, device=0)
torch.cuda.synchronize()
host = dev.cpu()
merged = merge_film(host)
rays = torch.tensor([st["rays_closest"], st["rays_any"], st["camera_samples"]], dtype=torch.int64)
dist.all_reduce(rays)
if rank == 0:
    whole = Film(gpu, res)
    stw = si.render_parallel(scene, whole, smp(), pipeline=A.FTN_PIPELINE_WAVEFRONT)
    np.savez(out_path, merged=merged.numpy(), whole=whole.pixels, world=world, rays=rays.numpy(),
             rays_whole=np.array([stw["rays_closest"], stw["rays_any"], stw["camera_samples"]]), spill=stw["spill_samples"])
dist.barrier()
dist.destroy_process_group()
