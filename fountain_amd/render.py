"""`python -m fountain_amd.render scene.pbrt [-o out.exr] [--samples N]`: the reference's `render` binary
(src/bin/render.rs:16-104) over the MI355X library -- parse the scene file, PathIntegrator::new(5, 1.0), render, write
Film::into_spectrum_buffer as an OpenEXR file.

--threads of the reference has no meaning here; --gpu picks the HIP device.  With `--gpus N` (the program starts its N ranks itself,
fountain_amd/launch.py) or started under `python -m torch.distributed.run --nproc-per-node N …` every rank renders the film tiles r, r+N, … on GPU LOCAL_RANK and the films are merged with the frame's single
reduce (RCCL when every rank has its own GPU, gloo with --dist-backend gloo); rank 0 writes the image.  --exact-stream renders with the reference's own
per-tile RandomSampler stream (one lane per 16x16 tile: for validation, slow); the default re-seeds per (pixel, sample) so that
samples run in parallel (samplers).
"""
import argparse
import sys
import time

from . import _abi as A
from .api import PathIntegrator, PbrtScene, SamplerIntegrator, default_backend, write_exr


def main(argv=None):
    ap = argparse.ArgumentParser(prog="fountain_amd.render")
    ap.add_argument("scene_file")
    ap.add_argument("-o", "--output", dest="image_name", default=None)
    ap.add_argument("--samples", type=int, default=None)
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--exact-stream", action="store_true")
    ap.add_argument("--max-depth", type=int, default=5)          # render.rs:79 hard-codes PathIntegrator::new(5, 1.0)
    ap.add_argument("--rr-threshold", type=float, default=1.0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
    ap.add_argument("--gpus", type=int, default=None, help="render on N GPUs of this node: film tiles r, r+N, ... per rank, one reduce at the end")
    opts = ap.parse_args(argv)
    import os
    from .launch import spawn_ranks, world_from_env
    env_world = world_from_env()
    if env_world is None and opts.gpus is not None and opts.gpus > 1:
        # started plainly: start the N ranks as a child process (before anything here has touched HIP) and hand back its exit code
        return spawn_ranks(opts.gpus, sys.argv[1:] if argv is None else list(argv), module="fountain_amd.render")
    world, rank = (env_world[0], env_world[1]) if env_world is not None else (1, 0)
    if opts.gpus is not None and opts.gpus != world:
        print("error: --gpus %d but the launcher started WORLD_SIZE %d ranks" % (opts.gpus, world), file=sys.stderr)
        return 2
    if world > 1:
        opts.gpu = int(os.environ.get("LOCAL_RANK", "0")) if opts.dist_backend == "nccl" else opts.gpu

    be = default_backend()
    parsed = PbrtScene(opts.scene_file, be)
    filename = opts.image_name or parsed.film_name
    if ".exr" not in filename:
        raise SystemExit("output must be an .exr file (render.rs:53)")
    scene = parsed.create_scene(device=opts.gpu)
    film = parsed.film()
    sampler = parsed.sampler(opts.samples, indexed=not opts.exact_stream)
    integrator = SamplerIntegrator(parsed.camera, PathIntegrator.new(opts.max_depth, opts.rr_threshold))
    info = scene.info()
    print("scene: %d primitives, %d BVH nodes, %d lights" % (info["n_prims"], info["n_nodes"], info["n_lights"]), file=sys.stderr)
    t0 = time.time()
    if world > 1:
        import torch
        import torch.distributed as dist
        from .distributed import merge_film, tile_shard
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if opts.dist_backend == "nccl":
            torch.cuda.set_device(opts.gpu)
            dist.init_process_group("nccl", device_id=torch.device("cuda", opts.gpu))
        else:
            dist.init_process_group("gloo")
        st = integrator.render_parallel(scene, film, sampler, tiles=tile_shard(rank, world), device=opts.gpu)
        t = torch.from_numpy(film.pixels)
        if opts.dist_backend == "nccl":
            t = t.cuda(opts.gpu)
        merge_film(t)
        film.pixels[...] = t.cpu().numpy()
        dist.barrier()
        dist.destroy_process_group()
        if rank != 0:
            return 0
    else:
        st = integrator.render_parallel(scene, film, sampler, device=opts.gpu)
    dt = time.time() - t0
    rays = st["rays_closest"] + st["rays_any"]
    print("Completed rendering in %.3f s (%.1f Mrays/s%s)" % (dt, rays / dt / 1e6, " on rank 0 of %d" % world if world > 1 else ""), file=sys.stderr)
    img, (w, h) = film.into_spectrum_buffer()
    write_exr(filename, img, be)
    return 0


if __name__ == "__main__":
    sys.exit(main())
