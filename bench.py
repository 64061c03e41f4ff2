#!/usr/bin/env python3
"""Headline benchmark: Mrays/s (primary + secondary) of the wavefront path tracer on the 10M-triangle synthetic scene
(BASELINE.json config 5: 2309 baked copies of rounded_cube, 4096x4096 film, PathIntegrator(5, 1.0), env-map light).

  python bench.py --gpus N --steps K --warmup W
  N > 1: either under a launcher (python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
  bench.py --gpus N ...) or plainly -- then bench.py starts its N ranks itself as a child process (fountain_amd/launch.py) and relays
  rank 0's line.  A launcher whose WORLD_SIZE differs from --gpus is an error (exit code 2).

A STEP is one pass of the hot path over one batch of camera samples: 16 N samples per pixel of the 4096x4096 film (--spp-per-gpu 16),
with the film's 65,536 tiles interleaved over the N ranks -- so every GPU traces the same 268 M paths per step whatever N is
(weak scaling of the 4096-spp job; `value` is the whole-job rate).  The batch is sized for the part's memory: one wavefront of 268 M
paths keeps ~88 GB of path state in HBM and is 18 % faster per ray than sixteen wavefronts of 17 M.  The film stays resident in HBM (a torch tensor handed to
ftn_render_device by pointer); at the end of the timed region the ranks' films are summed by ONE RCCL reduce.
Prints one JSON line on rank 0."""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--copies", type=int, default=int(os.environ.get("FTN_BENCH_COPIES", "2309")), help="mesh copies (2309 = 10,002,588 triangles)")
    ap.add_argument("--res", type=int, default=int(os.environ.get("FTN_BENCH_RES", "4096")))
    ap.add_argument("--spp-per-gpu", type=int, default=int(os.environ.get("FTN_BENCH_SPP_PER_GPU", "16")), help="samples per pixel one GPU renders per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tiles", type=int, default=int(os.environ.get("FTN_BENCH_CPU_TILES", "96")))
    ap.add_argument("--dist-backend", default=os.environ.get("FTN_BENCH_DIST_BACKEND", "nccl"), choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI, one GPU per rank (the measured configuration); gloo = rehearsal of the N-rank path (see --share-gpu)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal on a one-GPU box: every rank renders on GPU 0, films merged over gloo on the host (implies --dist-backend gloo: RCCL refuses two ranks on one device)")
    ap.add_argument("--scaling", default=os.environ.get("FTN_BENCH_SCALING", "weak"), choices=["weak", "strong"],
                    help="weak (default): every GPU renders --spp-per-gpu samples of its tiles per step, the job's work grows with N; strong: a step is --spp-per-gpu samples of the WHOLE film whatever N is, each GPU renders its 1/N of the tiles")
    ap.add_argument("--no-count-step", action="store_true", help="skip the untimed counting step (tools/make_traffic_json.py: every dispatch of the run then belongs to the timed steps)")
    ap.add_argument("--launch-check", action="store_true", help="start the ranks, form the process group, print the result line's n_gpus -- no rendering (CPU test of the launcher)")
    args = ap.parse_args()

    # ---- N ranks: either a launcher started us (WORLD_SIZE set), or we start the ranks ourselves -- as a child process, before
    # torch / HIP are touched -- and relay rank 0's line
    from fountain_amd.launch import spawn_ranks, world_from_env
    env_world = world_from_env()
    if env_world is None and args.gpus > 1:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:], script=os.path.abspath(__file__), json_only=True))
    world, rank, local_rank = env_world if env_world is not None else (1, 0, 0)
    if world != args.gpus:
        print("error: --gpus %d but the launcher started WORLD_SIZE %d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)

    import torch
    import torch.distributed as dist

    use_dist = world > 1 or os.environ.get("FTN_BENCH_FORCE_DIST") == "1"      # the latter: rehearse the collective path on one GPU
    if args.share_gpu:
        local_rank = 0
        if args.dist_backend != "gloo":
            if "--dist-backend" in sys.argv[1:] or os.environ.get("FTN_BENCH_DIST_BACKEND"):
                print("error: --share-gpu puts every rank on GPU 0 and RCCL refuses two ranks on one device: use --dist-backend gloo", file=sys.stderr)
                sys.exit(2)
            args.dist_backend = "gloo"
    if use_dist:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.dist_backend == "nccl" and not args.launch_check:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    if args.launch_check:
        t = torch.tensor([1.0, float(rank)])
        if use_dist:
            dist.all_reduce(t)
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": int(t[0]), "rank_sum": int(t[1]), "world_size_env": world}), flush=True)
        if use_dist:
            dist.destroy_process_group()
        return

    from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, default_backend, scenes, _abi as A
    from fountain_amd.distributed import merge_film, rank_spread, step_samples, tile_shard
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    gpu = default_backend()
    host_merge = use_dist and args.dist_backend == "gloo"      # rehearsal: the reduce runs over gloo on a host copy of the film

    # ---- scene (host assembly + BVH::build + upload; not timed: SURVEY 8(d))
    t0 = time.time()
    builder, cam, res = scenes.instanced_cubes(gpu, n_copies=args.copies, res=(args.res, args.res))
    desc, keep = builder.build_desc()
    from fountain_amd.api import Scene
    scene = Scene(gpu, desc, keep, device=local_rank)
    info = scene.info()
    build_s = time.time() - t0
    film = Film(gpu, res)
    n_tiles = film.tile_count()
    tiles = tile_shard(rank, world)
    integ = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    dev_film = torch.zeros((film.height, film.width, 4), dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    spp_per_step = step_samples(args.scaling, world, args.spp_per_gpu)
    total_spp = 4096

    def step(i, count=False):
        smp = RandomSampler(total_spp, 0, indexed=True, first_sample=(i * spp_per_step) % total_spp, sample_count=spp_per_step)
        return integ.render_device(scene, film, smp, dev_film.data_ptr(), stream.cuda_stream, tiles=tiles,
                                   pipeline=A.FTN_PIPELINE_WAVEFRONT, device=local_rank, count_traffic=count)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # one counted step: algorithmic bytes of the dominant kernel (nodes / triangles per ray are data dependent: SURVEY 8(d)).
    # count_traffic = 2: tally exactly the rays each kernel traces in production (MIS rays toward the environment light only need
    # hit / miss and go through the any-hit kernel: ftn_stats.mis_rays_any_hit)
    cst = step(0, count=2) if not args.no_count_step else None
    dev_film.zero_()
    for i in range(args.warmup):
        step(i)
    def merge():
        if host_merge:
            h = dev_film.cpu()
            merge_film(h)
            dev_film.copy_(h)
        else:
            merge_film(dev_film)              # the single end-of-frame reduce (RCCL over xGMI)

    if use_dist:
        merge()                               # untimed: the first large reduce also sets up RCCL's channels / buffers
    dev_film.zero_()
    barrier()
    t0 = time.perf_counter()
    rays = 0
    tot = {"trace_ms": 0.0, "trace_launches": 0, "any_ms": 0.0, "any_launches": 0, "shade_ms": 0.0, "shade_launches": 0, "sort_ms": 0.0, "kernel_ms": 0.0, "camera_samples": 0}
    for i in range(args.steps):
        st = step(args.warmup + i)
        rays += st["rays_closest"] + st["rays_any"]
        for k in tot:
            tot[k] += st[k]
    cam_samples = tot["camera_samples"]
    torch.cuda.synchronize(dev)
    t_render = time.perf_counter() - t0                    # this rank's K steps (host clock around its own device work)
    merge()
    torch.cuda.synchronize(dev)
    t_merge = time.perf_counter() - t0 - t_render          # the end-of-frame reduce as this rank saw it (includes waiting for the slowest rank)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed, float(rays), float(cam_samples)], dtype=torch.float64, device="cpu" if host_merge else dev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays, cam_samples = int(t[1]), int(t[2])
    # per-rank spread: where a scaling loss comes from (a slow rank, or the reduce)
    per_rank = rank_spread({"device_ms_per_step": tot["kernel_ms"] / max(args.steps, 1), "render_wall_ms": t_render * 1e3, "merge_ms": t_merge * 1e3},
                           device="cpu" if host_merge else dev)

    if rank == 0:
        mrays = rays / elapsed / 1e6
        n_steps = max(args.steps, 1)
        out = {
            "metric": "Mrays/s (primary+secondary) at fixed spp", "value": round(mrays, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed * 1e3 / n_steps, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "config 5 scene: %d baked copies of rounded_cube = %d triangles (%d BVH nodes, depth %d), %dx%d film, "
                                   "PathIntegrator(max_depth 5, rr 1.0), 1024^2 env-map light; one step = %d spp over the whole film"
                                   % (args.copies, info["n_prims"], info["n_nodes"], info["max_depth"], res[0], res[1], spp_per_step),
                       "tiles": n_tiles, "tiles_per_gpu": (n_tiles + world - 1) // world, "camera_samples_per_step": cam_samples // n_steps,
                       "rays_per_step": rays // n_steps, "pipeline": "wavefront", "sampler": "indexed xoshiro256+",
                       "per_gpu_workload": per_gpu_workload(args), "scene_build_s": round(build_s, 1), "device_ms_per_step": round(tot["kernel_ms"] / n_steps, 3),
                       "device_bytes": {k: v for k, v in info.items() if k.endswith("_bytes")}},
            # min / max over the ranks: device time of one step (HIP events inside the library), wall time of the K steps, and the single film
            # reduce at the end of the timed region (as each rank saw it: it includes waiting for the slowest rank)
            "ranks": {k: {"min": round(v["min"], 3), "max": round(v["max"], 3)} for k, v in per_rank.items()},
            "roofline": roofline(args, cst, tot, n_steps, spp_per_step),
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(desc, cam, film, args.cpu_tiles, n_tiles)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


def per_gpu_workload(args):
    """what ONE GPU renders per step (the PMC profile under profiles/ is of exactly this share)"""
    if args.scaling == "strong" and args.gpus > 1:
        return "copies=%d,res=%d,spp=%d,tiles=1/%d" % (args.copies, args.res, max(1, args.spp_per_gpu), args.gpus)
    return "copies=%d,res=%d,spp_per_gpu=%d" % (args.copies, args.res, max(1, args.spp_per_gpu))


def source_hash():
    """hash of the kernel sources the PMC profile was taken on (same list as tools/make_traffic_json.py)"""
    import hashlib
    tj = os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic.json")
    try:
        srcs = json.load(open(tj))["sources"]
    except Exception:
        return None
    h = hashlib.sha256()
    for rel in srcs:
        h.update(open(os.path.join(ROOT, rel), "rb").read())
    return h.hexdigest()[:16]


PROFILE_ROUND = "r03"
HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def roofline(args, cst, tot, n_steps, spp_per_step):
    """The dominant kernel (k_wf_trace4<closest>) against the HBM roof, three ways that must not be mixed up:
      * achieved / frac / traffic: bytes that really cross the HBM interface per launch -- from PMC counters, collected OFFLINE over this same
        command by tools/make_traffic_json.py (profiles/<round>/traffic.json, used only while the kernel sources still hash to what was
        profiled) -- divided by the launch duration measured LIVE in this run (HIP events on the launch stream).  frac <= 1 by construction;
      * algorithmic: bytes the walk asks the memory system for (128 B per four-box record + 48 B per triangle + ray i/o; SURVEY 8(d) restated
        for the records this kernel reads), most of which L1 / L2 serve -- a rate above the HBM peak only says the caches work;
      * what binds: neither the HBM roof nor instruction issue -- the latency of a ray's chain of dependent fetches at the occupancy the
        registers and LDS stacks allow (`bound`, `binds`, `issue`).
    The same three figures are given for the other kernel groups of a step and for the step as a whole."""
    avg_ms = tot["trace_ms"] / max(tot["trace_launches"], 1)
    r = {"bound": "hbm", "kernel": "k_wf_trace4<closest> (128-byte four-box records)", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
         "avg_launch_ms": round(avg_ms, 4), "launches_per_step": tot["trace_launches"] // n_steps}
    if cst is not None:
        rec_c = cst["quad_records"] - cst["quad_records_any"]
        prims_c = cst["prims_tested"] - cst["prims_tested_any"]
        rays_c = cst["rays_closest"] - cst["mis_rays_any_hit"]          # rays the closest-hit launches traced
        launches = max(cst["trace_launches"], 1)
        alg = (128.0 * rec_c + 48.0 * prims_c + (32.0 + 16.0) * rays_c) / launches
        r["algorithmic"] = {"bytes_per_launch": int(alg), "GB_per_s": round(alg / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None,
                            "records_per_ray": round(rec_c / max(rays_c, 1), 2), "prims_per_ray": round(prims_c / max(rays_c, 1), 3), "rays_per_launch": int(rays_c // launches),
                            "note": "128 B per four-box record + 48 B per triangle test + 48 B of ray i/o; served mostly by L1 / L2, hence not a fraction of the HBM roof"}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "traffic.json")))
    except Exception:
        tj = None
    # the profile is of ONE GPU's share of a step (N-GPU runs give every GPU the same share: --spp-per-gpu samples of its tiles)
    same_work = tj is not None and tj.get("per_gpu_workload") == per_gpu_workload(args)
    if tj is None or tj.get("source_hash") != source_hash() or not same_work:
        r["note"] = "no PMC profile of this build / workload under profiles/%s (run tools/make_traffic_json.py on the GPU box): HBM-side figures omitted" % PROFILE_ROUND
        return r
    dk = tj["dominant_kernel"]
    r["traffic"] = int(dk["hbm_bytes_per_launch"])
    r["achieved"] = round(dk["hbm_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9, 1) if avg_ms > 0 else None
    r["frac"] = round(r["achieved"] / HBM_PEAK_GBS, 4) if r["achieved"] is not None else None
    r["l2_hit_rate"] = round(dk["l2_hit_rate"], 3)
    r["valu_busy"] = round(dk["valu_busy"], 3) if dk.get("valu_busy") is not None else None
    r["wave_cycles_waiting_on_memory"] = round(dk["wave_cycles_waiting_on_memory"], 3) if dk.get("wave_cycles_waiting_on_memory") is not None else None
    # VALU issue slots x lane utilisation.  valu_busy = the fraction of the launch's SIMD cycles in which a VALU instruction issues
    # (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE)); lanes_per_valu_inst = the lanes live in the average VALU instruction
    # (SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU); their product / 64 is the fraction of the chip's lane-issue slots that do work
    if dk.get("valu_busy") is not None and dk.get("lanes_per_valu_inst") is not None:
        r["issue"] = {"valu_busy": round(dk["valu_busy"], 3), "lanes_per_valu_inst": round(dk["lanes_per_valu_inst"], 1), "of_lanes": 64,
                      "frac": round(dk["valu_busy"] * dk["lanes_per_valu_inst"] / 64.0, 4)}
    # What binds: neither roof.  Probes on this kernel (profiles/r03): +31 % dependent VALU instructions per
    # record step cost +1.7 % time (not issue bound, whatever the busy counter says); records at half-line granularity or a third fewer
    # record fetches per ray change nothing (not bytes, not the cached part of the chain).  A ray's walk is a chain of dependent fetches of
    # which ~5.5 miss L2 (about four of them the triangle reads of its leaf tests); the launch takes chain latency x rays / rays in flight,
    # and registers + LDS stacks fix the rays in flight (5 waves per SIMD).
    r["bound"] = "memory_latency"
    r["hbm"] = {"achieved": r["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": r["frac"], "traffic": r["traffic"]}
    r["binds"] = ("co-limited at 5 waves per SIMD: the latency of each ray's chain of dependent fetches (~5.5 L2 misses per ray), VALU issue (%.0f %% of the SIMD "
                  "cycles, %.0f of 64 lanes live) and the L1 path (~1.2 scattered lane-loads per clock and CU); the HBM interface runs at %.0f %% of its peak.  Taking "
                  "pressure off one of the three put it on another: +31 %% instructions cost +1.7 %% time, a third fewer record fetches or triangle fetches cost "
                  "nothing less (profiles/r03/g_*, h_*, j_*); achieved / peak / frac / traffic are the HBM figures the bench contract asks for"
                  % (100.0 * (dk.get("valu_busy") or 0.0), dk.get("lanes_per_valu_inst") or 0.0, 100.0 * (r["frac"] or 0.0)))
    r["traffic_source"] = "profiles/%s/traffic.json (rocprofv3 --pmc over this command, production kernels only; FETCH_SIZE x 2 + WRITE_SIZE, cross-checked with TCC_EA0_RDREQ_128B x 128 B; counted where L2 meets the fabric, so lines the 256 MB Infinity Cache serves are included: an upper bound of the DRAM bytes)" % PROFILE_ROUND
    live = {"closest": tot["trace_ms"] / n_steps, "any_hit": tot["any_ms"] / n_steps, "shade": tot["shade_ms"] / n_steps, "sort": tot["sort_ms"] / n_steps}
    groups = {}
    step_bytes = 0.0
    for g, v in tj["groups"].items():
        step_bytes += v["hbm_bytes_per_step"]
        ms = live.get(g)
        groups[g] = {"ms_per_step": round(ms, 2) if ms is not None else None, "hbm_GB_per_step": round(v["hbm_bytes_per_step"] / 1e9, 2),
                     "hbm_frac": round(v["hbm_bytes_per_step"] / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms else None, "l2_hit_rate": round(v["l2_hit_rate"], 3),
                     "valu_busy": round(v["valu_busy"], 3) if v.get("valu_busy") is not None else None}
    step_ms = tot["kernel_ms"] / n_steps
    if "other" in groups and groups["other"]["ms_per_step"] is None:        # generate / accumulate / resets: what the four timed groups leave of the step's device time
        oms = max(step_ms - sum(v for v in live.values()), 0.0)
        groups["other"]["ms_per_step"] = round(oms, 2)
        groups["other"]["hbm_frac"] = round(tj["groups"]["other"]["hbm_bytes_per_step"] / (oms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if oms > 0 else None
    if cst is not None and "shade" in groups:
        events = cst["rays_closest"] - cst["mis_rays_any_hit"]              # every ray of a closest-hit launch is classified and shaded once
        groups["shade"]["shading_events_per_step"] = int(events)
        groups["shade"]["hbm_bytes_per_event"] = round(tj["groups"]["shade"]["hbm_bytes_per_step"] / max(events, 1), 1)
    r["groups"] = groups
    r["whole_step"] = {"device_ms": round(step_ms, 2), "hbm_GB": round(step_bytes / 1e9, 1), "hbm_frac": round(step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if step_ms > 0 else None}
    return r


def cpu_baseline(desc, cam, film, n_cpu_tiles, n_tiles):
    """The CPU oracle (a C++ restatement of fountain's CPU path, NOT fountain itself: no Rust toolchain here) timed on a
    bounded sample of the same workload: n_cpu_tiles tiles spread evenly over the film, 1 spp, all host cores."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_loader import oracle_backend
    from fountain_amd import PathIntegrator, RandomSampler, SamplerIntegrator, Film
    from fountain_amd.api import Scene
    orc = oracle_backend(det=False)
    t0 = time.time()
    scene = Scene(orc, desc, None)
    build_s = time.time() - t0
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    try:                                               # a cgroup CPU quota (not visible in the affinity mask) bounds the useful thread count
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = min(cores, max(1, int(round(float(quota) / float(period)))))
    except Exception:
        pass
    cores = max(1, min(cores, int(os.environ.get("FTN_BENCH_CPU_THREADS", "64"))))
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    smp = RandomSampler(4096, 0, indexed=True, first_sample=0, sample_count=1)
    res = (film.desc.full_resolution[0], film.desc.full_resolution[1])
    # grow the tile sample until one run takes about 10-20 s of wall time
    n_cpu_tiles = min(n_tiles, max(n_cpu_tiles, 8 * cores))
    for _ in range(4):
        stride = max(1, n_tiles // n_cpu_tiles)
        st = si.render_parallel(scene, Film(orc, res), smp, tiles=(stride // 2, stride, n_cpu_tiles), n_threads=cores)
        secs = st["kernel_ms"] * 1e-3
        if secs >= 8.0 or n_cpu_tiles >= n_tiles:
            break
        n_cpu_tiles = int(min(n_tiles, n_cpu_tiles * min(8.0, 14.0 / max(secs, 1e-3))))
    rays = st["rays_closest"] + st["rays_any"]
    secs = st["kernel_ms"] * 1e-3
    return {"value": round(rays / secs / 1e6, 3), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": "%d of the %d film tiles (tile stride %d), 1 spp, same scene/integrator; %.1f s of wall time on %d threads; C++ restatement of fountain's CPU path "
                      "(oracle/), transcendentals from libm; oracle BVH build %.0f s not counted" % (n_cpu_tiles, n_tiles, stride, secs, cores, build_s)}


if __name__ == "__main__":
    main()
