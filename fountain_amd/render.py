"""`python -m fountain_amd.render scene.pbrt [-o out.exr] [--samples N]`: the reference's `render` binary
(src/bin/render.rs:16-104) over the MI355X library -- parse the scene file, PathIntegrator::new(5, 1.0), render, write
Film::into_spectrum_buffer as an OpenEXR file.

--threads of the reference has no meaning here; --gpu picks the HIP device.  --exact-stream renders with the reference's own
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
    opts = ap.parse_args(argv)

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
    st = integrator.render_parallel(scene, film, sampler, device=opts.gpu)
    dt = time.time() - t0
    rays = st["rays_closest"] + st["rays_any"]
    print("Completed rendering in %.3f s (%.1f Mrays/s)" % (dt, rays / dt / 1e6), file=sys.stderr)
    img, (w, h) = film.into_spectrum_buffer()
    write_exr(filename, img, be)
    return 0


if __name__ == "__main__":
    sys.exit(main())
