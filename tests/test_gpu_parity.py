"""GPU parity tests proper (-m gpu): the HIP path through the C ABI against the CPU oracle on identical inputs.
Bar: bit-exact (f32 radiance is computed with the same operation order and the same deterministic math); the only
tolerated deviation is the last bit of film pixels that received a "spill" sample (a sample whose f32 film position
lands on a pixel boundary and is added to two pixels, film.rs:138-139) -- see DESIGN.md, film accumulation.
Against the libm build of the oracle (what rustc would link) the agreement is statistical; the measured RMSE is
asserted against the tolerance stated in each test."""
import numpy as np
import pytest

from fountain_amd import (DirectLightingIntegrator, WhittedIntegrator, Film, FountainError, PathIntegrator, PerspectiveCamera, RandomSampler,
                          SamplerIntegrator, SceneBuilder, _abi as A, make_rays, scenes)

pytestmark = pytest.mark.gpu
MEGA, WAVE = A.FTN_PIPELINE_MEGAKERNEL, A.FTN_PIPELINE_WAVEFRONT


def unit_dirs(n, seed):
    rng = np.random.default_rng(seed)
    d = rng.normal(size=(n, 3))
    return (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def assert_film_equal(px_gpu, px_ref, n_spill, what):
    """Bit-exact except (at most) the pixels touched by spill samples, which may differ by a few ulp."""
    diff = (bits(px_gpu) != bits(px_ref)).any(axis=-1)
    assert int(diff.sum()) <= 4 * n_spill, "%s: %d pixels differ, only %d spill samples" % (what, int(diff.sum()), n_spill)
    assert np.allclose(px_gpu, px_ref, rtol=2e-6, atol=1e-7), what


KERNELS = ["counting", "production"]


def render_pair(gpu, ref, make, integ, sampler, pipeline, kernels="counting", **kw):
    """The same scene through the GPU library and the oracle.  kernels = "counting": the GPU runs the counting build of the REFERENCE walk
    (k_wf_trace over 32-byte nodes, MIS rays through the closest-hit kernel) so that node / triangle tallies can be compared with the
    oracle's; "production": what ftn_render runs by default and bench.py times (k_wf_trace4 / k_wf_trace4_any_dual over four-box records,
    MIS rays toward infinite lights through the any-hit kernel) -- films and ray counts must equal the oracle's just the same, the tallies
    are not produced."""
    assert kernels in KERNELS
    out = []
    for be in (gpu, ref):
        b, cam, res = make(be)
        bk = dict(pipeline=pipeline, count_traffic=kernels == "counting") if be is gpu else dict(count_traffic=True)
        rgb, px, st, _ = scenes.render(be, b, cam, res, integ, sampler, backend_kwargs=bk, **kw)
        out.append((rgb, px, st))
    return out


# ------------------------------------------------------------------ traversal (Scene::intersect / intersect_test)
def cube_scene(be):
    P, N, F = scenes.rounded_cube_mesh()
    b = SceneBuilder(be)
    b.material("none")
    b.shape("trianglemesh", P=P, N=N, indices=F)
    return b.create_scene()


def test_rounded_cube_watertight_and_bit_exact(gpu, orc_det):
    """tests/tri_watertight.rs on the GPU, plus hit records / shading geometry / node+triangle counts equal to the oracle's"""
    sg, so = cube_scene(gpu), cube_scene(orc_det)
    rays = make_rays(np.zeros((1, 3), np.float32), unit_dirs(100000, 11))
    occ, sa = sg.intersect_test(rays)
    t, prim, bary, sc = sg.intersect(rays)
    assert occ.all() and (prim >= 0).all()
    to, po, _, sco = so.intersect(rays)
    oo, sao = so.intersect_test(rays)
    assert np.array_equal(bits(t), bits(to)) and np.array_equal(prim, po) and np.array_equal(occ, oo)
    assert sc["nodes_visited"] == sco["nodes_visited"] and sc["prims_tested"] == sco["prims_tested"]
    assert sa["nodes_visited"] == sao["nodes_visited"] and sa["prims_tested"] == sao["prims_tested"]
    assert np.allclose(bary.sum(1), 1.0, atol=1e-5)
    fg, fo = sg.intersect_full(rays[:20000]), so.intersect_full(rays[:20000])
    for sl in (slice(0, 9), slice(11, 14), slice(20, 24)):          # p, p_err, n | wo | shading_n, t
        assert np.array_equal(bits(fg[:, sl]), bits(fo[:, sl]))


def test_rays_from_outside_and_short_rays(gpu, orc_det):
    """misses, grazing rays, finite t_max (shadow-ray style) and rays starting on the surface"""
    sg, so = cube_scene(gpu), cube_scene(orc_det)
    rng = np.random.default_rng(2)
    o = rng.uniform(-30, 30, (50000, 3)).astype(np.float32)
    d = unit_dirs(50000, 3) * rng.uniform(0.1, 40, (50000, 1)).astype(np.float32)
    rays = make_rays(o, d, t_max=rng.choice([np.inf, 1.0 - 1e-4, 0.5], 50000).astype(np.float32))
    t, prim, _, _ = sg.intersect(rays)
    to, po, _, _ = so.intersect(rays)
    assert np.array_equal(bits(t), bits(to)) and np.array_equal(prim, po)
    assert 0 < (prim >= 0).sum() < len(prim)
    assert np.array_equal(sg.intersect_test(rays)[0], so.intersect_test(rays)[0])


def test_sphere_bvh_vs_oracle(gpu, orc_det):
    """src/bvh.rs:401-444 scene (100 random spheres) on the GPU: any-hit <=> closest-hit, hits equal to the oracle's"""
    def make(be):
        rng = np.random.default_rng(3)
        b = SceneBuilder(be)
        b.material("none")
        for c, r in zip(rng.uniform(-10, 10, (100, 3)), rng.uniform(0.5, 3.0, 100)):
            b.attribute_begin(); b.translate(c); b.shape("sphere", radius=float(r)); b.attribute_end()
        return b.create_scene()
    sg, so = make(gpu), make(orc_det)
    rays = make_rays(np.zeros((1, 3), np.float32), unit_dirs(5000, 9))
    t, prim, _, _ = sg.intersect(rays)
    occ, _ = sg.intersect_test(rays)
    to, po, _, _ = so.intersect(rays)
    assert np.array_equal(occ, prim >= 0)
    assert np.array_equal(bits(t), bits(to)) and np.array_equal(prim, po)
    assert np.array_equal(bits(sg.intersect_full(rays)[:, :9]), bits(so.intersect_full(rays)[:, :9]))


def test_empty_scene(gpu):
    sc = SceneBuilder(gpu).create_scene()
    t, prim, _, _ = sc.intersect(make_rays([(0, 0, 0)], unit_dirs(100, 1)))
    assert (prim == -1).all() and np.isinf(t).all()


# ------------------------------------------------------------------ tests/furnace.rs on the GPU
@pytest.mark.parametrize("name,integ,expected,eps", [
    ("path", PathIntegrator.new(10, 1.0), 2.0, 0.1),
    ("path_no_rr", PathIntegrator.new(10, 0.0), 2.0, 0.001),
    ("directlighting", DirectLightingIntegrator(3), 1.5, 1e-5)])
def test_furnace_reference_sampler(gpu, orc_det, name, integ, expected, eps):
    """The three assertions of tests/furnace.rs with the reference's per-tile RandomSampler stream: one serial lane per tile in the
    megakernel, and -- PathIntegrator -- one path per tile on the wavefront queues (wavefront_render_serial), production kernels."""
    (rgb, px, st), (_, pxo, sto) = render_pair(gpu, orc_det, scenes.furnace, integ, RandomSampler(128, 0), MEGA)
    assert np.abs(rgb - expected).max() <= eps
    assert np.array_equal(bits(px), bits(pxo))
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"]
    if name.startswith("path"):
        (rgbw, pxw, stw), _ = render_pair(gpu, orc_det, scenes.furnace, integ, RandomSampler(128, 0), WAVE, "production")
        assert np.array_equal(bits(pxw), bits(pxo)) and np.abs(rgbw - expected).max() <= eps
        assert stw["rays_closest"] == sto["rays_closest"] and stw["rays_any"] == sto["rays_any"] and stw["camera_samples"] == sto["camera_samples"]


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("pipeline", [MEGA, WAVE])
def test_furnace_indexed_sampler(gpu, orc_det, pipeline, kernels):
    (rgb, px, st), (_, pxo, sto) = render_pair(gpu, orc_det, scenes.furnace, PathIntegrator.new(10, 1.0), RandomSampler(128, 0, indexed=True), pipeline, kernels)
    assert np.abs(rgb - 2.0).max() <= 0.1
    assert_film_equal(px, pxo, st["spill_samples"], "furnace")
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"]


# ------------------------------------------------------------------ render parity
def _materials_scene(be):
    """every material / lobe the reference implements: OrenNayar matte, anisotropic metal, plastic, rough glass, mirror;
    point + distant + area lights; thin-lens camera"""
    b = SceneBuilder(be)
    b.light_source("point", I=(40, 40, 40), from_=(0, -2, 3))
    b.light_source("distant", L=(1.5, 1.4, 1.2), from_=(1, -1, 2), to=(0, 0, 0))
    b.material("matte", Kd=(0.6, 0.6, 0.6), sigma=25.0)
    scenes._quad(b, (-6, -6, -1), (6, -6, -1), (6, 6, -1), (-6, 6, -1))
    b.attribute_begin(); b.material("matte", Kd=(0, 0, 0)); b.area_light_source("diffuse", L=(8, 8, 8)); b.translate((0, 0, 4)); b.reverse_orientation(); b.shape("sphere", radius=0.5, zmin=-0.5, zmax=0.2); b.attribute_end()
    specs = [("metal", dict(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), uroughness=0.02, vroughness=0.2)), ("plastic", dict(Kd=(0.3, 0.1, 0.1), Ks=(0.4, 0.4, 0.4), roughness=0.05)),
             ("glass", dict(uroughness=0.1, vroughness=0.1, eta=1.5)), ("mirror", dict()), ("metal", dict(eta=(1.5, 1.0, 0.5), k=(3, 2.5, 2), roughness=0.3, remaproughness=False))]
    for i, (m, kw) in enumerate(specs):
        b.attribute_begin(); b.material(m, **kw); b.translate((-2.4 + 1.2 * i, 0.2 * i, -0.45)); b.rotate(30.0 * i, (0.2, 0.3, 1)); b.scale(1, 1, 0.9); b.shape("sphere", radius=0.55); b.attribute_end()
    cam = PerspectiveCamera.look_at(be, (0, -7, 2.5), (0, 0, -0.3), (0, 0, 1), (96, 64), fov=42.0, lens_radius=0.08, focal_dist=7.3)
    return b, cam, (96, 64)


SCENES = {
    "cornell": (lambda be: scenes.cornell(be, res=96), 8),
    "cube_env": (lambda be: scenes.rounded_cube_env(be, res=96, env_n=64), 8),
    "materials": (_materials_scene, 8),
    "cubes27": (lambda be: scenes.instanced_cubes(be, n_copies=27, res=(80, 80), env_n=32, lens_radius=0.3), 4),
}


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("pipeline", [MEGA, WAVE])
@pytest.mark.parametrize("scene", sorted(SCENES))
def test_render_matches_oracle(gpu, orc_det, scene, pipeline, kernels):
    make, spp = SCENES[scene]
    (rgb, px, st), (rgbo, pxo, sto) = render_pair(gpu, orc_det, make, PathIntegrator.new(5, 1.0), RandomSampler(spp, 0, indexed=True), pipeline, kernels)
    assert rgb.mean() > 0.01 and np.isfinite(rgb).all()
    assert_film_equal(px, pxo, st["spill_samples"], scene)
    for k in ("rays_closest", "rays_any", "camera_samples", "spill_samples") + (("nodes_visited", "prims_tested") if kernels == "counting" else ()):
        assert st[k] == sto[k], (k, st[k], sto[k])
    rmse = float(np.sqrt(((rgb.astype(np.float64) - rgbo) ** 2).mean()))
    assert rmse <= 1e-4        # BASELINE.json: per-pixel RMSE <= 1e-4 vs the CPU reference


@pytest.mark.parametrize("scene", ["cornell", "cube_env", "materials", "cubes27"])
def test_render_vs_libm_oracle(gpu, orc, scene):
    """Against the oracle built like the reference (transcendentals from libm): <= 1 ulp differences in sin/cos/acos/atan2
    can flip a branch, which moves a pixel by O(L/spp); otherwise pixels differ in the last bits.  Stated tolerance:
    BASELINE.json's per-pixel RMSE <= 1e-4 (measured on MI355X: 7e-9), with more than half of the pixels bit-identical."""
    make, spp = SCENES[scene]
    (rgb, px, st), (rgbo, pxo, sto) = render_pair(gpu, orc, make, PathIntegrator.new(5, 1.0), RandomSampler(spp, 0, indexed=True), WAVE, "production")
    same = (bits(px) == bits(pxo)).all(axis=-1).mean()
    rmse = float(np.sqrt(((rgb.astype(np.float64) - rgbo) ** 2).mean()))
    assert same >= 0.5 and rmse <= 1e-4, (same, rmse)


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("scene", ["cornell", "materials", "cubes27", "cube_env"])
def test_tile_serial_sampler_on_the_wavefront_queues(gpu, orc_det, scene, kernels):
    """The reference's RandomSampler (one Xoshiro stream per 16 x 16 tile, random.rs:61-67) through the queue pipeline: one path per tile
    in flight, retired paths added to the film in stream order, the next camera sample drawn from where the finished path left the
    stream.  Bit-equal to the oracle (and hence to the megakernel's serial lanes), clipped edge tiles and thin-lens cameras included."""
    make, spp = SCENES[scene]
    smp = RandomSampler(3, 0)
    (rgb, px, st), (rgbo, pxo, sto) = render_pair(gpu, orc_det, make, PathIntegrator.new(5, 1.0), smp, WAVE, kernels)
    assert_film_equal(px, pxo, st["spill_samples"], scene + ", tile-serial on the queues")
    for k in ("rays_closest", "rays_any", "camera_samples", "spill_samples") + (("nodes_visited", "prims_tested") if kernels == "counting" else ()):
        assert st[k] == sto[k], (k, st[k], sto[k])
    mega = scenes.render(gpu, *make(gpu), PathIntegrator.new(5, 1.0), smp, backend_kwargs=dict(pipeline=MEGA))
    assert_film_equal(px, mega[1], st["spill_samples"], scene + ", queues vs megakernel")


def test_tile_serial_queues_many_tiles(gpu, orc_det):
    """625 tiles (the last row and column clipped to 16 x 0 ... pixels by the 400 x 400 film): more tiles than one shade workgroup holds"""
    make = lambda be: scenes.cornell(be, res=400)
    (rgb, px, st), (_, pxo, sto) = render_pair(gpu, orc_det, make, PathIntegrator.new(3, 1.0), RandomSampler(1, 0), WAVE, "production")
    assert_film_equal(px, pxo, st["spill_samples"], "625 tiles, tile-serial on the queues")
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"] and st["camera_samples"] == 400 * 400


def test_tile_serial_reference_stream(gpu, orc_det):
    """The reference's exact RandomSampler (one Xoshiro256+ stream per 16x16 tile) on a triangle + sphere scene"""
    (rgb, px, st), (_, pxo, sto) = render_pair(gpu, orc_det, lambda be: scenes.cornell(be, res=32), PathIntegrator.new(5, 1.0), RandomSampler(2, 0), MEGA)
    assert_film_equal(px, pxo, st["spill_samples"], "cornell tile-serial")
    assert st["rays_closest"] == sto["rays_closest"]


def test_direct_lighting_with_mirror_chain(gpu, orc_det):
    (rgb, px, st), (_, pxo, sto) = render_pair(gpu, orc_det, lambda be: scenes.cornell(be, res=48), DirectLightingIntegrator(4), RandomSampler(4, 0, indexed=True), MEGA)
    assert_film_equal(px, pxo, st["spill_samples"], "cornell direct lighting")
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"]


@pytest.mark.parametrize("scene", ["cornell", "materials"])
def test_whitted_integrator(gpu, orc_det, scene):
    """integrator/whitted.rs: every light sampled once per hit, specular recursion through the mirror sphere; tile-serial and indexed streams"""
    make = (lambda be: scenes.cornell(be, res=48)) if scene == "cornell" else _materials_scene
    for smp in (RandomSampler(4, 0, indexed=True), RandomSampler(2, 0)):
        (rgb, px, st), (_, pxo, sto) = render_pair(gpu, orc_det, make, WhittedIntegrator(4), smp, MEGA)
        assert_film_equal(px, pxo, st["spill_samples"], scene + " whitted")
        assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"]


# ------------------------------------------------------------------ size-independent properties at BASELINE sizes
def test_config2_sample_ranges_and_tile_shards_compose(gpu):
    """Cornell 512x512 (config 2): interleaved tile shards (the multi-GPU decomposition) added into one film equal the unsharded
    render bit for bit, and so does the megakernel; rendering the samples in two calls ([0,8) then [8,16)) gives the same film up to
    the rounding of two XYZ conversions per pixel instead of one (merge_film_tile converts a tile's RGB sum once per call)."""
    b, cam, res = scenes.cornell(gpu, res=512)
    sc = b.create_scene()
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    whole = Film(gpu, res)
    st = si.render_parallel(sc, whole, RandomSampler(16, 0, indexed=True), pipeline=WAVE)
    assert st["camera_samples"] == 512 * 512 * 16
    shards = Film(gpu, res)
    for r in range(4):
        si.render_parallel(sc, shards, RandomSampler(16, 0, indexed=True), tiles=(r, 4, 0), pipeline=WAVE)
    assert_film_equal(shards.pixels, whole.pixels, st["spill_samples"], "tile shards")
    mega = Film(gpu, res)
    si.render_parallel(sc, mega, RandomSampler(16, 0, indexed=True), pipeline=MEGA)
    assert_film_equal(mega.pixels, whole.pixels, st["spill_samples"], "megakernel vs wavefront")
    halves = Film(gpu, res)
    for first in (0, 8):
        si.render_parallel(sc, halves, RandomSampler(16, 0, indexed=True, first_sample=first, sample_count=8), pipeline=WAVE)
    assert np.array_equal(halves.pixels[..., 3], whole.pixels[..., 3]) and np.allclose(halves.pixels, whole.pixels, rtol=1e-5, atol=1e-6)
    rgb, _ = whole.into_spectrum_buffer()
    assert np.isfinite(rgb).all() and 0.05 < rgb.mean() < 5.0
    # weights: every pixel received exactly 16 samples (+ spills)
    assert np.all(whole.pixels[..., 3] >= 16) and whole.pixels[..., 3].sum() == 512 * 512 * 16 + (whole.pixels[..., 3] - 16).sum()


def test_more_paths_than_one_wavefront_pass_holds(gpu, monkeypatch):
    """The wavefront pipeline keeps a bounded number of paths in flight (FTN_WF_PATHS_M Mi, 128 by default); with 16 Mi,
    1024 x 1024 x 20 spp = 21 M camera samples take two passes (16 + 4 samples per pixel).  Same film as the megakernel, which has
    no passes, bit for bit."""
    monkeypatch.setenv("FTN_WF_PATHS_M", "16")
    b, cam, res = scenes.cornell(gpu, res=1024)
    sc = b.create_scene()
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    wave, mega = Film(gpu, res), Film(gpu, res)
    st = si.render_parallel(sc, wave, RandomSampler(20, 0, indexed=True), pipeline=WAVE)
    sm = si.render_parallel(sc, mega, RandomSampler(20, 0, indexed=True), pipeline=MEGA)
    assert st["camera_samples"] == 1024 * 1024 * 20 == sm["camera_samples"]
    assert st["rays_closest"] == sm["rays_closest"] and st["rays_any"] == sm["rays_any"]
    assert_film_equal(wave.pixels, mega.pixels, st["spill_samples"], "two passes vs megakernel")


def test_furnace_256_energy(gpu):
    """config 1 geometry at 256x256x16: the furnace value 2 - 2^-(depth) holds for every pixel without Russian roulette"""
    b, cam, res = scenes.furnace(gpu, res=256)
    rgb, px, st, _ = scenes.render(gpu, b, cam, res, PathIntegrator.new(5, 0.0), RandomSampler(16, 0, indexed=True), backend_kwargs=dict(pipeline=WAVE))
    assert np.abs(rgb - (2.0 - 2.0 ** -5)).max() < 2e-2


def _tile_subset_parity(gpu, orc_det, make, spp, tiles, pipeline=WAVE, rgb_out=None):
    out = []
    for be in (gpu, orc_det):
        b, cam, res = make(be)
        sc = b.create_scene()
        film = Film(be, res)
        si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
        kw = dict(pipeline=pipeline) if be is gpu else {}
        st = si.render_parallel(sc, film, RandomSampler(spp, 0, indexed=True), tiles=tiles, **kw)
        out.append((film.pixels, st))
        if rgb_out is not None:
            rgb_out.append(film.into_spectrum_buffer()[0])
    (px, st), (pxo, sto) = out
    assert_film_equal(px, pxo, st["spill_samples"], "tile subset")
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"] and st["camera_samples"] == sto["camera_samples"]
    return px, st


def test_config3_full_resolution_tile_subset(gpu, orc_det):
    """BASELINE config 3 (rounded_cube.ply + env map, 1024x1024) at full resolution: 24 tiles spread over the film vs the oracle"""
    px, st = _tile_subset_parity(gpu, orc_det, lambda be: scenes.rounded_cube_env(be, res=1024, env_n=512), 8, (37, 171, 24))
    assert st["camera_samples"] == 24 * 256 * 8 and (px[..., 3] > 0).sum() >= 24 * 256


def test_config4_like_full_resolution_tile_subset(gpu, orc_det):
    """BASELINE config 4 shape: mesh copies with Trowbridge-Reitz metal, image env light, thin-lens depth of field at 1920x1080
    (68 tile rows, the last one clipped to 8 pixels): 20 tiles including clipped ones vs the oracle"""
    make = lambda be: scenes.instanced_cubes(be, n_copies=46, res=(1920, 1080), env_n=256, lens_radius=0.4)
    n_tiles = Film(gpu, (1920, 1080)).tile_count()
    assert n_tiles == 120 * 68
    px, st = _tile_subset_parity(gpu, orc_det, make, 4, (n_tiles - 120 * 2 - 7, 13, 20))     # the last two tile rows
    assert 0 < st["camera_samples"] < 20 * 256 * 4                                           # some tiles are 16x8


def test_config4_env_1024_roughness_001_tile_subset(gpu, orc_det):
    """BASELINE config 4 as SURVEY 8(d) states it: a 1024 x 1024 image environment light and Trowbridge-Reitz metal at the smooth end of
    its range (roughness 0.01: near-specular lobes, the largest pdf values the MIS weights meet), thin lens, 1920 x 1080; 16 tiles vs the oracle"""
    make = lambda be: scenes.instanced_cubes(be, n_copies=46, res=(1920, 1080), env_n=1024, lens_radius=0.4, metal_roughness=0.01)
    px, st = _tile_subset_parity(gpu, orc_det, make, 8, (120 * 20 + 17, 211, 16))
    assert st["camera_samples"] == 16 * 256 * 8


@pytest.mark.parametrize("config", ["3", "4", "5"])
def test_baseline_tile_subsets_vs_libm_oracle(gpu, orc, config):
    """The north-star tolerance on BASELINE-size workloads against the oracle built the way rustc would build the reference (sin / cos /
    acos / atan2 / ln from libm instead of the product's deterministic versions): per-pixel RMSE of the resolved RGB over the rendered tiles
    <= 1e-4 (BASELINE.json), production kernels.  A <= 1 ulp difference in a transcendental can flip a Russian-roulette or lobe choice,
    which moves one pixel by O(L / spp): the ray counts may therefore differ by a few rays, the films agree statistically."""
    if config == "3":
        make, spp, tiles = (lambda be: scenes.rounded_cube_env(be, res=1024, env_n=512)), 8, (37, 171, 24)
    elif config == "4":
        make, spp, tiles = (lambda be: scenes.instanced_cubes(be, n_copies=46, res=(1920, 1080), env_n=1024, lens_radius=0.4, metal_roughness=0.01)), 8, (120 * 20 + 17, 211, 16)
    else:
        make, spp, tiles = (lambda be: scenes.instanced_cubes(be, n_copies=2309, res=(4096, 4096))), 2, (256 * 100 + 31, 2731, 12)
    films = []
    for be in (gpu, orc):
        b, cam, res = make(be)
        film = Film(be, res)
        kw = dict(pipeline=WAVE) if be is gpu else {}
        SamplerIntegrator(cam, PathIntegrator.new(5, 1.0)).render_parallel(b.create_scene(), film, RandomSampler(spp, 0, indexed=True), tiles=tiles, **kw)
        films.append((film.into_spectrum_buffer()[0], film.pixels[..., 3] > 0))
    (rgb, m), (rgbo, mo) = films
    assert np.array_equal(m, mo) and m.sum() >= 12 * 256
    rmse = float(np.sqrt(((rgb[m].astype(np.float64) - rgbo[m]) ** 2).mean()))
    same = float((bits(rgb[m]) == bits(rgbo[m])).all(axis=-1).mean())
    print("config %s vs the libm oracle: RMSE %.3g over %d pixels, %.1f %% of them bit-identical" % (config, rmse, int(m.sum()), 100.0 * same))
    assert rmse <= 1e-4, (rmse, same)


def test_config5_full_size(gpu, orc_det):
    """BASELINE config 5 at its full size (2309 baked copies = 10,002,588 triangles, 4096x4096 film, the bench.py workload):
    (a) 12 tiles spread over the film against the oracle, bit for bit, with equal ray counts;
    (b) size-independent properties over the WHOLE film: two interleaved tile shards (what two GPUs would render) merged by
        addition equal the unsharded film, and a second render of the same samples is bit-identical (no race in the film sums)."""
    make = lambda be: scenes.instanced_cubes(be, n_copies=2309, res=(4096, 4096))
    n_tiles = Film(gpu, (4096, 4096)).tile_count()
    assert n_tiles == 256 * 256
    px, st = _tile_subset_parity(gpu, orc_det, make, 1, (256 * 100 + 31, 2731, 12))
    assert st["camera_samples"] == 12 * 256
    b, cam, res = make(gpu)
    sc = b.create_scene()
    assert sc.info()["n_prims"] == 10002588
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    smp = lambda: RandomSampler(4096, 0, indexed=True, first_sample=7, sample_count=1)
    whole, again, merged = Film(gpu, res), Film(gpu, res), Film(gpu, res)
    st_w = si.render_parallel(sc, whole, smp(), pipeline=WAVE)
    si.render_parallel(sc, again, smp(), pipeline=WAVE)
    st_a = si.render_parallel(sc, merged, smp(), tiles=(0, 2, 0), pipeline=WAVE)
    st_b = si.render_parallel(sc, merged, smp(), tiles=(1, 2, 0), pipeline=WAVE)
    assert np.array_equal(bits(whole.pixels), bits(again.pixels))
    assert st_a["rays_closest"] + st_b["rays_closest"] == st_w["rays_closest"] and st_a["rays_any"] + st_b["rays_any"] == st_w["rays_any"]
    assert st_w["camera_samples"] == 4096 * 4096
    assert_film_equal(merged.pixels, whole.pixels, st_w["spill_samples"], "config 5 shards")
    assert whole.pixels[..., 3].min() >= 1.0


@pytest.mark.parametrize("config", ["2: 64 spp", "3: 256 spp", "4-like: 1024 spp", "5: 4096 spp"])
def test_baseline_configs_at_their_full_spp_on_a_few_tiles(gpu, orc_det, config):
    """The other BASELINE-size tests cut the sample count; samples of a pixel are added in order, so the FULL count matters for the film
    sums.  Here every config runs at its own spp (64 / 256 / 1024 / 4096) on a handful of tiles -- what the oracle finishes in seconds --
    and the films and ray counts must equal the oracle's bit for bit."""
    if config.startswith("2"):
        make, spp, tiles, n = (lambda be: scenes.cornell(be, res=512)), 64, (5 * 32 + 9, 97, 8), 8
    elif config.startswith("3"):
        make, spp, tiles, n = (lambda be: scenes.rounded_cube_env(be, res=1024, env_n=512)), 256, (64 * 30 + 28, 130, 6), 6
    elif config.startswith("4"):
        make, spp, tiles, n = (lambda be: scenes.instanced_cubes(be, n_copies=46, res=(1920, 1080), env_n=256, lens_radius=0.4)), 1024, (120 * 30 + 55, 7, 3), 3
    else:
        make, spp, tiles, n = (lambda be: scenes.instanced_cubes(be, n_copies=2309, res=(4096, 4096))), 4096, (256 * 120 + 140, 517, 2), 2
    px, st = _tile_subset_parity(gpu, orc_det, make, spp, tiles)
    assert st["camera_samples"] == n * 256 * spp
    # every pixel of the chosen tiles got all its samples (a jitter of exactly 0 puts a sample on two pixels, film.rs:138-139: counted as spill)
    assert (px[..., 3] == float(spp)).sum() >= n * 256 - 4 * st["spill_samples"]


def test_ray_queue_sort_changes_nothing(gpu, monkeypatch):
    """The wavefront pipeline sorts its secondary-ray queues for coherence; the order rays are traced in must not change
    a single bit of the film or a single counter: same render with the sort disabled."""
    make = lambda be: scenes.instanced_cubes(be, n_copies=27, res=(160, 160), env_n=32)     # > 16 K rays per bounce, so the sort really runs
    b, cam, res = make(gpu)
    sc = b.create_scene()
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    films, stats = [], []
    for flag in ("1", "0"):
        monkeypatch.setenv("FTN_WF_SORT", flag)
        f = Film(gpu, res)
        stats.append(si.render_parallel(sc, f, RandomSampler(2, 0, indexed=True), pipeline=WAVE, count_traffic=True))
        films.append(f.pixels)
    assert np.array_equal(bits(films[0]), bits(films[1]))
    for k in ("rays_closest", "rays_any", "nodes_visited", "prims_tested", "camera_samples"):
        assert stats[0][k] == stats[1][k], k


def test_any_hit_launches_beside_the_closest_hit_ones_change_nothing(gpu, monkeypatch):
    """From the second bounce on, the any-hit trace of a bounce runs on a second (low-priority) stream beside the closest-hit trace
    (it fills the GPU while the other kernel drains).  Same film, same counters as with one stream."""
    b, cam, res = scenes.instanced_cubes(gpu, n_copies=27, res=(160, 160), env_n=32)
    sc = b.create_scene()
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    films, stats = [], []
    for flag in ("1", "0"):
        monkeypatch.setenv("FTN_WF_OVERLAP", flag)
        for rep in range(2):                       # twice: the streams and events are reused from call to call
            f = Film(gpu, res)
            stats.append(si.render_parallel(sc, f, RandomSampler(2, 0, indexed=True), pipeline=WAVE, count_traffic=2))
            films.append(f.pixels)
    for k in range(1, 4):
        assert np.array_equal(bits(films[0]), bits(films[k]))
        for c in ("rays_closest", "rays_any", "nodes_visited", "prims_tested", "camera_samples", "mis_rays_any_hit"):
            assert stats[0][c] == stats[k][c], c


def test_mis_rays_through_the_any_hit_kernel_change_nothing(gpu, orc_det, monkeypatch):
    """estimate_direct's BSDF-sampled ray toward an INFINITE light only needs hit / miss (integrator/mod.rs:367-384: a hit primitive can
    only contribute if it is THIS area light), so outside the reference-order counting build it goes through the any-hit kernel.
    Same film bit for bit, same ray counts in the reference's accounting, and identical to the oracle."""
    make = lambda be: scenes.instanced_cubes(be, n_copies=27, res=(160, 160), env_n=32)
    b, cam, res = make(gpu)
    sc = b.create_scene()
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    films, stats = {}, {}
    for flag in ("1", "0"):
        monkeypatch.setenv("FTN_MIS_ANY", flag)
        f = Film(gpu, res)
        stats[flag] = si.render_parallel(sc, f, RandomSampler(2, 0, indexed=True), pipeline=WAVE)
        films[flag] = f.pixels
    monkeypatch.delenv("FTN_MIS_ANY")
    assert np.array_equal(bits(films["1"]), bits(films["0"]))
    assert stats["1"]["mis_rays_any_hit"] > 0 and stats["0"]["mis_rays_any_hit"] == 0
    assert stats["1"]["rays_closest"] == stats["0"]["rays_closest"] and stats["1"]["rays_any"] == stats["0"]["rays_any"]
    counted = si.render_parallel(sc, Film(gpu, res), RandomSampler(2, 0, indexed=True), pipeline=WAVE, count_traffic=True)
    assert counted["mis_rays_any_hit"] == 0 and counted["rays_closest"] == stats["1"]["rays_closest"]          # reference-order tally
    prod = si.render_parallel(sc, Film(gpu, res), RandomSampler(2, 0, indexed=True), pipeline=WAVE, count_traffic=2)
    assert prod["mis_rays_any_hit"] == stats["1"]["mis_rays_any_hit"] and prod["nodes_visited"] < counted["nodes_visited"]   # early exits
    bo, camo, _ = make(orc_det)
    fo = Film(orc_det, res)
    sto = SamplerIntegrator(camo, PathIntegrator.new(5, 1.0)).render_parallel(bo.create_scene(), fo, RandomSampler(2, 0, indexed=True))
    assert_film_equal(films["1"], fo.pixels, stats["1"]["spill_samples"], "mis-any vs oracle")
    assert sto["rays_closest"] == stats["1"]["rays_closest"] and sto["rays_any"] == stats["1"]["rays_any"]


def _env_only_scene(be, res=(96, 80)):
    """every material the path supports, lit by ONE infinite light (rotated 16x16 map)"""
    rng = np.random.default_rng(7)
    b = SceneBuilder(be)
    b.attribute_begin(); b.rotate(37.0, (0.3, 0.2, 1.0)); b.light_source("infinite", texels=(rng.random((16, 16, 3)) ** 3 * 2).astype(np.float32)); b.attribute_end()
    b.material("matte", Kd=(0.5, 0.45, 0.4), sigma=20.0)
    b.shape("trianglemesh", P=[(-8, -8, -1), (8, -8, -1), (8, 8, -1), (-8, 8, -1)], indices=[0, 1, 2, 0, 2, 3])
    mats = [("matte", dict(Kd=(0.7, 0.2, 0.2))), ("mirror", dict(Kr=(0.9, 0.9, 0.9))), ("plastic", dict(Kd=(0.2, 0.5, 0.3), Ks=(0.4, 0.4, 0.4), roughness=0.1)),
            ("metal", dict(eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=0.05)), ("glass", dict(uroughness=0.2, vroughness=0.3, eta=1.5, remaproughness=False))]
    for i, (name, kw) in enumerate(mats):
        b.attribute_begin(); b.material(name, **kw); b.translate((-3.0 + 1.5 * i, 0.5 * (i % 2), 0.0)); b.shape("sphere", radius=0.7); b.attribute_end()
    cam = PerspectiveCamera.look_at(be, (0.5, -7.0, 2.0), (0.0, 0.0, 0.0), (0, 0, 1), res, fov=45.0)
    return b.create_scene(), cam, res


def test_environment_only_shading_kernels_change_nothing(gpu, orc_det, monkeypatch):
    """A scene whose only light is an InfiniteAreaLight is shaded by kernels specialised for it (light record in the kernel arguments,
    other light kinds compiled out: DESIGN.md).  Bit-identical film and counts to the generic kernels, with the MIS rays in either
    trace kernel, and identical to the oracle."""
    sc, cam, res = _env_only_scene(gpu)
    si = SamplerIntegrator(cam, PathIntegrator.new(6, 1.0))
    films, stats = {}, {}
    for env in ("1", "0"):
        for mis in ("1", "0"):
            monkeypatch.setenv("FTN_SHADE_ENV", env); monkeypatch.setenv("FTN_MIS_ANY", mis)
            f = Film(gpu, res)
            stats[env + mis] = si.render_parallel(sc, f, RandomSampler(4, 3, indexed=True), pipeline=WAVE)
            films[env + mis] = f.pixels
    monkeypatch.delenv("FTN_SHADE_ENV"); monkeypatch.delenv("FTN_MIS_ANY")
    for k in ("10", "01", "00"):
        assert np.array_equal(bits(films["11"]), bits(films[k])), k
        assert all(stats["11"][c] == stats[k][c] for c in ("rays_closest", "rays_any", "camera_samples")), k
    sco, camo, _ = _env_only_scene(orc_det)
    fo = Film(orc_det, res)
    sto = SamplerIntegrator(camo, PathIntegrator.new(6, 1.0)).render_parallel(sco, fo, RandomSampler(4, 3, indexed=True))
    assert_film_equal(films["11"], fo.pixels, stats["11"]["spill_samples"], "env-only vs oracle")
    assert sto["rays_closest"] == stats["11"]["rays_closest"] and sto["rays_any"] == stats["11"]["rays_any"]


@pytest.mark.parametrize("pipeline", [WAVE, MEGA])
def test_wide_box_filter_then_default_filter(gpu, orc_det, pipeline):
    """BoxFilter of radius 1.25: every sample lands on 4-9 pixels, i.e. almost everything goes through the spill accumulators (float
    atomics: sums agree to rounding, weights exactly).  Then the default radius twice on the same scene handle: the spill accumulators
    must have been cleared after the wide render and are skipped while they are known to be clean; bit-exact both times."""
    sc, cam, res = _env_only_scene(gpu, res=(64, 48))
    sco, camo, _ = _env_only_scene(orc_det, res=(64, 48))
    si, sio = SamplerIntegrator(cam, PathIntegrator.new(4, 1.0)), SamplerIntegrator(camo, PathIntegrator.new(4, 1.0))
    for radius, exact in ((1.25, False), (0.5, True), (0.5, True)):
        f, fo = Film(gpu, res, (0.1, 0.2, 0.9, 1.0)), Film(orc_det, res, (0.1, 0.2, 0.9, 1.0))
        for film in (f, fo):
            film.desc.filter_radius[0] = radius; film.desc.filter_radius[1] = radius
        st = si.render_parallel(sc, f, RandomSampler(3, 1, indexed=True), pipeline=pipeline)
        sto = sio.render_parallel(sco, fo, RandomSampler(3, 1, indexed=True))
        assert st["camera_samples"] == sto["camera_samples"] and st["spill_samples"] == sto["spill_samples"]
        assert np.array_equal(f.pixels[..., 3], fo.pixels[..., 3])                      # filter_weight_sum: whole numbers
        if exact:
            assert_film_equal(f.pixels, fo.pixels, st["spill_samples"], "default filter after a wide one")
        else:
            assert st["spill_samples"] > 0.9 * st["camera_samples"]
            assert np.allclose(f.pixels, fo.pixels, rtol=1e-5, atol=1e-6)


# ------------------------------------------------------------------ error behaviour
def test_specular_glass_reports_unsupported(gpu):
    b = SceneBuilder(gpu)
    b.light_source("point", I=(10, 10, 10), from_=(0, 0, 5))
    b.material("glass", remaproughness=False)
    b.shape("sphere", radius=1.0)
    cam = PerspectiveCamera.look_at(gpu, (0, -4, 0), (0, 0, 0), (0, 0, 1), (16, 16), fov=40.0)
    for pl in (MEGA, WAVE):
        with pytest.raises(FountainError) as e:
            scenes.render(gpu, b, cam, (16, 16), PathIntegrator.new(5, 1.0), RandomSampler(2, 0, indexed=True), backend_kwargs=dict(pipeline=pl))
        assert e.value.code == A.FTN_ERR_UNSUPPORTED


def test_null_material_pass_through(gpu, orc_det):
    """path.rs:77-81: a primitive without material is skipped without counting a bounce"""
    def make(be):
        b, cam, res = scenes.cornell(be, res=48)
        b.attribute_begin(); b.material("none"); b.translate((0, -1.5, 0)); b.shape("sphere", radius=0.5); b.attribute_end()
        return b, cam, res
    for pl, kernels in ((MEGA, "counting"), (WAVE, "counting"), (WAVE, "production")):
        (rgb, px, st), (_, pxo, sto) = render_pair(gpu, orc_det, make, PathIntegrator.new(3, 1.0), RandomSampler(4, 0, indexed=True), pl, kernels)
        assert_film_equal(px, pxo, st["spill_samples"], "null material")
        assert st["rays_closest"] == sto["rays_closest"]


# ------------------------------------------------------------------ round-1 review findings
@pytest.mark.parametrize("pipeline", [MEGA, WAVE])
def test_tile_selection_cache_full_empty_full(gpu, pipeline):
    """full film, then a tile range that selects nothing, then the full film again on the SAME scene handle: the third call must
    render what the first did (the cached tile list used to survive the empty call under the old key)."""
    b, cam, res = scenes.cornell(gpu, res=48)
    scene = b.create_scene()
    si = SamplerIntegrator(cam, PathIntegrator(3, 1.0))
    smp = RandomSampler(2, 0, indexed=True)
    films = []
    for tiles in (None, (1000, 1, 1000), None, (2, 3, 2), None):
        film = Film(gpu, res)
        st = si.render_parallel(scene, film, smp, tiles=tiles, pipeline=pipeline)
        films.append((film.pixels.copy(), st))
    assert films[1][1]["camera_samples"] == 0 and not films[1][0].any()
    assert films[0][1]["camera_samples"] == 48 * 48 * 2
    for k in (2, 4):
        assert films[k][1]["camera_samples"] == films[0][1]["camera_samples"]
        assert np.array_equal(bits(films[k][0]), bits(films[0][0]))
    assert films[3][1]["camera_samples"] == 2 * 256 * 2


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("pipeline", [MEGA, WAVE])
@pytest.mark.parametrize("depth", [0, 1, 2])
def test_depth_limits_zero_one_two(gpu, orc_det, pipeline, depth, kernels):
    """max_depth 0 (emission / environment only: camera rays arrive at the depth limit), 1 and 2, on the scene with every light kind and
    on the environment-lit cube: the depth limit travels through the active queue's entry flags (k_wf_classify), not through the path state."""
    integ, smp = PathIntegrator(depth, 1.0), RandomSampler(2, 0, indexed=True)
    for make, what in ((lambda be: scenes.cornell(be, res=32), "cornell"), (lambda be: scenes.rounded_cube_env(be, res=32, env_n=64), "rounded_cube_env")):
        (rg, pg, sg), (ro, po, so) = render_pair(gpu, orc_det, make, integ, smp, pipeline, kernels)
        assert sg["rays_closest"] == so["rays_closest"] and sg["rays_any"] == so["rays_any"]
        assert_film_equal(pg, po, sg["spill_samples"], "%s, max_depth %d" % (what, depth))


def test_wavefront_paths_deeper_than_255_bounces(gpu, orc_det):
    """max_depth 300, no Russian roulette, albedo-0.5 furnace: paths run the full 300 bounces; the wavefront's bounce counter used to
    be 8 bits wide.  Bit-exact against the oracle and the megakernel."""
    integ, smp = PathIntegrator(300, 0.0), RandomSampler(2, 0, indexed=True)
    (rg, pg, sg), (ro, po, so) = render_pair(gpu, orc_det, lambda be: scenes.furnace(be, res=16), integ, smp, WAVE, "production")
    assert sg["rays_closest"] == so["rays_closest"] and sg["rays_closest"] >= 16 * 16 * 2 * 300
    assert_film_equal(pg, po, sg["spill_samples"], "furnace, depth 300")
    with pytest.raises(FountainError):
        scenes.render(gpu, *scenes.furnace(gpu, res=16), PathIntegrator(70000, 0.0), smp, backend_kwargs=dict(pipeline=WAVE))


# ------------------------------------------------------------------ production traversal kernels (four-box records, ftn_trace4.hip)
def _ray_mix(n, seed, lo=-30.0, hi=30.0):
    """random rays + the exceptional ones: axis-parallel directions (exact zeros, negative zeros) from origins on round coordinates"""
    rng = np.random.default_rng(seed)
    o = rng.uniform(lo, hi, (n, 3)).astype(np.float32)
    d = unit_dirs(n, seed + 1) * rng.uniform(0.1, 40, (n, 1)).astype(np.float32)
    k = n // 8
    d[:k, 0] = 0.0
    d[k // 2:k, 1] = -0.0
    d[k:k + k // 2] = np.array([0.0, 0.0, -1.0], np.float32)
    o[:k] = np.round(o[:k])
    o[k:k + k // 2, 2] = 25.0
    t_max = rng.choice([np.inf, 1.0 - 1e-4, 0.5], n).astype(np.float32)
    return make_rays(o, d, t_max=t_max)


@pytest.mark.parametrize("scene", ["cube", "cubes27", "cornell"])
def test_four_box_kernels_equal_the_oracle_on_ray_batches(gpu, orc_det, scene):
    """ftn_intersect / ftn_intersect_test without statistics run the production kernels (k_wf_trace4 / k_wf_trace4_any over 128-byte
    four-box records; rays with a zero direction component are handed to the reference-order kernel): t, primitive, barycentrics and
    occlusion bit-equal to the oracle's and to the counting build of the reference walk."""
    if scene == "cube":
        sg, so = cube_scene(gpu), cube_scene(orc_det)
    elif scene == "cubes27":
        sg, so = (scenes.instanced_cubes(be, n_copies=27, res=(16, 16), env_n=8, spacing=14.0)[0].create_scene() for be in (gpu, orc_det))
    else:
        sg, so = (scenes.cornell(be, res=16)[0].create_scene() for be in (gpu, orc_det))
    lo, hi = (-30.0, 30.0) if scene != "cornell" else (-1.5, 1.5)
    rays = _ray_mix(60000, 21, lo, hi)
    if scene == "cornell":
        rays[:, 3:6] *= 0.05
    t, prim, bary, none = sg.intersect(rays, stats=False)
    assert none is None
    tc, pc, bc, stc = sg.intersect(rays)
    to, po, bo, sto = so.intersect(rays)
    assert 0 < (po >= 0).sum() < len(po)
    assert np.array_equal(bits(t), bits(to)) and np.array_equal(prim, po)
    assert np.array_equal(bits(tc), bits(to)) and np.array_equal(pc, po) and stc["nodes_visited"] == sto["nodes_visited"]
    assert np.array_equal(bits(bary), bits(bc))          # (the oracle's intersect does not return barycentrics: the full records below pin them)
    occ, _ = sg.intersect_test(rays, stats=False)
    occ_o, _ = so.intersect_test(rays)
    assert np.array_equal(occ, occ_o) and 0 < occ.sum() < len(occ)
    full_g, full_o = sg.intersect_full(rays[:8000]), so.intersect_full(rays[:8000])
    assert np.array_equal(bits(full_g[:, :9]), bits(full_o[:, :9])) and np.array_equal(bits(full_g[:, 20:24]), bits(full_o[:, 20:24]))


def test_occlusion_records_edge_cases(gpu, orc_det, monkeypatch):
    """k_wf_trace8_any (eight-box occlusion records) where its special paths run: leaves of several primitives (coincident triangles share a
    centroid, so the reference's build cannot split them: the leaf's exact box comes from `oct_xbox` and its primitives are walked in order),
    flat and tiny triangles next to huge ones (step exponents, zero-extent axes), rays with direction components down to 1e-30 and origins
    far outside the scene (beyond the range the conservative test is proven for: handed to the reference-order kernel), a two-level LDS
    stack (every push spills).  Occlusion bit-equal to the oracle's and to the four-box kernel's."""
    def make(be):
        rng = np.random.default_rng(17)
        b = SceneBuilder(be); b.material("matte")
        P, N, F = scenes.rounded_cube_mesh()
        b.shape("trianglemesh", P=P, N=N, indices=F)
        tri = np.array([(2.0, 12.0, 1.0), (6.0, 12.0, 1.5), (3.0, 12.5, 6.0)], np.float32)
        for k in range(5):                                           # five triangles with one centroid: a leaf of five primitives
            b.shape("trianglemesh", P=np.roll(tri, k % 3, axis=0), indices=[0, 1, 2])
        for _ in range(60):
            c = rng.uniform(-25, 25, 3); sz = 10.0 ** rng.uniform(-3, 1.2)
            Pt = (c + rng.normal(size=(3, 3)) * sz).astype(np.float32)
            if rng.random() < 0.3: Pt[:, rng.integers(3)] = np.float32(c[0])          # flat along one axis
            b.shape("trianglemesh", P=Pt, indices=[0, 1, 2])
        b.shape("trianglemesh", P=[(-400, -400, -30), (400, -400, -30), (0, 500, -30)], indices=[0, 1, 2])     # one huge triangle
        return b.create_scene()
    sg, so = make(gpu), make(orc_det)
    assert sg.info()["oct_bytes"] > 0
    rays = _ray_mix(40000, 5)
    rng = np.random.default_rng(6)
    extra = _ray_mix(6000, 7)
    extra[:2000, 3] = np.float32(1e-30) * np.sign(rng.normal(size=2000)).astype(np.float32)       # |1/d| = 1e30: outside the proven range
    extra[2000:4000, 0:3] *= np.float32(1e13)                                                         # origins far away (|o| > 2^40)
    extra[2000:4000, 3:6] = -extra[2000:4000, 0:3] / np.float32(1e12)                                 # ... aiming at the scene
    extra[4000:, 5] = np.float32(3e-20)
    rays = np.concatenate([rays, extra]).astype(np.float32)
    occ_o, _ = so.intersect_test(rays)
    occ, _ = sg.intersect_test(rays, stats=False)
    assert np.array_equal(occ, occ_o) and 0 < occ.sum() < len(occ)
    monkeypatch.setenv("FTN_T8", "0")                                # the four-box any-hit kernel on the same rays
    occ4, _ = sg.intersect_test(rays, stats=False)
    monkeypatch.delenv("FTN_T8")
    assert np.array_equal(occ4, occ_o)
    monkeypatch.setenv("FTN_T8_ENTRIES", "2")
    assert np.array_equal(sg.intersect_test(rays, stats=False)[0], occ_o)


def test_four_box_kernels_render_like_the_two_record_kernels(gpu, orc_det, monkeypatch):
    """the same render with the four-box kernels (default), with a two-level LDS stack (nearly every push spills to global memory), and
    with FTN_TRACE4=0 (round 1's kernels): one film, one set of counters; and it is the oracle's film"""
    make = lambda be: scenes.instanced_cubes(be, n_copies=27, res=(160, 160), env_n=32)
    b, cam, res = make(gpu)
    sc = b.create_scene()
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    films, stats = [], []
    for env in ({}, {"FTN_T4_ENTRIES": "2", "FTN_T4_ENTRIES_ANY": "2"}, {"FTN_T4_BURST": "4", "FTN_T4_WG": "3"}, {"FTN_TRACE4": "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        f = Film(gpu, res)
        stats.append(si.render_parallel(sc, f, RandomSampler(2, 0, indexed=True), pipeline=WAVE, count_traffic=2))
        films.append(f.pixels)
        for k in env:
            monkeypatch.delenv(k)
    for k in range(1, 4):
        assert np.array_equal(bits(films[0]), bits(films[k])), k
        for c in ("rays_closest", "rays_any", "camera_samples", "mis_rays_any_hit"):
            assert stats[0][c] == stats[k][c], (k, c)
    assert stats[0]["quad_records"] == stats[1]["quad_records"] > 0 and stats[3]["quad_records"] == 0
    # one 128-byte record fetch decides two levels: fewer fetches than the two-record walk visits nodes
    assert stats[0]["quad_records"] * 2 < stats[3]["nodes_visited"]
    bo, camo, _ = make(orc_det)
    fo = Film(orc_det, res)
    sto = SamplerIntegrator(camo, PathIntegrator.new(5, 1.0)).render_parallel(bo.create_scene(), fo, RandomSampler(2, 0, indexed=True))
    assert_film_equal(films[0], fo.pixels, stats[0]["spill_samples"], "four-box kernels vs oracle")
    assert sto["rays_closest"] == stats[0]["rays_closest"] and sto["rays_any"] == stats[0]["rays_any"]


def test_furnace_path_no_rr_on_the_wavefront_pipeline(gpu, orc_det):
    """tests/furnace.rs `path_no_rr` (PathIntegrator(10, 0.0) in the albedo-0.5 furnace) on the pipeline the bench times, with the indexed
    sampler stream.  With max_depth 10 the estimator's mean is 2 - 2^-10 (the series 1 + 1/2 + 1/4 + ... is cut after ten scattering
    terms), which sits 2.3e-5 inside the reference's +-0.001 window around 2.0 -- the reference's assertion holds for its own stream at
    128 spp but is no statement about another stream's Monte-Carlo noise.  So: 1024 spp (noise ~3e-5 per pixel), every pixel within
    +-0.001 of the true mean 2 - 2^-10 and the image mean within 1e-4 of it; bit-exact against the oracle; every path takes all 11 segments (plus the MIS rays of its direct-lighting estimates)."""
    integ, smp = PathIntegrator.new(10, 0.0), RandomSampler(1024, 0, indexed=True)
    (rgb, px, st), (rgbo, pxo, sto) = render_pair(gpu, orc_det, scenes.furnace, integ, smp, WAVE)
    mean = 2.0 - 2.0 ** -10
    assert np.abs(rgb - mean).max() <= 0.001, float(np.abs(rgb - mean).max())
    assert abs(float(rgb.mean()) - mean) <= 1e-4 and np.abs(rgb - 2.0).max() <= 0.002
    assert_film_equal(px, pxo, st["spill_samples"], "furnace path_no_rr, wavefront")
    assert st["rays_closest"] == sto["rays_closest"] >= 16 * 16 * 1024 * 11 and st["rays_any"] == sto["rays_any"]
    prod = scenes.render(gpu, *scenes.furnace(gpu), integ, smp, backend_kwargs=dict(pipeline=WAVE))     # production kernels (no counting build)
    assert np.array_equal(bits(prod[1]), bits(px))


@pytest.mark.parametrize("pipeline,kernels", [(MEGA, "counting"), (WAVE, "counting"), (WAVE, "production")])
@pytest.mark.parametrize("shape", [(6, 3), (3, 6), (8, 4), (5, 7), (1, 4), (12, 1), (100, 37), (1, 1), (2, 2), (7, 7), (16, 16), (33, 33)])
def test_environment_maps_of_any_size(gpu, orc_det, shape, pipeline, kernels):
    """non-square and non-power-of-two environment maps (infinite.rs:63-77: only pyramid level 0 is ever read; the (height, width) name
    swap of compute_distribution reproduced as written): importance sampling, pdf and Le bit-equal to the oracle's.  Square maps read
    the cell records (DLight::cells: a cell's 3 x 3 texel neighbourhood, wrapped at the borders, and its function value in one line)"""
    from test_oracle_integration import _env_scene
    rng = np.random.default_rng(shape[0] * 131 + shape[1])
    tex = (rng.random(shape + (3,)) ** 3 * 2).astype(np.float32)
    tex[shape[0] // 2, shape[1] // 3] = 30.0
    (rgb, px, st), (rgbo, pxo, sto) = render_pair(gpu, orc_det, lambda be: _env_scene(be, tex), PathIntegrator.new(4, 1.0), RandomSampler(4, 0, indexed=True), pipeline, kernels)
    assert_film_equal(px, pxo, st["spill_samples"], "env map %dx%d" % shape)
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"] and rgb.max() > 0.5


def test_launch_schedules_change_nothing(gpu, monkeypatch):
    """the control parameters of the traversal launches -- the camera rays' lockstep launch (a wave re-arms only when all 64 lanes are done),
    the 2 x 2 pixel-block queue order of full tiles, re-arm thresholds, leaf batches -- decide WHEN a ray is walked, never what it hits:
    same film bit for bit, same counts"""
    sc, cam, res = _env_only_scene(gpu, res=(128, 96))
    si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    out = {}
    for name, env in (("default", {}), ("no lockstep", {"FTN_T4_REFILL0": "24", "FTN_T4_LEAF_BATCH0": "16", "FTN_T4_BURST0": "2"}), ("row order", {"FTN_GEN_BLOCKS": "0"}),
                      ("odd", {"FTN_T4_REFILL": "7", "FTN_T8_REFILL1": "60", "FTN_T4_LEAF_BATCH0": "64", "FTN_T4_BURST0": "1", "FTN_T8_LEAF_BATCH1": "3"})):
        for k, v in env.items(): monkeypatch.setenv(k, v)
        f = Film(gpu, res)
        st = si.render_parallel(sc, f, RandomSampler(8, 1, indexed=True), pipeline=WAVE)
        out[name] = (f.pixels, st["rays_closest"], st["rays_any"])
        for k in env: monkeypatch.delenv(k)
    for name, v in out.items():
        assert np.array_equal(bits(v[0]), bits(out["default"][0])) and v[1:] == out["default"][1:], name


def test_path_record_layouts_change_nothing(gpu, orc_det, monkeypatch):
    """where the path integrator keeps a path's state between two events -- SoA arrays (FTN_WF_BR=0, FTN_WF_PD=0), throughput + radiance in one
    32-byte record, the pending direct-light terms in one 64-byte record (FTN_WF_PD=1), with the MIS ray's direction behind them (2) and the
    any-hit kernels' results in the record's last word (3, the default: an any-hit kernel then only stores "occluded", the event that queued
    the ray having written "not occluded") -- and whether its Xoshiro stream is carried or replayed decide where bytes live, never their
    value: same film bit for bit, same counts, and the default equals the oracle.  Scenes: triangles under an environment map (shadow and
    MIS rays through k_wf_trace8_any), spheres of every material under one (the four-box any-hit kernel), the all-materials scene (area
    lights: MIS rays through the closest-hit kernel)"""

    def cubes(be):
        b, cam, res = scenes.instanced_cubes(be, n_copies=5, res=(96, 96), env_n=32)
        return b.create_scene(), cam, res

    def materials(be):
        b, cam, res = _materials_scene(be)
        return b.create_scene(), cam, res

    for make, smp in ((cubes, RandomSampler(8, 2, indexed=True)), (lambda be: _env_only_scene(be, res=(96, 64)), RandomSampler(8, 2, indexed=True)), (materials, RandomSampler(4, 5, indexed=True))):
        sc, cam, res = make(gpu)
        si = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
        out = {}
        for name, env in (("default", {}), ("soa", {"FTN_WF_BR": "0", "FTN_WF_PD": "0"}), ("pd1", {"FTN_WF_PD": "1"}), ("pd2", {"FTN_WF_PD": "2"}), ("pd3 br0", {"FTN_WF_PD": "3", "FTN_WF_BR": "0"}),
                          ("carried streams", {"FTN_RNG_REPLAY": "0"}), ("two streams", {"FTN_WF_OVERLAP": "1"})):
            for k, v in env.items(): monkeypatch.setenv(k, v)
            f = Film(gpu, res)
            st = si.render_parallel(sc, f, smp, pipeline=WAVE)
            out[name] = (f.pixels, st["rays_closest"], st["rays_any"], st["spill_samples"])
            for k in env: monkeypatch.delenv(k)
        for name, v in out.items():
            assert np.array_equal(bits(v[0]), bits(out["default"][0])) and v[1:3] == out["default"][1:3], name
        sco, camo, _ = make(orc_det)
        fo = Film(orc_det, res)
        sto = SamplerIntegrator(camo, PathIntegrator.new(5, 1.0)).render_parallel(sco, fo, smp)
        assert_film_equal(out["default"][0], fo.pixels, out["default"][3], "path records vs oracle")
        assert out["default"][1:3] == (sto["rays_closest"], sto["rays_any"])


def test_environment_cell_records_change_nothing(gpu, monkeypatch):
    """the cell records of a square environment map (built at scene creation; FTN_ENV_CELLS=0: not built; FTN_ENV_CELLS_USE=0: built but the
    environment-only kernels read the plain tables): same film, same counts, on both pipelines"""
    films = {}
    for build, use in (("1", "1"), ("1", "0"), ("0", "1")):
        monkeypatch.setenv("FTN_ENV_CELLS", build); monkeypatch.setenv("FTN_ENV_CELLS_USE", use)
        sc, cam, res = _env_only_scene(gpu)
        if build == "1": big = sc.info()["lights_bytes"]
        else: assert sc.info()["lights_bytes"] <= big - 16 * 16 * 128
        for pl in (WAVE, MEGA):
            f = Film(gpu, res)
            st = SamplerIntegrator(cam, PathIntegrator.new(6, 1.0)).render_parallel(sc, f, RandomSampler(4, 3, indexed=True), pipeline=pl)
            films[(build, use, pl)] = (f.pixels, st["rays_closest"], st["rays_any"])
    ref = films[("1", "1", WAVE)]
    for k, v in films.items():
        assert v[1] == ref[1] and v[2] == ref[2], k
        diff = (bits(v[0]) != bits(ref[0])).any(axis=-1).sum()
        assert diff == 0 or k[2] == MEGA, k                       # (the two pipelines may differ in the last bit of spill pixels only)
        assert np.allclose(v[0], ref[0], rtol=2e-6, atol=1e-7), k


# ------------------------------------------------------------------ DirectLightingIntegrator / WhittedIntegrator through the wavefront queues
def _mirror_hall(be):
    """two facing mirrors, a mirror sphere and matte / plastic / metal objects under point + area lights: specular chains up to the depth limit"""
    b = SceneBuilder(be)
    b.light_source("point", I=(25, 25, 25), from_=(0.3, -0.2, 2.6))
    b.material("matte", Kd=(0.6, 0.6, 0.6))
    scenes._quad(b, (-3, -3, 0), (3, -3, 0), (3, 3, 0), (-3, 3, 0))
    b.material("mirror", Kr=(0.9, 0.9, 0.9))
    scenes._quad(b, (-3, 3, 0), (3, 3, 0), (3, 3, 3), (-3, 3, 3))
    scenes._quad(b, (3, -3, 0), (-3, -3, 0), (-3, -3, 3), (3, -3, 3))
    b.attribute_begin(); b.material("mirror"); b.translate((1.2, 0.5, 0.6)); b.shape("sphere", radius=0.6); b.attribute_end()
    b.attribute_begin(); b.material("plastic", Kd=(0.3, 0.1, 0.1), Ks=(0.4, 0.4, 0.4), roughness=0.05); b.translate((-1.0, 0.2, 0.5)); b.shape("sphere", radius=0.5); b.attribute_end()
    b.attribute_begin(); b.material("metal", eta=(0.2, 0.92, 1.1), k=(3.9, 2.45, 2.14), roughness=0.2); b.translate((0.0, 1.6, 0.4)); b.shape("sphere", radius=0.4); b.attribute_end()
    b.attribute_begin(); b.material("matte", Kd=(0, 0, 0)); b.area_light_source("diffuse", L=(6, 6, 6)); b.translate((-1.5, -1.0, 2.5)); b.shape("sphere", radius=0.3); b.attribute_end()
    cam = PerspectiveCamera.look_at(be, (0.5, -2.6, 1.6), (0.2, 1.0, 0.7), (0, 0, 1), (88, 64), fov=60.0)
    return b, cam, (88, 64)


@pytest.mark.parametrize("kernels", KERNELS)
@pytest.mark.parametrize("scene", ["cornell", "materials", "mirror_hall", "cubes27", "furnace"])
@pytest.mark.parametrize("which", ["direct", "whitted"])
def test_direct_lighting_and_whitted_on_the_wavefront_pipeline(gpu, orc_det, scene, which, kernels):
    """the other two estimators as wavefront stages (k_wf_shade_dl): the nested product f * Li(child) * |cos| / pdf kept per depth and folded
    innermost first, so films are bit-equal to the oracle's recursion and to the megakernel; ray counts equal"""
    make = {"cornell": lambda be: scenes.cornell(be, res=64), "materials": _materials_scene, "mirror_hall": _mirror_hall,
            "cubes27": lambda be: scenes.instanced_cubes(be, n_copies=27, res=(96, 96), env_n=32), "furnace": scenes.furnace}[scene]
    integ = DirectLightingIntegrator(5) if which == "direct" else WhittedIntegrator(5)
    smp = RandomSampler(4, 0, indexed=True)
    (rgb, px, st), (rgbo, pxo, sto) = render_pair(gpu, orc_det, make, integ, smp, WAVE, kernels)
    assert_film_equal(px, pxo, st["spill_samples"], "%s %s, wavefront" % (scene, which))
    assert st["rays_closest"] == sto["rays_closest"] and st["rays_any"] == sto["rays_any"] and st["camera_samples"] == sto["camera_samples"]
    assert np.isfinite(rgb).all() and rgb.max() > 0
    mega = scenes.render(gpu, *make(gpu), integ, smp, backend_kwargs=dict(pipeline=MEGA))
    assert_film_equal(px, mega[1], st["spill_samples"], "%s %s, wavefront vs megakernel" % (scene, which))
    auto = scenes.render(gpu, *make(gpu), integ, smp)                          # AUTO picks the wavefront for the indexed sampler
    assert np.array_equal(bits(auto[1]), bits(scenes.render(gpu, *make(gpu), integ, smp, backend_kwargs=dict(pipeline=WAVE))[1]))


def test_wavefront_direct_lighting_limits(gpu, orc_det):
    """Whitted with many lights on the queues (one shadow-ray slot and one mask bit per light: up to 32), bit-equal to the oracle and the
    megakernel; beyond 32 lights an explicit request is refused and AUTO leaves the scene to the megakernel; the reference's tile-serial
    sampler is not taken by these stages."""
    def many_lights(be, n):
        b, cam, res = scenes.cornell(be, res=32)
        for k in range(n):
            b.light_source("point", I=(1 + 0.1 * k, 1, 1), from_=(0.07 * k - 0.5, 0.03 * k, 0.5))
        return b, cam, res
    smp = RandomSampler(2, 0, indexed=True)
    n_area = len(many_lights(gpu, 0)[0].create_scene().lights()[0])            # the Cornell box's emitter: one light per triangle
    for n in (4, 11, 32 - n_area):                               # the last one: exactly 32 lights
        (rgb, px, st), (_, pxo, sto) = render_pair(gpu, orc_det, lambda be: many_lights(be, n), WhittedIntegrator(3), smp, WAVE, "production")
        assert_film_equal(px, pxo, st["spill_samples"], "Whitted, %d lights" % (n + n_area))
        assert st["rays_any"] == sto["rays_any"] and st["rays_closest"] == sto["rays_closest"]
        m = scenes.render(gpu, *many_lights(gpu, n), WhittedIntegrator(3), smp, backend_kwargs=dict(pipeline=MEGA))
        assert np.array_equal(bits(px), bits(m[1]))
    with pytest.raises(FountainError) as e:
        scenes.render(gpu, *many_lights(gpu, 33 - n_area), WhittedIntegrator(3), smp, backend_kwargs=dict(pipeline=WAVE))
    assert e.value.code == A.FTN_ERR_UNSUPPORTED
    a = scenes.render(gpu, *many_lights(gpu, 33 - n_area), WhittedIntegrator(3), smp)                                       # AUTO -> megakernel
    m = scenes.render(gpu, *many_lights(gpu, 33 - n_area), WhittedIntegrator(3), smp, backend_kwargs=dict(pipeline=MEGA))
    assert np.array_equal(bits(a[1]), bits(m[1]))
    scenes.render(gpu, *many_lights(gpu, 40), DirectLightingIntegrator(3), smp, backend_kwargs=dict(pipeline=WAVE))   # one light per hit: any number of lights
    with pytest.raises(FountainError):
        scenes.render(gpu, *scenes.cornell(gpu, res=32), DirectLightingIntegrator(3), RandomSampler(2, 0), backend_kwargs=dict(pipeline=WAVE))
