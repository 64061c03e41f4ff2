"""Host-side mirror of fountain's render() surface over the C ABI (include/fountain_hip.h).

Names follow the reference so that tests read like the reference's own:
  Transform            src/geometry/transform.rs
  SceneBuilder         src/loaders/pbrt.rs:86-330 (PbrtSceneBuilder) + src/loaders/constructors.rs (defaults)
  PerspectiveCamera    src/camera/mod.rs:72-115, make_camera defaults src/loaders/pbrt.rs:426-466
  Film                 src/film.rs:18-223, make_film src/loaders/pbrt.rs:487-505
  RandomSampler        src/sampler/random.rs
  PathIntegrator       src/integrator/path.rs:10-20
  DirectLightingIntegrator  src/integrator/direct_lighting.rs:20-25
  SamplerIntegrator    src/integrator/mod.rs:22-25, render / render_parallel :206-227

No arithmetic happens in this module: every number is produced by the shared library behind `Backend`
(the HIP product by default).  The class is also instantiated by tests/ over the CPU oracle's orc_*
twins; this package itself never loads anything from oracle/.
"""
import ctypes as C
import os

import numpy as np

from . import _abi as A


class FountainError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("fountain error %d: %s" % (code, msg))
        self.code = code


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


class Backend:
    """A loaded shared library exposing the ftn_* (product) or orc_* (oracle) entry points."""

    def __init__(self, path, prefix="ftn_", is_oracle=False):
        if not os.path.exists(path):
            raise FountainError(A.FTN_ERR_INTERNAL, "shared library %s is missing: run `python -c 'import __graft_entry__ as g; g.build()'`" % path)
        self.lib = C.CDLL(path)
        self.path = path
        self.prefix = prefix
        self.is_oracle = is_oracle
        le = getattr(self.lib, prefix + "last_error")
        le.restype = C.c_char_p
        if not is_oracle:                     # the structs below are read and written by the library: a build against another header is refused here
            have = self.lib.ftn_abi_version() if hasattr(self.lib, "ftn_abi_version") else 0
            if have != A.FTN_ABI_VERSION:
                raise FountainError(A.FTN_ERR_INTERNAL, "%s reports ABI version %d, this binding was written for %d: rebuild the library" % (path, have, A.FTN_ABI_VERSION))
        for name in ("transform_scale", "transform_rotate", "transform_perspective", "sphere_init", "camera_perspective"):
            fn = getattr(self.lib, prefix + name)
            if name == "transform_scale":
                fn.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
            elif name == "transform_rotate":
                fn.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
            elif name == "transform_perspective":
                fn.argtypes = [C.c_float, C.c_float, C.c_float, C.c_void_p]
            elif name == "sphere_init":
                fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p]
            elif name == "camera_perspective":
                fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_void_p]

    def fn(self, name):
        return getattr(self.lib, self.prefix + name)

    def check(self, rc):
        if rc != 0:
            msg = self.fn("last_error")()
            raise FountainError(rc, msg.decode() if msg else "")

    def call(self, name, *args):
        self.check(self.fn(name)(*args))


_default_backend = None


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so under torch/lib with the same SONAME as /opt/rocm's.  Two HIP
    runtimes in one process each try to own the device ("No HIP GPUs are available" from whichever initialises second), so
    when torch is installed its copy is loaded first and libfountain_hip.so binds to it; device pointers and streams of torch
    tensors are then valid in ftn_render_device.  FTN_HIP_RUNTIME=system skips this."""
    if os.environ.get("FTN_HIP_RUNTIME", "torch") != "torch":
        return
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.origin:
        return
    lib = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(lib):
        try:
            C.CDLL(lib, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def default_backend():
    """The HIP product library; raises loudly if it has not been built."""
    global _default_backend
    if _default_backend is None:
        here = os.path.dirname(os.path.abspath(__file__))
        _share_hip_runtime_with_torch()
        _default_backend = Backend(os.path.join(here, os.environ.get("FTN_LIB", "libfountain_hip.so")), "ftn_", False)
    return _default_backend


# --------------------------------------------------------------------------- Transform
class Transform:
    def __init__(self, backend, raw=None):
        self.be = backend
        self.raw = raw if raw is not None else A.ftn_transform()

    @classmethod
    def identity(cls, be):
        t = cls(be)
        be.call("transform_identity", C.byref(t.raw))
        return t

    @classmethod
    def translate(cls, be, delta):
        t = cls(be)
        be.call("transform_translate", _f3(delta), C.byref(t.raw))
        return t

    @classmethod
    def scale(cls, be, sx, sy, sz):
        t = cls(be)
        be.call("transform_scale", sx, sy, sz, C.byref(t.raw))
        return t

    @classmethod
    def rotate(cls, be, angle_deg, axis):
        t = cls(be)
        be.call("transform_rotate", angle_deg, _f3(axis), C.byref(t.raw))
        return t

    @classmethod
    def look_at(cls, be, pos, look, up):
        t = cls(be)
        be.call("transform_look_at", _f3(pos), _f3(look), _f3(up), C.byref(t.raw))
        return t

    @classmethod
    def camera_look_at(cls, be, pos, look, up):  # transform.rs:58-60
        return cls.look_at(be, pos, look, up).inverse()

    @classmethod
    def from_flat(cls, be, m16):
        t = cls(be)
        be.call("transform_from_flat", (C.c_float * 16)(*[float(x) for x in m16]), C.byref(t.raw))
        return t

    @classmethod
    def perspective(cls, be, fov, near, far):
        t = cls(be)
        be.call("transform_perspective", fov, near, far, C.byref(t.raw))
        return t

    def __mul__(self, other):
        t = Transform(self.be)
        self.be.call("transform_mul", C.byref(self.raw), C.byref(other.raw), C.byref(t.raw))
        return t

    def inverse(self):
        t = Transform(self.be)
        self.be.call("transform_inverse", C.byref(self.raw), C.byref(t.raw))
        return t

    def then(self, nxt):  # transform.rs:137-139
        return nxt * self

    def swaps_handedness(self):
        return bool(self.be.fn("transform_swaps_handedness")(C.byref(self.raw)))

    def _apply(self, name, v):
        out = (C.c_float * 3)()
        self.be.call(name, C.byref(self.raw), _f3(v), out)
        return np.array(out[:], dtype=np.float32)

    def point(self, p):
        return self._apply("transform_point", p)

    def vector(self, v):
        return self._apply("transform_vector", v)

    def normal(self, n):
        return self._apply("transform_normal", n)

    def points(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 3)
        out = np.empty_like(arr)
        self.be.call("transform_points", C.byref(self.raw), C.c_size_t(arr.shape[0]), _fptr(arr), _fptr(out))
        return out

    def normals(self, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float32).reshape(-1, 3)
        out = np.empty_like(arr)
        self.be.call("transform_normals", C.byref(self.raw), C.c_size_t(arr.shape[0]), _fptr(arr), _fptr(out))
        return out

    def matrix(self):
        return np.array(self.raw.m[:], dtype=np.float32).reshape(4, 4).T  # row-major view [row][col]

    def copy(self):
        raw = A.ftn_transform()
        C.memmove(C.byref(raw), C.byref(self.raw), C.sizeof(raw))
        return Transform(self.be, raw)


# --------------------------------------------------------------------------- scene building
def load_ply_ascii(path):
    """Minimal ASCII PLY reader for `x y z [nx ny nz]` vertices and triangular `vertex_indices` faces
    (the layout of the reference's data/rounded_cube.ply; src/loaders/constructors.rs:94-190)."""
    with open(path, "r") as f:
        lines = f.read().split("\n")
    assert lines[0].strip() == "ply"
    nv = nf = 0
    props = []
    i = 1
    cur = None
    while lines[i].strip() != "end_header":
        tok = lines[i].split()
        if tok[0] == "element":
            cur = tok[1]
            if cur == "vertex":
                nv = int(tok[2])
            elif cur == "face":
                nf = int(tok[2])
        elif tok[0] == "property" and cur == "vertex":
            props.append(tok[-1])
        i += 1
    i += 1
    verts = np.array([[float(x) for x in lines[i + k].split()] for k in range(nv)], dtype=np.float32)
    i += nv
    faces = []
    for k in range(nf):
        tok = lines[i + k].split()
        if int(tok[0]) != 3:
            raise ValueError("Face with unsupported vertex count %s found" % tok[0])
        faces.append([int(tok[1]), int(tok[2]), int(tok[3])])
    col = {p: j for j, p in enumerate(props)}
    P = verts[:, [col["x"], col["y"], col["z"]]]
    N = verts[:, [col["nx"], col["ny"], col["nz"]]] if "nx" in col else None
    return np.ascontiguousarray(P), (np.ascontiguousarray(N) if N is not None else None), np.array(faces, dtype=np.uint32)


class SceneBuilder:
    """PbrtSceneBuilder (src/loaders/pbrt.rs:86-330): a graphics state (material, area light, orientation), a
    transform stack, and shape / light statements; create_scene() -> BVH::build + Scene::new."""

    def __init__(self, backend=None):
        self.be = backend or default_backend()
        self.materials = []
        self.area_emit = []
        self.prims = []          # (kind, index, material, area_emit)
        self.tri_indices = []
        self.tri_mesh = []
        self.P = []
        self.N = []
        self.UV = []
        self.S = []              # per-vertex shading tangents (zeros for meshes without)
        self.any_normals = False
        self.any_uvs = False
        self.any_tangents = False
        self.n_vertices = 0
        self.n_triangles = 0
        self.meshes = []
        self.spheres = []
        self.lights = []
        self.envmaps = []
        self.textures = []           # flat ftn_texture array
        self.images = []             # (texels [h,w,3] float32 after scale/flip, wrap)
        self.material_textures = []  # one [a, b, s0, s1, s2] per material (-1 = constant)
        self._spectrum_textures = {}
        self._float_textures = {}
        # GraphicsState: default material = make_matte(defaults) (pbrt.rs:88-96, constructors.rs:192-196)
        self._state = [dict(material=self._add_material(A.FTN_MAT_MATTE, a=(0.5, 0.5, 0.5), s0=0.0), area=-1, rev=False)]
        self._tf = [Transform.identity(self.be)]

    # -- graphics state
    def attribute_begin(self):
        self._state.append(dict(self._state[-1]))
        self._tf.append(self._tf[-1].copy())

    def attribute_end(self):
        self._state.pop()
        self._tf.pop()

    def identity(self):
        self._tf[-1] = Transform.identity(self.be)

    def translate(self, v):
        self._tf[-1] = self._tf[-1] * Transform.translate(self.be, v)

    def scale(self, sx, sy, sz):
        self._tf[-1] = self._tf[-1] * Transform.scale(self.be, sx, sy, sz)

    def rotate(self, angle_deg, axis):
        self._tf[-1] = self._tf[-1] * Transform.rotate(self.be, angle_deg, axis)

    def concat_transform(self, t):
        self._tf[-1] = self._tf[-1] * t

    def reverse_orientation(self):
        self._state[-1]["rev"] = True      # sets, does not toggle (pbrt.rs:203-205)

    # -- textures (pbrt.rs:362-385, constructors.rs:247-318)
    def _lookup_texture(self, name):
        """lookup_texture (pbrt.rs:142-147): spectrum textures first, then float textures -> (index, is_float)."""
        if name in self._spectrum_textures:
            return self._spectrum_textures[name], False
        if name in self._float_textures:
            return self._float_textures[name], True
        raise ValueError("TextureError: " + name)

    def _new_texture(self, kind, is_float, value=(0, 0, 0), tex1=-1, tex2=-1, image=-1, mapping=(1.0, 1.0, 0.0, 0.0)):
        t = A.ftn_texture()
        t.kind, t.is_float = kind, 1 if is_float else 0
        t.value = (C.c_float * 3)(*value)
        t.tex1, t.tex2, t.image = tex1, tex2, image
        t.su, t.sv, t.du, t.dv = mapping
        self.textures.append(t)
        return len(self.textures) - 1

    def _texture_or_const(self, v, is_float):
        """get_texture_or_const (loaders/mod.rs:213-225): a named texture of the right output type, or a constant."""
        if isinstance(v, str):
            idx, f = self._lookup_texture(v)
            if f != is_float:
                raise ValueError("ParamError: texture %s has the wrong output type" % v)
            return idx
        val = (float(v),) * 3 if is_float else tuple(float(x) for x in v)
        return self._new_texture(A.FTN_TEX_CONSTANT, is_float, value=val)

    def texture(self, name, ty, cls, **kw):
        """`Texture "name" "ty" "cls" ...`: (spectrum|color|float, checkerboard), (spectrum|color, uv | imagemap).
        imagemap takes `filename` (an OpenEXR file) or `texels` ([h, w, 3], first row = top of the image), `wrap`, `scale`."""
        ty = "spectrum" if ty == "color" else ty
        if (ty, cls) not in (("spectrum", "checkerboard"), ("spectrum", "uv"), ("float", "checkerboard"), ("spectrum", "imagemap")):
            raise ValueError("UnknownName(%s %s)" % (ty, cls))
        is_float = ty == "float"
        if kw.get("mapping", "uv") != "uv":
            raise ValueError("Unknown mapping type " + kw["mapping"])
        mapping = (kw.get("uscale", 1.0), kw.get("vscale", 1.0), kw.get("udelta", 0.0), kw.get("vdelta", 0.0))
        if cls == "checkerboard":
            if "tex1" not in kw or "tex2" not in kw:
                raise ValueError("ParamError: checkerboard needs tex1 and tex2 (constructors.rs:265-266)")
            t1, t2 = self._texture_or_const(kw["tex1"], is_float), self._texture_or_const(kw["tex2"], is_float)
            idx = self._new_texture(A.FTN_TEX_CHECKERBOARD, is_float, tex1=t1, tex2=t2, mapping=mapping)
        elif cls == "uv":
            idx = self._new_texture(A.FTN_TEX_UV, False, mapping=mapping)
        else:   # make_imagemap_spect (constructors.rs:295-318) -> load_mipmap (imageio/mod.rs:81-124): scale, flip_y = true
            tex = kw.get("texels")
            if tex is None:
                tex = read_exr(kw["filename"], self.be)
            wrap = {"repeat": A.FTN_WRAP_REPEAT, "black": A.FTN_WRAP_BLACK, "clamp": A.FTN_WRAP_CLAMP}[kw.get("wrap", "repeat")]
            tex = np.array(tex, np.float32, order="C")                 # (a copy: the gamma step works in place)
            if kw.get("gamma", False):                                  # imageio/mod.rs:86-107: given -> as given; default false for .exr input
                self.be.call("image_inverse_gamma", tex.ctypes.data_as(C.c_void_p), C.c_size_t(tex.size))
            tex = tex * np.float32(kw.get("scale", 1.0))
            tex = np.ascontiguousarray(tex[::-1])
            self.images.append((tex, wrap))
            idx = self._new_texture(A.FTN_TEX_IMAGE, False, image=len(self.images) - 1, mapping=mapping)
        (self._float_textures if is_float else self._spectrum_textures)[name] = idx
        return idx

    def _slot(self, v, default, is_float):
        """get_texture_or_default (loaders/mod.rs:227-236): value -> (constant, texture index).  A texture of the other output
        type fails the conversion and the default constant is used, as in the reference."""
        if isinstance(v, str):
            idx, f = self._lookup_texture(v)
            return (default, idx) if f == is_float else (default, -1)
        return (v, -1)

    def _add_material(self, type_, a=(0, 0, 0), b=(0, 0, 0), s0=0.0, s1=0.0, s2=0.0, remap=True, defaults=None):
        d = defaults or {}
        (a, ta), (b, tb) = self._slot(a, d.get("a", (0, 0, 0)), False), self._slot(b, d.get("b", (0, 0, 0)), False)
        (s0, t0), (s1, t1), (s2, t2) = self._slot(s0, d.get("s0", 0.0), True), self._slot(s1, d.get("s1", 0.0), True), self._slot(s2, d.get("s2", 0.0), True)
        m = A.ftn_material()
        m.type = type_
        m.remap_roughness = 1 if remap else 0
        m.a = (C.c_float * 3)(*a)
        m.b = (C.c_float * 3)(*b)
        m.s0, m.s1, m.s2 = s0, s1, s2
        self.materials.append(m)
        self.material_textures.append([ta, tb, t0, t1, t2])
        return len(self.materials) - 1

    def material(self, name, **kw):
        """Defaults from src/loaders/constructors.rs:192-236."""
        if name == "matte":
            idx = self._add_material(A.FTN_MAT_MATTE, a=kw.get("Kd", (0.5, 0.5, 0.5)), s0=kw.get("sigma", 0.0),
                                     defaults=dict(a=(0.5, 0.5, 0.5), s0=0.0))
        elif name == "mirror":
            idx = self._add_material(A.FTN_MAT_MIRROR, a=kw.get("Kr", (0.9, 0.9, 0.9)), defaults=dict(a=(0.9, 0.9, 0.9)))
        elif name == "metal":
            rough = kw.get("roughness", 0.01)
            u = kw.get("uroughness", None)
            v = kw.get("vroughness", None)
            if u is None or v is None:
                u = v = rough
            idx = self._add_material(A.FTN_MAT_METAL, a=kw["eta"], b=kw["k"], s1=u, s2=v, remap=kw.get("remaproughness", True),
                                     defaults=dict(s1=0.01, s2=0.01))
        elif name == "plastic":
            idx = self._add_material(A.FTN_MAT_PLASTIC, a=kw.get("Kd", (0.25, 0.25, 0.25)), b=kw.get("Ks", (0.25, 0.25, 0.25)),
                                     s1=kw.get("roughness", 0.1), remap=kw.get("remaproughness", True),
                                     defaults=dict(a=(0.25, 0.25, 0.25), b=(0.25, 0.25, 0.25), s1=0.1))
        elif name == "glass":
            idx = self._add_material(A.FTN_MAT_GLASS, a=kw.get("Kr", (1, 1, 1)), b=kw.get("Kt", (1, 1, 1)), s0=kw.get("eta", 1.5),
                                     s1=kw.get("uroughness", 0.0), s2=kw.get("vroughness", 0.0), remap=kw.get("remaproughness", True),
                                     defaults=dict(a=(1, 1, 1), b=(1, 1, 1), s0=1.5, s1=0.0, s2=0.0))
        elif name == "none":
            idx = -1
        else:
            raise ValueError("unknown material " + name)
        self._state[-1]["material"] = idx
        return idx

    def area_light_source(self, name="diffuse", L=(1.0, 1.0, 1.0)):
        assert name == "diffuse"
        self.area_emit.append(tuple(float(x) for x in L))
        self._state[-1]["area"] = len(self.area_emit) - 1

    # -- shapes
    def shape(self, name, **kw):
        st = self._state[-1]
        tf = self._tf[-1]
        if name == "sphere":
            radius = kw.get("radius", 1.0)
            s = A.ftn_sphere()
            w2o = tf.inverse()
            self.be.call("sphere_init", C.byref(tf.raw), C.byref(w2o.raw), 1 if st["rev"] else 0, radius,
                         kw.get("zmin", -radius), kw.get("zmax", radius), kw.get("phimax", 360.0), C.byref(s))
            self.spheres.append(s)
            self.prims.append((A.FTN_SHAPE_SPHERE, len(self.spheres) - 1, st["material"], st["area"]))
        elif name in ("trianglemesh", "plymesh"):
            if name == "plymesh":
                P, N, idx = load_ply_ascii(kw["filename"])
                UV = None
            else:
                P = np.asarray(kw["P"], dtype=np.float32).reshape(-1, 3)
                N = np.asarray(kw["N"], dtype=np.float32).reshape(-1, 3) if kw.get("N") is not None else None
                UV = np.asarray(kw["uv"], dtype=np.float32).reshape(-1, 2) if kw.get("uv") is not None else None
                idx = np.asarray(kw["indices"], dtype=np.uint32).reshape(-1, 3)
            S = np.asarray(kw["S"], dtype=np.float32).reshape(-1, 3) if kw.get("S") is not None else None      # constructors.rs:63
            self.add_mesh(P, N, UV, idx, tf, st["rev"], st["material"], st["area"], S=S)
        else:
            raise ValueError("unknown shape " + name)

    def add_mesh(self, P, N, UV, idx, tf, rev, material, area, S=None):
        """TriangleMesh::new (src/shapes/triangle.rs:29-74): vertices, normals and tangents go to world space."""
        Pw = tf.points(P)
        Nw = tf.normals(N) if N is not None else None
        Sw = np.stack([tf.vector(t) for t in S]).astype(np.float32) if S is not None else None
        m = A.ftn_mesh()
        m.has_tangents = 1 if S is not None else 0
        m.has_normals = 1 if N is not None else 0
        m.has_uvs = 1 if UV is not None else 0
        m.reverse_orientation = 1 if rev else 0
        m.flip_normals = 1 if (rev ^ tf.swaps_handedness()) else 0
        mesh_id = len(self.meshes)
        self.meshes.append(m)
        base = self.n_vertices
        nv = Pw.shape[0]
        self.P.append(Pw)
        self.N.append(Nw if Nw is not None else np.zeros((nv, 3), np.float32))
        self.UV.append(UV if UV is not None else np.zeros((nv, 2), np.float32))
        self.S.append(Sw if Sw is not None else np.zeros((nv, 3), np.float32))
        self.any_tangents |= S is not None
        self.any_normals |= N is not None
        self.any_uvs |= UV is not None
        self.n_vertices += nv
        idx = np.asarray(idx, dtype=np.uint32).reshape(-1, 3)
        nt = idx.shape[0]
        self.tri_indices.append(idx + np.uint32(base))
        self.tri_mesh.append(np.full(nt, mesh_id, np.uint32))
        first = self.n_triangles
        self.n_triangles += nt
        self.prims.append(("trirange", first, nt, material, area))

    # -- lights (src/loaders/constructors.rs:320-359)
    def light_source(self, name, **kw):
        l = A.ftn_light()
        l.envmap = -1
        tf = self._tf[-1]
        if name == "point":
            I = np.float32(kw.get("I", (1, 1, 1))) * np.float32(kw.get("scale", (1, 1, 1)))
            frm = kw.get("from_", (0, 0, 0))
            l2w = Transform.translate(self.be, frm)
            l.type = A.FTN_LIGHT_POINT
            l.rgb = (C.c_float * 3)(*I)
            l.v = (C.c_float * 3)(*l2w.point((0, 0, 0)))
            l.light_to_world = l2w.raw
        elif name == "distant":
            L = np.float32(kw.get("L", (1, 1, 1))) * np.float32(kw.get("scale", (1, 1, 1)))
            frm = np.float32(kw.get("from_", (0, 0, 0)))
            to = np.float32(kw.get("to", (0, 0, 1)))
            d = frm - to
            # DistantLight::new normalises (distant.rs:23-31); done by the library's vector helpers
            n = self._normalize(d)
            l.type = A.FTN_LIGHT_DISTANT
            l.rgb = (C.c_float * 3)(*L)
            l.v = (C.c_float * 3)(*n)
            l.light_to_world = Transform.identity(self.be).raw
        elif name == "infinite":
            l.type = A.FTN_LIGHT_INFINITE
            l.light_to_world = tf.raw
            tex = kw.get("texels", None)
            if tex is None:   # new_uniform (infinite.rs:43-61): 1x1 map
                tex = np.float32(kw.get("L", (1, 1, 1))).reshape(1, 1, 3)
            tex = np.ascontiguousarray(tex, dtype=np.float32)
            self.envmaps.append(tex)
            l.envmap = len(self.envmaps) - 1
        else:
            raise ValueError("unknown light " + name)
        self.lights.append(l)

    def _normalize(self, v):
        # cgmath normalize in f32: v * (1/|v|)
        v = np.float32(v)
        mag = np.sqrt(np.float32(np.float32(v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]), dtype=np.float32)
        return v * (np.float32(1.0) / mag)

    # -- finish
    def build_desc(self):
        """Flatten into an ftn_scene_desc; returns (desc, keepalive)."""
        keep = []
        chunks = []
        for p in self.prims:
            if p[0] == "trirange":
                _, first, nt, mat, area = p
                c = np.empty(nt, dtype=PRIM_DTYPE)
                c["k"] = A.FTN_SHAPE_TRIANGLE
                c["i"] = np.arange(first, first + nt, dtype=np.uint32)
                c["m"] = mat
                c["a"] = area
            else:
                c = np.array([p], dtype=PRIM_DTYPE)
            chunks.append(c)
        pa = np.ascontiguousarray(np.concatenate(chunks)) if chunks else np.zeros(0, dtype=PRIM_DTYPE)
        d = A.ftn_scene_desc()
        d.n_prims = pa.shape[0]
        d.prims = pa.ctypes.data_as(C.POINTER(A.ftn_prim))
        keep.append(pa)
        ti = np.ascontiguousarray(np.concatenate(self.tri_indices)) if self.tri_indices else np.zeros((0, 3), np.uint32)
        tm = np.ascontiguousarray(np.concatenate(self.tri_mesh)) if self.tri_mesh else np.zeros(0, np.uint32)
        P = np.ascontiguousarray(np.concatenate(self.P)) if self.P else np.zeros((0, 3), np.float32)
        N = np.ascontiguousarray(np.concatenate(self.N)) if self.N else np.zeros((0, 3), np.float32)
        UV = np.ascontiguousarray(np.concatenate(self.UV)) if self.UV else np.zeros((0, 2), np.float32)
        S = np.ascontiguousarray(np.concatenate(self.S)) if self.S else np.zeros((0, 3), np.float32)
        keep += [ti, tm, P, N, UV, S]
        d.n_triangles = ti.shape[0]
        d.tri_indices = ti.ctypes.data_as(C.POINTER(C.c_uint32))
        d.tri_mesh = tm.ctypes.data_as(C.POINTER(C.c_uint32))
        d.n_vertices = P.shape[0]
        d.P = _fptr(P)
        d.N = _fptr(N) if self.any_normals else None
        d.UV = _fptr(UV) if self.any_uvs else None
        d.S = _fptr(S) if self.any_tangents else None
        for name, items, typ in (("meshes", self.meshes, A.ftn_mesh), ("spheres", self.spheres, A.ftn_sphere),
                                 ("materials", self.materials, A.ftn_material), ("lights", self.lights, A.ftn_light)):
            arr = (typ * max(len(items), 1))(*items)
            keep.append(arr)
            setattr(d, "n_" + name, len(items))
            setattr(d, name, arr)
        ae = np.ascontiguousarray(np.array(self.area_emit, dtype=np.float32).reshape(-1, 3))
        keep.append(ae)
        d.n_area_emit = ae.shape[0]
        d.area_emit = _fptr(ae)
        envs = (A.ftn_envmap * max(len(self.envmaps), 1))()
        for i, t in enumerate(self.envmaps):
            envs[i].height, envs[i].width = t.shape[0], t.shape[1]
            envs[i].texels = _fptr(t)
        keep += [envs, self.envmaps]
        d.n_envmaps = len(self.envmaps)
        d.envmaps = envs
        if self.textures:
            tarr = (A.ftn_texture * len(self.textures))(*self.textures)
            marr = (A.ftn_material_textures * max(len(self.materials), 1))()
            for i, slots in enumerate(self.material_textures):
                marr[i].a, marr[i].b, marr[i].s0, marr[i].s1, marr[i].s2 = slots
            iarr = (A.ftn_image * max(len(self.images), 1))()
            for i, (tex, wrap) in enumerate(self.images):
                iarr[i].height, iarr[i].width, iarr[i].wrap = tex.shape[0], tex.shape[1], wrap
                iarr[i].texels = _fptr(tex)
            keep += [tarr, marr, iarr, self.images]
            d.n_textures, d.textures, d.material_textures = len(self.textures), tarr, marr
            d.n_images, d.images = len(self.images), iarr
        return d, keep

    def create_scene(self, device=0):
        d, keep = self.build_desc()
        return Scene(self.be, d, keep, device)


class PbrtScene:
    """A parsed .pbrt file (loaders/pbrt.rs PbrtHeader + PbrtSceneBuilder evaluated by the library's C++ reader):
    .camera / .film / .sampler mirror make_camera / make_film / make_sampler; create_scene() -> BVH::build + upload."""

    def __init__(self, path, backend=None):
        self.be = be = backend or default_backend()
        if be.is_oracle:
            raise ValueError("scene files are read by the product library only")
        self.handle = C.c_void_p()
        load = be.lib.ftn_pbrt_load
        load.argtypes = [C.c_char_p, C.POINTER(C.c_void_p)]
        rc = load(os.fsencode(path), C.byref(self.handle))
        if rc != 0:
            err = be.lib.ftn_pbrt_last_error
            err.restype = C.c_char_p
            raise FountainError(rc, err().decode())
        for name, typ in (("scene", A.ftn_scene_desc), ("camera", A.ftn_camera_desc), ("film", A.ftn_film_desc)):
            f = getattr(be.lib, "ftn_pbrt_" + name)
            f.restype = C.POINTER(typ)
            f.argtypes = [C.c_void_p]
        be.lib.ftn_pbrt_samples_per_pixel.argtypes = [C.c_void_p]
        be.lib.ftn_pbrt_film_name.argtypes = [C.c_void_p]
        be.lib.ftn_pbrt_film_name.restype = C.c_char_p
        self.desc = be.lib.ftn_pbrt_scene(self.handle).contents
        self.samples_per_pixel = be.lib.ftn_pbrt_samples_per_pixel(self.handle)
        self.film_name = be.lib.ftn_pbrt_film_name(self.handle).decode()
        self.camera = PerspectiveCamera.__new__(PerspectiveCamera)
        self.camera.be = be
        self.camera.desc = be.lib.ftn_pbrt_camera(self.handle).contents
        self._film_desc = be.lib.ftn_pbrt_film(self.handle).contents

    def film(self):
        """A fresh Film (make_film, pbrt.rs:487-505)."""
        return Film.from_desc(self.be, self._film_desc)

    def sampler(self, override_samples=None, **kw):
        """make_sampler (pbrt.rs:468-485): RandomSampler::new_with_seed(spp, 0)."""
        return RandomSampler.new_with_seed(override_samples or self.samples_per_pixel, 0, **kw)

    def create_scene(self, device=0):
        return Scene(self.be, self.desc, [self], device)

    def __del__(self):
        try:
            if self.handle:
                d = self.be.lib.ftn_pbrt_destroy
                d.argtypes = [C.c_void_p]
                d.restype = None
                d(self.handle)
                self.handle = None
        except Exception:
            pass


def load_ply(path, backend=None):
    """make_triangle_mesh_from_ply's reader (constructors.rs:94-190) -> (P [nv,3], N or None, UV or None, idx [nt,3])."""
    be = backend or default_backend()
    f = be.lib.ftn_ply_load
    f.argtypes = [C.c_char_p] + [C.c_void_p] * 8
    nv, nt, hn, huv = C.c_uint32(), C.c_uint32(), C.c_int(), C.c_int()

    def call(*arrs):
        rc = f(os.fsencode(path), C.cast(C.byref(nv), C.c_void_p), C.cast(C.byref(nt), C.c_void_p), *arrs,
               C.cast(C.byref(hn), C.c_void_p), C.cast(C.byref(huv), C.c_void_p))
        if rc != 0:
            err = be.lib.ftn_pbrt_last_error
            err.restype = C.c_char_p
            raise FountainError(rc, err().decode())
    call(None, None, None, None)
    P = np.empty((nv.value, 3), np.float32)
    N = np.empty((nv.value, 3), np.float32)
    UV = np.empty((nv.value, 2), np.float32)
    idx = np.empty((nt.value, 3), np.uint32)
    call(*[a.ctypes.data_as(C.c_void_p) for a in (P, N, UV, idx)])
    return P, (N if hn.value else None), (UV if huv.value else None), idx


class Scene:
    """Scene (src/scene/mod.rs:14-18) as an opaque device-resident handle."""

    def __init__(self, be, desc, keep, device=0):
        self.be = be
        self._keep = keep
        self.desc = desc
        self.handle = C.c_void_p()
        if be.is_oracle:
            be.call("scene_create", C.byref(desc), C.byref(self.handle))
        else:
            be.call("scene_create", C.byref(desc), C.c_int(device), C.byref(self.handle))

    def __del__(self):
        try:
            if self.handle:
                self.be.fn("scene_destroy")(self.handle)
                self.handle = None
        except Exception:
            pass

    def info(self):
        nn, npr, nl, md = C.c_uint32(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        wb = (C.c_float * 6)()
        self.be.call("scene_info", self.handle, C.byref(nn), C.byref(npr), C.byref(nl), C.byref(md), wb)
        out = dict(n_nodes=nn.value, n_prims=npr.value, n_lights=nl.value, max_depth=md.value,
                   world_bound=np.array(wb[:], dtype=np.float32))
        if not self.be.is_oracle:             # device bytes per array (ftn_scene_memory_info)
            m = A.ftn_scene_memory()
            self.be.call("scene_memory_info", self.handle, C.byref(m))
            out.update({k + "_bytes": int(getattr(m, k)) for k, _ in m._fields_})
        return out

    def nodes(self):
        i = self.info()
        nodes = np.zeros(i["n_nodes"], dtype=NODE_DTYPE)
        order = np.zeros(i["n_prims"], dtype=np.uint32)
        self.be.call("scene_get_nodes", self.handle, nodes.ctypes.data_as(C.c_void_p), order.ctypes.data_as(C.c_void_p))
        return nodes, order

    def lights(self):
        n = self.info()["n_lights"]
        kind = np.zeros(max(n, 1), np.int32)
        prim = np.zeros(max(n, 1), np.int32)
        self.be.call("scene_get_lights", self.handle, kind.ctypes.data_as(C.c_void_p), prim.ctypes.data_as(C.c_void_p))
        return kind[:n], prim[:n]

    # Scene::intersect / intersect_test for a batch of rays: rays [n,8] = o, d, t_max, time
    # stats=True asks for node / primitive tallies: the HIP library then runs the counting build of the REFERENCE-order walk (tallies
    # equal the oracle's); stats=False runs the production kernels (four-box records, ftn_trace4.hip) and returns None for the tallies
    def intersect(self, rays, n_threads=8, stats=True):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        t = np.empty(n, np.float32)
        prim = np.empty(n, np.int32)
        bary = np.zeros((n, 3), np.float32)
        st = A.ftn_stats()
        args = [self.handle, _fptr(rays), C.c_size_t(n), _fptr(t), prim.ctypes.data_as(C.c_void_p), _fptr(bary), C.byref(st) if stats else None]
        if self.be.is_oracle:
            args.append(C.c_int(n_threads))
        self.be.call("intersect", *args)
        return t, prim, bary, (st.as_dict() if stats else None)

    def intersect_test(self, rays, n_threads=8, stats=True):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        occ = np.empty(n, np.uint8)
        st = A.ftn_stats()
        args = [self.handle, _fptr(rays), C.c_size_t(n), occ.ctypes.data_as(C.c_void_p), C.byref(st) if stats else None]
        if self.be.is_oracle:
            args.append(C.c_int(n_threads))
        self.be.call("intersect_test", *args)
        return occ.astype(bool), (st.as_dict() if stats else None)

    def intersect_full(self, rays, n_threads=8):
        rays = np.ascontiguousarray(rays, dtype=np.float32).reshape(-1, 8)
        n = rays.shape[0]
        out = np.empty((n, 24), np.float32)
        args = [self.handle, _fptr(rays), C.c_size_t(n), _fptr(out)]
        if self.be.is_oracle:
            args.append(C.c_int(n_threads))
        self.be.call("intersect_full", *args)
        return out


PRIM_DTYPE = np.dtype([("k", "<u4"), ("i", "<u4"), ("m", "<i4"), ("a", "<i4")])
NODE_DTYPE = np.dtype([("bmin", "<f4", 3), ("bmax", "<f4", 3), ("idx", "<u4"), ("n_prims", "<u2"), ("axis", "u1"), ("is_leaf", "u1")])


def make_rays(o, d, t_max=np.inf, time=0.0):
    o = np.asarray(o, np.float32).reshape(-1, 3)
    d = np.asarray(d, np.float32).reshape(-1, 3)
    n = max(o.shape[0], d.shape[0])
    r = np.empty((n, 8), np.float32)
    r[:, 0:3] = o
    r[:, 3:6] = d
    r[:, 6] = t_max
    r[:, 7] = time
    return r


# --------------------------------------------------------------------------- camera / film / sampler / integrators
class PerspectiveCamera:
    def __init__(self, be, camera_to_world, full_resolution, screen_window=None, shutter=(0.0, 1.0),
                 lens_radius=0.0, focal_dist=1e6, fov=90.0):
        """PerspectiveCamera::new (camera/mod.rs:85-114); defaults of make_camera (loaders/pbrt.rs:426-466)."""
        self.be = be
        xres, yres = full_resolution
        if screen_window is None:
            aspect = np.float32(xres) / np.float32(yres)
            if aspect > 1.0:
                screen_window = (-aspect, -1.0, aspect, 1.0)
            else:
                screen_window = (-1.0, np.float32(-1.0) / aspect, 1.0, np.float32(1.0) / aspect)
        self.desc = A.ftn_camera_desc()
        be.call("camera_perspective", C.byref(camera_to_world.raw), (C.c_int32 * 2)(xres, yres),
                (C.c_float * 4)(*[float(x) for x in screen_window]), (C.c_float * 2)(*shutter),
                lens_radius, focal_dist, fov, C.byref(self.desc))

    @classmethod
    def look_at(cls, be, eye, look, up, full_resolution, **kw):
        """`LookAt` + `Camera "perspective"`: camera_tf = look_at(..), cam2world = inverse (pbrt.rs:431-433, :586-590)."""
        # the header CTM is identity * look_at (eval_transform_stmt multiplies even by the identity, which turns -0 into +0)
        return cls(be, (Transform.identity(be) * Transform.look_at(be, eye, look, up)).inverse(), full_resolution, **kw)


class Film:
    def __init__(self, be, resolution, crop_window=(0.0, 0.0, 1.0, 1.0)):
        """Film::new with BoxFilter::default(), crop_window = (min.x, min.y, max.x, max.y) (film.rs:43-83)."""
        self.be = be
        self.desc = A.ftn_film_desc()
        be.call("film_init", (C.c_int32 * 2)(*resolution), (C.c_float * 4)(*crop_window), C.byref(self.desc))
        c = self.desc.crop
        self.width, self.height = c[2] - c[0], c[3] - c[1]
        self.pixels = np.zeros((self.height, self.width, 4), np.float32)   # Pixel{xyz, filter_weight_sum}

    @classmethod
    def from_desc(cls, be, desc):
        f = cls.__new__(cls)
        f.be = be
        f.desc = A.ftn_film_desc()
        C.memmove(C.byref(f.desc), C.byref(desc), C.sizeof(desc))
        c = f.desc.crop
        f.width, f.height = c[2] - c[0], c[3] - c[1]
        f.pixels = np.zeros((f.height, f.width, 4), np.float32)
        return f

    def sample_bounds(self):
        out = (C.c_int32 * 4)()
        self.be.call("film_sample_bounds", C.byref(self.desc), out)
        return tuple(out[:])

    def tile_count(self):
        n = C.c_uint32()
        self.be.call("film_tile_count", C.byref(self.desc), C.byref(n))
        return n.value

    def into_spectrum_buffer(self):
        """film.rs:195-210 -> (rgb [h,w,3], (w,h))."""
        rgb = np.empty((self.height, self.width, 3), np.float32)
        self.be.call("film_resolve", self.pixels.ctypes.data_as(C.c_void_p), C.c_size_t(self.width * self.height), _fptr(rgb))
        return rgb, (self.width, self.height)


def _io_error(be, rc):
    err = be.lib.ftn_imageio_last_error
    err.restype = C.c_char_p
    return FountainError(rc, err().decode())


def write_exr(path, rgb, backend=None):
    """write_exr (imageio/exr.rs:47-87): rgb [h, w, 3] float32 -> scanline OpenEXR with FLOAT R, G, B channels."""
    be = backend or default_backend()
    rgb = np.ascontiguousarray(rgb, dtype=np.float32)
    h, w = rgb.shape[0], rgb.shape[1]
    f = be.lib.ftn_exr_write
    f.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
    rc = f(os.fsencode(path), rgb.ctypes.data_as(C.c_void_p), w, h)
    if rc != 0:
        raise _io_error(be, rc)


def read_exr(path, backend=None):
    """read_exr (imageio/exr.rs:11-45) -> rgb [h, w, 3] float32."""
    be = backend or default_backend()
    f = be.lib.ftn_exr_read
    f.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p]
    w, h = C.c_uint32(), C.c_uint32()
    rc = f(os.fsencode(path), C.cast(C.byref(w), C.c_void_p), C.cast(C.byref(h), C.c_void_p), None)
    if rc != 0:
        raise _io_error(be, rc)
    rgb = np.empty((h.value, w.value, 3), np.float32)
    rc = f(os.fsencode(path), C.cast(C.byref(w), C.c_void_p), C.cast(C.byref(h), C.c_void_p), rgb.ctypes.data_as(C.c_void_p))
    if rc != 0:
        raise _io_error(be, rc)
    return rgb


def film_resolve_device(be, device_pixels_ptr, n_pixels, device_rgb_ptr, stream_ptr=0):
    """Film::into_spectrum_buffer (film.rs:195-210) with both buffers in HBM (device pointers)."""
    f = be.lib.ftn_film_resolve_device
    f.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
    be.check(f(C.c_void_p(device_pixels_ptr), n_pixels, C.c_void_p(device_rgb_ptr), C.c_void_p(stream_ptr)))


class RandomSampler:
    """RandomSampler::new_with_seed(spp, seed) (sampler/random.rs:11-18). `indexed=True` selects the re-seeded
    per-(pixel, sample) variant behind the same Sampler surface (FTN_SAMPLER_INDEXED)."""

    def __init__(self, samples_per_pixel, seed=0, indexed=False, first_sample=0, sample_count=0):
        self.desc = A.ftn_sampler_desc()
        self.desc.kind = A.FTN_SAMPLER_INDEXED if indexed else A.FTN_SAMPLER_TILE_SERIAL
        self.desc.samples_per_pixel = samples_per_pixel
        self.desc.seed = seed
        self.desc.first_sample = first_sample
        self.desc.sample_count = sample_count

    @classmethod
    def new_with_seed(cls, spp, seed, **kw):
        return cls(spp, seed, **kw)


class PathIntegrator:
    def __init__(self, max_depth, rr_threshold):
        self.desc = A.ftn_integrator_desc()
        self.desc.kind = A.FTN_INTEGRATOR_PATH
        self.desc.max_depth = max_depth
        self.desc.rr_threshold = rr_threshold

    @classmethod
    def new(cls, max_depth, rr_threshold):
        return cls(max_depth, rr_threshold)


class DirectLightingIntegrator:
    def __init__(self, max_depth):
        self.desc = A.ftn_integrator_desc()
        self.desc.kind = A.FTN_INTEGRATOR_DIRECT_LIGHTING
        self.desc.max_depth = max_depth
        self.desc.rr_threshold = 0.0


class WhittedIntegrator:
    """WhittedIntegrator{max_depth} (integrator/whitted.rs:11-13)."""

    def __init__(self, max_depth):
        self.desc = A.ftn_integrator_desc()
        self.desc.kind = A.FTN_INTEGRATOR_WHITTED
        self.desc.max_depth = max_depth
        self.desc.rr_threshold = 0.0


class SamplerIntegrator:
    """SamplerIntegrator{camera, radiance} (integrator/mod.rs:22-25)."""

    def __init__(self, camera, radiance):
        self.camera = camera
        self.radiance = radiance
        self.last_stats = None

    def render_parallel(self, scene, film, sampler, tiles=None, pipeline=A.FTN_PIPELINE_AUTO, device=-1,
                        count_traffic=False, n_threads=0):
        """integrator/mod.rs:218-227: renders into film.pixels (added, as merge_film_tile does)."""
        be = scene.be
        tr = A.ftn_tile_range()
        if tiles is not None:
            tr.first, tr.stride, tr.count = tiles
        else:
            tr.first, tr.stride, tr.count = 0, 1, 0
        st = A.ftn_stats()
        if be.is_oracle:
            be.call("render", scene.handle, C.byref(self.camera.desc), C.byref(film.desc), C.byref(sampler.desc),
                    C.byref(self.radiance.desc), C.byref(tr), C.c_int(n_threads), C.c_int(1 if count_traffic else 0),
                    film.pixels.ctypes.data_as(C.c_void_p), C.byref(st))
        else:
            opt = A.ftn_render_options()
            opt.pipeline, opt.device, opt.count_traffic = pipeline, device, int(count_traffic)
            be.call("render", scene.handle, C.byref(self.camera.desc), C.byref(film.desc), C.byref(sampler.desc),
                    C.byref(self.radiance.desc), C.byref(tr), C.byref(opt),
                    film.pixels.ctypes.data_as(C.c_void_p), C.byref(st))
        self.last_stats = st.as_dict()
        return self.last_stats

    render = render_parallel   # render() differs only in tile scheduling (integrator/mod.rs:206-216)

    def render_device(self, scene, film, sampler, device_pixels_ptr, stream_ptr=0, tiles=None,
                      pipeline=A.FTN_PIPELINE_AUTO, device=-1, count_traffic=False):
        """Film resident in HBM: device_pixels_ptr is a device pointer to height*width ftn_pixel."""
        be = scene.be
        tr = A.ftn_tile_range()
        tr.first, tr.stride, tr.count = tiles if tiles is not None else (0, 1, 0)
        st = A.ftn_stats()
        opt = A.ftn_render_options()
        opt.pipeline, opt.device, opt.count_traffic = pipeline, device, int(count_traffic)
        be.call("render_device", scene.handle, C.byref(self.camera.desc), C.byref(film.desc), C.byref(sampler.desc),
                C.byref(self.radiance.desc), C.byref(tr), C.byref(opt), C.c_void_p(device_pixels_ptr),
                C.c_void_p(stream_ptr), C.byref(st))
        self.last_stats = st.as_dict()
        return self.last_stats
