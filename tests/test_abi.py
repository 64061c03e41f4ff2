"""The C-ABI library loads and exports every symbol include/fountain_hip.h declares; struct layouts match the ctypes
mirror.  No compute calls: this runs without a GPU."""
import ctypes as C
import os
import re

import pytest

from fountain_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "fountain_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ftn_[a-z0-9_]+)\s*\(", src)))


def test_struct_sizes():
    for name, size in A.SIZES.items():
        assert C.sizeof(getattr(A, name)) == size, name


def test_header_and_mirror_agree():
    assert header_functions() == sorted(A.DECLARED_FUNCTIONS)


def test_library_exports_every_declared_symbol(ftn):
    for name in header_functions():
        assert hasattr(ftn.lib, name), "libfountain_hip.so does not export %s" % name


def test_version_and_error_strings(ftn):
    v = ftn.lib.ftn_version
    v.restype = C.c_char_p
    assert b"gfx950" in v()
    assert ftn.lib.ftn_abi_version() == A.FTN_ABI_VERSION
    header = open(os.path.join(ROOT, "include", "fountain_hip.h")).read()
    assert int(re.search(r"#define\s+FTN_ABI_VERSION\s+(\d+)", header).group(1)) == A.FTN_ABI_VERSION


def test_oracle_exports_twins(orc):
    for name in header_functions():
        twin = "orc_" + name[4:]
        if name in ("ftn_render_device", "ftn_device_count", "ftn_version", "ftn_abi_version", "ftn_scene_memory_info", "ftn_bvh_build", "ftn_bvh_quads", "ftn_bvh_octs", "ftn_test_math"):
            continue   # device-only / covered by orc_scene_get_nodes
        if name.startswith(("ftn_pbrt_", "ftn_ply_", "ftn_exr_", "ftn_imageio_")) or name in ("ftn_film_resolve_device", "ftn_test_mipmap_level", "ftn_test_texture_eval"):
            continue   # host-side file ingestion: checked against SceneBuilder in tests/test_pbrt_loader.py
        assert hasattr(orc.lib, twin), twin


def test_compute_without_gpu_fails_loudly(ftn):
    """The product has no CPU fallback: without a device, scene creation reports FTN_ERR_NO_DEVICE."""
    if ftn.fn("device_count")() > 0:
        pytest.skip("a GPU is present")
    from fountain_amd import FountainError, scenes
    b, cam, res = scenes.furnace(ftn)          # host-side constructors work without a GPU
    with pytest.raises(FountainError) as e:
        b.create_scene()
    assert e.value.code == A.FTN_ERR_NO_DEVICE


def test_product_sources_do_not_reach_into_the_oracle():
    """oracle/ is test infrastructure: nothing under fountain_amd/ may include, link or load it (a product path that routes through
    the CPU restatement would void every parity claim)."""
    import glob, os, re
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "fountain_amd")
    bad = []
    for path in glob.glob(os.path.join(root, "**", "*"), recursive=True):
        if not os.path.isfile(path) or not path.endswith((".py", ".cpp", ".h", ".hpp", ".hip", "Makefile")):
            continue
        text = open(path, encoding="utf-8", errors="replace").read()
        for m in re.finditer(r'liboracle|orc_[a-z_]+\s*\(|#include\s*"[^"]*oracle|oracle_loader', text):
            bad.append((os.path.relpath(path, root), m.group(0)))
    assert not bad, bad
