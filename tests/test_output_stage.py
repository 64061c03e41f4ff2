"""Output stage: the OpenEXR writer / reader behind write_exr / read_exr (imageio/exr.rs:11-87) and, on the
GPU, Film::into_spectrum_buffer in HBM.  The file layout is checked by an independent reader written here from the OpenEXR
file-layout specification; the RLE + HALF read path by an independent encoder."""
import os
import struct

import numpy as np
import pytest

from fountain_amd import FountainError, PbrtScene, SceneBuilder, read_exr, write_exr
from fountain_amd import _abi as A


def _independent_exr_read(path):
    b = open(path, "rb").read()
    assert b[:4] == bytes([0x76, 0x2f, 0x31, 0x01]) and struct.unpack("<I", b[4:8])[0] == 2
    p, attrs = 8, {}
    while b[p] != 0:
        e = b.index(0, p); name = b[p:e].decode(); p = e + 1
        e = b.index(0, p); typ = b[p:e].decode(); p = e + 1
        (n,) = struct.unpack("<i", b[p:p + 4]); p += 4
        attrs[name] = (typ, b[p:p + n]); p += n
    p += 1
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    assert attrs["compression"][1] == b"\0" and attrs["lineOrder"][1] == b"\0"
    ch, q, raw = [], 0, attrs["channels"][1]
    while raw[q] != 0:
        e = raw.index(0, q); nm = raw[q:e].decode(); q = e + 1
        typ, lin, xs, ys = struct.unpack("<iB3xii", raw[q:q + 16]); q += 16
        ch.append((nm, typ)); assert xs == 1 and ys == 1
    assert [c[0] for c in ch] == sorted(c[0] for c in ch), "channels must be stored alphabetically"
    offs = struct.unpack("<%dQ" % h, b[p:p + 8 * h])
    img = np.zeros((h, w, 3), np.float32)
    for off in offs:
        y, n = struct.unpack("<ii", b[off:off + 8])
        assert n == 4 * w * len(ch)
        row = np.frombuffer(b[off + 8:off + 8 + n], "<f4").reshape(len(ch), w)
        for k, (nm, typ) in enumerate(ch):
            assert typ == 2
            img[y - y0, :, "RGB".index(nm)] = row[k]
    return img, attrs


def test_exr_writer_layout_and_bits(ftn, tmp_path):
    rng = np.random.default_rng(5)
    img = rng.standard_normal((37, 53, 3)).astype(np.float32) * 100
    img[0, 0] = (0.0, np.float32(1e-42), np.inf)          # zero, a denormal, an infinity survive untouched
    path = str(tmp_path / "a.exr")
    write_exr(path, img, ftn)
    got, attrs = _independent_exr_read(path)
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))
    assert attrs["name"][1] == b"image"                    # Layer::new("image", ..) exr.rs:74
    back = read_exr(path, ftn)
    assert np.array_equal(back.view(np.uint32), img.view(np.uint32))


def _rle_block(raw):
    """OpenEXR RLE: de-interleave into two halves, delta-predict, then run-length encode (runs as literals here and there)."""
    n = len(raw)
    t = bytes(raw[0::2]) + bytes(raw[1::2])
    d = bytearray(n); d[0] = t[0]
    for i in range(1, n):
        d[i] = (t[i] - t[i - 1] + 128) & 255
    out, i = bytearray(), 0
    while i < n:
        j = i
        while j + 1 < n and d[j + 1] == d[i] and j - i < 126:
            j += 1
        if j - i >= 2:
            out += struct.pack("b", j - i) + bytes([d[i]]); i = j + 1
        else:
            k = i
            while k < n and k - i < 127 and not (k + 2 < n and d[k] == d[k + 1] == d[k + 2]):
                k += 1
            out += struct.pack("b", -(k - i)) + bytes(d[i:k]); i = k
    return bytes(out)


@pytest.mark.parametrize("half", [False, True])
def test_exr_reader_rle_and_half(ftn, tmp_path, half):
    """The file kind the reference's own writer produces (RLE scanlines), plus HALF channels (read_exr's F16 arm)."""
    w, h = 24, 9
    rng = np.random.default_rng(9)
    img = rng.random((h, w, 3)).astype(np.float32)
    img[2:5] = 0.25                                            # long runs
    if half:
        img = img.astype(np.float16).astype(np.float32)
    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chl = b"".join(c.encode() + b"\0" + struct.pack("<iB3xii", 1 if half else 2, 0, 1, 1) for c in "BGR") + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    hdr = bytes([0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0]) + attr("channels", "chlist", chl) + attr("compression", "compression", b"\1") + \
        attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + \
        attr("pixelAspectRatio", "float", struct.pack("<f", 1)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1)) + b"\0"
    blocks = []
    for y in range(h):
        raw = b"".join(img[y, :, "RGB".index(c)].astype("<f2" if half else "<f4").tobytes() for c in "BGR")
        comp = _rle_block(raw)
        data = comp if len(comp) < len(raw) else raw           # OpenEXR stores the raw block when compression does not help
        blocks.append(struct.pack("<ii", y, len(data)) + data)
    off, offs = len(hdr) + 8 * h, []
    for blk in blocks:
        offs.append(off); off += len(blk)
    path = tmp_path / "r.exr"
    path.write_bytes(hdr + struct.pack("<%dQ" % h, *offs) + b"".join(blocks))
    got = read_exr(str(path), ftn)
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))


@pytest.mark.parametrize("comp,lines", [(2, 1), (3, 16)])
def test_exr_reader_zip(ftn, tmp_path, comp, lines):
    """ZIPS (one scanline per block) and ZIP (16 per block, the last one short): the usual encodings of PBRT scenes' EXR textures"""
    import zlib
    w, h = 21, 37
    rng = np.random.default_rng(comp)
    img = (rng.random((h, w, 3)) * np.linspace(0.1, 4, w)[None, :, None]).astype(np.float32)
    def attr(name, typ, data):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(data)) + data
    chl = b"".join(c.encode() + b"\0" + struct.pack("<iB3xii", 2, 0, 1, 1) for c in "BGR") + b"\0"
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    hdr = bytes([0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0]) + attr("channels", "chlist", chl) + attr("compression", "compression", bytes([comp])) + \
        attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box) + attr("lineOrder", "lineOrder", b"\0") + \
        attr("pixelAspectRatio", "float", struct.pack("<f", 1)) + attr("screenWindowCenter", "v2f", struct.pack("<2f", 0, 0)) + \
        attr("screenWindowWidth", "float", struct.pack("<f", 1)) + b"\0"
    blocks = []
    for y0 in range(0, h, lines):
        raw = b"".join(img[y, :, "RGB".index(c)].astype("<f4").tobytes() for y in range(y0, min(y0 + lines, h)) for c in "BGR")
        t = bytes(raw[0::2]) + bytes(raw[1::2])                                      # two halves
        d = bytearray(len(t)); d[0] = t[0]
        for i in range(1, len(t)):
            d[i] = (t[i] - t[i - 1] + 128) & 255                                     # byte predictor
        z = zlib.compress(bytes(d))
        data = z if len(z) < len(raw) else raw
        blocks.append(struct.pack("<ii", y0, len(data)) + data)
    off, offs = len(hdr) + 8 * len(blocks), []
    for blk in blocks:
        offs.append(off); off += len(blk)
    path = tmp_path / "z.exr"
    path.write_bytes(hdr + struct.pack("<%dQ" % len(blocks), *offs) + b"".join(blocks))
    got = read_exr(str(path), ftn)
    assert np.array_equal(got.view(np.uint32), img.view(np.uint32))


def test_exr_errors(ftn, tmp_path):
    with pytest.raises(FountainError):
        read_exr(str(tmp_path / "missing.exr"), ftn)
    (tmp_path / "junk.exr").write_bytes(b"not an exr file at all")
    with pytest.raises(FountainError) as e:
        read_exr(str(tmp_path / "junk.exr"), ftn)
    assert e.value.code == A.FTN_ERR_INVALID_ARGUMENT


def test_infinite_light_mapname(ftn, tmp_path):
    """make_infinite_area_light with "mapname" (constructors.rs:339-359 -> load_mipmap imageio/mod.rs:81-106): EXR texels * scale[0]."""
    from test_pbrt_loader import assert_same_desc
    rng = np.random.default_rng(2)
    tex = rng.random((8, 8, 3)).astype(np.float32)
    write_exr(str(tmp_path / "sky.exr"), tex, ftn)
    (tmp_path / "s.pbrt").write_text('Camera "perspective"\nWorldBegin\nRotate 30 0 1 0\n'
                                      'LightSource "infinite" "string mapname" "sky.exr" "rgb scale" [2.5 9 9]\nShape "sphere"\nWorldEnd\n')
    ps = PbrtScene(str(tmp_path / "s.pbrt"), ftn)
    b = SceneBuilder(ftn)
    b.rotate(30, (0, 1, 0))
    b.light_source("infinite", texels=tex * np.float32(2.5))
    b.shape("sphere")
    d, keep = b.build_desc()
    assert_same_desc(ps.desc, d)


@pytest.mark.gpu
def test_device_film_and_spectrum_buffer_match_host(gpu, tmp_path):
    """ftn_render_device + ftn_film_resolve_device (film resident in HBM) == ftn_render + host into_spectrum_buffer, bit for bit;
    then the CLI end to end on the reference's furnace scene file."""
    import torch
    from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, film_resolve_device, scenes
    b, cam, res = scenes.cornell(gpu, 96)
    scene = b.create_scene()
    integ = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
    host = Film(gpu, res)
    integ.render_parallel(scene, host, RandomSampler(4, indexed=True))
    rgb_host, _ = host.into_spectrum_buffer()
    dev = torch.zeros((res[1], res[0], 4), dtype=torch.float32, device="cuda:0")
    rgb = torch.empty((res[1], res[0], 3), dtype=torch.float32, device="cuda:0")
    stream = torch.cuda.current_stream().cuda_stream
    integ.render_device(scene, Film(gpu, res), RandomSampler(4, indexed=True), dev.data_ptr(), stream)
    film_resolve_device(gpu, dev.data_ptr(), res[0] * res[1], rgb.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(dev.cpu().numpy().view(np.uint32), host.pixels.view(np.uint32))
    assert np.array_equal(rgb.cpu().numpy().view(np.uint32), rgb_host.view(np.uint32))

    from fountain_amd import render as cli
    out = str(tmp_path / "furnace.exr")
    golden = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "furnace_empty.pbrt")
    assert cli.main([golden, "-o", out, "--samples", "16", "--max-depth", "10", "--rr-threshold", "0.0", "--exact-stream"]) == 0
    img = read_exr(out, gpu)
    assert img.shape == (16, 16, 3)
    assert np.all(np.abs(img - 2.0) <= 1e-3)                     # tests/furnace.rs path_no_rr: L -> 1/(1-0.5) = 2, eps 0.001


@pytest.mark.gpu
def test_rccl_film_merge_single_rank(gpu):
    """The multi-GPU merge (fountain_amd.distributed.merge_film) over the real RCCL backend, with the one rank this box has:
    render a tile shard into a device film, reduce, compare with the host render.  (N > 1 semantics: tests/test_multigpu_gloo.py.)"""
    import torch
    import torch.distributed as dist
    from fountain_amd import Film, PathIntegrator, RandomSampler, SamplerIntegrator, scenes
    from fountain_amd.distributed import merge_film, tile_shard
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(29600 + os.getpid() % 300))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        b, cam, res = scenes.cornell(gpu, 64)
        scene = b.create_scene()
        integ = SamplerIntegrator(cam, PathIntegrator.new(5, 1.0))
        dev = torch.zeros((res[1], res[0], 4), dtype=torch.float32, device="cuda:0")
        integ.render_device(scene, Film(gpu, res), RandomSampler(2, indexed=True), dev.data_ptr(), torch.cuda.current_stream().cuda_stream,
                            tiles=tile_shard(0, 1))
        # with world_size 1 merge_film is the identity; force the collective itself as well
        merged = merge_film(dev)
        dist.reduce(dev, dst=0, op=dist.ReduceOp.SUM)
        dist.barrier()
        torch.cuda.synchronize()
        host = Film(gpu, res)
        integ.render_parallel(scene, host, RandomSampler(2, indexed=True))
        assert np.array_equal(merged.cpu().numpy().view(np.uint32), host.pixels.view(np.uint32))
    finally:
        dist.destroy_process_group()
