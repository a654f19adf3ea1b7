"""The C++ host front end (gradientdomain-mitsuba_amd/host): scene-XML subset reader and the reference CLI's flags.
CPU tests use --parse-only (no GPU); the GPU test renders through the CLI and through the Python mirror and expects
the same bytes, since both marshal the same scene into the same C-ABI calls."""
import json
import os
import subprocess
import textwrap

import numpy as np
import pytest

from gradientdomain_mitsuba_amd import _build, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
XML = os.path.join(ROOT, "scenes", "cornell_box.xml")


@pytest.fixture(scope="module")
def cli():
    _build.build()
    return _build.HOST_BIN


def run(cli, *args):
    return subprocess.run([cli] + list(args), capture_output=True, text=True)


def test_parse_only_matches_the_python_scene(cli):
    r = run(cli, "--parse-only", "-D", "width=1280", "-D", "height=720", "-D", "spp=16", XML)
    assert r.returncode == 0, r.stderr
    d = json.loads(r.stdout)
    sc = scenes.cornell_box(1280, 720)
    assert d["triangles"] == sc.ntri == 32 and d["materials"] == 4 and d["emitters"] == 1
    assert (d["width"], d["height"], d["sampleCount"], d["maxDepth"]) == (1280, 720, 16, -1)
    assert d["fovX"] == pytest.approx(sc.fov_x, rel=1e-9) and d["cameraOrigin"] == [278, 273, -800]
    assert d["firstVertex"] == list(sc.verts[0][:3])
    assert json.loads(run(cli, "--parse-only", XML).stdout)["width"] == 512            # <default> values


def scene_with(tmp_path, body, film='<film type="multifilm"><integer name="width" value="8"/><integer name="height" value="4"/><string name="fileFormat" value="pfm"/><rfilter type="box"/></film>'):
    p = tmp_path / "s.xml"
    p.write_text(textwrap.dedent('''<scene version="0.5.0">
        <integrator type="gpt"/>
        <sensor type="perspective"><float name="fov" value="40"/>
            <transform name="toWorld"><lookat origin="0,0,-5" target="0,0,0" up="0,1,0"/></transform>
            <sampler type="independent"><integer name="sampleCount" value="2"/></sampler>%s</sensor>
        %s
        </scene>''') % (film, body))
    return str(p)


LIGHT = '<shape type="rectangle"><emitter type="area"><rgb name="radiance" value="3"/></emitter></shape>'


def test_transforms_compose_in_document_order(cli, tmp_path):
    # scale then translate: the rectangle's corner (-1,-1,0) -> (-2,-2,0) -> (8,-2,0); translate then scale would give (18,-2,0)
    s = scene_with(tmp_path, '<shape type="rectangle"><transform name="toWorld"><scale value="2"/><translate x="10"/></transform></shape>' + LIGHT)
    d = json.loads(run(cli, "--parse-only", s).stdout)
    assert d["firstVertex"] == [8, -2, 0] and d["triangles"] == 4 and d["materials"] == 2
    s = scene_with(tmp_path, '<shape type="rectangle"><transform name="toWorld"><rotate z="1" angle="90"/></transform></shape>' + LIGHT)
    assert np.allclose(json.loads(run(cli, "--parse-only", s).stdout)["firstVertex"], [1, -1, 0], atol=1e-12)
    s = scene_with(tmp_path, '<shape type="cube"><bsdf type="roughconductor"><float name="alpha" value="0.2"/><rgb name="eta" value="1,1,1"/><rgb name="k" value="2,2,2"/><string name="distribution" value="ggx"/></bsdf></shape>' + LIGHT)
    assert json.loads(run(cli, "--parse-only", s).stdout)["triangles"] == 14
    s = scene_with(tmp_path, '<shape type="cube"><bsdf type="dielectric"><string name="intIOR" value="water"/><float name="extIOR" value="1.0"/></bsdf></shape>' + LIGHT)
    assert json.loads(run(cli, "--parse-only", s).stdout)["materials"] == 2
    s = scene_with(tmp_path, '<bsdf type="twosided" id="w"><bsdf type="diffuse"/></bsdf><shape type="rectangle"><ref id="w"/></shape>' + LIGHT)
    assert json.loads(run(cli, "--parse-only", s).stdout)["materials"] == 2
    # OBJ normals (obj.cpp createMesh + TriMesh::computeNormals): a square pyramid with shared apex/base vertices
    pyr = tmp_path / "pyr.obj"
    pyr.write_text("v 0 0 0\nv 1 0 0\nv 1 0 1\nv 0 0 1\nv 0.5 1 0.5\nf 1 5 2\nf 2 5 3\nf 3 5 4\nf 4 5 1\n")
    sh = lambda extra: scene_with(tmp_path, '<shape type="obj"><string name="filename" value="%s"/>%s</shape>' % (pyr, extra) + LIGHT)
    d = json.loads(run(cli, "--parse-only", sh("")).stdout)
    assert d["smoothTriangles"] == 4                                        # no vn in the file: angle-weighted vertex normals over shared vertices
    n = np.array(d["firstNormal"]); assert abs(np.linalg.norm(n) - 1) < 1e-8 and n[1] > 0 and n[0] < 0 and n[2] < 0   # corner (0,0,0) leans outwards
    assert json.loads(run(cli, "--parse-only", sh('<boolean name="faceNormals" value="true"/>')).stdout)["smoothTriangles"] == 0
    nf = np.array(json.loads(run(cli, "--parse-only", sh('<boolean name="flipNormals" value="true"/>')).stdout)["firstNormal"])
    assert np.allclose(nf, -n, atol=1e-9)                                  # computed normals are negated, the winding stays
    pyr.write_text("v 0 0 0\nv 1 0 0\nv 0.5 1 0.5\nvn 0 0 -1\nvn 0.6 0 -0.8\nf 1//1 3//2 2//1\n")
    d = json.loads(run(cli, "--parse-only", sh('<transform name="toWorld"><scale x="2" y="1" z="1"/></transform>')).stdout)
    assert d["smoothTriangles"] == 1 and np.allclose(d["firstNormal"], [0, 0, -1])
    unit = np.array([0.6 / 2, 0, -0.8]); unit /= np.linalg.norm(unit)       # normals go through the inverse transpose
    pyr.write_text("v 0 0 0\nv 1 0 0\nv 0.5 1 0.5\nvn 0.6 0 -0.8\nf 1//1 3//1 2//1\n")
    d = json.loads(run(cli, "--parse-only", sh('<transform name="toWorld"><scale x="2" y="1" z="1"/></transform>')).stdout)
    assert np.allclose(d["firstNormal"], unit, atol=1e-8)
    pyr.write_text("v 0 0 0\nv 1 0 0\nv 0.5 1 0\nvn 0 0 -1\nf 1//1 3//1 2//1\n")        # normals equal to the face normal: the flat path
    assert json.loads(run(cli, "--parse-only", sh("")).stdout)["smoothTriangles"] == 0
    # <emitter type="constant">: radiance and its place in the emitter list (XML order)
    s = scene_with(tmp_path, '<emitter type="constant"><rgb name="radiance" value="0.5, 0.75, 1"/></emitter><shape type="rectangle"/>' + LIGHT)
    assert json.loads(run(cli, "--parse-only", s).stdout)["environment"] == [0.5, 0.75, 1, 0]
    s = scene_with(tmp_path, LIGHT + '<shape type="rectangle"/><emitter type="constant"/>')
    assert json.loads(run(cli, "--parse-only", s).stdout)["environment"] == [1, 1, 1, 1]
    s = scene_with(tmp_path, '<emitter type="point"><point name="position" x="1" y="2" z="3"/><rgb name="intensity" value="5, 6, 7"/></emitter><shape type="rectangle"/>')
    assert json.loads(run(cli, "--parse-only", s).stdout)["emitters"] == 1               # a point light alone is a valid scene
    s = scene_with(tmp_path, '<emitter type="point"><point name="position" x="1" y="2" z="3"/><transform name="toWorld"><translate x="1"/></transform></emitter><shape type="rectangle"/>')
    r = run(cli, "--parse-only", s); assert r.returncode == 1 and "Only one of the parameters" in r.stderr
    s = scene_with(tmp_path, '<shape type="rectangle"/><emitter type="constant"/>')          # the environment alone lights the scene
    assert json.loads(run(cli, "--parse-only", s).stdout)["emitters"] == 0


def _write_serialized(path, meshes, version=4, double=False):
    """Mitsuba's compressed mesh format (TriMesh::serialize, trimesh.cpp:789-870): per mesh header 0x041C + version, then a zlib
    stream: flags, [name\\0 (v4)], u64 vertex count, u64 triangle count, positions, [normals], u32 indices; for several meshes
    an offset dictionary at the end (u64 offsets in v4, u32 in v3) + u32 count."""
    import struct, zlib
    blob, offsets = b"", []
    fl = "d" if double else "f"
    for (pos, nrm, idx) in meshes:
        offsets.append(len(blob))
        flags = (0x2000 if double else 0x1000) | (0x0001 if nrm is not None else 0)
        body = struct.pack("<I", flags) + (b"mesh\0" if version == 4 else b"") + struct.pack("<QQ", len(pos), len(idx))
        body += struct.pack("<%d%s" % (3 * len(pos), fl), *np.asarray(pos, float).ravel())
        if nrm is not None:
            body += struct.pack("<%d%s" % (3 * len(pos), fl), *np.asarray(nrm, float).ravel())
        body += struct.pack("<%dI" % (3 * len(idx)), *np.asarray(idx, int).ravel())
        blob += struct.pack("<HH", 0x041C, version) + zlib.compress(body)
    if len(meshes) > 1:
        blob += b"".join(struct.pack("<Q" if version == 4 else "<I", o) for o in offsets) + struct.pack("<I", len(meshes))
    open(path, "wb").write(blob)


def test_serialized_meshes(cli, tmp_path):
    """`<shape type="serialized">` (serialized.cpp, TriMesh::loadCompressed): v3/v4, float/double, shapeIndex through the offset
    dictionary, normals through the inverse transpose, the winding swap under a mirroring transform, computed normals."""
    quad = ([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], None, [[0, 1, 2], [0, 2, 3]])
    pyr_pos = [[0, 0, 0], [1, 0, 0], [1, 0, 1], [0, 0, 1], [0.5, 1, 0.5]]
    pyr = (pyr_pos, None, [[0, 4, 1], [1, 4, 2], [2, 4, 3], [3, 4, 0]])
    tilted = ([[0, 0, 0], [1, 0, 0], [0.5, 1, 0]], [[0.6, 0, -0.8]] * 3, [[0, 2, 1]])
    sh = lambda fn, extra="": scene_with(tmp_path, '<shape type="serialized"><string name="filename" value="%s"/>%s</shape>' % (fn, extra) + LIGHT)
    for version, dbl in ((3, False), (4, False), (4, True)):
        fn = str(tmp_path / ("m%d%d.serialized" % (version, dbl)))
        _write_serialized(fn, [quad, pyr, tilted], version, dbl)
        d0 = json.loads(run(cli, "--parse-only", sh(fn)).stdout)
        assert d0["triangles"] == 2 + 2 and d0["smoothTriangles"] == 0 and d0["firstVertex"] == [0, 0, 0]     # shape 0: the planar quad (+ the light)
        d1 = json.loads(run(cli, "--parse-only", sh(fn, '<integer name="shapeIndex" value="1"/>')).stdout)
        assert d1["triangles"] == 4 + 2 and d1["smoothTriangles"] == 4                                        # pyramid: angle-weighted vertex normals
        d1f = json.loads(run(cli, "--parse-only", sh(fn, '<integer name="shapeIndex" value="1"/><boolean name="faceNormals" value="true"/>')).stdout)
        assert d1f["smoothTriangles"] == 0
        d2 = json.loads(run(cli, "--parse-only", sh(fn, '<integer name="shapeIndex" value="2"/><transform name="toWorld"><scale x="2"/></transform>')).stdout)
        n = np.array([0.3, 0, -0.8]); n /= np.linalg.norm(n)
        assert d2["smoothTriangles"] == 1 and np.allclose(d2["firstNormal"], n, atol=1e-6)
        d2m = json.loads(run(cli, "--parse-only", sh(fn, '<integer name="shapeIndex" value="2"/><transform name="toWorld"><scale x="-1"/></transform>')).stdout)
        assert d2m["firstVertex"] == [-0.5, 1, 0]                         # mirrored: the first two vertices of the triangle are swapped
        r = run(cli, "--parse-only", sh(fn, '<integer name="shapeIndex" value="7"/>'))
        assert r.returncode == 1 and "out of range" in r.stderr
    for blob, needle in ((b"\x34\x12\x04\x00garbage", "invalid file format"), (b"\x04\x1c\x04\x00garbage", "old version of Mitsuba"),
                         (b"\x1c\x04\x09\x00garbage", "incompatible file version"), (b"\x1c\x04\x04\x00garbage", "corrupt")):
        bad = str(tmp_path / "bad.serialized"); open(bad, "wb").write(blob)
        r = run(cli, "--parse-only", sh(bad)); assert r.returncode == 1 and needle in r.stderr, r.stderr


@pytest.mark.parametrize("body,film,needle", [
    ('<shape type="sphere"/>' + LIGHT, None, "shape \"sphere\" is not carried"),
    ('<shape type="rectangle"><bsdf type="plastic"/></shape>' + LIGHT, None, "not carried"),
    ('<shape type="rectangle"><bsdf type="dielectric"><string name="intIOR" value="unobtainium"/></bsdf></shape>' + LIGHT, None, "Unable to find an IOR"),
    ('<shape type="rectangle"><bsdf type="twosided"><bsdf type="dielectric"/></bsdf></shape>' + LIGHT, None, "transmission component"),
    ('<shape type="rectangle"><bsdf type="twosided"/></shape>' + LIGHT, None, "nested one-sided material is required"),
    ('<shape type="rectangle"><bsdf type="conductor"/></shape>' + LIGHT, None, "explicit eta and k"),
    ('<shape type="rectangle"/>', None, "no emitter"),
    ('<emitter type="constant"/><emitter type="constant"/>' + LIGHT, None, "Only one environment emitter"),
    (LIGHT, '<film type="hdrfilm"><rfilter type="box"/></film>', "without MultiFilm"),
    (LIGHT, '<film type="multifilm"><string name="fileFormat" value="pfm"/><rfilter type="sinc"/></film>', "rfilter \"sinc\" is not carried"),
    ('<emitter type="sunsky"/>' + LIGHT, None, "not carried"),
    ('<emitter type="envmap"/>' + LIGHT, None, "missing filename"),
    ('<shape type="rectangle"><ref id="nope"/></shape>' + LIGHT, None, "not found"),
])
def test_unsupported_input_is_an_error_with_a_reason(cli, tmp_path, body, film, needle):
    s = scene_with(tmp_path, body, film) if film else scene_with(tmp_path, body)
    r = run(cli, "--parse-only", s)
    assert r.returncode == 1 and needle in r.stderr, r.stderr


def test_cli_errors(cli, tmp_path):
    assert run(cli).returncode == 1 and "no scene file" in run(cli).stderr
    assert "undefined parameter" in run(cli, "--parse-only", scene_with(tmp_path, '<shape type="rectangle"><transform name="toWorld"><translate x="$dx"/></transform></shape>' + LIGHT)).stderr
    bad = tmp_path / "bad.xml"; bad.write_text("<scene><shape></scene>")
    assert run(cli, "--parse-only", str(bad)).returncode == 1


def read_exr(path):
    """Minimal independent reader (numpy + zlib) for the scanline RGB OpenEXR files MultiFilm's default format produces here: uncompressed,
    ZIPS (1 line per chunk) or ZIP (16 lines per chunk; OpenEXR's default and what the reference's Bitmap::writeOpenEXR writes)."""
    import struct
    import zlib
    b = open(path, "rb").read()
    assert struct.unpack("<II", b[:8]) == (20000630, 2)
    i, attrs = 8, {}
    while b[i] != 0:
        e = b.index(b"\0", i); name = b[i:e].decode(); i = e + 1
        e = b.index(b"\0", i); typ = b[i:e].decode(); i = e + 1
        (sz,) = struct.unpack("<i", b[i:i + 4]); i += 4
        attrs[name] = (typ, b[i:i + sz]); i += sz
    i += 1
    comp = attrs["compression"][1][0]
    assert comp in (0, 2, 3) and attrs["lineOrder"][1] == b"\0"
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    w, h = x1 - x0 + 1, y1 - y0 + 1
    ch, j, names, types = attrs["channels"][1], 0, [], []
    while ch[j] != 0:
        e = ch.index(b"\0", j); names.append(ch[j:e].decode()); j = e + 1
        types.append(struct.unpack("<i", ch[j:j + 4])[0]); j += 16
    assert names == ["B", "G", "R"] and len(set(types)) == 1
    dt = {1: "<f2", 2: "<f4"}[types[0]]
    lines = 16 if comp == 3 else 1
    chunks = (h + lines - 1) // lines
    offs = struct.unpack("<%dQ" % chunks, b[i:i + 8 * chunks])
    img = np.zeros((h, w, 3), np.float32)
    line_bytes = w * 3 * np.dtype(dt).itemsize
    for c in range(chunks):
        yy, sz = struct.unpack("<ii", b[offs[c]:offs[c] + 8])
        n = min(lines, h - yy)
        assert yy == c * lines
        raw = b[offs[c] + 8:offs[c] + 8 + sz]
        if comp != 0 and sz != n * line_bytes:                       # ImfZip: zlib -> undo the delta predictor -> interleave the two halves
            t = np.frombuffer(zlib.decompress(raw), np.uint8).astype(np.int64)
            assert t.size == n * line_bytes
            t[1:] -= 128
            t = (np.cumsum(t) % 256).astype(np.uint8)
            half = (t.size + 1) // 2
            out = np.empty(t.size, np.uint8); out[0::2] = t[:half]; out[1::2] = t[half:]
            raw = out.tobytes()
        assert len(raw) == n * line_bytes
        blk = np.frombuffer(raw, dt).reshape(n, 3, w)
        img[yy:yy + n, :, 2], img[yy:yy + n, :, 1], img[yy:yy + n, :, 0] = blk[:, 0], blk[:, 1], blk[:, 2]
    return img, attrs


def write_pfm(path, a):
    with open(path, "wb") as f:
        f.write(("PF\n%d %d\n-1.0\n" % (a.shape[1], a.shape[0])).encode()); f.write(a[::-1].astype("<f4").tobytes())


def test_exr_writer_round_trip(cli, tmp_path):
    rng = np.random.default_rng(0)
    a = (rng.standard_normal((7, 13, 3)) * 10).astype(np.float32)
    a[0, 0] = [0.0, 65504.0, 1e-7]; a[1, 1] = [1e6, -1e-3, 6.1e-5]        # half extremes: max, subnormal, overflow -> inf
    write_pfm(str(tmp_path / "a.pfm"), a)
    assert run(cli, "--pfm2exr", str(tmp_path / "a.pfm"), str(tmp_path / "f.exr"), "float32").returncode == 0
    img, attrs = read_exr(str(tmp_path / "f.exr"))
    assert np.array_equal(img, a) and attrs["channels"][0] == "chlist"
    assert run(cli, "--pfm2exr", str(tmp_path / "a.pfm"), str(tmp_path / "h.exr"), "float16").returncode == 0
    imgh, _ = read_exr(str(tmp_path / "h.exr"))
    with np.errstate(over="ignore"):
        assert np.array_equal(imgh, a.astype(np.float16).astype(np.float32))                                    # round-to-nearest-even, like numpy
    assert attrs["compression"][1] == b"\x03"                             # the default: ZIP, as OpenEXR's Header defaults (bitmap.cpp:3197)


def test_exr_zip_writer_reader_and_rgbe(cli, tmp_path):
    """ZIP / ZIPS / uncompressed scanline OpenEXR: written by the host's writer, read back by an independent numpy + zlib reader AND by the
    host's texture / environment-map reader (`--tex2pfm`); ragged sizes (the last ZIP block shorter than 16 lines, odd byte counts), a
    compressible and an incompressible image (a block that does not shrink is stored raw); the Radiance RGBE writer of fileFormat=rgbe."""
    rng = np.random.default_rng(4)
    for (h, w) in ((37, 21), (16, 5), (1, 1), (50, 64)):
        smooth = np.stack([np.add.outer(np.arange(h), np.arange(w)) * 0.01 + c for c in range(3)], -1).astype(np.float32)
        noisy = rng.standard_normal((h, w, 3)).astype(np.float32)
        for name, a in (("smooth", smooth), ("noisy", noisy)):
            write_pfm(str(tmp_path / "a.pfm"), a)
            for fmt in ("float32", "float16"):
                want = a if fmt == "float32" else a.astype(np.float16).astype(np.float32)
                for cmp in ("zip", "zips", "none"):
                    out = str(tmp_path / ("%s_%s_%s.exr" % (name, fmt, cmp)))
                    assert run(cli, "--pfm2exr", str(tmp_path / "a.pfm"), out, fmt, cmp).returncode == 0
                    img, attrs = read_exr(out)
                    assert attrs["compression"][1][0] == {"zip": 3, "zips": 2, "none": 0}[cmp] and np.array_equal(img, want), (h, w, name, fmt, cmp)
                    r = run(cli, "--tex2pfm", out, str(tmp_path / "back.pfm"))
                    assert r.returncode == 0, r.stderr
                    assert np.array_equal(read_pfm(str(tmp_path / "back.pfm")), want), (h, w, name, fmt, cmp)
        if (h, w) == (50, 64):
            assert os.path.getsize(str(tmp_path / "smooth_float32_zip.exr")) < 0.5 * os.path.getsize(str(tmp_path / "smooth_float32_none.exr"))
    # RGBE: shared exponent, ~1 % relative accuracy on the largest channel
    a = np.abs(rng.standard_normal((9, 14, 3))).astype(np.float32) * 5
    write_pfm(str(tmp_path / "a.pfm"), a)
    assert run(cli, "--pfm2exr", str(tmp_path / "a.pfm"), str(tmp_path / "a.rgbe"), "rgbe").returncode == 0
    b = open(str(tmp_path / "a.rgbe"), "rb").read()
    hdr_end = b.index(b"+X 14\n") + 6
    assert b.startswith(b"#?RGBE\n") and b"-Y 9 +X 14" in b[:hdr_end]
    px = np.frombuffer(b[hdr_end:], np.uint8).reshape(9, 14, 4).astype(np.float64)
    dec = px[..., :3] * np.exp2(px[..., 3:4] - 136.0)
    assert np.abs(dec - a).max() <= a.max(axis=-1, keepdims=True).max() / 100.0
    bad = tmp_path / "x.jpg"; bad.write_bytes(b"\xff\xd8\xff\xe0" + b"\0" * 32)
    r = run(cli, "--tex2pfm", str(bad), str(tmp_path / "back.pfm"))
    assert r.returncode == 1 and "JPEG" in r.stderr


def test_exr_reader_takes_the_data_window_origin(cli, tmp_path):
    """A scanline chunk stores its y in dataWindow coordinates: a cropped render (dataWindow.min != 0: common for textures and environment
    maps cut out of a larger frame) must decode into rows 0..h-1, not be rejected as truncated.  The file is this build's own, with its
    data / display windows and every chunk's y moved by (+11, +7) in place."""
    import struct
    rng = np.random.default_rng(9)
    a = rng.standard_normal((37, 21, 3)).astype(np.float32)
    write_pfm(str(tmp_path / "a.pfm"), a)
    for cmp in ("zip", "zips", "none"):
        out = str(tmp_path / ("w_%s.exr" % cmp))
        assert run(cli, "--pfm2exr", str(tmp_path / "a.pfm"), out, "float32", cmp).returncode == 0
        b = bytearray(open(out, "rb").read())
        i = 8
        while b[i] != 0:
            e = b.index(b"\0", i); name = bytes(b[i:e]).decode(); i = e + 1
            e = b.index(b"\0", i); i = e + 1
            (sz,) = struct.unpack("<i", b[i:i + 4]); i += 4
            if name in ("dataWindow", "displayWindow"):
                x0, y0, x1, y1 = struct.unpack("<4i", b[i:i + 16])
                b[i:i + 16] = struct.pack("<4i", x0 + 11, y0 + 7, x1 + 11, y1 + 7)
            i += sz
        i += 1
        lines = 16 if cmp == "zip" else 1
        chunks = (37 + lines - 1) // lines
        for off in struct.unpack("<%dQ" % chunks, b[i:i + 8 * chunks]):
            (yy,) = struct.unpack("<i", b[off:off + 4])
            b[off:off + 4] = struct.pack("<i", yy + 7)
        open(out, "wb").write(bytes(b))
        r = run(cli, "--tex2pfm", out, str(tmp_path / "back.pfm"))
        assert r.returncode == 0, r.stderr
        assert np.array_equal(read_pfm(str(tmp_path / "back.pfm")), a), cmp


def read_pfm(path):
    with open(path, "rb") as f:
        assert f.readline().strip() == b"PF"
        w, h = map(int, f.readline().split())
        assert float(f.readline()) < 0
        return np.frombuffer(f.read(), "<f4").reshape(h, w, 3)[::-1]


@pytest.mark.gpu
def test_cli_render_equals_python_mirror(cli, tmp_path, gpu_required):
    import gradientdomain_mitsuba_amd.gpt as G
    dest = str(tmp_path / "cbox")
    r = run(cli, "-o", dest, "-D", "width=48", "-D", "height=40", "-D", "spp=6", "-D", "maxDepth=6", "-p", "1", "-b", "32", XML)
    assert r.returncode == 0, r.stderr
    assert "Writing image" in r.stdout and "Using HIP" in r.stdout and "Execution time" in r.stdout
    out = G.GradientPathIntegrator(maxDepth=6).render(G.Scene(scenes.cornell_box(48, 40)), 6)
    for suffix in G.BUFFER_NAMES:
        img = read_pfm(dest + suffix + ".pfm")
        assert img.shape == (40, 48, 3)
        assert np.array_equal(img, out[suffix]), suffix
    assert os.path.exists(dest + "-log.txt") and "Render time" in open(dest + "-log.txt").read()
    stats = open(dest + "-stats.txt").read()                     # Statistics::getStats layout (statistics.cpp:152-272)
    assert stats.startswith("-" * 60 + "\n * Loaded plugins :") and stats.endswith("-" * 60)
    assert "\n  * General :\n    -  Normal rays traced : " in stats and "    -  Shadow rays traced : " in stats
    assert "\n  * Gradient Path Tracer :\n    -  Average path length : " in stats and " K / 11.52 K)" in stats    # 48 x 40 x 6 paths
    assert run(cli, "-o", dest, "-x", "-D", "width=48", "-D", "height=40", XML).stdout.startswith("Skipping")
    # MultiFilm's default output format: OpenEXR (float32 here; the default componentFormat is float16)
    xml32 = str(tmp_path / "exr.xml")
    open(xml32, "w").write(open(XML).read().replace('<string name="fileFormat" value="pfm"/>', '<string name="fileFormat" value="openexr"/>'))
    import shutil; shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    r = run(cli, "-o", dest + "e", "-D", "width=48", "-D", "height=40", "-D", "spp=6", "-D", "maxDepth=6", xml32)
    assert r.returncode == 0, r.stderr
    for suffix in G.BUFFER_NAMES:
        img, attrs = read_exr(dest + "e" + suffix + ".exr")
        assert np.array_equal(img, out[suffix]) and b"Render time" in attrs["log"][1]
    # the same box under a constant environment emitter (declared last: entry 1 of the emitter list)
    xmlenv = str(tmp_path / "env.xml")
    open(xmlenv, "w").write(open(XML).read().replace("</scene>", '<emitter type="constant"><rgb name="radiance" value="0.6, 0.8, 1.1"/></emitter></scene>'))
    r = run(cli, "-o", dest + "v", "-D", "width=48", "-D", "height=40", "-D", "spp=6", "-D", "maxDepth=6", xmlenv)
    assert r.returncode == 0, r.stderr
    oute = G.GradientPathIntegrator(maxDepth=6).render(G.Scene(scenes.cornell_box(48, 40, environment=(0.6, 0.8, 1.1))), 6)
    for suffix in G.BUFFER_NAMES:
        assert np.array_equal(read_pfm(dest + "v" + suffix + ".pfm"), oute[suffix]), suffix
    assert not np.array_equal(oute["-final"], out["-final"])
    # an OBJ sphere with per-vertex normals in the box: the CLI (obj reader -> gdpt_scene_create_ex) against the Python mirror
    sph = scenes.cornell_box(48, 40, "smooth")
    nt = sph.ntri
    first = 10 + 2                                        # the scene builder puts the two spheres after the 5 walls (10 triangles)
    with open(str(tmp_path / "meshes" / "sphere.obj"), "w") as f:
        k = 0
        for t in range(10, nt - 2):                       # both spheres (everything between the walls and the light)
            for j in range(3):
                f.write("v %.17g %.17g %.17g\n" % tuple(sph.verts[t][3 * j:3 * j + 3])); f.write("vn %.17g %.17g %.17g\n" % tuple(sph.normals[t][3 * j:3 * j + 3]))
            f.write("f %d//%d %d//%d %d//%d\n" % (k + 1, k + 1, k + 2, k + 2, k + 3, k + 3)); k += 3
    assert first
    import re
    xs = open(XML).read()
    # drop the two box shapes, add the sphere mesh with the tall sphere's material on all of it
    xs = re.sub(r'<shape type="obj">\s*<string name="filename" value="meshes/cbox_(small|large)box.obj"/>.*?</shape>', "", xs, flags=re.S)
    xs = xs.replace("</scene>", '<shape type="obj"><string name="filename" value="meshes/sphere.obj"/><bsdf type="diffuse"><rgb name="reflectance" value="0.725, 0.71, 0.68"/></bsdf></shape></scene>')
    xsm = str(tmp_path / "smooth.xml"); open(xsm, "w").write(xs)
    r = run(cli, "-o", dest + "s", "-D", "width=48", "-D", "height=40", "-D", "spp=4", "-D", "maxDepth=5", xsm)
    assert r.returncode == 0, r.stderr
    assert json.loads(run(cli, "--parse-only", "-D", "width=48", "-D", "height=40", xsm).stdout)["smoothTriangles"] == nt - 12
    img = read_pfm(dest + "s-final.pfm")
    assert np.isfinite(img).all() and img.max() > 0
    # <rfilter type="gaussian"> (Mitsuba's default film filter) through the CLI == the Python mirror with the same filter
    xg = str(tmp_path / "gauss.xml")
    open(xg, "w").write(open(XML).read().replace('<rfilter type="box"/>', '<rfilter type="gaussian"><float name="stddev" value="0.4"/></rfilter>'))
    assert '<rfilter type="gaussian">' in open(xg).read()
    r = run(cli, "-o", dest + "g", "-D", "width=32", "-D", "height=24", "-D", "spp=2", "-D", "maxDepth=4", xg)
    assert r.returncode == 0, r.stderr
    scg = scenes.cornell_box(32, 24); scg.rfilter = (scenes.RFILTER_GAUSSIAN, 0.4, 0.0)
    outg = G.GradientPathIntegrator(maxDepth=4).render(G.Scene(scg), 2)
    for suffix in G.BUFFER_NAMES:
        a, b = read_pfm(dest + "g" + suffix + ".pfm"), outg[suffix]
        assert np.allclose(a, b, rtol=1e-4, atol=1e-6), suffix          # fp32 images of fp64 sums accumulated by atomics in free order, then a solve
    bad = run(cli, "-o", dest, "-D", "width=16", "-D", "height=16", "-D", "maxDepth=0", XML)
    assert bad.returncode == 1 and "maxDepth" in bad.stderr


@pytest.mark.gpu
def test_cli_strips_over_devices_equal_one_device(cli, tmp_path, gpu_required):
    """gdpt_mitsuba -p N: the C++ host shards the frame into N row strips (one thread, scene copy and film per strip; halo rows packed,
    copied device to device and unpacked; develop per strip; gather and reconstruction on the first device -- host/gdpt_host.hpp
    renderStrips).  On a one-GPU box the strips share the device (ordinals wrap around): the data path is the same.  The image must
    not depend on N beyond the rounding of the border sums; ray statistics are identical."""
    base = str(tmp_path / "one")
    args = ["-D", "width=64", "-D", "height=50", "-D", "spp=5", "-D", "maxDepth=7"]
    r1 = run(cli, "-o", base, *args, XML)
    assert r1.returncode == 0, r1.stderr
    ref = {sfx: read_pfm(base + sfx + ".pfm") for sfx in ("-final", "-throughput", "-dx", "-dy", "-direct")}
    stats1 = open(base + "-stats.txt").read()
    for n, extra in ((3, ["-p", "3"]), (2, ["--devices", "0,0"]), (4, ["-p", "4"])):
        dest = str(tmp_path / ("strips%d" % n))
        r = run(cli, "-o", dest, *args, *extra, XML)
        assert r.returncode == 0, r.stderr
        log = open(dest + "-log.txt").read()
        assert ("%d strips" % n) in log and log.count("strip rows [") == n and "halo " in log
        assert open(dest + "-stats.txt").read() == stats1                                   # same rays, same paths
        for sfx, img in ref.items():
            got = read_pfm(dest + sfx + ".pfm")
            tol = 5e-5 if sfx == "-final" else 1e-6                                          # -final went through the fp32 CG
            assert np.allclose(got, img, rtol=tol, atol=tol * float(np.abs(img).max())), (n, sfx)
    # a film with the default (gaussian) reconstruction filter: strips render the filter's reach themselves, nothing is exchanged
    xg = str(tmp_path / "gauss.xml")
    import shutil
    shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    open(xg, "w").write(open(XML).read().replace('<rfilter type="box"/>', '<rfilter type="gaussian"/>'))
    g1, g2 = str(tmp_path / "g1"), str(tmp_path / "g2")
    assert run(cli, "-o", g1, "-D", "width=40", "-D", "height=30", "-D", "spp=3", "-D", "maxDepth=5", xg).returncode == 0
    r = run(cli, "-o", g2, "-D", "width=40", "-D", "height=30", "-D", "spp=3", "-D", "maxDepth=5", "-p", "2", xg)
    assert r.returncode == 0, r.stderr
    assert "halo 0 bytes" in open(g2 + "-log.txt").read()
    for sfx in ("-throughput", "-dx", "-dy", "-direct"):
        a, b = read_pfm(g1 + sfx + ".pfm"), read_pfm(g2 + sfx + ".pfm")
        assert np.allclose(a, b, rtol=1e-6, atol=1e-6 * float(np.abs(a).max())), sfx
    bad = run(cli, "-o", base, *args, "--devices", "0,7", XML)
    assert bad.returncode == 1 and "out of range" in bad.stderr


def write_pfm(path, img):
    img = np.asarray(img, np.float32)
    with open(path, "wb") as f:
        f.write(b"PF\n%d %d\n-1.0\n" % (img.shape[1], img.shape[0]))
        f.write(img[::-1].tobytes())


def write_png(path, img8):
    """A minimal 8-bit RGB PNG (filter type 0 and 1 rows alternating) for the reader's test."""
    import struct
    import zlib
    h, w = img8.shape[:2]
    raw = b""
    for y in range(h):
        row = img8[y].astype(np.int32).reshape(-1)
        if y % 2 == 0:
            raw += b"\x00" + img8[y].tobytes()
        else:                                                # Sub filter: difference to the pixel on the left
            left = np.concatenate([np.zeros(3, np.int32), row[:-3]])
            raw += b"\x01" + ((row - left) & 255).astype(np.uint8).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    open(path, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))


@pytest.mark.gpu
def test_cli_bitmap_textures_equal_python_mirror(cli, tmp_path, gpu_required):
    """`<texture type="bitmap">` on a diffuse reflectance through the scene reader (PFM and 8-bit sRGB PNG files, OBJ `vt` coordinates with
    flipTexCoords, filterType / wrapMode / uscale) == the Python mirror given the same texels and coordinates; the default filterType is `ewa`."""
    import shutil
    import gradientdomain_mitsuba_amd.gpt as G
    shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    sc = scenes.cornell_box(40, 30)
    v = np.asarray(sc.verts).reshape(-1, 3, 3)
    rgb = scenes.checker_rgb(12, 9, 5).astype(np.float32).astype(np.float64)           # what a float32 PFM holds
    write_pfm(str(tmp_path / "floor.pfm"), rgb)
    img8 = (scenes.checker_rgb(10, 6, 8) * 255).astype(np.uint8)
    write_png(str(tmp_path / "back.png"), img8)
    # the floor as an OBJ with texture coordinates (vt): u = x / 552.8 * 2, v = z / 559.2 * 1.5
    with open(str(tmp_path / "meshes" / "floor_uv.obj"), "w") as f:
        k = 0
        for t in (0, 1):
            for j in range(3):
                f.write("v %.17g %.17g %.17g\n" % tuple(v[t, j]))
                f.write("vt %.17g %.17g\n" % (v[t, j, 0] / 552.8 * 2.0, v[t, j, 2] / 559.2 * 1.5))
            f.write("f %d/%d %d/%d %d/%d\n" % (k + 1, k + 1, k + 2, k + 2, k + 3, k + 3)); k += 3
    xml = open(XML).read()
    xml = xml.replace('<string name="filename" value="meshes/cbox_floor.obj"/>\n\t\t<boolean name="faceNormals" value="true"/>\n\t\t<ref id="white"/>',
                      '<string name="filename" value="meshes/floor_uv.obj"/>\n\t\t<boolean name="faceNormals" value="true"/>\n\t\t<bsdf type="diffuse"><texture type="bitmap" name="reflectance">'
                      '<string name="filename" value="floor.pfm"/><string name="filterType" value="bilinear"/><string name="wrapMode" value="mirror"/><float name="uscale" value="1.5"/></texture></bsdf>')
    xml = xml.replace('<string name="filename" value="meshes/cbox_back.obj"/>\n\t\t<boolean name="faceNormals" value="true"/>\n\t\t<ref id="white"/>',
                      '<string name="filename" value="meshes/cbox_back.obj"/>\n\t\t<boolean name="faceNormals" value="true"/>\n\t\t<bsdf type="diffuse"><texture type="bitmap" name="reflectance">'
                      '<string name="filename" value="back.png"/><string name="filterType" value="nearest"/></texture></bsdf>')
    assert "floor.pfm" in xml and "back.png" in xml
    xt = str(tmp_path / "tex.xml"); open(xt, "w").write(xml)
    dest = str(tmp_path / "tex")
    r = run(cli, "-o", dest, "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", xt)
    assert r.returncode == 0, r.stderr
    # the Python mirror of the same scene
    nt = sc.ntri
    floor_m = len(sc.materials); sc.materials.append(scenes.diffuse((0.5, 0.5, 0.5)))
    back_m = len(sc.materials); sc.materials.append(scenes.diffuse((0.5, 0.5, 0.5)))
    tm = np.array(sc.tri_material, np.int32).copy(); tm[0:2] = floor_m; tm[4:6] = back_m
    sc.tri_material = tm
    uvs = np.zeros((nt, 6)); has = np.zeros(nt, np.uint8)
    for t in (0, 1):
        for j in range(3):
            uvs[t, 2 * j] = v[t, j, 0] / 552.8 * 2.0
            uvs[t, 2 * j + 1] = 1 - v[t, j, 2] / 559.2 * 1.5                      # flipTexCoords (obj.cpp:306-307)
        has[t] = 1
    sc.uvs, sc.tri_has_uv = uvs, has
    srgb = np.array([(i * float(np.float32(1.0) / np.float32(255))) for i in range(256)])
    lin = np.where(srgb <= 0.04045, srgb * (1.0 / 12.92), ((srgb + 0.055) * (1.0 / 1.055)) ** 2.4)      # fmtconv.cpp:1093-1098
    sc.textures = [scenes.bitmap_texture(rgb, wrap=scenes.TEXWRAP_MIRROR, filter=scenes.TEXFILTER_BILINEAR, uscale=1.5),
                   scenes.bitmap_texture(lin[img8], filter=scenes.TEXFILTER_NEAREST)]
    mt = [-1] * len(sc.materials); mt[floor_m], mt[back_m] = 0, 1
    sc.material_textures = mt
    out = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(sc), 4)
    for suffix in G.BUFFER_NAMES:
        img = read_pfm(dest + suffix + ".pfm")
        assert np.allclose(img, out[suffix], rtol=2e-6, atol=1e-7), suffix         # (pow() of the sRGB table: glibc here, numpy there)
    plain = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(scenes.cornell_box(40, 30)), 4)
    assert not np.allclose(plain["-throughput"], out["-throughput"], rtol=1e-2, atol=1e-3)
    # without a filterType the reference's default applies: ewa (bitmap.cpp:213), with maxAnisotropy as given == the Python mirror again
    ewa = str(tmp_path / "ewa.xml"); open(ewa, "w").write(xml.replace('<string name="filterType" value="bilinear"/>', '<float name="maxAnisotropy" value="4"/>'))
    r = run(cli, "-o", dest + "e", "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", ewa)
    assert r.returncode == 0, r.stderr
    sc.textures[0] = scenes.bitmap_texture(rgb, wrap=scenes.TEXWRAP_MIRROR, filter=scenes.TEXFILTER_EWA, uscale=1.5, maxAnisotropy=4.0)
    oute = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(sc), 4)
    for suffix in G.BUFFER_NAMES:
        assert np.allclose(read_pfm(dest + "e" + suffix + ".pfm"), oute[suffix], rtol=2e-6, atol=1e-7), suffix
    assert not np.allclose(oute["-throughput"], out["-throughput"], rtol=1e-3, atol=1e-5)
    bad = str(tmp_path / "badf.xml"); open(bad, "w").write(xml.replace('value="bilinear"', 'value="cubic"'))
    r = run(cli, "-o", dest + "x", "-D", "width=16", "-D", "height=12", bad)
    assert r.returncode == 1 and "Invalid filter type" in r.stderr


@pytest.mark.gpu
def test_cli_envmap_emitter_equals_python_mirror(cli, tmp_path, gpu_required):
    """`<emitter type="envmap">` through the scene reader (a PFM file, `scale`, a rotating `toWorld`, its place in the emitter list) == the
    Python mirror given the same map."""
    import shutil
    import gradientdomain_mitsuba_amd.gpt as G
    shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    img = scenes.sky_map(24, 12).astype(np.float32).astype(np.float64)
    write_pfm(str(tmp_path / "sky.pfm"), img)
    xml = open(XML).read().replace("</scene>", '<emitter type="envmap"><string name="filename" value="sky.pfm"/><float name="scale" value="0.75"/>'
                                   '<transform name="toWorld"><rotate y="1" angle="40"/></transform></emitter></scene>')
    xe = str(tmp_path / "env.xml"); open(xe, "w").write(xml)
    dest = str(tmp_path / "env")
    r = run(cli, "-o", dest, "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", xe)
    assert r.returncode == 0, r.stderr
    sc = scenes.cornell_box(40, 30)
    a = np.deg2rad(40.0)
    sc.environment_map = dict(rgb=img, scale=0.75, toWorld=np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]]), index=len(sc.emitters))
    out = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(sc), 4)
    for suffix in G.BUFFER_NAMES:
        assert np.allclose(read_pfm(dest + suffix + ".pfm"), out[suffix], rtol=2e-6, atol=1e-7), suffix
    plain = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(scenes.cornell_box(40, 30)), 4)
    assert not np.allclose(plain["-throughput"], out["-throughput"], rtol=1e-2, atol=1e-3)
    # the same map as a ZIP-compressed float32 OpenEXR file (what real scene files ship) renders the same bytes as its PFM copy
    assert run(cli, "--pfm2exr", str(tmp_path / "sky.pfm"), str(tmp_path / "sky.exr"), "float32", "zip").returncode == 0
    xz = str(tmp_path / "envz.xml"); open(xz, "w").write(xml.replace("sky.pfm", "sky.exr"))
    r = run(cli, "-o", dest + "z", "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", xz)
    assert r.returncode == 0, r.stderr
    for suffix in G.BUFFER_NAMES:
        assert np.array_equal(read_pfm(dest + "z" + suffix + ".pfm"), read_pfm(dest + suffix + ".pfm")), suffix


@pytest.mark.gpu
def test_cli_rectangle_light_equals_python_mirror(cli, tmp_path, gpu_required):
    """A `rectangle` shape with an area emitter through the scene reader (toWorld = scale, rotate, translate; once with flipNormals) ==
    the Python mirror: the two triangles of Rectangle::createTriMesh, texture coordinates, and the emitter sampled as the shape samples itself."""
    import shutil
    import gradientdomain_mitsuba_amd.gpt as G
    shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    for flipped in (False, True):
        ang = 10.0 if not flipped else -170.0                      # rotation about x by ang + 90 degrees: the light faces down into the box either way
        shape = ('<shape type="rectangle"><transform name="toWorld"><scale x="40" y="25"/><rotate x="1" angle="%g"/><translate x="200" y="420" z="250"/></transform>'
                 '%s<emitter type="area"><rgb name="radiance" value="20, 15, 10"/></emitter></shape>') % (ang + 90.0, '<boolean name="flipNormals" value="true"/>' if flipped else "")
        xml = open(XML).read().replace("</scene>", shape + "</scene>")
        xr = str(tmp_path / ("rect%d.xml" % flipped)); open(xr, "w").write(xml)
        dest = str(tmp_path / ("rect%d" % flipped))
        r = run(cli, "-o", dest, "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", xr)
        assert r.returncode == 0, r.stderr
        a = np.deg2rad(ang + 90.0)
        Rx = np.array([[1, 0, 0], [0, np.cos(a), -np.sin(a)], [0, np.sin(a), np.cos(a)]])
        L = Rx @ np.diag([40.0, 25.0, 1.0])
        T = np.concatenate([L, np.array([[200.0], [420.0], [250.0]])], 1)          # 3x4 toWorld
        M = T.copy()
        if flipped: M[:, 2] = -M[:, 2]                                              # toWorld * scale(1, 1, -1)
        n = np.linalg.inv(M[:, :3]).T @ np.array([0.0, 0.0, 1.0]); n /= np.linalg.norm(n)
        P = lambda x, y: T @ np.array([x, y, 0.0, 1.0])
        v = [P(-1, -1), P(1, -1), P(1, 1), P(-1, 1)]
        order = [[0, 1, 2], [2, 3, 0]]
        if flipped: order = [[0, 2, 1], [2, 0, 3]]                                   # the reader swaps the last two vertices of a flipped triangle
        sc = scenes.cornell_box(40, 30)
        first = sc.ntri
        q = [(0, 0), (1, 0), (1, 1), (0, 1)]
        sc.verts = np.concatenate([np.asarray(sc.verts, np.float64).reshape(-1, 9), np.array([[v[i] for i in o] for o in order]).reshape(2, 9)])
        gray = len(sc.materials); sc.materials = list(sc.materials) + [scenes.diffuse((0.5, 0.5, 0.5))]
        sc.tri_material = np.concatenate([np.asarray(sc.tri_material, np.int32), np.full(2, gray, np.int32)])
        uvs = np.zeros((first + 2, 6)); has = np.zeros(first + 2, np.uint8)
        for t, o in enumerate(order):
            uvs[first + t] = np.array([q[i] for i in o], float).reshape(6); has[first + t] = 1
        sc.uvs, sc.tri_has_uv = uvs, has
        sc.emitters = list(sc.emitters) + [(first, 2, (20.0, 15.0, 10.0), M, tuple(n))]
        out = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(sc), 4)
        for suffix in G.BUFFER_NAMES:
            assert np.allclose(read_pfm(dest + suffix + ".pfm"), out[suffix], rtol=2e-6, atol=1e-7), (flipped, suffix)


@pytest.mark.gpu
def test_cli_gbdpt_integrator_equals_python_mirror(cli, tmp_path, gpu_required):
    """`<integrator type="gbdpt">` through the C++ host (GBDPTIntegrator::render, gbdpt.cpp:140-262: seven MultiFilm buffers, both
    reconstructions) == the Python mirror over the same C-ABI; refused scopes carry their reason to the command line; a mirror in the scene renders."""
    import gradientdomain_mitsuba_amd.gpt as G
    import gradientdomain_mitsuba_amd.gbdpt as B
    xs = open(XML).read().replace('<integrator type="gpt">', '<integrator type="gbdpt">')
    xb = str(tmp_path / "bd.xml"); open(xb, "w").write(xs)
    import shutil; shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    dest = str(tmp_path / "bd")
    r = run(cli, "-o", dest, "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=6", xb)
    assert r.returncode == 0, r.stderr
    integ = B.GBDPTIntegrator(maxDepth=6)
    out = integ.render(G.Scene(scenes.cornell_box(40, 30)), 4)
    names = integ.outNames()
    assert names[0] == "-L1" and all(os.path.exists(dest + n + ".pfm") for n in names)
    for n in names:
        img = read_pfm(dest + n + ".pfm")
        ref = out[n].astype(np.float32)
        # (the sums of a pixel are fp64 atomics in free order: equal to rounding of the fp32 images and of the two solves fed by them)
        assert img.shape == (30, 40, 3) and np.allclose(img, ref, rtol=2e-4, atol=2e-6), n
    assert "Render time" in open(dest + "-log.txt").read()
    bad = run(cli, "-o", dest + "x", "-D", "width=16", "-D", "height=16", "-D", "spp=1", xb.replace("bd.xml", "bd.xml"), "-D", "maxDepth=21")       # (the records hold subpaths of up to 20 + 2 vertices since round 5: 12 until then)
    assert bad.returncode == 1 and "maxDepth" in bad.stderr
    xm = str(tmp_path / "mirror.xml")
    open(xm, "w").write(xs.replace("</scene>", '<shape type="rectangle"><transform name="toWorld"><scale value="50"/><translate x="270" y="200" z="300"/></transform><bsdf type="conductor"><rgb name="eta" value="1,1,1"/><rgb name="k" value="3,3,3"/></bsdf></shape></scene>'))
    # (a Dirac BSDF was refused until round 4; now such samples take the general form of the shift -- gbdpt_general.hip.h -- and the scene renders)
    ok = run(cli, "-o", dest + "m", "-D", "width=16", "-D", "height=16", "-D", "spp=1", xm)
    assert ok.returncode == 0, ok.stderr
    assert np.isfinite(read_pfm(dest + "m-L2.pfm")).all() and read_pfm(dest + "m-primal.pfm").max() > 0


@pytest.mark.gpu
def test_cli_thinlens_sensor_equals_python_mirror(cli, tmp_path, gpu_required):
    """`<sensor type="thinlens">` with apertureRadius / focusDistance through the scene reader == the Python mirror with the same lens; a zero aperture
    radius becomes Epsilon as in thinlens.cpp:134-138 (and then renders what the pinhole renders, up to the two extra random numbers per sample)."""
    import shutil
    import gradientdomain_mitsuba_amd.gpt as G
    shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    src = open(XML).read()
    assert '<sensor type="perspective">' in src
    xml = src.replace('<sensor type="perspective">', '<sensor type="thinlens"><float name="apertureRadius" value="30"/><float name="focusDistance" value="900"/>')
    xl = str(tmp_path / "lens.xml"); open(xl, "w").write(xml)
    dest = str(tmp_path / "lens")
    r = run(cli, "-o", dest, "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", xl)
    assert r.returncode == 0, r.stderr
    sc = scenes.cornell_box(40, 30); sc.thinlens = (30.0, 900.0)
    out = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(sc), 4)
    for suffix in G.BUFFER_NAMES:
        assert np.allclose(read_pfm(dest + suffix + ".pfm"), out[suffix], rtol=2e-6, atol=1e-7), suffix
    pin = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(scenes.cornell_box(40, 30)), 4)
    assert not np.allclose(out["-throughput"], pin["-throughput"], rtol=1e-3)
    xz = str(tmp_path / "lens0.xml"); open(xz, "w").write(xml.replace('value="30"', 'value="0"'))
    r = run(cli, "-o", str(tmp_path / "lens0"), "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", xz)
    assert r.returncode == 0, r.stderr
    sc0 = scenes.cornell_box(40, 30); sc0.thinlens = (1e-7, 900.0)
    out0 = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(sc0), 4)
    assert np.allclose(read_pfm(str(tmp_path / "lens0") + "-throughput.pfm"), out0["-throughput"], rtol=2e-6, atol=1e-7)


@pytest.mark.gpu
def test_cli_shutter_interval_equals_python_mirror(cli, tmp_path, gpu_required):
    """`shutterOpen` / `shutterClose` on the sensor through the scene reader (Sensor::Sensor, sensor.cpp:26-38) == the Python mirror with the same interval
    (every sample draws its time sample, gpt.cpp:1265-1267); a closing time before the opening time ends with the reference's message."""
    import shutil
    import gradientdomain_mitsuba_amd.gpt as G
    shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    src = open(XML).read()
    xml = src.replace('<sensor type="perspective">', '<sensor type="perspective"><float name="shutterOpen" value="0.25"/><float name="shutterClose" value="0.5"/>')
    xs = str(tmp_path / "shutter.xml"); open(xs, "w").write(xml)
    dest = str(tmp_path / "shutter")
    r = run(cli, "-o", dest, "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", xs)
    assert r.returncode == 0, r.stderr
    sc = scenes.cornell_box(40, 30); sc.shutter = (0.25, 0.5)
    out = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(sc), 4)
    for suffix in G.BUFFER_NAMES:
        assert np.allclose(read_pfm(dest + suffix + ".pfm"), out[suffix], rtol=2e-6, atol=1e-7), suffix
    still = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(scenes.cornell_box(40, 30)), 4)
    assert not np.allclose(out["-throughput"], still["-throughput"], rtol=1e-3)
    xb = str(tmp_path / "shutter_bad.xml"); open(xb, "w").write(xml.replace('value="0.5"', 'value="0.125"'))
    bad = run(cli, "-o", dest + "b", "-D", "width=16", "-D", "height=16", "-D", "spp=1", xb)
    assert bad.returncode == 1 and "Shutter opening time" in bad.stderr


@pytest.mark.gpu
def test_cli_crop_window_equals_python_mirror(cli, tmp_path, gpu_required):
    """`cropOffsetX/Y` + `cropWidth/Height` on the film through the scene reader (film.cpp:34-48; perspective.cpp:126-163) == the Python mirror with the same
    window: the written images have the crop's size; a window that leaves the film ends with the reference's message."""
    import shutil
    import gradientdomain_mitsuba_amd.gpt as G
    shutil.copytree(os.path.join(ROOT, "scenes", "meshes"), str(tmp_path / "meshes"))
    src = open(XML).read()
    assert '<film type="multifilm">' in src
    xml = src.replace('<film type="multifilm">', '<film type="multifilm"><integer name="cropOffsetX" value="9"/><integer name="cropOffsetY" value="5"/>'
                      '<integer name="cropWidth" value="24"/><integer name="cropHeight" value="16"/>')
    xc = str(tmp_path / "crop.xml"); open(xc, "w").write(xml)
    dest = str(tmp_path / "crop")
    r = run(cli, "-o", dest, "-D", "width=40", "-D", "height=30", "-D", "spp=4", "-D", "maxDepth=5", xc)
    assert r.returncode == 0, r.stderr
    sc = scenes.cornell_box(24, 16); sc.crop = (9, 5, 40, 30)
    out = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(sc), 4)
    for suffix in G.BUFFER_NAMES:
        img = read_pfm(dest + suffix + ".pfm")
        assert img.shape[:2] == (16, 24)
        assert np.allclose(img, out[suffix], rtol=2e-6, atol=1e-7), suffix
    whole = G.GradientPathIntegrator(maxDepth=5).render(G.Scene(scenes.cornell_box(24, 16)), 4)
    assert not np.allclose(out["-throughput"], whole["-throughput"], rtol=1e-3)
    xb = str(tmp_path / "crop_bad.xml"); open(xb, "w").write(xml.replace('name="cropOffsetX" value="9"', 'name="cropOffsetX" value="20"'))
    bad = run(cli, "-o", dest + "b", "-D", "width=40", "-D", "height=30", "-D", "spp=1", xb)
    assert bad.returncode == 1 and "Invalid crop window specification!" in bad.stderr
