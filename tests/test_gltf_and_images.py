"""Rows N1 / N3 / N4 of SURVEY.md §8f: the C++ glTF loader with the behaviour of src/gltf_loader.rs, the image
writers, and the progressive completion summary.  Fixtures: tests/golden/*.gltf|glb (authored for this repository
by tests/golden/make_gltf_fixtures.py; the reference's tests hold no glTF file)."""
import os
import struct
import zlib

import numpy as np
import pytest

from gpu_raytracer_amd import host, hostpack as H, scenes, types as T

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _tri_pos(sc):
    v = sc.vertices["position"]
    return np.stack([v[sc.triangles[k]] for k in ("v0_index", "v1_index", "v2_index")], 1)


@pytest.mark.parametrize("loader", ["gltf", "glb"])
def test_cornell12_fixture_equals_the_harness_scene(loader):
    s = host.load_gltf(os.path.join(GOLD, "cornell12.gltf")) if loader == "gltf" else host.load_glb(open(os.path.join(GOLD, "cornell12.glb"), "rb").read())
    r = scenes.cornell12()
    assert s.n_triangles == 12 and len(s.spheres) == 0
    # per-primitive vertex deduplication (src/gltf_loader.rs:307-330): 6 quads x 4 corners, shared corners are NOT merged across primitives
    assert len(s.vertices) == 24
    assert s.materials.tobytes() == r.materials.tobytes()
    assert s.lights.tobytes() == r.lights.tobytes()          # point light at (0,0.9,0), range inf -> f16 inf
    assert s.camera.tobytes() == r.camera.tobytes()          # yfov pi/4 rad -> 45 degrees, camera looks down -Z
    np.testing.assert_array_equal(_tri_pos(s), _tri_pos(r))
    np.testing.assert_array_equal(s.triangles["material_id"], r.triangles["material_id"])


def test_cornell12_gltf_renders_like_the_harness_scene_on_the_oracle(oracle_mod):
    s = host.load_gltf(os.path.join(GOLD, "cornell12.gltf"))
    a = oracle_mod.render_frame(oracle_mod.PackedScene(s), 96, 96)
    b = oracle_mod.render_frame(oracle_mod.PackedScene(scenes.cornell12()), 96, 96)
    np.testing.assert_array_equal(a["rgb"].view(np.uint32), b["rgb"].view(np.uint32))
    np.testing.assert_array_equal(a["prim"], b["prim"])


def test_modes_fixture_topologies_indices_transforms_and_materials():
    s = host.load_gltf(os.path.join(GOLD, "modes.gltf"))
    # strip 4 + fan 3 + three indexed quads (u8, u16, u32 with a dangling 7th index) 6 + non-indexed soup 1; LINES skipped
    assert s.n_triangles == 14
    tp = _tri_pos(s)
    # node chain: root(scale 1) -> node0 (T(1,2,3) * R(90deg about Y) * S(2,1,0.5)); local (x,y,z) -> (1 + 0.5 z, 2 + y, 3 - 2 x)
    def xf(p):
        p = np.asarray(p, np.float64)
        return np.stack([1 + 0.5 * p[..., 2], 2 + p[..., 1], 3 - 2 * p[..., 0]], -1)
    strip = np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0], [1, 1, 0], [2, 0, 0], [2, 1, 0]], np.float64)
    want_strip = [xf(strip[[0, 1, 2]]), xf(strip[[1, 3, 2]]), xf(strip[[2, 3, 4]]), xf(strip[[3, 5, 4]])]  # odd triangles swap v1/v2
    np.testing.assert_allclose(tp[:4], want_strip, atol=1e-5)
    fan = np.array([[0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1.5, 1], [-1, 1, 1]], np.float64)
    np.testing.assert_allclose(tp[4:7], [xf(fan[[0, 1, 2]]), xf(fan[[0, 2, 3]]), xf(fan[[0, 3, 4]])], atol=1e-5)
    quad = np.array([[0, 0, 2], [1, 0, 2], [1, 1, 2], [0, 1, 2]], np.float64)
    for k in (7, 9, 11):
        np.testing.assert_allclose(tp[k:k + 2], [xf(quad[[0, 1, 2]]), xf(quad[[0, 2, 3]])], atol=1e-5)
    np.testing.assert_allclose(tp[13], xf(np.array([[0, 0, 3], [1, 0, 3], [0, 1, 3]])), atol=1e-5)
    # material ids: 0,1,2,3, none -> 0, out-of-range 99 -> 0 (material_map lookup fails, unwrap_or(0), :296-298)
    assert s.triangles["material_id"].tolist() == [0] * 4 + [1] * 3 + [2] * 2 + [3] * 2 + [0] * 2 + [0]
    # vertex dedup is per primitive: the three quad primitives each add their own 4 vertices
    assert len(s.vertices) == 6 + 5 + 4 + 4 + 4 + 3
    m0, m1, m2, m3 = s.materials
    assert m0.tobytes()[:12] == np.array([0.2, 0.4, 0.6], np.float32).tobytes()
    want0 = H.material_new((0.2, 0.4, 0.6), 0.75, 0.25, (0.1, 0.2, 0.3), 1.33, 0.6)
    assert int(m0["metallic_roughness_f16"]) == int(want0["metallic_roughness_f16"]) and int(m0["ior_transmission_f16"]) == int(want0["ior_transmission_f16"])
    np.testing.assert_array_equal(m0["emission"], np.array([0.1, 0.2, 0.3], np.float32))
    assert m0["specular_factor"] == np.float32(0.5) and m0["thickness_factor"] == np.float32(0.2) and m0["attenuation_distance"] == np.float32(4.0)
    np.testing.assert_array_equal(m0["attenuation_color"], np.array([0.5, 0.6, 0.7], np.float32))
    assert m1["material_type"] == 1 and m1["glossiness_factor"] == np.float32(0.7)  # Material::specular_glossiness
    np.testing.assert_array_equal(m1["specular_color"], np.array([0.4, 0.5, 0.6], np.float32))
    assert int(m1["metallic_roughness_f16"]) >> 16 == H.f16_bits(np.float32(1.0) - np.float32(0.7))
    assert m2.tobytes() == H.material_new((1, 1, 1), 1.0, 1.0, (0, 0, 0), 1.5, 0.0).tobytes()  # glTF defaults
    assert m3["texture_indices"].tolist() == [2, 0, 1, 3] + [0xFFFFFFFF] * 4
    # lights: spot under node2 (translated by (0,3,0) in node0's frame), directional under it rotated 45 deg about X
    assert s.lights["light_type"].tolist() == [2, 0]
    np.testing.assert_allclose(s.lights[0]["position"], xf(np.array([0, 3, 0])), atol=1e-5)
    assert s.lights[0]["intensity"] == 3.0 and (int(s.lights[0]["range_packed"]) & 0xFFFF) == H.f16_bits(25.0)
    assert int(s.lights[0]["cone_angles_packed"]) == H.f16_bits(0.2) | (H.f16_bits(0.6) << 16)
    assert s.lights[1]["intensity"] == 2.0 and np.allclose(s.lights[1]["color"], [1.0, 0.9, 0.8])
    np.testing.assert_allclose(np.linalg.norm(s.lights[1]["direction"]), 1.0, atol=1e-6)
    # orthographic camera -> fov 45 (:243-245); direction / up are normalised although the node scales
    cam = s.meta["cameras"][0]
    assert cam["fov"] == 45.0
    np.testing.assert_allclose(np.linalg.norm(cam["direction"]), 1.0, atol=1e-6)
    np.testing.assert_allclose(cam["position"], xf(np.array([0.5, 0.25, 4.0])), atol=1e-5)
    # scene selection: index 1 roots at node 0 directly (same geometry since node 3 is an identity wrapper)
    assert host.load_gltf(os.path.join(GOLD, "modes.gltf"), 1).n_triangles == 14
    with pytest.raises(host.GltfError, match="Scene 7 not found"):
        host.load_gltf(os.path.join(GOLD, "modes.gltf"), 7)


def test_loader_errors(tmp_path):
    with pytest.raises(host.GltfError, match="IoError"):
        host.load_gltf(str(tmp_path / "missing.gltf"))
    bad = tmp_path / "bad.gltf"
    bad.write_text("{ not json")
    with pytest.raises(host.GltfError, match="GltfError"):
        host.load_gltf(str(bad))
    nopos = tmp_path / "nopos.gltf"
    nopos.write_text('{"asset":{"version":"2.0"},"scenes":[{"nodes":[0]}],"nodes":[{"mesh":0}],"meshes":[{"primitives":[{"attributes":{}}]}]}')
    with pytest.raises(host.GltfError, match="Primitive missing position data"):
        host.load_gltf(str(nopos))
    noscene = tmp_path / "noscene.gltf"
    noscene.write_text('{"asset":{"version":"2.0"}}')
    with pytest.raises(host.GltfError, match="No scenes found"):
        host.load_gltf(str(noscene))
    with pytest.raises(host.GltfError):
        host.load_glb(b"glTF" + struct.pack("<II", 1, 12))


def test_image_writers_round_trip(tmp_path):
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (37, 53, 4), dtype=np.uint8)
    host.write_image(tmp_path / "a.ppm", img)
    raw = (tmp_path / "a.ppm").read_bytes()
    assert raw.startswith(b"P6\n53 37\n255\n")
    np.testing.assert_array_equal(np.frombuffer(raw[len(b"P6\n53 37\n255\n"):], np.uint8).reshape(37, 53, 3), img[..., :3])
    host.write_image(tmp_path / "a.png", img)
    png = (tmp_path / "a.png").read_bytes()
    assert png[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, hdr = 8, b"", None
    while pos < len(png):
        (n,), typ = struct.unpack(">I", png[pos:pos + 4]), png[pos + 4:pos + 8]
        data = png[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", png[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data)
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        if typ == b"IDAT":
            idat += data
        pos += 12 + n
    assert hdr == (53, 37, 8, 6, 0, 0, 0)
    rows = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(37, 1 + 53 * 4)
    assert (rows[:, 0] == 0).all()
    np.testing.assert_array_equal(rows[:, 1:].reshape(37, 53, 4), img)


def _read_exr(raw):
    """Minimal reader for what write_exr emits, following the OpenEXR file layout: magic, version, attributes until an
    empty name, the line offset table, one uncompressed block per scanline with its channels stored as planes in
    alphabetical order.  Returns (attributes, H x W x 3 float32 in R G B order)."""
    assert raw[:4] == bytes([0x76, 0x2F, 0x31, 0x01]) and struct.unpack("<I", raw[4:8])[0] == 2
    pos, attrs = 8, {}
    while raw[pos] != 0:
        end = raw.index(b"\0", pos)
        name = raw[pos:end].decode()
        pos = end + 1
        end = raw.index(b"\0", pos)
        typ = raw[pos:end].decode()
        pos = end + 1
        (n,) = struct.unpack("<I", raw[pos:pos + 4])
        attrs[name] = (typ, raw[pos + 4:pos + 4 + n])
        pos += 4 + n
    pos += 1
    assert attrs["compression"] == ("compression", b"\0") and attrs["lineOrder"] == ("lineOrder", b"\0")
    x0, y0, x1, y1 = struct.unpack("<4i", attrs["dataWindow"][1])
    assert attrs["dataWindow"][0] == "box2i" and attrs["displayWindow"] == attrs["dataWindow"] and (x0, y0) == (0, 0)
    w, h = x1 + 1, y1 + 1
    chl, names, q = attrs["channels"][1], [], 0
    assert attrs["channels"][0] == "chlist"
    while chl[q] != 0:
        end = chl.index(b"\0", q)
        names.append(chl[q:end].decode())
        pixel_type, p_linear, xs, ys = struct.unpack("<iB3xii", chl[end + 1:end + 17])
        assert (pixel_type, xs, ys) == (2, 1, 1)  # FLOAT, no subsampling
        q = end + 17
    assert names == sorted(names) == ["B", "G", "R"] and q == len(chl) - 1
    offsets = struct.unpack(f"<{h}Q", raw[pos:pos + 8 * h])
    img = np.zeros((h, w, 3), np.float32)
    for y in range(h):
        o = offsets[y]
        yy, nbytes = struct.unpack("<iI", raw[o:o + 8])
        assert yy == y and nbytes == 12 * w
        planes = np.frombuffer(raw[o + 8:o + 8 + nbytes], "<f4").reshape(3, w)
        img[y, :, 2], img[y, :, 1], img[y, :, 0] = planes[0], planes[1], planes[2]
    assert offsets[-1] + 8 + 12 * w == len(raw)
    return attrs, img


def test_exr_writer_keeps_every_bit(tmp_path):
    """The float image goes to OpenEXR (scanline, uncompressed, FLOAT) unchanged - NaN payloads, infinities, denormals
    and negative zero included - so float parity diffs can be inspected outside this repository."""
    rng = np.random.default_rng(8)
    img = rng.standard_normal((23, 41, 3)).astype(np.float32)
    img.view(np.uint32)[0, :6, 0] = [0x7FC00001, 0xFF800000, 0x7F800000, 0x00000001, 0x80000000, 0x7F7FFFFF]
    host.write_image(tmp_path / "a.exr", img)
    attrs, back = _read_exr((tmp_path / "a.exr").read_bytes())
    np.testing.assert_array_equal(back.view(np.uint32), img.view(np.uint32))
    assert struct.unpack("<f", attrs["pixelAspectRatio"][1])[0] == 1.0 and attrs["screenWindowCenter"][0] == "v2f"
    host.write_image(tmp_path / "one.exr", img[:1, :1])
    assert np.array_equal(_read_exr((tmp_path / "one.exr").read_bytes())[1].view(np.uint32), img[:1, :1].view(np.uint32))
    with pytest.raises(ValueError):
        host.write_image(tmp_path / "bad.exr", np.zeros((4, 4, 4), np.float32))
    with pytest.raises(OSError):
        host.write_image(tmp_path / "no_such_dir" / "a.exr", img)


@pytest.mark.gpu
def test_gltf_scene_through_the_whole_drop_in_path_on_gpu(gpu_ctx, oracle_mod, tmp_path):
    """glTF file -> GltfLoader -> BvhBuilder -> BufferManager -> ComputeRenderer::run_compute (rt_dispatch_tile per tile
    and channel) -> main_fs combine -> PNG; equals the oracle's frame of the same scene."""
    scene = host.load_gltf(os.path.join(GOLD, "cornell12.gltf"))
    w, h = 640, 360
    n_dispatches, n_calls = host.render_progressive(gpu_ctx, scene, w, h)
    assert n_dispatches == 5 * 3 * 3 and n_calls == 1  # 15 tiles <= 16: all in one run_compute call (TileHelper)
    comb = gpu_ctx.read_rgba8_combined()
    ref = oracle_mod.render_frame(oracle_mod.PackedScene(scene, use_bvh=False), w, h)
    np.testing.assert_array_equal(comb, ref["combined"])
    t = host.progressive_timing()
    assert t["calls"] == 1 and t["tiles"] == 15 and t["total_ms"] > 0 and t["p50_ms"] <= t["p99_ms"] and t["tiles_per_s"] > 0
    host.write_image(tmp_path / "cornell.png", comb)
    assert (tmp_path / "cornell.png").stat().st_size > w * h * 4


def _rt_render_binary():
    from gpu_raytracer_amd import build as rt_build
    return rt_build.build_examples(verbose=False)


def test_native_front_end_builds_and_fails_loudly_without_a_device():
    """examples/rt_render.cpp (C++ over the C ABI and the host mirror) builds with g++ alone; on a box without a HIP
    device it must stop with the library's error, not render on the CPU."""
    import subprocess
    exe = _rt_render_binary()
    out = subprocess.run([exe, "--help"], capture_output=True, text=True)
    assert out.returncode == 0 and "--gltf" in out.stdout
    out = subprocess.run([exe, "--bogus"], capture_output=True, text=True)
    assert out.returncode == 2
    import torch
    if not torch.cuda.is_available():
        out = subprocess.run([exe, "--size", "32x32", "--out", os.devnull], capture_output=True, text=True)
        assert out.returncode == 4 and "no CPU fallback" in out.stderr


@pytest.mark.gpu
def test_native_front_end_renders_the_gltf_fixture_like_the_oracle(oracle_mod, tmp_path):
    """The same drop-in path as above, driven by the native C++ front end: its PPM equals the oracle's combined frame;
    and the extended mode through the same binary equals the Python-driven render."""
    import subprocess
    exe = _rt_render_binary()
    w, h = 300, 200
    ppm = tmp_path / "cornell.ppm"
    out = subprocess.run([exe, "--gltf", os.path.join(GOLD, "cornell12.gltf"), "--size", f"{w}x{h}", "--out", str(ppm)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "progressive rendering complete: 6 tiles (3x2) in 1 calls" in out.stdout
    raw = ppm.read_bytes()
    header = f"P6\n{w} {h}\n255\n".encode()
    assert raw.startswith(header)
    img = np.frombuffer(raw[len(header):], np.uint8).reshape(h, w, 3)
    scene = host.load_gltf(os.path.join(GOLD, "cornell12.gltf"))
    ref = oracle_mod.render_frame(oracle_mod.PackedScene(scene, use_bvh=False), w, h)
    np.testing.assert_array_equal(img, ref["combined"][..., :3])
    fly = subprocess.run([exe, "--fly", "5", "--size", "160x120", "--out", str(tmp_path / "fly.ppm")], capture_output=True, text=True)
    assert fly.returncode == 0 and "fly-through: 5 frames of 160x120" in fly.stdout, fly.stderr
    # the same five camera steps through the Python mirror: identical last frame
    cam = scenes.default_scene().camera
    for f in range(5):
        cam = host.camera_move(host.camera_rotate(cam, 1.0, -0.5), 0.05, -0.05)
    ref5 = oracle_mod.render_frame(oracle_mod.PackedScene(scenes.default_scene(), use_bvh=False), 160, 120, camera=cam)
    raw5 = (tmp_path / "fly.ppm").read_bytes()
    np.testing.assert_array_equal(np.frombuffer(raw5[len(b"P6\n160 120\n255\n"):], np.uint8).reshape(120, 160, 3), ref5["combined"][..., :3])
    ppm2 = tmp_path / "ext.ppm"
    out = subprocess.run([exe, "--size", "160x120", "--spp", "4", "--bounces", "2", "--out", str(ppm2)], capture_output=True, text=True)
    assert out.returncode == 0 and "extended mode: 4 spp, 2 bounces" in out.stdout, out.stderr
    ext = oracle_mod.render_extended(oracle_mod.PackedScene(scenes.default_scene(), use_bvh=False), 160, 120, 4, 2)
    raw2 = ppm2.read_bytes()
    img2 = np.frombuffer(raw2[len(b"P6\n160 120\n255\n"):], np.uint8).reshape(120, 160, 3)
    exp = np.clip(np.floor(np.clip(ext["rgb"], 0, 1) * 255.0 + 0.5), 0, 255).astype(np.uint8)
    np.testing.assert_array_equal(img2, exp)
    exr = tmp_path / "ext.exr"
    out = subprocess.run([exe, "--size", "160x120", "--spp", "4", "--bounces", "2", "--out", str(exr)], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    np.testing.assert_array_equal(_read_exr(exr.read_bytes())[1].view(np.uint32), ext["rgb"].view(np.uint32))


def test_loader_survives_mutated_files_under_asan_ubsan(tmp_path):
    """The loader parses untrusted files: 600 deterministic mutations of the fixtures (byte flips, deletions,
    duplications, truncation, hostile numbers, stray JSON punctuation) through an AddressSanitizer + UBSan build of the
    header-only loader (tests/fuzz_gltf.cpp).  Every input must be either loaded or rejected with a GltfError."""
    import random
    import re
    import shutil
    import subprocess
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    exe = tmp_path / "fuzz_gltf"
    cc = subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-I" + os.path.join(root, "include"),
                         "-I" + os.path.join(root, "gpu_raytracer_amd", "csrc"), os.path.join(here, "fuzz_gltf.cpp"), "-o", str(exe)],
                        capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    rnd = random.Random(20240607)

    def mutate(b):
        b = bytearray(b)
        if len(b) < 4:
            return bytes(b)
        k = rnd.choice(["flip", "del", "dup", "num", "trunc", "ins"])
        if k == "flip":
            for _ in range(rnd.randint(1, 8)):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif k == "del":
            i = rnd.randrange(len(b))
            del b[i:min(len(b), i + rnd.randint(1, 64))]
        elif k == "dup":
            i = rnd.randrange(len(b))
            b[i:i] = b[i:min(len(b), i + rnd.randint(1, 64))]
        elif k == "num":
            s = bytes(b)
            ms = list(re.finditer(rb"\d+", s))
            if ms:
                m = rnd.choice(ms)
                b = bytearray(s[:m.start()] + rnd.choice([b"0", b"-1", b"4294967295", b"99999999999999999999", b"1e309", b"2147483648", b"65536", b"3"]) + s[m.end():])
        elif k == "trunc":
            b = b[:rnd.randrange(len(b))]
        else:
            i = rnd.randrange(len(b))
            b[i:i] = rnd.choice([b"{", b"}", b"[", b"]", b'"', b",", b":", b"\\", b"\x00", b"null", b"true"])
        return bytes(b)

    shutil.copy(os.path.join(GOLD, "modes.bin"), tmp_path / "modes.bin")
    files = []
    for name in ("cornell12.gltf", "cornell12.glb", "modes.gltf"):
        data = open(os.path.join(GOLD, name), "rb").read()
        stem, ext = name.split(".")
        for i in range(200):
            d = data
            for _ in range(rnd.randint(1, 3)):
                d = mutate(d)
            p = tmp_path / f"{stem}_{i}.{ext}"
            p.write_bytes(d)
            files.append(str(p))
    env = dict(os.environ, UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", ASAN_OPTIONS="detect_leaks=1")
    run = subprocess.run([str(exe)] + files, capture_output=True, text=True, env=env, timeout=600)
    assert run.returncode == 0, run.stderr[-4000:]
    m = re.search(r"loaded (\d+), rejected (\d+)", run.stdout)
    assert m and int(m.group(1)) + int(m.group(2)) == len(files) and int(m.group(2)) > 100 and int(m.group(1)) > 5


def _scene_to_glb(scene):
    """A GLB of `scene`'s triangles: one primitive per material, u32 indices into one shared POSITION accessor per
    primitive, diffuse metallic-roughness materials with the scene's albedo (written here, for the test, with the same
    conventions as make_gltf_fixtures.py)."""
    import json
    v = scene.vertices["position"]
    blob, views, accessors, prims, mats = b"", [], [], [], []
    order = []
    for m in range(len(scene.materials)):
        sel = np.nonzero(scene.triangles["material_id"] == m)[0]
        if len(sel) == 0:
            continue
        order.append(sel)
        corners = np.stack([scene.triangles[k][sel] for k in ("v0_index", "v1_index", "v2_index")], 1).reshape(-1)
        uniq, inv = np.unique(corners, return_inverse=True)
        pos = np.ascontiguousarray(v[uniq], np.float32)
        idx = inv.astype(np.uint32)
        for arr, comp, typ in ((pos, 5126, "VEC3"), (idx, 5125, "SCALAR")):
            views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": arr.nbytes})
            blob += arr.tobytes() + b"\x00" * (-arr.nbytes % 4)
            acc = {"bufferView": len(views) - 1, "componentType": comp, "count": int(arr.shape[0]), "type": typ}
            if typ == "VEC3":
                acc["min"], acc["max"] = pos.min(0).tolist(), pos.max(0).tolist()
            accessors.append(acc)
        prims.append({"attributes": {"POSITION": len(accessors) - 2}, "indices": len(accessors) - 1, "material": len(mats), "mode": 4})
        a = scene.materials[m]["albedo"]
        mats.append({"pbrMetallicRoughness": {"baseColorFactor": [float(a[0]), float(a[1]), float(a[2]), 1.0], "metallicFactor": 0.0, "roughnessFactor": 1.0}})
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": [{"mesh": 0}], "meshes": [{"primitives": prims}],
           "materials": mats, "accessors": accessors, "bufferViews": views, "buffers": [{"byteLength": len(blob)}]}
    js = json.dumps(doc, separators=(",", ":")).encode()
    js += b" " * (-len(js) % 4)
    return b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(blob)) + struct.pack("<II", len(js), 0x4E4F534A) + js + \
        struct.pack("<II", len(blob), 0x004E4942) + blob, np.concatenate(order)


def test_loader_at_scale_round_trips_a_60k_triangle_scene():
    """u32 indices, megabytes of binary chunk, tens of thousands of deduplicated vertices: the loaded triangles are the
    exported ones, in primitive order."""
    scene = scenes.random_soup(60000, seed=8, size=0.2, n_materials=5)
    glb, order = _scene_to_glb(scene)
    s = host.load_glb(glb)
    assert s.n_triangles == 60000 and len(s.materials) == 5
    np.testing.assert_array_equal(_tri_pos(s), _tri_pos(scene)[order])
    np.testing.assert_array_equal(s.triangles["material_id"], scene.triangles["material_id"][order])
    for m in range(5):
        np.testing.assert_array_equal(s.materials[m]["albedo"], scene.materials[m]["albedo"])


@pytest.mark.gpu
def test_loaded_60k_scene_renders_like_the_original_geometry(gpu_ctx, oracle_mod):
    """The same file through the whole path on the GPU: glTF -> loader -> upload -> render equals the oracle's frame of the
    loaded scene (bit-exact hit distances; primitive ids only differ at equal-t ties, see test_gpu_parity)."""
    scene = scenes.random_soup(60000, seed=8, size=0.2, n_materials=5)
    glb, _ = _scene_to_glb(scene)
    s = host.load_glb(glb)
    s = type(scene)(s.name, s.spheres, scene.lights, s.vertices, s.triangles, s.materials, scene.camera)  # the file carries no lights / camera
    w, h = 160, 100
    ref = oracle_mod.render_frame(oracle_mod.PackedScene(s), w, h)
    gpu_ctx.upload_scene(s)
    gpu_ctx.render(w, h, s.camera, mode=0)
    prim, t = gpu_ctx.read_hits()
    np.testing.assert_array_equal(t.view(np.uint32), ref["t"].view(np.uint32))
    assert (prim != ref["prim"]).mean() <= 1e-3
    same = prim == ref["prim"]
    np.testing.assert_array_equal(gpu_ctx.read_rgb32f()[same].view(np.uint32), ref["rgb"][same].view(np.uint32))
