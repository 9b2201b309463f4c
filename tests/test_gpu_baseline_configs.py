"""GPU tests that run the HIP path, through the C ABI, on the three BASELINE.json configurations the parity suite did
not reach at their full size (VERDICT r01 "configs_untested"):

  configs[1]  Cornell box 1920x1080, 64 spp, primary rays only
  configs[3]  Sponza 3840x2160, 256 spp, 8 bounces, image-tiled across 8 GPUs   -> one rank's share on this GPU
  configs[4]  Bistro 3840x2160, 64 spp, 8 GPUs (extended mode = the wavefront pipeline) -> one rank's share

Where the CPU oracle can reach, the comparison is bit for bit: the whole Cornell frame (12 triangles: brute force is
affordable), and crops of the two 4K frames (oracle_render_extended_region: a pixel's samples depend only on its
coordinates in the full frame).  Beyond that the size-independent properties: determinism, finiteness, segment
accounting, and that a pixel does not depend on how the tiles are partitioned (src/compute.rs:194-209 tile geometry).
Tolerance of the stated gate: |d rgb| <= 2e-3 for >= 99.9 % of pixels; asserted here: identical bits.
"""
import os
import sys
import zlib

import numpy as np
import pytest

from gpu_raytracer_amd import scenes
from gpu_raytracer_amd import types as T

pytestmark = pytest.mark.gpu

TILE = 32  # bench.py's tile size for the partitioned frames


def _owned_mask(w, h, tile, rank, world):
    tx, ty = (w + tile - 1) // tile, (h + tile - 1) // tile
    idx = np.arange(tx * ty).reshape(ty, tx)
    m = np.kron((idx % world == rank).astype(np.uint8), np.ones((tile, tile), np.uint8)).astype(bool)
    return m[:h, :w]


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ---------------------------------------------------------------------------------------------------------
# configs[1]: Cornell box 1920x1080 64 spp, primary rays only
# ---------------------------------------------------------------------------------------------------------
def test_cornell_1080p_reference_semantics_full_frame(gpu_ctx, oracle_mod):
    """Parity mode at the full size: the reference has no spp, every sample is the pixel-centre ray, so the
    64-spp image IS the single-sample image.  Whole frame, both reference modes, against the brute-force oracle."""
    scene = scenes.cornell12()
    w, h = 1920, 1080
    packed = oracle_mod.PackedScene(scene, use_bvh=False)
    gpu_ctx.upload_scene(scene)
    for mode in (0, 1):
        ref = oracle_mod.render_frame(packed, w, h, mode=mode)
        st = gpu_ctx.render(w, h, scene.camera, mode=mode, spp=64)
        prim, t = gpu_ctx.read_hits()
        np.testing.assert_array_equal(prim, ref["prim"])
        np.testing.assert_array_equal(_bits(t), _bits(ref["t"]))
        np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f()), _bits(ref["rgb"]))
        np.testing.assert_array_equal(gpu_ctx.read_rgba8_combined(), ref["combined"])
        assert st["rays"] == w * h


def test_cornell_1080p_64spp_primary_only_full_frame_vs_oracle(gpu_ctx, oracle_mod):
    """Extended mode, 64 jittered samples per pixel, 0 bounces ("primary rays only"): the whole 1920x1080 frame bit for
    bit against the CPU statement, with the shadow segments of the one light and without them."""
    scene = scenes.cornell12()
    w, h, spp = 1920, 1080, 64
    packed = oracle_mod.PackedScene(scene, use_bvh=False)
    gpu_ctx.upload_scene(scene)
    # primary rays only (no shadow segments): the whole frame
    ref = oracle_mod.render_extended(packed, w, h, spp, 0, flags=oracle_mod.EXT_NO_SHADOWS)
    st = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=0, no_shadows=True, tile_size=TILE)
    assert (st["primary_rays"], st["continuation_rays"], st["shadow_rays"]) == (w * h * spp, 0, 0) and ref["segments"]["camera"] == w * h * spp
    # primary rays only over a 12-triangle tree: the one-pass kernel (samples in registers, one store per pixel) by rule, round 3
    assert st["flags"] & T.STAT_SINGLE_PASS and not st["flags"] & T.STAT_MEGAKERNEL_FALLBACK
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f()), _bits(ref["rgb"]))
    # ... and the queue pipeline, which RT_FLAG_KERNEL_PIPELINE keeps, gives the same frame
    st = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=0, no_shadows=True, tile_size=TILE, kernel_pipeline=True)
    assert not st["flags"] & T.STAT_SINGLE_PASS and (st["primary_rays"], st["shadow_rays"]) == (w * h * spp, 0)
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f()), _bits(ref["rgb"]))
    # with the shadow segment toward the one light: the centre 640x360 of the frame on the CPU, the whole frame on the GPU
    x0, y0, rw, rh = 640, 360, 640, 360
    ref = oracle_mod.render_extended(packed, w, h, spp, 0, region=(x0, y0, rw, rh))
    frames = []
    for pipeline in (False, True):
        st = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=0, tile_size=TILE, kernel_pipeline=pipeline)
        assert bool(st["flags"] & T.STAT_SINGLE_PASS) == (not pipeline)
        assert st["primary_rays"] == w * h * spp and 0 < st["shadow_rays"] < st["primary_rays"]
        frames.append((_bits(gpu_ctx.read_rgb32f()).copy(), st["shadow_rays"]))
        np.testing.assert_array_equal(frames[-1][0][y0:y0 + rh, x0:x0 + rw], _bits(ref["rgb"]))
    np.testing.assert_array_equal(frames[0][0], frames[1][0])
    assert frames[0][1] == frames[1][1]


# ---------------------------------------------------------------------------------------------------------
# the headline configuration itself (BASELINE.json metric; configs[2]'s scene at 64 spp): sponza-like 1920x1080, 64 spp, 4 bounces
# ---------------------------------------------------------------------------------------------------------
def test_sponza_1080p_64spp_4_bounces_headline_frame_vs_oracle(gpu_ctx, oracle_mod):
    """The frame every bench.py number is quoted on, rendered as bench.py renders it (32-pixel tiles): its CRC is the one bench.py
    prints and expects, and crops of it - a corner at each end, the image centre, the neighbourhood of the grazed-edge pixel - carry
    the CPU statement's bits (oracle_render_extended_region: a pixel's samples depend only on its coordinates in the full frame).
    Pixel (844, 563) is the one place where the reference's own two paths disagree (tests/test_oracle_extended.py): the HIP path
    gives its brute-force value there, the reference-format BVH walk of the oracle the other one."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    scene = scenes.sponza_like()
    w, h, spp, bounces = 1920, 1080, 64, 4
    gpu_ctx.upload_scene(scene)
    st = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=TILE)
    a = gpu_ctx.read_rgb32f()
    assert st["primary_rays"] == w * h * spp and st["rays"] == st["primary_rays"] + st["continuation_rays"] + st["shadow_rays"]
    assert zlib.crc32(a.tobytes()) & 0xFFFFFFFF == bench.HEADLINE_FRAME_CRC == 4012668657
    packed = oracle_mod.PackedScene(scene)  # the reference-format chunked BVH (> 100k triangles), walked as shader/src/bvh.rs does
    for (x0, y0, rw, rh) in ((0, 0, 6, 4), (957, 538, 6, 4), (w - 6, h - 4, 6, 4), (842, 562, 5, 3)):
        ref = oracle_mod.render_extended(packed, w, h, spp, bounces, region=(x0, y0, rw, rh))["rgb"]
        got = a[y0:y0 + rh, x0:x0 + rw]
        differ = np.argwhere((_bits(got) != _bits(ref)).any(axis=-1))
        expected = [[563 - y0, 844 - x0]] if (x0 <= 844 < x0 + rw and y0 <= 563 < y0 + rh) else []
        assert differ.tolist() == expected, (x0, y0, differ.tolist())
    # the grazed-edge pixel: the brute-force path's value (pinned against the oracle's brute-force walk in tests/test_oracle_extended.py)
    assert _bits(a[563, 844]).tolist() == [1066366301, 1057316370, 1044521191]


# ---------------------------------------------------------------------------------------------------------
# configs[3]: Sponza 3840x2160 256 spp 8 bounces across 8 GPUs -> rank 0's share
# ---------------------------------------------------------------------------------------------------------
def test_sponza_4k_256spp_8_bounces_rank_share(gpu_ctx, oracle_mod):
    scene = scenes.sponza_like()
    w, h, spp, bounces = 3840, 2160, 256, 8
    gpu_ctx.upload_scene(scene)
    st = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=TILE, tile_rank=0, tile_world=8)
    a = gpu_ctx.read_rgb32f()
    own = _owned_mask(w, h, TILE, 0, 8)
    # segment accounting: every owned pixel starts spp paths; a path has at most `bounces` continuation segments
    assert st["pixels"] == own.sum() and st["primary_rays"] == int(own.sum()) * spp
    assert st["rays"] == st["primary_rays"] + st["continuation_rays"] + st["shadow_rays"]
    assert st["primary_rays"] < st["continuation_rays"] <= bounces * st["primary_rays"] and st["shadow_rays"] > st["primary_rays"]
    assert np.isfinite(a).all() and (a >= 0).all() and (a[own].max(-1) > 0).mean() > 0.99
    # determinism
    st2 = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=TILE, tile_rank=0, tile_world=8)
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f())[own], _bits(a)[own])
    assert st2["rays"] == st["rays"]
    # a crop against the CPU statement (which walks the reference-format chunked BVH): two owned 32x32 tiles' corners
    packed = oracle_mod.PackedScene(scene)
    for (x0, y0) in ((0, 0), (8 * TILE * 7, TILE * 33)):  # tile (0,0) and tile column 56 of row 33: 33*120+56 = 4016 = 8*502
        assert own[y0, x0]
        ref = oracle_mod.render_extended(packed, w, h, spp, bounces, region=(x0, y0, 6, 4))
        np.testing.assert_array_equal(_bits(a[y0:y0 + 4, x0:x0 + 6]), _bits(ref["rgb"]))
    # partition invariance: the same pixels out of a 3-rank partition of the frame (another tile -> rank map)
    gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=TILE, tile_rank=0, tile_world=3)
    both = own & _owned_mask(w, h, TILE, 0, 3)
    assert both.any()
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f())[both], _bits(a)[both])


def test_sponza_small_frame_8_bounces_vs_oracle(gpu_ctx, oracle_mod):
    """The 8-bounce setting of configs[3] on a frame the CPU statement finishes in seconds: segment counts and bits."""
    scene = scenes.sponza_like()
    w, h, spp, bounces = 48, 27, 4, 8
    ref = oracle_mod.render_extended(oracle_mod.PackedScene(scene), w, h, spp, bounces)
    gpu_ctx.upload_scene(scene)
    st = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces)
    seg = ref["segments"]
    assert (st["primary_rays"], st["continuation_rays"], st["shadow_rays"]) == (seg["camera"], seg["continuation"], seg["shadow"])
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f()), _bits(ref["rgb"]))


# ---------------------------------------------------------------------------------------------------------
# configs[4]: Bistro 3840x2160 64 spp across 8 GPUs, extended mode (the wavefront pipeline) -> rank 0's share
# ---------------------------------------------------------------------------------------------------------
def test_bistro_4k_64spp_wavefront_rank_share_and_small_frame(gpu_ctx, oracle_mod):
    scene = scenes.bistro_like()
    gpu_ctx.upload_scene(scene)
    packed = oracle_mod.PackedScene(scene)
    # small frame in mode 2 against the CPU statement (380-triangle mesh-order leaves: the slow, faithful traversal)
    sw, sh, sspp, bounces = 32, 18, 2, 8
    ref = oracle_mod.render_extended(packed, sw, sh, sspp, bounces)
    st = gpu_ctx.render(sw, sh, scene.camera, mode=2, spp=sspp, max_bounces=bounces)
    seg = ref["segments"]
    assert (st["primary_rays"], st["continuation_rays"], st["shadow_rays"]) == (seg["camera"], seg["continuation"], seg["shadow"])
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f()), _bits(ref["rgb"]))
    # rank 0 of 8's share of the 4K 64-spp frame
    w, h, spp = 3840, 2160, 64
    st = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=TILE, tile_rank=0, tile_world=8)
    a = gpu_ctx.read_rgb32f()
    own = _owned_mask(w, h, TILE, 0, 8)
    assert st["pixels"] == own.sum() and st["primary_rays"] == int(own.sum()) * spp
    assert st["rays"] == st["primary_rays"] + st["continuation_rays"] + st["shadow_rays"]
    assert 0 < st["continuation_rays"] <= bounces * st["primary_rays"]
    assert np.isfinite(a).all() and (a >= 0).all()
    st2 = gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=TILE, tile_rank=0, tile_world=8)
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f())[own], _bits(a)[own])
    assert st2["rays"] == st["rays"]
    # a crop of the 4K frame against the CPU statement
    ref = oracle_mod.render_extended(packed, w, h, spp, bounces, region=(0, 0, 4, 2))
    np.testing.assert_array_equal(_bits(a[0:2, 0:4]), _bits(ref["rgb"]))
    # partition invariance against the unpartitioned frame at a lower sample count is not possible (samples differ with
    # spp), so: the same share out of a 2-rank partition on the tiles both partitions give to rank 0
    gpu_ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=TILE, tile_rank=0, tile_world=2)
    both = own & _owned_mask(w, h, TILE, 0, 2)
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f())[both], _bits(a)[both])
