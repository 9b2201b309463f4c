"""Camera beams (round 3; wavefront.hip k_wf_beams / k_wf_trace_camera): the camera segments of an 8x8 pixel block test the triangles
its pyramid touches directly instead of walking the tree.  Which triangle is hit must not change: frames with the beams (the default)
and with RT_FLAG_NO_BEAMS carry the same bits and count the same segments, on the headline scene (where 10 % of the blocks have no
list and walk the tree), on scenes made of exact ties (shared edges seen edge-on: the lowest triangle index wins, and the record that
goes with it - the first version kept the old record on lanes accepted through the tie clause, one wrong pixel per 2 M segments),
with spheres, tile-edge blocks, NaN cameras, and against the CPU oracle."""
import numpy as np
import pytest

from gpu_raytracer_amd import hostpack as H
from gpu_raytracer_amd import scenes
from test_gpu_adversarial import _grid, _scene

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _both(ctx, scene, w, h, spp, bounces, camera=None, **kw):
    cam = scene.camera if camera is None else camera
    a = ctx.render(w, h, cam, mode=2, spp=spp, max_bounces=bounces, kernel_pipeline=True, **kw)
    img = ctx.read_rgb32f().copy()
    b = ctx.render(w, h, cam, mode=2, spp=spp, max_bounces=bounces, kernel_pipeline=True, no_beams=True, **kw)
    ref = ctx.read_rgb32f()
    assert (a["primary_rays"], a["continuation_rays"], a["shadow_rays"]) == (b["primary_rays"], b["continuation_rays"], b["shadow_rays"])
    diff = np.argwhere((_bits(img) != _bits(ref)).any(-1))
    assert diff.size == 0, f"{scene.name}: {len(diff)} pixels differ between beams and tree walk, first at (x, y) = {tuple(diff[0][::-1])}"
    return img, a


def test_headline_scene_primary_rays_and_full_paths(gpu_ctx):
    sc = scenes.sponza_like()
    gpu_ctx.upload_scene(sc)
    _both(gpu_ctx, sc, 1920, 1080, 9, 0, no_shadows=True)  # 18.7 M camera segments: ~9 exact ties between neighbouring triangles among them
    counts = gpu_ctx.debug_beams(1 << 20)
    lists = counts[counts != 0xFFFFFFFF]
    assert 0.02 < 1.0 - len(lists) / len(counts) < 0.25 and 8 < lists.mean() < 40 and lists.max() <= 128, (len(counts), len(lists), lists.mean())
    _both(gpu_ctx, sc, 960, 540, 4, 4, tile_size=32)
    _both(gpu_ctx, sc, 1001, 333, 3, 2, tile_size=50)  # ragged tiles: blocks that straddle the image edge


def test_exact_ties_along_shared_edges(gpu_ctx, oracle_mod):
    """A fan of triangles around an axis through the camera: every sample near the axis hits two to many triangles at the same t; a
    tessellated plane seen head-on from a camera on a grid line; the lowest index must win with its own record (normal, material)."""
    n = 24
    ang = np.linspace(0, 2 * np.pi, n + 1)
    fan = np.array([[(0.0, 0.0, -4.0), (np.cos(a0), np.sin(a0), -4.0 - 0.3 * (k % 3)), (np.cos(a1), np.sin(a1), -4.0 - 0.3 * ((k + 1) % 3))]
                    for k, (a0, a1) in enumerate(zip(ang[:-1], ang[1:]))], np.float32)
    floor, ids = _grid(16, 16, z=-6.0, size=8.0)
    sc = _scene("ties", np.concatenate([fan, floor]), [k % 4 for k in range(n)] + list(ids), camera=H.camera(position=(0.0, 0.0, 0.0)))
    gpu_ctx.upload_scene(sc)
    for w, h, spp in ((64, 64, 16), (257, 131, 5)):
        img, st = _both(gpu_ctx, sc, w, h, spp, 2)
        ref = oracle_mod.render_extended(oracle_mod.PackedScene(sc, use_bvh=False), w, h, spp, 2)
        np.testing.assert_array_equal(_bits(img), _bits(ref["rgb"]))


def test_spheres_soups_and_degenerate_cameras(gpu_ctx, oracle_mod):
    sc = scenes.random_soup(20000, seed=4, n_spheres=3, n_lights=3)
    gpu_ctx.upload_scene(sc)
    img, st = _both(gpu_ctx, sc, 320, 200, 4, 3)
    ref = oracle_mod.render_extended(oracle_mod.PackedScene(sc, use_bvh=False), 96, 64, 2, 2)
    gpu_ctx.render(96, 64, sc.camera, mode=2, spp=2, max_bounces=2, kernel_pipeline=True)
    np.testing.assert_array_equal(_bits(gpu_ctx.read_rgb32f()), _bits(ref["rgb"]))
    for cam in (H.camera(position=(np.nan, 0, 5)), H.camera(fov=179.9), H.camera(fov=0.0), H.camera(up=(0, 0, -1)), H.camera(position=(0, 0, 3e30)),
                H.camera(direction=(0, 0, 0)), H.camera(position=(0.3, 0.2, -3.0))):  # (the last one sits inside the soup)
        a = gpu_ctx.render(120, 80, cam, mode=2, spp=2, max_bounces=1, kernel_pipeline=True)
        img = _bits(gpu_ctx.read_rgb32f()).copy()
        gpu_ctx.render(120, 80, cam, mode=2, spp=2, max_bounces=1, kernel_pipeline=True, no_beams=True)
        np.testing.assert_array_equal(img, _bits(gpu_ctx.read_rgb32f()))
