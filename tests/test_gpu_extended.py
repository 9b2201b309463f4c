"""GPU tests of the extended mode (jittered spp, shadow rays, bounces) against its CPU statement.

Parity for this mode is UNPINNED (no reference implementation exists); what is checked is that the HIP
kernel and the CPU specification agree — bit for bit, because both keep the same f32 operation order,
IEEE divide/sqrt and explicit-fma sin/cos — and that the mode reduces to the pinned mode 1.
Tolerance fallback (BASELINE.md gate): |d rgb| <= 2e-3 for >= 99.9 % of pixels.
"""
import numpy as np
import pytest

from gpu_raytracer_amd import scenes

pytestmark = pytest.mark.gpu


KERNELS = {"wavefront": {}, "state_machine": {"kernel_sm": True}, "nested": {"kernel_v1": True}}


def _gpu_ext(ctx, scene, w, h, spp, bounces, **kw):
    ctx.upload_scene(scene)
    st = ctx.render(w, h, kw.pop("camera", scene.camera), mode=2, spp=spp, max_bounces=bounces, **kw)
    return ctx.read_rgb32f(), st


@pytest.mark.parametrize("name", ["default", "cornell12", "single_triangle", "empty"])
def test_extended_reduces_to_mode1_on_gpu(gpu_ctx, name):
    scene = scenes.SCENES[name]()
    rgb2, st = _gpu_ext(gpu_ctx, scene, 160, 96, 1, 0, no_shadows=True)
    gpu_ctx.render(160, 96, scene.camera, mode=1)
    rgb1 = gpu_ctx.read_rgb32f()
    np.testing.assert_array_equal(rgb2.view(np.uint32), rgb1.view(np.uint32))
    assert st["rays"] == st["primary_rays"] == 160 * 96 and st["shadow_rays"] == 0


@pytest.mark.parametrize("kernel", list(KERNELS))
@pytest.mark.parametrize("name,w,h,spp,bounces", [
    ("default", 96, 64, 8, 4), ("cornell12", 64, 64, 16, 4), ("cornell12", 40, 40, 3, 1), ("single_triangle", 64, 40, 4, 2)])
def test_extended_bit_exact_vs_cpu_statement(gpu_ctx, oracle_mod, name, w, h, spp, bounces, kernel):
    """All three device implementations (wavefront pipeline = default, state-machine megakernel, nested-loop
    megakernel) against the CPU statement: identical segment counts and identical bits."""
    scene = scenes.SCENES[name]()
    ref = oracle_mod.render_extended(oracle_mod.PackedScene(scene, use_bvh=False), w, h, spp, bounces)
    rgb, st = _gpu_ext(gpu_ctx, scene, w, h, spp, bounces, **KERNELS[kernel])
    seg = ref["segments"]
    assert (st["primary_rays"], st["continuation_rays"], st["shadow_rays"]) == (seg["camera"], seg["continuation"], seg["shadow"])
    assert st["rays"] == seg["camera"] + seg["continuation"] + seg["shadow"]
    np.testing.assert_array_equal(rgb.view(np.uint32), ref["rgb"].view(np.uint32))


@pytest.mark.parametrize("kernel", list(KERNELS))
def test_extended_soup_with_spheres_glass_metal(gpu_ctx, oracle_mod, kernel):
    scene = scenes.random_soup(400, seed=21, size=0.7, n_spheres=3, n_lights=3)  # materials: metal, glass, emissive, diffuse
    ref = oracle_mod.render_extended(oracle_mod.PackedScene(scene, use_bvh=False), 72, 48, 6, 5, frame_seed=77)
    rgb, st = _gpu_ext(gpu_ctx, scene, 72, 48, 6, 5, frame_seed=77, **KERNELS[kernel])
    d = np.abs(rgb - ref["rgb"]).max(-1)
    assert (d > 2e-3).mean() <= 1e-3
    np.testing.assert_array_equal(rgb.view(np.uint32), ref["rgb"].view(np.uint32))


def test_extended_sponza_like_vs_cpu_statement(gpu_ctx, oracle_mod):
    scene = scenes.sponza_like()
    w, h, spp, bounces = 64, 36, 2, 2
    ref = oracle_mod.render_extended(oracle_mod.PackedScene(scene), w, h, spp, bounces)
    rgb, st = _gpu_ext(gpu_ctx, scene, w, h, spp, bounces)
    d = np.abs(rgb - ref["rgb"]).max(-1)
    assert (d > 2e-3).mean() <= 1e-3, f"outliers {(d > 2e-3).mean()}"
    seg = ref["segments"]
    assert abs(st["rays"] - (seg["camera"] + seg["continuation"] + seg["shadow"])) <= 0.001 * st["rays"]


def test_extended_full_size_properties(gpu_ctx):
    """Headline workload shape (sponza-like 1080p) at 4 spp: determinism, finiteness, tile partition invariance,
    spp convergence, and ray accounting."""
    scene = scenes.sponza_like()
    gpu_ctx.upload_scene(scene)
    w, h = 1920, 1080
    st = gpu_ctx.render(w, h, scene.camera, mode=2, spp=4, max_bounces=4)
    a = gpu_ctx.read_rgb32f()
    assert st["primary_rays"] == w * h * 4 and st["rays"] == st["primary_rays"] + st["continuation_rays"] + st["shadow_rays"]
    assert st["continuation_rays"] > st["primary_rays"] and st["shadow_rays"] > 0
    assert np.isfinite(a).all() and (a >= 0).all()
    gpu_ctx.render(w, h, scene.camera, mode=2, spp=4, max_bounces=4)
    np.testing.assert_array_equal(gpu_ctx.read_rgb32f().view(np.uint32), a.view(np.uint32))
    # a pixel's samples do not depend on which rank renders its tile
    acc = np.zeros_like(a)
    for rank in range(2):
        gpu_ctx.render(w, h, scene.camera, mode=2, spp=4, max_bounces=4, tile_rank=rank, tile_world=2)
        part = gpu_ctx.read_rgb32f()
        for tile in range(rank, 15 * 9, 2):
            ox, oy = (tile % 15) * 128, (tile // 15) * 128
            acc[oy:oy + 128, ox:ox + 128] = part[oy:oy + 128, ox:ox + 128]
    np.testing.assert_array_equal(acc.view(np.uint32), a.view(np.uint32))
    gpu_ctx.render(w, h, scene.camera, mode=2, spp=16, max_bounces=4)
    b = gpu_ctx.read_rgb32f()
    assert abs(a.mean() - b.mean()) < 0.02 * b.mean()
    # the megakernels produce the very same image as the wavefront pipeline
    for kw in ({"kernel_sm": True}, {"kernel_v1": True}):
        gpu_ctx.render(w, h, scene.camera, mode=2, spp=4, max_bounces=4, **kw)
        np.testing.assert_array_equal(gpu_ctx.read_rgb32f().view(np.uint32), a.view(np.uint32))


@pytest.mark.parametrize("n_lights", [9, 32, 40])
def test_extended_many_lights(gpu_ctx, oracle_mod, n_lights):
    """Shadow-queue windows grow with the light count (>= 128 slots per light), visibility is one bit per light in a
    32-bit word, and more than 32 lights take the state-machine megakernel: all bit-exact against the CPU statement.

    Record: this test was red once, in gpurun_out/r02_pytest7.log (round 2, 17:32), at exactly 32 lights, 13 % of
    the pixels.  The tree then held an uncommitted experiment - k_wf_shade handing k_wf_finish the mask of contributing lights PACKED
    BESIDE the visibility bits in the same 32-bit word of the vertex record (profiles/ab_r02.json, "k_wf_shade hands k_wf_finish the
    mask of contributing lights", kept: false).  The word has room for both only below 32 lights; at RT_WF_MAX_LIGHTS the two fields
    overlapped and lights were dropped or lit wrongly.  The experiment measured no gain and was removed in 9a2666d seven minutes later,
    before anything of it was committed; the visibility word has been visibility only ever since (`1u << li`, li < 32)."""
    scene = scenes.random_soup(600, seed=5, size=0.6, n_spheres=1, n_lights=n_lights)
    ref = oracle_mod.render_extended(oracle_mod.PackedScene(scene, use_bvh=False), 64, 40, 3, 2, frame_seed=3)
    rgb, st = _gpu_ext(gpu_ctx, scene, 64, 40, 3, 2, frame_seed=3)
    seg = ref["segments"]
    assert (st["primary_rays"], st["continuation_rays"], st["shadow_rays"]) == (seg["camera"], seg["continuation"], seg["shadow"])
    np.testing.assert_array_equal(rgb.view(np.uint32), ref["rgb"].view(np.uint32))


def test_extended_image_independent_of_batching_and_tiles(rt_api, monkeypatch):
    """The wavefront pipeline's scheduling knobs - samples per batch, tile size, tile partition - only change WHEN a
    path's next step runs: same bits for 1, 3 and all samples per batch, for 16/64/128-pixel tiles."""
    scene = scenes.sponza_like()
    w, h, spp, bounces = 480, 270, 7, 3
    images = {}
    for batch in ("1", "3", "0"):  # 0 = the library's own choice
        monkeypatch.setenv("RT_WF_BATCH", batch)
        with rt_api.Context() as ctx:  # a fresh context: an allocation is reused for the same frame shape
            ctx.upload_scene(scene)
            ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces)
            images["batch" + batch] = ctx.read_rgb32f()
    monkeypatch.delenv("RT_WF_BATCH")
    with rt_api.Context() as ctx:
        ctx.upload_scene(scene)
        for ts in (16, 64, 128):
            ctx.render(w, h, scene.camera, mode=2, spp=spp, max_bounces=bounces, tile_size=ts)
            images[f"tile{ts}"] = ctx.read_rgb32f()
    ref = images["batch0"]
    assert np.isfinite(ref).all()
    for k, img in images.items():
        np.testing.assert_array_equal(img.view(np.uint32), ref.view(np.uint32), err_msg=k)

