"""The light grids of the extended mode's shadow stage are built when a frame first needs them, not by rt_upload_scene* (VERDICT r02
item 5): the reference's flow - SceneState::replace_with_gltf (src/scene.rs:87-119), then ComputeRenderer::run_compute every frame
(src/compute.rs:12-50) - only ever renders modes 0/1, which trace no shadow segments, and must not pay 45-50 ms and 7.5 GB for them.
Through the C ABI: rt_stats.grid_bytes / grid_build_ms, rt_prepare."""
import time

import numpy as np
import pytest

from gpu_raytracer_amd import scenes

pytestmark = pytest.mark.gpu


def test_upload_and_reference_frames_build_no_grids_and_the_first_extended_frame_does(gpu_ctx):
    sc = scenes.sponza_like()
    gpu_ctx.upload_scene(sc)  # (a context's first upload also pays one-time module loading)
    t0 = time.perf_counter()
    gpu_ctx.upload_scene(sc)
    st = gpu_ctx.render(1920, 1080, sc.camera, mode=1)
    ms = (time.perf_counter() - t0) * 1e3
    print(f"rt_upload_scene (262,144 triangles) + one mode-1 frame at 1080p: {ms:.1f} ms")
    assert ms < 30.0  # measured 11-12 ms; 55-60 ms when the upload built the grids
    for mode in (0, 1):
        st = gpu_ctx.render(640, 360, sc.camera, mode=mode)
        assert st["grid_bytes"] == 0 and st["grid_build_ms"] == 0.0
    assert gpu_ctx.debug_shadow_grid()["lights_with_grid"] == 0
    # frames that trace no shadow segments through the lists do not build them either
    for kw in ({"no_shadows": True}, {"no_shadow_grid": True}, {"kernel_sm": True}):
        st = gpu_ctx.render(320, 180, sc.camera, mode=2, spp=2, max_bounces=2, **kw)
        assert st["grid_bytes"] == 0, kw
    ref = gpu_ctx.read_rgb32f().copy()  # (the state machine walks the BVH for its shadow segments)
    st = gpu_ctx.render(320, 180, sc.camera, mode=2, spp=2, max_bounces=2)
    assert st["grid_bytes"] > 0 and st["grid_build_ms"] > 0.0
    assert gpu_ctx.debug_shadow_grid()["lights_with_grid"] == len(sc.lights) and gpu_ctx.debug_shadow_grid()["bytes"] == st["grid_bytes"]
    np.testing.assert_array_equal(gpu_ctx.read_rgb32f().view(np.uint32), ref.view(np.uint32))
    built = (st["grid_bytes"], st["grid_build_ms"])
    st = gpu_ctx.render(320, 180, sc.camera, mode=2, spp=2, max_bounces=2)
    assert (st["grid_bytes"], st["grid_build_ms"]) == built  # once per scene
    gpu_ctx.upload_scene(sc)  # a new scene: the grids are gone with the old one
    assert gpu_ctx.stats()["grid_bytes"] == 0 and gpu_ctx.debug_shadow_grid()["lights_with_grid"] == 0


def test_prepare_builds_them_ahead_of_time(gpu_ctx):
    sc = scenes.cornell12()
    with pytest.raises(Exception):
        from gpu_raytracer_amd import api
        with api.Context() as fresh:
            fresh.prepare()  # nothing uploaded
    gpu_ctx.upload_scene(sc)
    with pytest.raises(Exception):
        gpu_ctx.prepare(what=6)  # unknown bits
    gpu_ctx.prepare()
    st = gpu_ctx.stats()
    assert st["grid_bytes"] > 0 and gpu_ctx.debug_shadow_grid()["lights_with_grid"] == 1
    gpu_ctx.prepare()  # idempotent
    assert gpu_ctx.stats()["grid_bytes"] == st["grid_bytes"] and gpu_ctx.stats()["grid_build_ms"] == st["grid_build_ms"]


def test_quality_tree_is_an_option_for_scenes_that_stay(gpu_ctx, rt_api):
    """rt_prepare(RT_PREPARE_QUALITY_TREE): the host builder's tree (binned SAH + insertion-based optimisation) in place of the one
    rt_upload_scene built on the device: same frames in every mode, rt_stats.tree_build says which tree is in use, the light grids
    (they hold triangle records in leaf order) go with the old tree and come back on demand."""
    sc = scenes.sponza_like()
    gpu_ctx.upload_scene(sc)
    assert gpu_ctx.stats()["tree_build"] == 2  # built on the device
    frames = {}
    for mode, kw in ((1, {}), (2, dict(spp=3, max_bounces=3))):
        gpu_ctx.render(640, 360, sc.camera, mode=mode, **kw)
        frames[mode] = gpu_ctx.read_rgb32f().copy()
    before = gpu_ctx.stats()
    assert before["grid_bytes"] > 0
    gpu_ctx.prepare(rt_api.PREPARE_QUALITY_TREE)
    st = gpu_ctx.stats()
    assert st["tree_build"] == 0 and st["grid_bytes"] == 0 and st["bvh_nodes"] > 0 and st["bvh_nodes"] != before["bvh_nodes"]
    chk = gpu_ctx.debug_check_bvh()
    assert chk["failures"] == 0 and chk["placed_once"] == sc.n_triangles and chk["method"] == 0
    for mode, kw in ((1, {}), (2, dict(spp=3, max_bounces=3))):
        gpu_ctx.render(640, 360, sc.camera, mode=mode, **kw)
        np.testing.assert_array_equal(gpu_ctx.read_rgb32f().view(np.uint32), frames[mode].view(np.uint32))
    assert gpu_ctx.stats()["grid_bytes"] > 0  # rebuilt by the mode-2 frame
    gpu_ctx.prepare(rt_api.PREPARE_QUALITY_TREE | rt_api.PREPARE_SHADOW_GRIDS)  # idempotent
    assert gpu_ctx.stats()["tree_build"] == 0
    gpu_ctx.upload_scene(sc)  # a new upload is a device build again
    assert gpu_ctx.stats()["tree_build"] == 2


def test_stage_times_are_those_of_the_light_grid_launches(gpu_ctx):
    """RT_FLAG_STAGE_TIMES (bench.py's roofline of the dominant kernel): every k_wf_shadow_grid launch of the frame is timed with HIP events on
    the stream it is launched on: one launch per batch and depth, their sum a part of the frame's kernel time; frames without the flag report none."""
    sc = scenes.sponza_like()
    gpu_ctx.upload_scene(sc)
    st = gpu_ctx.render(640, 360, sc.camera, mode=2, spp=4, max_bounces=3, stage_times=True)
    ms, launches = gpu_ctx.debug_stage_times()
    assert launches in (4, 8) and 0.0 < ms < st["kernel_ms"]  # (3 + 1) depths x one or two batches
    gpu_ctx.render(640, 360, sc.camera, mode=2, spp=4, max_bounces=3)
    assert gpu_ctx.debug_stage_times() == (0.0, 0)
    gpu_ctx.render(640, 360, sc.camera, mode=2, spp=4, max_bounces=3, stage_times=True, no_shadow_grid=True)
    assert gpu_ctx.debug_stage_times() == (0.0, 0)  # no light-grid launches in that frame
