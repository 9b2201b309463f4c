// rt_render — headless native front end over the C ABI (include/rt_hip.h) and the C++ mirror of the reference's host
// types (gpu_raytracer_amd/csrc/host/raytracer_host.hpp).
//
// It walks the reference application's own call stack without a window: SceneState::new / load_from_gltf
// (src/scene.rs:20-69) -> BufferManager::update (src/buffers.rs:157-377) -> ComputeRenderer::run_compute, one call per
// presented frame until the progressive image is complete (src/compute.rs:12-251; TileHelper::calculate_tiles_per_frame
// tiles per call, three colour-channel dispatches per tile) -> the display combine of main_fs (shader/src/lib.rs:383-388)
// -> an image file (the reference has no image output; PLAN.md lists it as future), and prints the completion summary of
// src/compute.rs:320-363.  With --spp/--bounces it renders the extended mode through rt_render instead.
//
// Build (done by __graft_entry__.build()):
//   g++ -std=c++17 -O2 examples/rt_render.cpp -Iinclude -Igpu_raytracer_amd/csrc -Lgpu_raytracer_amd -lrt_hip
//       -Wl,-rpath,'$ORIGIN/../gpu_raytracer_amd' -o build/rt_render
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../include/rt_hip.h"
#include "../include/rt_shared.h"
#include "host/gltf_loader.hpp"
#include "host/image_io.hpp"
#include "host/raytracer_host.hpp"

using namespace raytracer;

static double now_ms() {
    using namespace std::chrono;
    return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

static int usage(const char* argv0, int rc) {
    std::fprintf(rc ? stderr : stdout,
                 "usage: %s [--gltf FILE.gltf|.glb] [--size WxH] [--out FILE.png|.ppm|.exr] [--spp N --bounces B] [--fly FRAMES] [--device D]\n"
                 "  without --gltf the reference's default scene (6 spheres, 2 triangles, 1 light) is rendered;\n"
                 "  without --spp the reference path runs: progressive 128x128 tiles, three channel dispatches per tile;\n"
                 "  with --spp N the extended mode (jittered samples, shadow rays, --bounces B, default 4) runs through rt_render;\n"
                 "  with --fly FRAMES a scripted fly-through (CameraController deltas of src/input.rs) renders FRAMES whole frames with\n"
                 "  the reference semantics, reads each back and reports frames per second; the last frame is written.\n",
                 argv0);
    return rc;
}

int main(int argc, char** argv) {
    std::string gltf, out = "out.png";
    uint32_t width = 800, height = 600, spp = 0, bounces = 4, fly = 0;
    int device = 0;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { return i + 1 < argc ? argv[++i] : nullptr; };
        if (a == "--help" || a == "-h") return usage(argv[0], 0);
        const char* v = next();
        if (!v) return usage(argv[0], 2);
        if (a == "--gltf") gltf = v;
        else if (a == "--out") out = v;
        else if (a == "--size") {
            if (std::sscanf(v, "%ux%u", &width, &height) != 2 || !width || !height) return usage(argv[0], 2);
        } else if (a == "--spp") spp = (uint32_t)std::atoi(v);
        else if (a == "--bounces") bounces = (uint32_t)std::atoi(v);
        else if (a == "--fly") fly = (uint32_t)std::atoi(v);
        else if (a == "--device") device = std::atoi(v);
        else return usage(argv[0], 2);
    }

    // ---- scene (src/scene.rs) ----
    SceneState scene;
    if (gltf.empty()) {
        scene = SceneState::new_();
    } else if (GltfError e = scene_state_load_from_gltf(gltf, scene)) {
        std::fprintf(stderr, "glTF load failed: %s\n", e.message.c_str()); // the reference falls back to the default scene here (src/main.rs:173-182);
        return 3;                                                          // a batch tool should not silently render something else
    }
    std::printf("scene: %zu spheres, %zu triangles, %zu vertices, %zu materials, %zu lights, %zu reference-format BVH nodes\n", scene.spheres.size(),
                scene.triangles.size(), scene.vertices.size(), scene.materials.size(), scene.lights.size(), scene.bvh_nodes.size());

    // ---- device + buffers (src/renderer.rs device creation, src/buffers.rs) ----
    rt_ctx* ctx = nullptr;
    int rc = rt_create(&ctx, &device, 1);
    if (rc != RT_OK) {
        std::fprintf(stderr, "rt_create failed (%d): %s\n", rc, rt_last_error(nullptr));
        return 4;
    }
    auto fail = [&](const char* what, int code) {
        std::fprintf(stderr, "%s failed (%d): %s\n", what, code, rt_last_error(ctx));
        rt_destroy(ctx);
        return 5;
    };
    BufferManager buffers;
    std::vector<uint8_t> rgba8((size_t)width * height * 4);

    if (fly > 0) {
        // ---- scripted fly-through: what dragging the mouse and holding W/D does in the reference (src/main.rs:150-186 ->
        // CameraController), one whole frame per step through rt_render (all tiles, three channels in one pass) + read-back ----
        rc = buffers.update(ctx, scene, nullptr);
        if (rc != RT_OK) return fail("BufferManager::update", rc);
        rt_render_params p;
        std::memset(&p, 0, sizeof p);
        p.width = width;
        p.height = height;
        p.spp = 1;
        p.mode = RT_MODE_LEGACY;
        p.tile_world = 1;
        double device_ms = 0.0;
        const double t0 = now_ms();
        for (uint32_t f = 0; f < fly; f++) {
            camera_controller::rotate_camera(scene.camera, (f / 60) % 2 ? -1.0 : 1.0, (f / 30) % 2 ? 0.5 : -0.5); // sweeps left and right, nods
            camera_controller::move_camera(scene.camera, 0.05f, (f / 60) % 2 ? 0.05f : -0.05f);                  // creeps forward, strafes
            p.camera = scene.camera;
            rc = rt_render(ctx, &p);
            if (rc != RT_OK) return fail("rt_render", rc);
            rt_stats st;
            rt_get_stats(ctx, &st);
            device_ms += st.kernel_ms;
            rc = rt_read_rgba8_combined(ctx, rgba8.data(), rgba8.size());
            if (rc != RT_OK) return fail("rt_read_rgba8_combined", rc);
        }
        const double total = now_ms() - t0;
        std::printf("fly-through: %u frames of %ux%u in %.1f ms = %.0f frames/s including the read-back (%.3f ms of device time per frame)\n", fly, width, height,
                    total, fly / (total * 1e-3), device_ms / fly);
    } else if (spp == 0) {
        // ---- the reference's progressive loop (src/main.rs:278-279 -> compute.rs:12) ----
        ProgressiveState progressive;
        progressive.resize(width, height);
        std::vector<double> call_ms;
        bool done = false;
        const double t0 = now_ms();
        while (!done) {
            const double c0 = now_ms();
            rc = ComputeRenderer::run_compute(ctx, buffers, scene, progressive, &done);
            if (rc != RT_OK) return fail("run_compute", rc);
            call_ms.push_back(now_ms() - c0);
        }
        const double total = now_ms() - t0;
        rc = rt_read_rgba8_combined(ctx, rgba8.data(), rgba8.size()); // main_fs: (red.x, green.y, blue.z, 1)
        if (rc != RT_OK) return fail("rt_read_rgba8_combined", rc);
        // completion summary, src/compute.rs:320-363
        std::sort(call_ms.begin(), call_ms.end());
        auto pct = [&](double p) { return call_ms[std::min(call_ms.size() - 1, (size_t)(p * (double)call_ms.size()))]; };
        const uint32_t tiles = progressive.tiles_x * progressive.tiles_y;
        std::printf("progressive rendering complete: %u tiles (%ux%u) in %zu calls of %u, %.2f ms total, %.0f tiles/s, per call P50 %.3f P95 %.3f P99 %.3f ms\n",
                    tiles, progressive.tiles_x, progressive.tiles_y, call_ms.size(), progressive.tiles_per_frame, total, tiles / (total * 1e-3), pct(0.50),
                    pct(0.95), pct(0.99));
    } else {
        // ---- extended mode through the frame-level entry point ----
        rc = buffers.update(ctx, scene, nullptr);
        if (rc != RT_OK) return fail("BufferManager::update", rc);
        rt_render_params p;
        std::memset(&p, 0, sizeof p);
        p.camera = scene.camera;
        p.width = width;
        p.height = height;
        p.spp = spp;
        p.max_bounces = bounces;
        p.mode = RT_MODE_EXTENDED;
        p.tile_world = 1;
        rc = rt_render(ctx, &p);
        if (rc != RT_OK) return fail("rt_render", rc);
        rt_stats st;
        rt_get_stats(ctx, &st);
        std::printf("extended mode: %u spp, %u bounces: %.2f ms on the device, %.1f M segments (%.1f camera, %.1f continuation, %.1f shadow), %.0f Mrays/s\n", spp,
                    bounces, st.kernel_ms, st.rays / 1e6, st.primary_rays / 1e6, st.continuation_rays / 1e6, st.shadow_rays / 1e6,
                    st.rays / (st.kernel_ms * 1e3));
        rc = rt_read_rgba8_combined(ctx, rgba8.data(), rgba8.size());
        if (rc != RT_OK) return fail("rt_read_rgba8_combined", rc);
    }

    const bool ppm = out.size() > 4 && out.substr(out.size() - 4) == ".ppm";
    const bool exr = out.size() > 4 && out.substr(out.size() - 4) == ".exr";
    bool ok;
    if (exr) { // the float image, every bit of it
        std::vector<float> rgb((size_t)width * height * 3);
        rc = rt_read_rgb32f(ctx, rgb.data(), rgb.size());
        if (rc != RT_OK) return fail("rt_read_rgb32f", rc);
        ok = write_exr(out.c_str(), rgb.data(), width, height);
    } else
        ok = ppm ? write_ppm(out.c_str(), rgba8.data(), width, height) : write_png(out.c_str(), rgba8.data(), width, height);
    rt_destroy(ctx);
    if (!ok) {
        std::fprintf(stderr, "cannot write %s\n", out.c_str());
        return 6;
    }
    std::printf("wrote %s (%ux%u)\n", out.c_str(), width, height);
    return 0;
}
