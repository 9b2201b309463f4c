// rt_host_api.cpp — C wrappers (include/rt_host.h) around the C++ host mirror raytracer_host.hpp.
#include "../../include/rt_host.h"

#include "host/raytracer_host.hpp"
#include "host/gltf_loader.hpp"
#include "host/image_io.hpp"

using namespace raytracer;

namespace {
thread_local double g_timing[7] = {0, 0, 0, 0, 0, 0, 0};
}

struct rt_host_scene {
    LoadedScene scene;
};

extern "C" {

void rt_host_material_new(rt_material* out, const float albedo[3], float metallic, float roughness, const float emission[3], float ior,
                          float transmission) {
    *out = material::new_(albedo, metallic, roughness, emission, ior, transmission);
}

void rt_host_light_new(rt_light* out, uint32_t light_type, const float position[3], const float direction[3], const float color[3],
                       float intensity, float range, float inner_cone, float outer_cone) {
    if (light_type == 0) *out = light::directional(direction, color, intensity);
    else if (light_type == 1) *out = light::point(position, color, intensity, range);
    else *out = light::spot(position, direction, color, intensity, range, inner_cone, outer_cone);
}

void rt_host_push_constants_new(rt_push_constants* out, const float resolution[2], const rt_camera* camera, uint32_t triangle_count,
                                uint32_t material_count, const uint32_t tile_offset[2], const uint32_t tile_size[2],
                                const uint32_t total_tiles[2], uint32_t triangles_per_buffer, const rt_scene_metadata_offsets* offsets,
                                uint32_t color_channel, uint32_t wavefront_mode, uint32_t current_bounce, uint32_t max_bounce,
                                uint32_t frame_seed) {
    if (wavefront_mode == 0)
        *out = push_constants::new_(resolution, *camera, triangle_count, material_count, tile_offset, tile_size, total_tiles,
                                    triangles_per_buffer, *offsets, color_channel);
    else
        *out = push_constants::new_wavefront(resolution, *camera, triangle_count, material_count, tile_offset, tile_size, total_tiles,
                                             triangles_per_buffer, *offsets, color_channel, current_bounce, max_bounce, frame_seed);
}

void rt_host_camera_rotate(rt_camera* camera, double delta_x, double delta_y) {
    if (camera) raytracer::camera_controller::rotate_camera(*camera, delta_x, delta_y);
}
void rt_host_camera_move(rt_camera* camera, float forward, float right) {
    if (camera) raytracer::camera_controller::move_camera(*camera, forward, right);
}

void rt_host_tile_count(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t* tiles_x, uint32_t* tiles_y) {
    TileHelper::calculate_tile_count(width, height, tile_size, tiles_x, tiles_y);
}

uint32_t rt_host_tiles_per_frame(uint32_t total_tiles) { return TileHelper::calculate_tiles_per_frame(total_tiles); }

int rt_host_default_scene(rt_sphere* spheres, uint32_t* n_spheres, rt_triangle* triangles, uint32_t* n_triangles, rt_vertex* vertices,
                          uint32_t* n_vertices, rt_material* materials, uint32_t* n_materials, rt_light* lights, uint32_t* n_lights,
                          rt_camera* cam) {
    std::vector<Sphere> s;
    std::vector<Triangle> t;
    std::vector<Vertex> v;
    std::vector<Material> m;
    std::vector<Light> l;
    SceneBuilder::build_default_scene(s, t, v, m, l);
    if (*n_spheres < s.size() || *n_triangles < t.size() || *n_vertices < v.size() || *n_materials < m.size() || *n_lights < l.size()) return -1;
    std::copy(s.begin(), s.end(), spheres);
    std::copy(t.begin(), t.end(), triangles);
    std::copy(v.begin(), v.end(), vertices);
    std::copy(m.begin(), m.end(), materials);
    std::copy(l.begin(), l.end(), lights);
    *n_spheres = (uint32_t)s.size();
    *n_triangles = (uint32_t)t.size();
    *n_vertices = (uint32_t)v.size();
    *n_materials = (uint32_t)m.size();
    *n_lights = (uint32_t)l.size();
    if (cam) *cam = camera::new_();
    return 0;
}

int rt_host_bvh_build(const rt_triangle* triangles, uint32_t n_triangles, const rt_vertex* vertices, uint32_t n_vertices, rt_bvh_node* nodes,
                      uint32_t* n_nodes, uint32_t* triangle_indices, uint32_t* n_indices) {
    if (!n_nodes || !n_indices) return RT_ERR_BAD_ARG;
    for (uint32_t i = 0; i < n_triangles; i++)
        if (triangles[i].v0_index >= n_vertices || triangles[i].v1_index >= n_vertices || triangles[i].v2_index >= n_vertices) return RT_ERR_BAD_ARG;
    std::vector<Triangle> t(triangles, triangles + n_triangles);
    std::vector<Vertex> v(vertices, vertices + n_vertices);
    BvhResult r = BvhBuilder::build(t, v);
    if (nodes) {
        if (*n_nodes < r.nodes.size() || *n_indices < r.triangle_indices.size()) return RT_ERR_BAD_ARG;
        std::copy(r.nodes.begin(), r.nodes.end(), nodes);
        if (triangle_indices) std::copy(r.triangle_indices.begin(), r.triangle_indices.end(), triangle_indices);
    }
    *n_nodes = (uint32_t)r.nodes.size();
    *n_indices = (uint32_t)r.triangle_indices.size();
    return RT_OK;
}

int rt_host_pack_scene_metadata(const rt_sphere* spheres, uint32_t n_spheres, const rt_light* lights, uint32_t n_lights, const rt_bvh_node* nodes,
                                uint32_t n_nodes, const uint32_t* tri_indices, uint32_t n_indices, const rt_vertex* vertices, uint32_t n_vertices,
                                uint32_t* combined, size_t capacity_words, rt_scene_metadata_offsets* offsets) {
    std::vector<uint32_t> buf;
    SceneMetadataOffsets off = BufferManager::pack_scene_metadata(
        std::vector<Sphere>(spheres, spheres + n_spheres), std::vector<Light>(lights, lights + n_lights), std::vector<BvhNode>(nodes, nodes + n_nodes),
        std::vector<uint32_t>(tri_indices, tri_indices + n_indices), std::vector<Vertex>(vertices, vertices + n_vertices), buf);
    if (buf.size() > capacity_words) return RT_ERR_BAD_ARG;
    if (!buf.empty()) std::memcpy(combined, buf.data(), buf.size() * 4);
    if (offsets) *offsets = off;
    return RT_OK;
}

int rt_host_render_progressive(rt_ctx* ctx, const rt_sphere* spheres, uint32_t n_spheres, const rt_light* lights, uint32_t n_lights,
                               const rt_vertex* vertices, uint32_t n_vertices, const rt_triangle* triangles, uint32_t n_triangles,
                               const rt_material* materials, uint32_t n_materials, const rt_camera* cam, uint32_t width, uint32_t height,
                               uint32_t* n_dispatches, uint32_t* n_calls) {
    if (!ctx || !cam || width == 0 || height == 0) return RT_ERR_BAD_ARG;
    SceneState scene;
    scene.spheres.assign(spheres, spheres + n_spheres);
    scene.lights.assign(lights, lights + n_lights);
    scene.vertices.assign(vertices, vertices + n_vertices);
    scene.triangles.assign(triangles, triangles + n_triangles);
    scene.materials.assign(materials, materials + n_materials);
    scene.camera = *cam;
    for (const auto& t : scene.triangles)
        if (t.v0_index >= n_vertices || t.v1_index >= n_vertices || t.v2_index >= n_vertices) return RT_ERR_BAD_ARG;
    scene.rebuild_bvh();
    BufferManager buffers;
    ProgressiveState progressive;
    progressive.resize(width, height);
    uint32_t calls = 0;
    bool done = false;
    std::vector<double> call_ms;
    const auto t_start = std::chrono::steady_clock::now();
    while (!done) {
        const auto t0 = std::chrono::steady_clock::now();
        int rc = ComputeRenderer::run_compute(ctx, buffers, scene, progressive, &done);
        if (rc != RT_OK) return rc;
        call_ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        if (++calls > 1000000u) return RT_ERR_INTERNAL;
    }
    { // print_completion_summary's numbers (src/compute.rs:320-363), returned instead of printed
        const double total = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count();
        std::sort(call_ms.begin(), call_ms.end());
        auto pct = [&](double p) { return call_ms.empty() ? 0.0 : call_ms[std::min(call_ms.size() - 1, (size_t)(p * (double)call_ms.size()))]; };
        const double tiles = (double)progressive.tiles_x * progressive.tiles_y;
        g_timing[0] = total;
        g_timing[1] = (double)calls;
        g_timing[2] = tiles;
        g_timing[3] = total > 0 ? tiles / (total / 1000.0) : 0.0;
        g_timing[4] = pct(0.50);
        g_timing[5] = pct(0.95);
        g_timing[6] = pct(0.99);
    }
    if (n_dispatches) *n_dispatches = progressive.tiles_x * progressive.tiles_y * 3;
    if (n_calls) *n_calls = calls;
    return RT_OK;
}

static int gltf_finish(GltfLoader& loader, GltfError e, int scene_index, rt_host_scene** out, char* err, size_t err_len) {
    std::unique_ptr<rt_host_scene> h(new rt_host_scene());
    if (!e) e = loader.extract_scene(scene_index, h->scene);
    if (e) {
        if (err && err_len) std::snprintf(err, err_len, "%s", e.message.c_str());
        return e.kind == GltfError::IoError ? RT_HOST_ERR_IO : e.kind == GltfError::ValidationError ? RT_HOST_ERR_VALIDATION : RT_HOST_ERR_GLTF;
    }
    *out = h.release();
    return RT_OK;
}

int rt_host_gltf_load(const char* path, int scene_index, rt_host_scene** out, char* err, size_t err_len) {
    if (!path || !out) return RT_ERR_BAD_ARG;
    *out = nullptr;
    GltfLoader loader;
    GltfError e = GltfLoader::load_from_path(path, loader);
    return gltf_finish(loader, e, scene_index, out, err, err_len);
}

int rt_host_gltf_load_glb(const uint8_t* data, size_t len, int scene_index, rt_host_scene** out, char* err, size_t err_len) {
    if (!data || !out) return RT_ERR_BAD_ARG;
    *out = nullptr;
    GltfLoader loader;
    GltfError e = GltfLoader::load_from_glb(data, len, loader);
    return gltf_finish(loader, e, scene_index, out, err, err_len);
}

void rt_host_scene_counts(const rt_host_scene* s, uint32_t counts[6]) {
    const LoadedScene& l = s->scene;
    counts[0] = (uint32_t)l.spheres.size();
    counts[1] = (uint32_t)l.lights.size();
    counts[2] = (uint32_t)l.vertices.size();
    counts[3] = (uint32_t)l.triangles.size();
    counts[4] = (uint32_t)l.materials.size();
    counts[5] = (uint32_t)l.cameras.size();
}

int rt_host_scene_copy(const rt_host_scene* s, rt_sphere* spheres, rt_light* lights, rt_vertex* vertices, rt_triangle* triangles,
                       rt_material* materials, rt_camera* cameras) {
    if (!s) return RT_ERR_BAD_ARG;
    const LoadedScene& l = s->scene;
    if (spheres) std::copy(l.spheres.begin(), l.spheres.end(), spheres);
    if (lights) std::copy(l.lights.begin(), l.lights.end(), lights);
    if (vertices) std::copy(l.vertices.begin(), l.vertices.end(), vertices);
    if (triangles) std::copy(l.triangles.begin(), l.triangles.end(), triangles);
    if (materials) std::copy(l.materials.begin(), l.materials.end(), materials);
    if (cameras) std::copy(l.cameras.begin(), l.cameras.end(), cameras);
    return RT_OK;
}

void rt_host_scene_free(rt_host_scene* s) { delete s; }

int rt_host_write_ppm(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!path || !rgba8 || !width || !height) return RT_ERR_BAD_ARG;
    return write_ppm(path, rgba8, width, height) ? RT_OK : RT_HOST_ERR_IO;
}

int rt_host_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height) {
    if (!path || !rgba8 || !width || !height) return RT_ERR_BAD_ARG;
    return write_png(path, rgba8, width, height) ? RT_OK : RT_HOST_ERR_IO;
}

int rt_host_write_exr(const char* path, const float* rgb32f, uint32_t width, uint32_t height) {
    if (!path || !rgb32f || !width || !height) return RT_ERR_BAD_ARG;
    return write_exr(path, rgb32f, width, height) ? RT_OK : RT_HOST_ERR_IO;
}

float rt_host_branchless_float_if_nonnan(int condition, float if_true, float if_false) { return branchless::float_if_nonnan(condition != 0, if_true, if_false); }
float rt_host_branchless_float_if(int condition, float if_true, float if_false, int* valid) {
    const branchless::FloatIf r = branchless::float_if(condition != 0, if_true, if_false);
    if (valid) *valid = r.valid ? 1 : 0;
    return r.value;
}
uint32_t rt_host_branchless_u32_if(int condition, uint32_t if_true, uint32_t if_false) { return branchless::u32_if(condition != 0, if_true, if_false); }

int rt_host_bvh_triangle(const rt_triangle* triangle, const rt_vertex* vertices, uint32_t n_vertices, float centroid[3], rt_aabb* box) {
    if (!triangle || !vertices) return RT_ERR_BAD_ARG;
    if (triangle->v0_index >= n_vertices || triangle->v1_index >= n_vertices || triangle->v2_index >= n_vertices) return RT_ERR_BAD_ARG;
    const BvhTriangle b = bvh_triangle::new_(*triangle, 0);
    if (centroid) bvh_triangle::centroid(b, vertices, centroid);
    if (box) *box = bvh_triangle::aabb(b, vertices);
    return RT_OK;
}

int rt_host_triangle_aabb(const rt_triangle* triangle, const rt_vertex* vertices, uint32_t n_vertices, rt_aabb* box) {
    if (!triangle || !vertices || !box) return RT_ERR_BAD_ARG;
    if (triangle->v0_index >= n_vertices || triangle->v1_index >= n_vertices || triangle->v2_index >= n_vertices) return RT_ERR_BAD_ARG;
    *box = BvhBuilder::triangle_aabb(*triangle, vertices);
    return RT_OK;
}

void rt_host_progressive_timing(double out[7]) {
    for (int i = 0; i < 7; i++) out[i] = g_timing[i];
}

} // extern "C"
