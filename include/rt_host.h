/*
 * rt_host.h — C entry points onto the C++ host mirror (gpu_raytracer_amd/csrc/host/raytracer_host.hpp),
 * so that non-C++ hosts and the Python tests can reach the reference-named host functions.
 * Each function names the reference function it mirrors.  These are conveniences above the boundary;
 * the drop-in boundary itself is rt_hip.h.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include "rt_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Material::new (shared/src/lib.rs:255-291): f16 packing metallic lo16 | roughness hi16, ior lo16 | transmission hi16 */
void rt_host_material_new(rt_material* out, const float albedo[3], float metallic, float roughness, const float emission[3],
                          float ior, float transmission);
/* Light::directional / point / spot (shared/src/lib.rs:497-586); light_type selects which */
void rt_host_light_new(rt_light* out, uint32_t light_type, const float position[3], const float direction[3], const float color[3],
                       float intensity, float range, float inner_cone, float outer_cone);
/* PushConstants::new (mode 0) / new_wavefront (mode != 0) (shared/src/lib.rs:1076-1134) */
void rt_host_push_constants_new(rt_push_constants* out, const float resolution[2], const rt_camera* camera, uint32_t triangle_count,
                                uint32_t material_count, const uint32_t tile_offset[2], const uint32_t tile_size[2],
                                const uint32_t total_tiles[2], uint32_t triangles_per_buffer, const rt_scene_metadata_offsets* offsets,
                                uint32_t color_channel, uint32_t wavefront_mode, uint32_t current_bounce, uint32_t max_bounce,
                                uint32_t frame_seed);
/* CameraController::rotate_camera / move_camera (src/input.rs:49-97): mouse delta in pixels, movement in units of
 * CAMERA_MOVE_SPEED.  For scripted fly-throughs. */
void rt_host_camera_rotate(rt_camera* camera, double delta_x, double delta_y);
void rt_host_camera_move(rt_camera* camera, float forward, float right);

/* TileHelper::calculate_tile_count / calculate_tiles_per_frame (shared/src/lib.rs:1187-1203) */
void rt_host_tile_count(uint32_t width, uint32_t height, uint32_t tile_size, uint32_t* tiles_x, uint32_t* tiles_y);
uint32_t rt_host_tiles_per_frame(uint32_t total_tiles);

/* branchless_float_if! / branchless_u32_if! (shared/src/lib.rs:1293-1326; never used by the shader, pinned by the
 * reference's unit tests :1333-1365): the `@nonnan` arm, the NaN-aware arm (returns the value, *valid = the macro's
 * second tuple element) and the integer select. */
float rt_host_branchless_float_if_nonnan(int condition, float if_true, float if_false);
float rt_host_branchless_float_if(int condition, float if_true, float if_false, int* valid);
uint32_t rt_host_branchless_u32_if(int condition, uint32_t if_true, uint32_t if_false);

/* BvhTriangle::centroid (src/bvh.rs:27-37) and BvhTriangleWithVertices::aabb (src/bvh.rs:47-55) of one triangle. */
int rt_host_bvh_triangle(const rt_triangle* triangle, const rt_vertex* vertices, uint32_t n_vertices, float centroid[3], rt_aabb* box);
/* BvhBuilder::triangle_aabb (src/bvh.rs:272-275): the per-triangle box the chunked build unions (:165). */
int rt_host_triangle_aabb(const rt_triangle* triangle, const rt_vertex* vertices, uint32_t n_vertices, rt_aabb* box);

/* SceneBuilder::build_default_scene (shared/src/lib.rs:1242-1286) + Camera::new.  Arrays must hold
 * 6 spheres, 2 triangles, 6 vertices, 4 materials, 1 light; returns 0, or -1 if a capacity is too small. */
int rt_host_default_scene(rt_sphere* spheres, uint32_t* n_spheres, rt_triangle* triangles, uint32_t* n_triangles, rt_vertex* vertices,
                          uint32_t* n_vertices, rt_material* materials, uint32_t* n_materials, rt_light* lights, uint32_t* n_lights,
                          rt_camera* camera);

/* BvhBuilder::build (src/bvh.rs:104-122): reference-format BVH.  Call with nodes == NULL for the counts. */
int rt_host_bvh_build(const rt_triangle* triangles, uint32_t n_triangles, const rt_vertex* vertices, uint32_t n_vertices,
                      rt_bvh_node* nodes, uint32_t* n_nodes, uint32_t* triangle_indices, uint32_t* n_indices);

/* BufferManager::update_scene_metadata's packing (src/buffers.rs:213-268): combined must hold
 * 5*n_spheres + 13*n_lights + 12*n_nodes + n_indices + 3*n_vertices words. */
int rt_host_pack_scene_metadata(const rt_sphere* spheres, uint32_t n_spheres, const rt_light* lights, uint32_t n_lights,
                                const rt_bvh_node* nodes, uint32_t n_nodes, const uint32_t* tri_indices, uint32_t n_indices,
                                const rt_vertex* vertices, uint32_t n_vertices, uint32_t* combined, size_t capacity_words,
                                rt_scene_metadata_offsets* offsets);

/* The reference's whole frame loop on an rt_ctx: BvhBuilder::build -> BufferManager::update_* ->
 * ComputeRenderer::run_compute repeated until the image is complete (src/compute.rs:12-251), i.e. one
 * rt_upload_scene_packed and tiles x 3 rt_dispatch_tile calls.  *n_dispatches receives the dispatch count. */
int rt_host_render_progressive(rt_ctx* ctx, const rt_sphere* spheres, uint32_t n_spheres, const rt_light* lights, uint32_t n_lights,
                               const rt_vertex* vertices, uint32_t n_vertices, const rt_triangle* triangles, uint32_t n_triangles,
                               const rt_material* materials, uint32_t n_materials, const rt_camera* camera, uint32_t width,
                               uint32_t height, uint32_t* n_dispatches, uint32_t* n_calls);

/* ---- row N1: glTF 2.0 / GLB loading with the behaviour of src/gltf_loader.rs (extract_scene :77-125) ---- */
typedef struct rt_host_scene rt_host_scene;
#define RT_HOST_ERR_IO (-11)         /* GltfError::IoError         */
#define RT_HOST_ERR_GLTF (-12)       /* GltfError::GltfError       */
#define RT_HOST_ERR_VALIDATION (-13) /* GltfError::ValidationError */
/* GltfLoader::load_from_path + extract_scene(scene_index < 0 ? None : Some(i)).  err receives the message. */
int rt_host_gltf_load(const char* path, int scene_index, rt_host_scene** out, char* err, size_t err_len);
/* GltfLoader::load_from_glb + extract_scene */
int rt_host_gltf_load_glb(const uint8_t* data, size_t len, int scene_index, rt_host_scene** out, char* err, size_t err_len);
/* counts[6] = spheres, lights, vertices, triangles, materials, cameras of the LoadedScene */
void rt_host_scene_counts(const rt_host_scene* s, uint32_t counts[6]);
/* copy the LoadedScene's Vecs into caller arrays (each may be NULL) */
int rt_host_scene_copy(const rt_host_scene* s, rt_sphere* spheres, rt_light* lights, rt_vertex* vertices, rt_triangle* triangles,
                       rt_material* materials, rt_camera* cameras);
void rt_host_scene_free(rt_host_scene* s);

/* ---- row N3: image output (the reference has none) ---- */
int rt_host_write_ppm(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);
int rt_host_write_png(const char* path, const uint8_t* rgba8, uint32_t width, uint32_t height);
/* OpenEXR (scanline, uncompressed, FLOAT R G B) of the packed float image rt_read_rgb32f returns: width*height*3 floats */
int rt_host_write_exr(const char* path, const float* rgb32f, uint32_t width, uint32_t height);

/* ---- row N4: the reference's progressive completion summary (src/compute.rs:320-363) for the last
 * rt_host_render_progressive on this thread: out[0] total ms, [1] calls, [2] tiles, [3] tiles/s,
 * [4] p50, [5] p95, [6] p99 of the per-call time in ms ---- */
void rt_host_progressive_timing(double out[7]);

#ifdef __cplusplus
}
#endif
#endif
