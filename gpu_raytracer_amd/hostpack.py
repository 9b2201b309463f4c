"""numpy harness for the reference's host-side constructors and packers.

These build the *inputs* (scene arrays, packed binding-1 buffer, push
constants) for tests and the bench.  They restate, in numpy, the same
functions the C++ host mirror (csrc/host/raytracer_host.hpp) implements, so the
two can be checked against each other:

  Material::new / diffuse / metallic / glass / emissive  shared/src/lib.rs:255-332
  Light::directional / point / spot                      shared/src/lib.rs:497-586
  PushConstants::new / new_wavefront / pack_*            shared/src/lib.rs:1076-1159
  TileHelper::calculate_tile_count / tiles_per_frame     shared/src/lib.rs:1187-1203
  SceneBuilder::build_default_scene                      shared/src/lib.rs:1242-1286
  BufferManager::update_scene_metadata (packing order)   src/buffers.rs:213-268
  BufferManager::update_triangles (3-buffer split)       src/buffers.rs:274-336
"""
import numpy as np

from . import types as T


def f16_bits(x):
    """half::f16::from_f32(x).to_bits() — round to nearest even (shared/src/lib.rs:250-252)."""
    with np.errstate(over="ignore"):
        return int(np.array(x, dtype=np.float32).astype(np.float16).view(np.uint16))


def material_new(albedo, metallic, roughness, emission, ior, transmission):
    m = np.zeros((), dtype=T.MATERIAL)
    m["albedo"] = albedo
    m["metallic_roughness_f16"] = f16_bits(metallic) | (f16_bits(roughness) << 16)
    m["emission"] = emission
    m["ior_transmission_f16"] = f16_bits(ior) | (f16_bits(transmission) << 16)
    m["specular_factor"] = 1.0
    m["specular_color"] = (1.0, 1.0, 1.0)
    m["attenuation_distance"] = np.inf
    m["attenuation_color"] = (1.0, 1.0, 1.0)
    m["thickness_factor"] = 0.0
    m["diffuse_factor"] = albedo
    m["glossiness_factor"] = np.float32(1.0) - np.float32(roughness)
    m["material_type"] = 0
    m["texture_indices"] = 0xFFFFFFFF
    return m


def material_diffuse(albedo):
    return material_new(albedo, 0.0, 1.0, (0, 0, 0), 1.5, 0.0)


def material_metallic(albedo, roughness):
    return material_new(albedo, 1.0, roughness, (0, 0, 0), 1.5, 0.0)


def material_glass(albedo, ior, transmission):
    return material_new(albedo, 0.0, 0.0, (0, 0, 0), ior, transmission)


def material_emissive(albedo, emission):
    return material_new(albedo, 0.0, 1.0, emission, 1.5, 0.0)


def _light(position, light_type, color, intensity, direction, rng, inner, outer):
    l = np.zeros((), dtype=T.LIGHT)
    l["position"] = position
    l["light_type"] = light_type
    l["color"] = color
    l["intensity"] = intensity
    l["direction"] = direction
    l["range_packed"] = f16_bits(rng)
    l["cone_angles_packed"] = f16_bits(inner) | (f16_bits(outer) << 16)
    return l


def light_directional(direction, color, intensity):
    return _light((0, 0, 0), 0, color, intensity, direction, np.inf, 0.0, 0.0)


def light_point(position, color, intensity, rng=np.inf):
    return _light(position, 1, color, intensity, (0, 0, 0), rng, 0.0, 0.0)


def light_spot(position, direction, color, intensity, rng, inner, outer):
    return _light(position, 2, color, intensity, direction, rng, inner, outer)


def camera(position=(0.0, 0.0, 5.0), direction=(0.0, 0.0, -1.0), up=(0.0, 1.0, 0.0), fov=45.0):
    """Camera::new defaults — shared/src/lib.rs:231-238."""
    c = np.zeros((), dtype=T.CAMERA)
    c["position"], c["direction"], c["up"], c["fov"] = position, direction, up, fov
    return c


def pack_tile_size(w, h):
    return (min(w, 65535) & 0xFFFF) | ((min(h, 65535) & 0xFFFF) << 16)


def pack_flags(channel, cur_bounce, max_bounce, mode):
    return (channel & 0xFF) | ((cur_bounce & 0xFF) << 8) | ((max_bounce & 0xFF) << 16) | ((mode & 0xFF) << 24)


def push_constants(resolution, cam, triangle_count, material_count, tile_offset, tile_size, total_tiles,
                   triangles_per_buffer, offsets, channel, mode=0, cur_bounce=0, max_bounce=4, frame_seed=0):
    """PushConstants::new (mode 0: pack_flags(channel, 0, 4, 0)) / new_wavefront (mode 1)."""
    pc = np.zeros((), dtype=T.PUSH_CONSTANTS)
    pc["resolution"] = resolution
    pc["camera"] = cam
    pc["triangle_count"] = triangle_count
    pc["material_count"] = material_count
    pc["tile_offset"] = tile_offset
    pc["tile_size_packed"] = pack_tile_size(*tile_size)
    pc["total_tiles"] = total_tiles
    pc["triangles_per_buffer"] = triangles_per_buffer
    pc["metadata_offsets"] = offsets
    pc["packed_flags"] = pack_flags(channel, cur_bounce, max_bounce, mode)
    pc["frame_seed"] = frame_seed
    return pc


def tile_count(width, height, tile_size=T.TILE_SIZE):
    return (width + tile_size - 1) // tile_size, (height + tile_size - 1) // tile_size


def tiles_per_frame(total_tiles):
    if total_tiles <= 16:
        v = total_tiles
    elif total_tiles <= 64:
        v = total_tiles // 8
    elif total_tiles <= 256:
        v = total_tiles // 32
    elif total_tiles <= 1024:
        v = total_tiles // 64
    else:
        v = 1
    return max(v, 1)


def legacy_to_indexed(legacy):
    """TriangleLegacy::to_indexed — shared/src/lib.rs:715-748 (bit-exact vertex dedup, first-seen order)."""
    verts, tris, seen = [], [], {}
    for v0, v1, v2, mat in legacy:
        idx = []
        for v in (v0, v1, v2):
            key = np.asarray(v, dtype=np.float32).tobytes()
            # PartialEq on f32 treats -0.0 == 0.0; normalise the key for that one case
            key = (np.asarray(v, dtype=np.float32) + np.float32(0.0)).tobytes()
            if key not in seen:
                seen[key] = len(verts)
                verts.append(tuple(np.asarray(v, dtype=np.float32)))
            idx.append(seen[key])
        tris.append((idx[0], idx[1], idx[2], mat))
    va = np.zeros(len(verts), dtype=T.VERTEX)
    if verts:
        va["position"] = np.asarray(verts, dtype=np.float32)
    ta = np.array(tris, dtype=T.TRIANGLE) if tris else np.zeros(0, dtype=T.TRIANGLE)
    return va, ta


def pack_scene_metadata(spheres, lights, bvh_nodes, tri_indices, vertices):
    """BufferManager::update_scene_metadata's concatenation + offsets (src/buffers.rs:213-268)."""
    parts = [np.ascontiguousarray(spheres).view(np.uint32).ravel(),
             np.ascontiguousarray(lights).view(np.uint32).ravel(),
             np.ascontiguousarray(bvh_nodes).view(np.uint32).ravel(),
             np.ascontiguousarray(tri_indices, dtype=np.uint32).ravel(),
             np.ascontiguousarray(vertices).view(np.uint32).ravel()]
    md = np.concatenate(parts) if sum(p.size for p in parts) else np.zeros(0, dtype=np.uint32)
    off = np.zeros((), dtype=T.SCENE_METADATA_OFFSETS)
    o = 0
    for name, part, count in (("spheres", parts[0], len(spheres)), ("lights", parts[1], len(lights)),
                              ("bvh_nodes", parts[2], len(bvh_nodes)),
                              ("triangle_indices", parts[3], len(tri_indices)),
                              ("vertices", parts[4], len(vertices))):
        off[name + "_offset"] = o
        off[name + "_count"] = count
        o += part.size
    return md, off


def split_triangles(triangles, per_buffer=T.REF_TRIANGLES_PER_BUFFER):
    """BufferManager::update_triangles' split (src/buffers.rs:274-336): up to three buffers."""
    bufs = []
    for i in range(3):
        bufs.append(np.ascontiguousarray(triangles[i * per_buffer:(i + 1) * per_buffer]))
    return bufs
