"""Generates the hand-authored glTF fixtures under tests/golden/ (SURVEY.md §8d C0):

  cornell12.gltf   Cornell box, 12 triangles in six primitives (floor, ceiling, back, left, right, light quad) of one
                   mesh with an embedded base64 buffer, a KHR_lights_punctual point light, a camera node at (0,0,3.4).
                   The geometry / materials / light / camera are the numbers of gpu_raytracer_amd.scenes.cornell12().
  cornell12.glb    the same asset as a binary GLB
  modes.gltf       one mesh exercising TRIANGLE_STRIP, TRIANGLE_FAN, u8 / u16 / u32 indices, a byteStride, a node
                   hierarchy with translation / rotation / scale and a matrix, all KHR material extensions, a
                   directional and a spot light, an orthographic camera

Run from the repo root:  python tests/golden/make_gltf_fixtures.py
These are inputs authored for this repository (no reference test holds a glTF file)."""
import base64, json, os, struct, sys
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from gpu_raytracer_amd import scenes  # noqa: E402


def pad4(b):
    return b + b"\x00" * (-len(b) % 4)


def cornell():
    s = scenes.cornell12()
    v = s.vertices["position"]
    quads = [s.triangles[i:i + 2] for i in range(0, 12, 2)]
    blob, views, accessors, prims = b"", [], [], []
    for q in quads:
        ids = []
        for t in q:
            ids += [int(t["v0_index"]), int(t["v1_index"]), int(t["v2_index"])]
        uniq = list(dict.fromkeys(ids))
        pos = np.array([v[i] for i in uniq], np.float32)
        idx = np.array([uniq.index(i) for i in ids], np.uint16)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": pos.nbytes})
        blob = pad4(blob + pos.tobytes())
        accessors.append({"bufferView": len(views) - 1, "componentType": 5126, "count": len(pos), "type": "VEC3",
                          "min": pos.min(0).tolist(), "max": pos.max(0).tolist()})
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": idx.nbytes})
        blob = pad4(blob + idx.tobytes())
        accessors.append({"bufferView": len(views) - 1, "componentType": 5123, "count": len(idx), "type": "SCALAR"})
        prims.append({"attributes": {"POSITION": len(accessors) - 2}, "indices": len(accessors) - 1, "material": int(q[0]["material_id"]), "mode": 4})
    mats = [{"pbrMetallicRoughness": {"baseColorFactor": [0.73, 0.73, 0.73, 1.0], "metallicFactor": 0.0, "roughnessFactor": 1.0}},
            {"pbrMetallicRoughness": {"baseColorFactor": [0.65, 0.05, 0.05, 1.0], "metallicFactor": 0.0, "roughnessFactor": 1.0}},
            {"pbrMetallicRoughness": {"baseColorFactor": [0.12, 0.45, 0.15, 1.0], "metallicFactor": 0.0, "roughnessFactor": 1.0}},
            {"pbrMetallicRoughness": {"baseColorFactor": [1.0, 1.0, 1.0, 1.0], "metallicFactor": 0.0, "roughnessFactor": 1.0},
             "emissiveFactor": [1.0, 1.0, 1.0]}]
    doc = {
        "asset": {"version": "2.0", "generator": "tests/golden/make_gltf_fixtures.py"},
        "extensionsUsed": ["KHR_lights_punctual"],
        "extensions": {"KHR_lights_punctual": {"lights": [{"type": "point", "color": [1.0, 1.0, 1.0], "intensity": 1.0}]}},
        "scene": 0, "scenes": [{"nodes": [0, 1, 2]}],
        "nodes": [{"mesh": 0, "name": "box"},
                  {"camera": 0, "translation": [0.0, 0.0, 3.4], "name": "camera"},
                  {"translation": [0.0, 0.9, 0.0], "extensions": {"KHR_lights_punctual": {"light": 0}}, "name": "light"}],
        "cameras": [{"type": "perspective", "perspective": {"yfov": 0.7853981633974483, "znear": 0.01, "aspectRatio": 1.0}}],
        "meshes": [{"primitives": prims}], "materials": mats, "accessors": accessors, "bufferViews": views,
    }
    return doc, blob


def write(doc, blob, stem):
    d = dict(doc)
    d["buffers"] = [{"byteLength": len(blob), "uri": "data:application/octet-stream;base64," + base64.b64encode(blob).decode()}]
    json.dump(d, open(os.path.join(HERE, stem + ".gltf"), "w"), indent=1)
    g = dict(doc)
    g["buffers"] = [{"byteLength": len(blob)}]
    js = pad4(json.dumps(g, separators=(",", ":")).encode()).replace(b"\x00", b" ")
    js = js + b" " * (-len(js) % 4)
    bn = pad4(blob)
    total = 12 + 8 + len(js) + 8 + len(bn)
    with open(os.path.join(HERE, stem + ".glb"), "wb") as f:
        f.write(b"glTF" + struct.pack("<II", 2, total))
        f.write(struct.pack("<II", len(js), 0x4E4F534A) + js)
        f.write(struct.pack("<II", len(bn), 0x004E4942) + bn)


def modes():
    blob, views, accessors = b"", [], []

    def add(arr, comp, typ, stride=None):
        nonlocal blob
        data = arr.tobytes()
        if stride:  # interleave with padding to exercise byteStride
            elem = arr.dtype.itemsize * arr.shape[1]
            data = b"".join(arr[i].tobytes() + b"\xAB" * (stride - elem) for i in range(len(arr)))
        view = {"buffer": 0, "byteOffset": len(blob), "byteLength": len(data)}
        if stride:
            view["byteStride"] = stride
        views.append(view)
        blob = pad4(blob + data)
        accessors.append({"bufferView": len(views) - 1, "componentType": comp, "count": len(arr), "type": typ})
        return len(accessors) - 1

    strip = np.array([[0, 0, 0], [0, 1, 0], [1, 0, 0], [1, 1, 0], [2, 0, 0], [2, 1, 0]], np.float32)
    fan = np.array([[0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1.5, 1], [-1, 1, 1]], np.float32)
    quad = np.array([[0, 0, 2], [1, 0, 2], [1, 1, 2], [0, 1, 2]], np.float32)
    a_strip, a_fan = add(strip, 5126, "VEC3", stride=20), add(fan, 5126, "VEC3")
    a_quad = add(quad, 5126, "VEC3")
    i8 = add(np.array([[0], [1], [2], [0], [2], [3]], np.uint8), 5121, "SCALAR")
    i16 = add(np.array([[0], [1], [2], [0], [2], [3]], np.uint16), 5123, "SCALAR")
    i32 = add(np.array([[0], [1], [2], [0], [2], [3], [1]], np.uint32), 5125, "SCALAR")  # trailing index: incomplete triangle dropped
    soup = add(np.array([[0, 0, 3], [1, 0, 3], [0, 1, 3], [5, 5, 5]], np.float32), 5126, "VEC3")  # non-indexed: 1 triangle + 1 dangling vertex
    prims = [{"attributes": {"POSITION": a_strip}, "mode": 5, "material": 0},
             {"attributes": {"POSITION": a_fan}, "mode": 6, "material": 1},
             {"attributes": {"POSITION": a_quad}, "indices": i8, "material": 2},
             {"attributes": {"POSITION": a_quad}, "indices": i16, "material": 3},
             {"attributes": {"POSITION": a_quad}, "indices": i32},
             {"attributes": {"POSITION": soup}, "mode": 4, "material": 99},
             {"attributes": {"POSITION": a_quad}, "mode": 1}]  # LINES: skipped
    mats = [{"pbrMetallicRoughness": {"baseColorFactor": [0.2, 0.4, 0.6, 1.0], "metallicFactor": 0.75, "roughnessFactor": 0.25},
             "emissiveFactor": [0.1, 0.2, 0.3],
             "extensions": {"KHR_materials_transmission": {"transmissionFactor": 0.6}, "KHR_materials_ior": {"ior": 1.33},
                            "KHR_materials_specular": {"specularFactor": 0.5, "specularColorFactor": [0.9, 0.8, 0.7]},
                            "KHR_materials_volume": {"thicknessFactor": 0.2, "attenuationDistance": 4.0, "attenuationColor": [0.5, 0.6, 0.7]}}},
            {"extensions": {"KHR_materials_pbrSpecularGlossiness": {"diffuseFactor": [0.3, 0.2, 0.1, 1.0], "specularFactor": [0.4, 0.5, 0.6], "glossinessFactor": 0.7}}},
            {},
            {"pbrMetallicRoughness": {"baseColorTexture": {"index": 2}, "metallicRoughnessTexture": {"index": 0}}, "normalTexture": {"index": 1}, "emissiveTexture": {"index": 3}}]
    doc = {
        "asset": {"version": "2.0"},
        "extensions": {"KHR_lights_punctual": {"lights": [{"type": "directional", "color": [1.0, 0.9, 0.8], "intensity": 2.0},
                                                            {"type": "spot", "intensity": 3.0, "range": 25.0, "spot": {"innerConeAngle": 0.2, "outerConeAngle": 0.6}}]}},
        "scenes": [{"nodes": [3]}, {"nodes": [0]}],
        "nodes": [{"mesh": 0, "translation": [1.0, 2.0, 3.0], "rotation": [0.0, 0.7071067811865476, 0.0, 0.7071067811865476], "scale": [2.0, 1.0, 0.5],
                   "children": [1, 2]},
                  {"camera": 0, "matrix": [1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0.5, 0.25, 4.0, 1]},
                  {"extensions": {"KHR_lights_punctual": {"light": 1}}, "translation": [0.0, 3.0, 0.0], "children": [4]},
                  {"children": [0], "scale": [1.0, 1.0, 1.0]},
                  {"extensions": {"KHR_lights_punctual": {"light": 0}}, "rotation": [0.3826834323650898, 0.0, 0.0, 0.9238795325112867]}],
        "cameras": [{"type": "orthographic", "orthographic": {"xmag": 1.0, "ymag": 1.0, "zfar": 10.0, "znear": 0.1}}],
        "meshes": [{"primitives": prims}], "materials": mats, "accessors": accessors, "bufferViews": views,
    }
    return doc, blob


if __name__ == "__main__":
    write(*cornell(), "cornell12")
    doc, blob = modes()
    d = dict(doc)
    d["buffers"] = [{"byteLength": len(blob), "uri": "modes.bin"}]
    json.dump(d, open(os.path.join(HERE, "modes.gltf"), "w"), indent=1)
    open(os.path.join(HERE, "modes.bin"), "wb").write(blob)
    os.remove(os.path.join(HERE, "cornell12.glb")) if False else None
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith((".gltf", ".glb", ".bin"))))
