// gltf_loader.hpp — glTF 2.0 / GLB loader with the behaviour of the reference's src/gltf_loader.rs
// (row N1 of SURVEY.md §8f).  The reference leans on the `gltf` and `image` crates; neither exists
// here, so this is a self-contained reader: a small JSON parser, base64 / external .bin / GLB buffers,
// node hierarchy, and the reference's extraction rules:
//   extract_scene            :77-125   default scene (or first), materials first, nodes depth-first
//   process_node             :187-227  transform = parent * local; mesh, camera, light, then children
//   convert_camera           :230-250  position = T*0, direction = normalize(T*(-Z)), up = normalize(T*Y),
//                                      fov = yfov in DEGREES, orthographic -> 45
//   convert_light            :253-284  KHR_lights_punctual: directional / point / spot, range default inf
//   process_primitive        :287-394  Triangles (indexed or not), TriangleFan, TriangleStrip (alternating
//                                      winding); vertices deduplicated PER PRIMITIVE by exact bit pattern of
//                                      the TRANSFORMED position, first-seen order; material default 0
//   convert_material         :397-489  metallic-roughness or KHR_materials_pbrSpecularGlossiness, emissive,
//                                      KHR_materials_transmission / ior / specular / volume, texture indices
//   get_accessor_data        :499-542  positions: byteStride honoured, must be 12-byte VEC3 float
//   get_indices_data         :546-594  u8 / u16 / u32, tightly packed
// Textures are parsed for their indices only: the reference loads them but its kernel never samples them
// (shader/src/lib.rs:34-35 `_textures`, `_texture_data`).
#ifndef RT_GLTF_LOADER_HPP
#define RT_GLTF_LOADER_HPP

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "raytracer_host.hpp"

namespace raytracer {

// GltfError — src/gltf_loader.rs:16-21
struct GltfError {
    enum Kind { None = 0, IoError, GltfError_, ValidationError, ImageError } kind = None;
    std::string message;
    explicit operator bool() const { return kind != None; }
};

// LoadedScene — src/gltf_loader.rs:42-52
struct LoadedScene {
    std::vector<Triangle> triangles;
    std::vector<Vertex> vertices;
    std::vector<Material> materials;
    std::vector<Sphere> spheres;
    std::vector<Light> lights;
    std::vector<Camera> cameras;
    std::vector<rt_texture_info> textures;
    std::vector<uint8_t> texture_data;
};

namespace json {

struct Value;
using Object = std::map<std::string, Value>;
using Array = std::vector<Value>;
struct Value {
    enum Type { Null, Bool, Number, String, Arr, Obj } type = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::shared_ptr<Array> arr;
    std::shared_ptr<Object> obj;

    bool is_object() const { return type == Obj; }
    bool is_array() const { return type == Arr; }
    bool is_number() const { return type == Number; }
    const Value* get(const char* key) const {
        if (type != Obj) return nullptr;
        auto it = obj->find(key);
        return it == obj->end() ? nullptr : &it->second;
    }
    size_t size() const { return type == Arr ? arr->size() : 0; }
    const Value& at(size_t i) const { return (*arr)[i]; }
    double number_or(double d) const { return type == Number ? num : d; }
};

class Parser {
  public:
    Parser(const char* p, size_t n) : p_(p), end_(p + n) {}
    bool parse(Value& out, std::string& err) {
        skip();
        if (!value(out, err, 0)) return false;
        skip();
        if (p_ != end_) {
            err = "trailing characters after JSON document";
            return false;
        }
        return true;
    }

  private:
    const char *p_, *end_;
    void skip() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\t' || *p_ == '\n' || *p_ == '\r')) p_++;
    }
    bool fail(std::string& err, const char* m) {
        err = m;
        return false;
    }
    bool value(Value& v, std::string& err, int depth) {
        if (depth > 256) return fail(err, "JSON nested too deeply");
        if (p_ >= end_) return fail(err, "unexpected end of JSON");
        switch (*p_) {
            case '{': {
                p_++;
                v.type = Value::Obj;
                v.obj = std::make_shared<Object>();
                skip();
                if (p_ < end_ && *p_ == '}') { p_++; return true; }
                for (;;) {
                    skip();
                    Value key;
                    if (p_ >= end_ || *p_ != '"' || !string(key.str, err)) return fail(err, "expected object key");
                    skip();
                    if (p_ >= end_ || *p_ != ':') return fail(err, "expected ':'");
                    p_++;
                    skip();
                    Value child;
                    if (!value(child, err, depth + 1)) return false;
                    (*v.obj)[key.str] = std::move(child);
                    skip();
                    if (p_ < end_ && *p_ == ',') { p_++; continue; }
                    if (p_ < end_ && *p_ == '}') { p_++; return true; }
                    return fail(err, "expected ',' or '}'");
                }
            }
            case '[': {
                p_++;
                v.type = Value::Arr;
                v.arr = std::make_shared<Array>();
                skip();
                if (p_ < end_ && *p_ == ']') { p_++; return true; }
                for (;;) {
                    skip();
                    Value child;
                    if (!value(child, err, depth + 1)) return false;
                    v.arr->push_back(std::move(child));
                    skip();
                    if (p_ < end_ && *p_ == ',') { p_++; continue; }
                    if (p_ < end_ && *p_ == ']') { p_++; return true; }
                    return fail(err, "expected ',' or ']'");
                }
            }
            case '"':
                v.type = Value::String;
                return string(v.str, err);
            case 't':
                if (end_ - p_ >= 4 && !std::strncmp(p_, "true", 4)) { p_ += 4; v.type = Value::Bool; v.b = true; return true; }
                return fail(err, "bad literal");
            case 'f':
                if (end_ - p_ >= 5 && !std::strncmp(p_, "false", 5)) { p_ += 5; v.type = Value::Bool; v.b = false; return true; }
                return fail(err, "bad literal");
            case 'n':
                if (end_ - p_ >= 4 && !std::strncmp(p_, "null", 4)) { p_ += 4; v.type = Value::Null; return true; }
                return fail(err, "bad literal");
            default: {
                const char* s = p_;
                if (p_ < end_ && (*p_ == '-' || *p_ == '+')) p_++;
                while (p_ < end_ && ((*p_ >= '0' && *p_ <= '9') || *p_ == '.' || *p_ == 'e' || *p_ == 'E' || *p_ == '-' || *p_ == '+')) p_++;
                if (p_ == s) return fail(err, "unexpected character in JSON");
                v.type = Value::Number;
                v.num = std::strtod(std::string(s, p_).c_str(), nullptr);
                return true;
            }
        }
    }
    bool string(std::string& out, std::string& err) {
        p_++; // opening quote
        out.clear();
        while (p_ < end_ && *p_ != '"') {
            if (*p_ == '\\') {
                if (++p_ >= end_) return fail(err, "bad escape");
                switch (*p_) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': { // BMP code point -> UTF-8 (names only; never used for data)
                        if (end_ - p_ < 5) return fail(err, "bad \\u escape");
                        unsigned cp = (unsigned)std::strtoul(std::string(p_ + 1, p_ + 5).c_str(), nullptr, 16);
                        p_ += 4;
                        if (cp < 0x80) out += (char)cp;
                        else if (cp < 0x800) { out += (char)(0xC0 | (cp >> 6)); out += (char)(0x80 | (cp & 0x3F)); }
                        else { out += (char)(0xE0 | (cp >> 12)); out += (char)(0x80 | ((cp >> 6) & 0x3F)); out += (char)(0x80 | (cp & 0x3F)); }
                        break;
                    }
                    default: out += *p_; break; // \" \\ \/
                }
                p_++;
            } else {
                out += *p_++;
            }
        }
        if (p_ >= end_) return fail(err, "unterminated string");
        p_++;
        return true;
    }
};

} // namespace json

// 4x4 column-major matrix with glam's operation order (Mat4::mul, transform_point3, transform_vector3)
struct Mat4 {
    float c[4][4]; // c[col][row]
    static Mat4 identity() {
        Mat4 m;
        std::memset(&m, 0, sizeof m);
        m.c[0][0] = m.c[1][1] = m.c[2][2] = m.c[3][3] = 1.0f;
        return m;
    }
    Mat4 mul(const Mat4& r) const { // self * r: column j = self.x*r[j].x + self.y*r[j].y + self.z*r[j].z + self.w*r[j].w
        Mat4 o;
        for (int j = 0; j < 4; j++)
            for (int i = 0; i < 4; i++) o.c[j][i] = ((c[0][i] * r.c[j][0] + c[1][i] * r.c[j][1]) + c[2][i] * r.c[j][2]) + c[3][i] * r.c[j][3];
        return o;
    }
    void transform_point3(const float v[3], float out[3]) const {
        for (int i = 0; i < 3; i++) out[i] = c[3][i] + (c[2][i] * v[2] + (c[1][i] * v[1] + c[0][i] * v[0]));
    }
    void transform_vector3(const float v[3], float out[3]) const {
        for (int i = 0; i < 3; i++) out[i] = c[2][i] * v[2] + (c[1][i] * v[1] + c[0][i] * v[0]);
    }
};

class GltfLoader {
  public:
    // load_from_path :55-63 (.gltf with external / data-URI buffers, or .glb)
    static GltfError load_from_path(const std::string& path, GltfLoader& out) {
        std::vector<uint8_t> bytes;
        if (!read_file(path, bytes)) return {GltfError::IoError, "cannot read " + path};
        std::string dir;
        size_t slash = path.find_last_of('/');
        if (slash != std::string::npos) dir = path.substr(0, slash + 1);
        if (bytes.size() >= 4 && !std::memcmp(bytes.data(), "glTF", 4)) return out.parse_glb(bytes.data(), bytes.size(), dir);
        return out.parse_json(reinterpret_cast<const char*>(bytes.data()), bytes.size(), dir, nullptr, 0);
    }
    // load_from_glb :66-74
    static GltfError load_from_glb(const uint8_t* data, size_t len, GltfLoader& out) { return out.parse_glb(data, len, ""); }

    size_t scene_count() const { return arr_size("scenes"); } // :606-608

    // extract_scene :77-125
    GltfError extract_scene(int scene_index, LoadedScene& out) const {
        out = LoadedScene();
        const json::Value* scenes = doc_.get("scenes");
        const json::Value* scene = nullptr;
        if (scene_index >= 0) {
            if (!scenes || (size_t)scene_index >= scenes->size()) return {GltfError::ValidationError, "Scene " + std::to_string(scene_index) + " not found"};
            scene = &scenes->at((size_t)scene_index);
        } else {
            const json::Value* def = doc_.get("scene");
            if (def && def->is_number() && scenes && (size_t)def->num < scenes->size()) scene = &scenes->at((size_t)def->num);
            else if (scenes && scenes->size() > 0) scene = &scenes->at(0);
            else return {GltfError::ValidationError, "No scenes found in glTF file"};
        }
        const json::Value* mats = doc_.get("materials");
        for (size_t i = 0; mats && i < mats->size(); i++) out.materials.push_back(convert_material(mats->at(i)));
        const json::Value* roots = scene->get("nodes");
        for (size_t i = 0; roots && i < roots->size(); i++) {
            GltfError e = process_node((size_t)roots->at(i).number_or(-1), Mat4::identity(), out, 0);
            if (e) return e;
        }
        return {};
    }

  private:
    json::Value doc_;
    std::vector<std::vector<uint8_t>> buffers_;

    static bool read_file(const std::string& path, std::vector<uint8_t>& out) {
        std::ifstream f(path, std::ios::binary);
        if (!f) return false;
        f.seekg(0, std::ios::end);
        std::streamoff n = f.tellg();
        if (n < 0) return false;
        f.seekg(0);
        out.resize((size_t)n);
        if (n) f.read(reinterpret_cast<char*>(out.data()), n);
        return (bool)f || n == 0;
    }
    static bool base64_decode(const std::string& s, size_t from, std::vector<uint8_t>& out) {
        auto val = [](char ch) -> int {
            if (ch >= 'A' && ch <= 'Z') return ch - 'A';
            if (ch >= 'a' && ch <= 'z') return ch - 'a' + 26;
            if (ch >= '0' && ch <= '9') return ch - '0' + 52;
            if (ch == '+' || ch == '-') return 62;
            if (ch == '/' || ch == '_') return 63;
            return -1;
        };
        uint32_t acc = 0;
        int bits = 0;
        out.clear();
        for (size_t i = from; i < s.size(); i++) {
            if (s[i] == '=' || s[i] == '\n' || s[i] == '\r') continue;
            int v = val(s[i]);
            if (v < 0) return false;
            acc = (acc << 6) | (uint32_t)v;
            bits += 6;
            if (bits >= 8) {
                bits -= 8;
                out.push_back((uint8_t)((acc >> bits) & 0xFF));
            }
        }
        return true;
    }
    size_t arr_size(const char* key) const {
        const json::Value* a = doc_.get(key);
        return a ? a->size() : 0;
    }
    const json::Value* item(const char* key, size_t i) const {
        const json::Value* a = doc_.get(key);
        return a && i < a->size() ? &a->at(i) : nullptr;
    }

    GltfError parse_glb(const uint8_t* d, size_t n, const std::string& dir) {
        if (n < 20 || std::memcmp(d, "glTF", 4)) return {GltfError::GltfError_, "not a GLB file"};
        uint32_t version, total;
        std::memcpy(&version, d + 4, 4);
        std::memcpy(&total, d + 8, 4);
        if (version != 2 || total > n) return {GltfError::GltfError_, "unsupported GLB version or truncated file"};
        size_t off = 12;
        const char* js = nullptr;
        size_t js_len = 0;
        const uint8_t* bin = nullptr;
        size_t bin_len = 0;
        while (off + 8 <= total) {
            uint32_t clen, ctype;
            std::memcpy(&clen, d + off, 4);
            std::memcpy(&ctype, d + off + 4, 4);
            off += 8;
            if (off + clen > total) return {GltfError::GltfError_, "GLB chunk exceeds file"};
            if (ctype == 0x4E4F534Au) { js = reinterpret_cast<const char*>(d + off); js_len = clen; }
            else if (ctype == 0x004E4942u && !bin) { bin = d + off; bin_len = clen; }
            off += (clen + 3u) & ~3u;
        }
        if (!js) return {GltfError::GltfError_, "GLB without JSON chunk"};
        return parse_json(js, js_len, dir, bin, bin_len);
    }

    GltfError parse_json(const char* js, size_t len, const std::string& dir, const uint8_t* glb_bin, size_t glb_bin_len) {
        std::string err;
        json::Parser p(js, len);
        if (!p.parse(doc_, err) || !doc_.is_object()) return {GltfError::GltfError_, "JSON: " + err};
        const json::Value* asset = doc_.get("asset");
        if (!asset || !asset->get("version")) return {GltfError::GltfError_, "missing asset.version"};
        const json::Value* bufs = doc_.get("buffers");
        buffers_.clear();
        for (size_t i = 0; bufs && i < bufs->size(); i++) {
            const json::Value& b = bufs->at(i);
            std::vector<uint8_t> data;
            const json::Value* uri = b.get("uri");
            if (uri && uri->type == json::Value::String) {
                const std::string& u = uri->str;
                if (u.compare(0, 5, "data:") == 0) {
                    size_t comma = u.find(',');
                    if (comma == std::string::npos || !base64_decode(u, comma + 1, data)) return {GltfError::GltfError_, "bad data URI in buffer " + std::to_string(i)};
                } else if (!read_file(dir + u, data)) {
                    return {GltfError::IoError, "cannot read buffer " + dir + u};
                }
            } else if (i == 0 && glb_bin) {
                data.assign(glb_bin, glb_bin + glb_bin_len);
            } else {
                return {GltfError::GltfError_, "buffer " + std::to_string(i) + " has no data"};
            }
            size_t want = (size_t)(b.get("byteLength") ? b.get("byteLength")->number_or(0) : 0);
            if (data.size() < want) return {GltfError::GltfError_, "buffer " + std::to_string(i) + " shorter than byteLength"};
            buffers_.push_back(std::move(data));
        }
        return {};
    }

    // node.transform().matrix(): explicit matrix, or T * R * S
    static Mat4 node_matrix(const json::Value& node) {
        Mat4 m = Mat4::identity();
        const json::Value* mat = node.get("matrix");
        if (mat && mat->size() == 16) {
            for (int j = 0; j < 4; j++)
                for (int i = 0; i < 4; i++) m.c[j][i] = (float)mat->at((size_t)(j * 4 + i)).number_or(0);
            return m;
        }
        float t[3] = {0, 0, 0}, r[4] = {0, 0, 0, 1}, s[3] = {1, 1, 1};
        if (const json::Value* v = node.get("translation")) for (size_t i = 0; i < 3 && i < v->size(); i++) t[i] = (float)v->at(i).number_or(0);
        if (const json::Value* v = node.get("rotation")) for (size_t i = 0; i < 4 && i < v->size(); i++) r[i] = (float)v->at(i).number_or(0);
        if (const json::Value* v = node.get("scale")) for (size_t i = 0; i < 3 && i < v->size(); i++) s[i] = (float)v->at(i).number_or(1);
        const float x = r[0], y = r[1], z = r[2], w = r[3];
        const float x2 = x + x, y2 = y + y, z2 = z + z;
        const float xx = x * x2, xy = x * y2, xz = x * z2, yy = y * y2, yz = y * z2, zz = z * z2, wx = w * x2, wy = w * y2, wz = w * z2;
        m.c[0][0] = (1.0f - (yy + zz)) * s[0]; m.c[0][1] = (xy + wz) * s[0]; m.c[0][2] = (xz - wy) * s[0]; m.c[0][3] = 0.0f;
        m.c[1][0] = (xy - wz) * s[1]; m.c[1][1] = (1.0f - (xx + zz)) * s[1]; m.c[1][2] = (yz + wx) * s[1]; m.c[1][3] = 0.0f;
        m.c[2][0] = (xz + wy) * s[2]; m.c[2][1] = (yz - wx) * s[2]; m.c[2][2] = (1.0f - (xx + yy)) * s[2]; m.c[2][3] = 0.0f;
        m.c[3][0] = t[0]; m.c[3][1] = t[1]; m.c[3][2] = t[2]; m.c[3][3] = 1.0f;
        return m;
    }

    static void normalize3(float v[3]) {
        const float inv = 1.0f / std::sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
        v[0] *= inv; v[1] *= inv; v[2] *= inv;
    }

    GltfError process_node(size_t index, const Mat4& parent, LoadedScene& out, int depth) const { // :187-227
        const json::Value* node = item("nodes", index);
        if (!node) return {GltfError::ValidationError, "node index out of range"};
        if (depth > 1024) return {GltfError::ValidationError, "node hierarchy too deep (cycle?)"};
        const Mat4 transform = parent.mul(node_matrix(*node));
        if (const json::Value* mi = node->get("mesh")) {
            const json::Value* mesh = item("meshes", (size_t)mi->number_or(-1));
            if (!mesh) return {GltfError::ValidationError, "mesh index out of range"};
            const json::Value* prims = mesh->get("primitives");
            for (size_t i = 0; prims && i < prims->size(); i++) {
                GltfError e = process_primitive(prims->at(i), transform, out);
                if (e) return e;
            }
        }
        if (const json::Value* ci = node->get("camera")) {
            const json::Value* cam = item("cameras", (size_t)ci->number_or(-1));
            if (cam) out.cameras.push_back(convert_camera(*cam, transform));
        }
        if (const json::Value* ext = node->get("extensions"))
            if (const json::Value* kl = ext->get("KHR_lights_punctual"))
                if (const json::Value* li = kl->get("light")) {
                    const json::Value* root_ext = doc_.get("extensions");
                    const json::Value* lights = root_ext && root_ext->get("KHR_lights_punctual") ? root_ext->get("KHR_lights_punctual")->get("lights") : nullptr;
                    size_t k = (size_t)li->number_or(-1);
                    if (lights && k < lights->size()) out.lights.push_back(convert_light(lights->at(k), transform));
                }
        const json::Value* children = node->get("children");
        for (size_t i = 0; children && i < children->size(); i++) {
            GltfError e = process_node((size_t)children->at(i).number_or(-1), transform, out, depth + 1);
            if (e) return e;
        }
        return {};
    }

    static Camera convert_camera(const json::Value& cam, const Mat4& t) { // :230-250
        Camera c;
        const float zero[3] = {0, 0, 0}, negz[3] = {0, 0, -1.0f}, y[3] = {0, 1.0f, 0};
        t.transform_point3(zero, c.position);
        t.transform_vector3(negz, c.direction);
        normalize3(c.direction);
        t.transform_vector3(y, c.up);
        normalize3(c.up);
        c.fov = 45.0f;
        const json::Value* persp = cam.get("perspective");
        const json::Value* type = cam.get("type");
        if (persp && !(type && type->str == "orthographic")) {
            const float yfov = (float)(persp->get("yfov") ? persp->get("yfov")->number_or(0.7853981633974483) : 0.7853981633974483);
            c.fov = yfov * (180.0f / 3.14159265358979323846f); // f32::to_degrees
        }
        return c;
    }

    static Light convert_light(const json::Value& l, const Mat4& t) { // :253-284
        const float zero[3] = {0, 0, 0}, negz[3] = {0, 0, -1.0f};
        float position[3], direction[3], color[3] = {1.0f, 1.0f, 1.0f};
        t.transform_point3(zero, position);
        t.transform_vector3(negz, direction);
        normalize3(direction);
        if (const json::Value* c = l.get("color")) for (size_t i = 0; i < 3 && i < c->size(); i++) color[i] = (float)c->at(i).number_or(1);
        const float intensity = (float)(l.get("intensity") ? l.get("intensity")->number_or(1) : 1.0);
        const float range = l.get("range") ? (float)l.get("range")->number_or(0) : std::numeric_limits<float>::infinity();
        const std::string type = l.get("type") ? l.get("type")->str : "point";
        if (type == "directional") return light::directional(direction, color, intensity);
        if (type == "spot") {
            const json::Value* s = l.get("spot");
            const float inner = (float)(s && s->get("innerConeAngle") ? s->get("innerConeAngle")->number_or(0) : 0.0);
            const float outer = (float)(s && s->get("outerConeAngle") ? s->get("outerConeAngle")->number_or(0.7853981633974483) : 0.7853981633974483);
            return light::spot(position, direction, color, intensity, range, inner, outer);
        }
        return light::point(position, color, intensity, range);
    }

    static void read3(const json::Value* v, float out[3], float dflt) {
        out[0] = out[1] = out[2] = dflt;
        for (size_t i = 0; v && i < 3 && i < v->size(); i++) out[i] = (float)v->at(i).number_or(dflt);
    }
    static float num(const json::Value* obj, const char* key, double dflt) {
        const json::Value* v = obj ? obj->get(key) : nullptr;
        return (float)(v ? v->number_or(dflt) : dflt);
    }

    static Material convert_material(const json::Value& gm) { // :397-489
        const json::Value* ext = gm.get("extensions");
        Material m;
        const json::Value* sg = ext ? ext->get("KHR_materials_pbrSpecularGlossiness") : nullptr;
        const json::Value* pbr = gm.get("pbrMetallicRoughness");
        if (sg) {
            float diffuse[3], specular[3];
            read3(sg->get("diffuseFactor"), diffuse, 1.0f);
            read3(sg->get("specularFactor"), specular, 1.0f);
            m = material::specular_glossiness(diffuse, specular, num(sg, "glossinessFactor", 1.0));
        } else {
            float albedo[3];
            read3(pbr ? pbr->get("baseColorFactor") : nullptr, albedo, 1.0f);
            const float zero[3] = {0, 0, 0};
            m = material::new_(albedo, num(pbr, "metallicFactor", 1.0), num(pbr, "roughnessFactor", 1.0), zero, 1.5f, 0.0f);
        }
        read3(gm.get("emissiveFactor"), m.emission, 0.0f);
        if (ext) {
            if (const json::Value* t = ext->get("KHR_materials_transmission")) material::set_transmission(m, num(t, "transmissionFactor", 0.0));
            if (const json::Value* i = ext->get("KHR_materials_ior")) material::set_ior(m, num(i, "ior", 1.5));
            if (const json::Value* s = ext->get("KHR_materials_specular")) {
                m.specular_factor = num(s, "specularFactor", 1.0);
                read3(s->get("specularColorFactor"), m.specular_color, 1.0f);
            }
            if (const json::Value* v = ext->get("KHR_materials_volume")) {
                m.thickness_factor = num(v, "thicknessFactor", 0.0);
                m.attenuation_distance = v->get("attenuationDistance") ? (float)v->get("attenuationDistance")->number_or(0) : std::numeric_limits<float>::infinity();
                read3(v->get("attenuationColor"), m.attenuation_color, 1.0f);
            }
        }
        uint32_t tex[8];
        for (auto& t : tex) t = 0xFFFFFFFFu;
        int k = 0;
        auto tex_index = [&](const json::Value* info) {
            if (info && info->get("index") && k < 8) tex[k++] = (uint32_t)info->get("index")->number_or(0);
        };
        tex_index(pbr ? pbr->get("baseColorTexture") : nullptr);
        tex_index(pbr ? pbr->get("metallicRoughnessTexture") : nullptr);
        tex_index(gm.get("normalTexture"));
        tex_index(gm.get("emissiveTexture"));
        std::memcpy(m.texture_indices, tex, sizeof tex);
        return m;
    }

    struct View {
        const uint8_t* data = nullptr;
        size_t len = 0, stride = 0, count = 0, elem = 0;
        int component = 0;
    };
    GltfError accessor_view(size_t index, View& v, const char* what) const {
        const json::Value* acc = item("accessors", index);
        if (!acc) return {GltfError::ValidationError, std::string(what) + ": accessor index out of range"};
        const json::Value* bvi = acc->get("bufferView");
        if (!bvi) return {GltfError::ValidationError, std::string(what) + " accessor missing buffer view"};
        const json::Value* bv = item("bufferViews", (size_t)bvi->number_or(-1));
        if (!bv) return {GltfError::ValidationError, "buffer view index out of range"};
        const size_t bi = (size_t)num(bv, "buffer", -1);
        if (bi >= buffers_.size()) return {GltfError::ValidationError, "buffer index out of range"};
        v.component = (int)num(acc, "componentType", 0);
        const std::string type = acc->get("type") ? acc->get("type")->str : "SCALAR";
        const size_t comps = type == "VEC3" ? 3 : type == "VEC2" ? 2 : type == "VEC4" ? 4 : type == "SCALAR" ? 1 : 16;
        const size_t csize = (v.component == 5120 || v.component == 5121) ? 1 : (v.component == 5122 || v.component == 5123) ? 2 : 4;
        v.elem = comps * csize;
        v.count = (size_t)num(acc, "count", 0);
        const size_t start = (size_t)num(bv, "byteOffset", 0) + (size_t)num(acc, "byteOffset", 0);
        v.stride = bv->get("byteStride") ? (size_t)bv->get("byteStride")->number_or(0) : v.elem;
        if (start > buffers_[bi].size()) return {GltfError::ValidationError, "Buffer access out of bounds"};
        v.data = buffers_[bi].data() + start;
        v.len = buffers_[bi].size() - start;
        return {};
    }

    GltfError process_primitive(const json::Value& prim, const Mat4& transform, LoadedScene& out) const { // :287-394
        uint32_t material_id = 0;
        if (const json::Value* mi = prim.get("material")) {
            const size_t k = (size_t)mi->number_or(0);
            material_id = k < out.materials.size() ? (uint32_t)k : 0u;
        }
        const json::Value* attrs = prim.get("attributes");
        const json::Value* pos = attrs ? attrs->get("POSITION") : nullptr;
        if (!pos) return {GltfError::ValidationError, "Primitive missing position data"};
        View pv;
        if (GltfError e = accessor_view((size_t)pos->number_or(-1), pv, "POSITION")) return e;
        if (pv.elem != 12 || pv.component != 5126) return {GltfError::ValidationError, "POSITION accessor is not VEC3 float"};
        std::vector<float> positions(pv.count * 3);
        for (size_t i = 0; i < pv.count; i++) {
            if (i * pv.stride + 12 > pv.len) return {GltfError::ValidationError, "Buffer access out of bounds"};
            std::memcpy(&positions[3 * i], pv.data + i * pv.stride, 12);
        }
        struct Key {
            uint32_t b[3];
            bool operator==(const Key& o) const { return b[0] == o.b[0] && b[1] == o.b[1] && b[2] == o.b[2]; }
        };
        struct KeyHash {
            size_t operator()(const Key& k) const { return ((size_t)k.b[0] * 0x9E3779B97F4A7C15ull) ^ ((size_t)k.b[1] << 21) ^ ((size_t)k.b[2] * 0xC2B2AE3D27D4EB4Full); }
        };
        std::unordered_map<Key, uint32_t, KeyHash> vertex_map; // exact bit pattern of the transformed position, per primitive
        auto vertex_index = [&](size_t i) -> uint32_t {
            float p[3];
            transform.transform_point3(&positions[3 * i], p);
            Key k;
            std::memcpy(k.b, p, 12);
            auto it = vertex_map.find(k);
            if (it != vertex_map.end()) return it->second;
            const uint32_t idx = (uint32_t)out.vertices.size();
            out.vertices.push_back(Vertex{{p[0], p[1], p[2]}});
            vertex_map.emplace(k, idx);
            return idx;
        };
        const int mode = (int)num(&prim, "mode", 4);
        const size_t npos = pv.count;
        auto tri = [&](uint32_t a, uint32_t b, uint32_t c) { out.triangles.push_back(triangle::new_indexed(a, b, c, material_id)); };
        if (mode == 4) { // Triangles
            if (const json::Value* ia = prim.get("indices")) {
                View iv;
                if (GltfError e = accessor_view((size_t)ia->number_or(-1), iv, "indices")) return e;
                std::vector<uint32_t> idx;
                const size_t csize = iv.component == 5121 ? 1 : iv.component == 5123 ? 2 : iv.component == 5125 ? 4 : 0;
                if (!csize) return {GltfError::ValidationError, "Unsupported index data type"};
                for (size_t i = 0; i < iv.count; i++) { // tightly packed, entries past the buffer end are dropped (:560-588)
                    if ((i + 1) * csize > iv.len) break;
                    uint32_t v = 0;
                    std::memcpy(&v, iv.data + i * csize, csize);
                    idx.push_back(v);
                }
                for (size_t i = 0; i + 3 <= idx.size(); i += 3) {
                    if (idx[i] >= npos || idx[i + 1] >= npos || idx[i + 2] >= npos) return {GltfError::ValidationError, "index exceeds POSITION count"};
                    const uint32_t a = vertex_index(idx[i]), b = vertex_index(idx[i + 1]), c = vertex_index(idx[i + 2]);
                    tri(a, b, c);
                }
            } else {
                for (size_t i = 0; i + 3 <= npos; i += 3) {
                    const uint32_t a = vertex_index(i), b = vertex_index(i + 1), c = vertex_index(i + 2);
                    tri(a, b, c);
                }
            }
        } else if (mode == 6) { // TriangleFan
            if (npos >= 3) {
                const uint32_t center = vertex_index(0);
                for (size_t i = 1; i + 1 < npos; i++) {
                    const uint32_t b = vertex_index(i), c = vertex_index(i + 1);
                    tri(center, b, c);
                }
            }
        } else if (mode == 5) { // TriangleStrip, alternating winding
            for (size_t i = 0; i + 2 < npos; i++) {
                const uint32_t a = vertex_index(i), b = vertex_index(i + 1), c = vertex_index(i + 2);
                if (i % 2 == 0) tri(a, b, c);
                else tri(a, c, b);
            }
        } // other modes: the reference prints a warning and skips the primitive
        return {};
    }
};

// SceneState::load_from_gltf — src/scene.rs:43-69: first camera of the file or Camera::new, then BvhBuilder::build
inline GltfError scene_state_load_from_gltf(const std::string& path, SceneState& out) {
    GltfLoader loader;
    if (GltfError e = GltfLoader::load_from_path(path, loader)) return e;
    LoadedScene ls;
    if (GltfError e = loader.extract_scene(-1, ls)) return e;
    out = SceneState();
    out.camera = ls.cameras.empty() ? camera::new_() : ls.cameras[0];
    out.spheres = std::move(ls.spheres);
    out.triangles = std::move(ls.triangles);
    out.vertices = std::move(ls.vertices);
    out.materials = std::move(ls.materials);
    out.lights = std::move(ls.lights);
    out.rebuild_bvh();
    return {};
}

} // namespace raytracer
#endif
