// image_io.hpp — write the combined rgba8 image to disk (row N3 of SURVEY.md §8f).  The reference has no image
// output (its PLAN.md lists "save image" as future work); this exists so parity diffs can be looked at.
// PPM (P6) and PNG (8-bit RGBA, stored-deflate: no compression library needed) for the combined rgba8 image;
// OpenEXR (scanline, uncompressed, 32-bit float R G B) for the float image, so float parity diffs keep every bit.
#ifndef RT_IMAGE_IO_HPP
#define RT_IMAGE_IO_HPP

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

namespace raytracer {

inline bool write_ppm(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    std::fprintf(f, "P6\n%u %u\n255\n", w, h);
    std::vector<uint8_t> row(3 * (size_t)w);
    for (uint32_t y = 0; y < h; y++) {
        for (uint32_t x = 0; x < w; x++) {
            const uint8_t* p = rgba8 + 4 * ((size_t)y * w + x);
            row[3 * x] = p[0];
            row[3 * x + 1] = p[1];
            row[3 * x + 2] = p[2];
        }
        if (std::fwrite(row.data(), 1, row.size(), f) != row.size()) {
            std::fclose(f);
            return false;
        }
    }
    return std::fclose(f) == 0;
}

namespace png_detail {
inline uint32_t crc32(const uint8_t* d, size_t n, uint32_t crc = 0) {
    static uint32_t table[256];
    static bool init = false;
    if (!init) {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        init = true;
    }
    crc = ~crc;
    for (size_t i = 0; i < n; i++) crc = table[(crc ^ d[i]) & 0xFF] ^ (crc >> 8);
    return ~crc;
}
inline void be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24));
    v.push_back((uint8_t)(x >> 16));
    v.push_back((uint8_t)(x >> 8));
    v.push_back((uint8_t)x);
}
inline void chunk(std::vector<uint8_t>& out, const char type[4], const std::vector<uint8_t>& data) {
    be32(out, (uint32_t)data.size());
    size_t start = out.size();
    out.insert(out.end(), type, type + 4);
    out.insert(out.end(), data.begin(), data.end());
    be32(out, crc32(out.data() + start, out.size() - start));
}
} // namespace png_detail

inline bool write_png(const char* path, const uint8_t* rgba8, uint32_t w, uint32_t h) {
    using namespace png_detail;
    std::vector<uint8_t> raw; // filter byte 0 + row
    raw.reserve((size_t)h * (4 * (size_t)w + 1));
    for (uint32_t y = 0; y < h; y++) {
        raw.push_back(0);
        raw.insert(raw.end(), rgba8 + 4 * (size_t)y * w, rgba8 + 4 * ((size_t)y + 1) * w);
    }
    std::vector<uint8_t> z = {0x78, 0x01}; // zlib header, stored blocks
    uint32_t a = 1, b = 0;                 // adler32
    for (uint8_t c : raw) {
        a = (a + c) % 65521u;
        b = (b + a) % 65521u;
    }
    size_t pos = 0;
    do {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((uint8_t)(n & 0xFF));
        z.push_back((uint8_t)(n >> 8));
        z.push_back((uint8_t)(~n & 0xFF));
        z.push_back((uint8_t)((~n >> 8) & 0xFF));
        z.insert(z.end(), raw.begin() + (long)pos, raw.begin() + (long)(pos + n));
        pos += n;
    } while (pos < raw.size());
    be32(z, (b << 16) | a);
    std::vector<uint8_t> out = {0x89, 'P', 'N', 'G', 0x0D, 0x0A, 0x1A, 0x0A};
    std::vector<uint8_t> ihdr;
    be32(ihdr, w);
    be32(ihdr, h);
    ihdr.insert(ihdr.end(), {8, 6, 0, 0, 0}); // 8 bit, RGBA
    chunk(out, "IHDR", ihdr);
    chunk(out, "IDAT", z);
    chunk(out, "IEND", {});
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    bool ok = std::fwrite(out.data(), 1, out.size(), f) == out.size();
    return (std::fclose(f) == 0) && ok;
}

// OpenEXR 2 single-part scanline file, NO_COMPRESSION, three FLOAT channels (stored alphabetically: B, G, R),
// one scanline per block, increasing y.  `rgb32f` is packed R G B per pixel, row-major (what rt_read_rgb32f returns).
namespace exr_detail {
inline void le32(std::vector<uint8_t>& v, uint32_t x) {
    for (int k = 0; k < 4; k++) v.push_back((uint8_t)(x >> (8 * k)));
}
inline void le64(std::vector<uint8_t>& v, uint64_t x) {
    for (int k = 0; k < 8; k++) v.push_back((uint8_t)(x >> (8 * k)));
}
inline void str(std::vector<uint8_t>& v, const char* s) {
    while (*s) v.push_back((uint8_t)*s++);
    v.push_back(0);
}
inline void f32(std::vector<uint8_t>& v, float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    le32(v, u);
}
inline void attr(std::vector<uint8_t>& v, const char* name, const char* type, const std::vector<uint8_t>& value) {
    str(v, name);
    str(v, type);
    le32(v, (uint32_t)value.size());
    v.insert(v.end(), value.begin(), value.end());
}
} // namespace exr_detail

inline bool write_exr(const char* path, const float* rgb32f, uint32_t w, uint32_t h) {
    using namespace exr_detail;
    if (!w || !h || w > 0x7FFFFFFFu || h > 0x7FFFFFFFu) return false;
    std::vector<uint8_t> head = {0x76, 0x2F, 0x31, 0x01, 2, 0, 0, 0}; // magic, version 2, no flags
    std::vector<uint8_t> v;
    for (const char* c : {"B", "G", "R"}) {
        str(v, c);
        le32(v, 2); // FLOAT
        v.insert(v.end(), {0, 0, 0, 0}); // pLinear + reserved
        le32(v, 1);
        le32(v, 1); // x / y sampling
    }
    v.push_back(0);
    attr(head, "channels", "chlist", v);
    attr(head, "compression", "compression", {0});
    v.clear();
    le32(v, 0);
    le32(v, 0);
    le32(v, w - 1);
    le32(v, h - 1);
    attr(head, "dataWindow", "box2i", v);
    attr(head, "displayWindow", "box2i", v);
    attr(head, "lineOrder", "lineOrder", {0});
    v.clear();
    f32(v, 1.0f);
    attr(head, "pixelAspectRatio", "float", v);
    attr(head, "screenWindowWidth", "float", v);
    v.clear();
    f32(v, 0.0f);
    f32(v, 0.0f);
    attr(head, "screenWindowCenter", "v2f", v);
    head.push_back(0); // end of header
    const uint64_t row_bytes = 12ull * w, block = 8 + row_bytes, first = head.size() + 8ull * h;
    for (uint32_t y = 0; y < h; y++) le64(head, first + (uint64_t)y * block);
    FILE* f = std::fopen(path, "wb");
    if (!f) return false;
    bool ok = std::fwrite(head.data(), 1, head.size(), f) == head.size();
    std::vector<uint8_t> row;
    row.reserve((size_t)block);
    for (uint32_t y = 0; y < h && ok; y++) {
        row.clear();
        le32(row, y);
        le32(row, (uint32_t)row_bytes);
        for (int c = 2; c >= 0; c--) // B, G, R planes of this scanline
            for (uint32_t x = 0; x < w; x++) f32(row, rgb32f[3 * ((size_t)y * w + x) + c]);
        ok = std::fwrite(row.data(), 1, row.size(), f) == row.size();
    }
    return (std::fclose(f) == 0) && ok;
}

} // namespace raytracer
#endif
