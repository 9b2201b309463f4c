// image_io.hpp — write the combined rgba8 image to disk (row N3 of SURVEY.md §8f).  The reference has no image
// output (its PLAN.md lists "save image" as future work); this exists so parity diffs can be looked at.
// PPM (P6) and PNG (8-bit RGBA, stored-deflate: no compression library needed).
#ifndef RT_IMAGE_IO_HPP
#define RT_IMAGE_IO_HPP

#include <cstdint>
#include <cstdio>
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

} // namespace raytracer
#endif
