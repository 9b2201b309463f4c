// half.h — IEEE binary16 <-> binary32 on the host.
// f16 -> f32 is exact (what spirv-std's f16_to_f32 / UnpackHalf2x16 does,
// shader/src/material.rs:26-38); f32 -> f16 rounds to nearest even (half::f16::from_f32,
// shared/src/lib.rs:250-252).
#ifndef RT_HALF_H
#define RT_HALF_H
#include <cstdint>
#include <cstring>

namespace rt {

inline float f16_bits_to_f32(uint16_t h) {
    uint32_t sign = (uint32_t)(h >> 15) << 31, exp = (h >> 10) & 0x1F, frac = h & 0x3FF, bits;
    if (exp == 0) {
        if (frac == 0) {
            bits = sign;
        } else { // subnormal: normalise
            int e = -1;
            do {
                e++;
                frac <<= 1;
            } while (!(frac & 0x400));
            bits = sign | ((uint32_t)(112 - e) << 23) | ((frac & 0x3FF) << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (frac << 13);
    } else {
        bits = sign | ((exp + 112) << 23) | (frac << 13);
    }
    float f;
    std::memcpy(&f, &bits, 4);
    return f;
}

inline uint16_t f32_to_f16_bits(float v) {
    uint32_t u;
    std::memcpy(&u, &v, 4);
    uint32_t sign = (u >> 16) & 0x8000u, absu = u & 0x7FFFFFFFu;
    if (absu > 0x7F800000u) return (uint16_t)(sign | 0x7E00u | ((absu >> 13) & 0x3FFu)); // NaN
    if (absu >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);                           // overflow -> inf
    if (absu < 0x33000001u) return (uint16_t)sign;                                        // underflow -> 0
    int32_t e = (int32_t)(absu >> 23) - 127;
    uint32_t m = (absu & 0x7FFFFFu) | 0x800000u;
    uint32_t shift = e < -14 ? (uint32_t)(13 + (-14 - e)) : 13u;
    uint32_t half_exp = e < -14 ? 0u : (uint32_t)(e + 15);
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) q++;
    return (uint16_t)(sign | (half_exp == 0 ? q : ((half_exp << 10) + (q - 0x400u))));
}

} // namespace rt
#endif
