// pcm_dev.hpp -- device primitives of the reference's wire codec, shared by the stand-alone kernels (pcm_kernels.hip)
// and the fused wire launch (banded_mfma.hip: wire_kernel):
//   quant1          modules/network/websocket.rs:248   (x.clamp(-1.0, 1.0) * 32767.0) as i16
//   b64_*           modules/network/websocket.rs:254   STANDARD.encode (RFC 4648, '=' padding)
#pragma once
#include <hip/hip_runtime.h>

namespace af {
namespace pcmdev {

__device__ __forceinline__ int quant1(float x)
{
    // websocket.rs:248: (x.clamp(-1.0, 1.0) * 32767.0) as i16
    if (x != x) return 0;                   // NaN survives clamp, `as i16` gives 0
    x = x < -1.0f ? -1.0f : x;
    x = x > 1.0f ? 1.0f : x;
    return (int)(x * 32767.0f);             // f32 product, truncation toward zero
}

// Four 6-bit values, one per byte of v, to their RFC 4648 characters with byte-parallel
// arithmetic: c = v + 'A' + 6 [v > 25] - 75 [v > 51] - 15 [v > 61] + 3 [v > 62].  v <= 63 keeps
// every byte sum below 256, and the subtracted bytes never exceed the added ones, so no carry
// or borrow crosses a byte.
__device__ __forceinline__ unsigned b64_chars4(unsigned v)
{
    const unsigned m25 = (v + 0x66666666u) & 0x80808080u;   // 0x80 where v >= 26
    const unsigned m51 = (v + 0x4C4C4C4Cu) & 0x80808080u;   //            v >= 52
    const unsigned m61 = (v + 0x42424242u) & 0x80808080u;   //            v >= 62
    const unsigned m62 = (v + 0x41414141u) & 0x80808080u;   //            v >= 63
    const unsigned add = v + 0x41414141u + (m25 >> 5) + (m25 >> 6) + (m62 >> 6) + (m62 >> 7);   // +6, +3
    const unsigned sub = (m51 >> 1) + (m51 >> 4) + (m51 >> 6) + (m51 >> 7)                      // 75
                         + (m61 >> 3) - (m61 >> 7);                                             // 15
    return add - sub;
}

// 12 bytes (three little-endian dwords = six little-endian i16 samples, websocket.rs:249) -> 16 characters, 4 per dword
__device__ __forceinline__ uint4 b64_encode_words(unsigned w0, unsigned w1, unsigned w2)
{
    // byte stream b0..b11 = w0 byte 0, 1, 2, 3, w1 byte 0 ...; triple t = bytes 3t, 3t+1, 3t+2
    auto triple = [](unsigned b0, unsigned b1, unsigned b2) {
        const unsigned t = (b0 << 16) | (b1 << 8) | b2;             // 24 bits, first byte on top
        // characters in memory order: byte 0 = bits 23..18, byte 1 = 17..12, byte 2 = 11..6, byte 3 = 5..0
        const unsigned v = ((t >> 18) & 63u) | (((t >> 12) & 63u) << 8) | (((t >> 6) & 63u) << 16) |
                           ((t & 63u) << 24);
        return b64_chars4(v);
    };
    uint4 o;
    o.x = triple(w0 & 0xffu, (w0 >> 8) & 0xffu, (w0 >> 16) & 0xffu);
    o.y = triple(w0 >> 24, w1 & 0xffu, (w1 >> 8) & 0xffu);
    o.z = triple((w1 >> 16) & 0xffu, w1 >> 24, w2 & 0xffu);
    o.w = triple((w2 >> 8) & 0xffu, (w2 >> 16) & 0xffu, w2 >> 24);
    return o;
}

// 6 samples -> 12 bytes (little-endian i16) -> 16 characters
__device__ __forceinline__ uint4 b64_encode16(const float (&x)[6])
{
    unsigned q[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) q[j] = (unsigned)quant1(x[j]) & 0xffffu;
    return b64_encode_words(q[0] | (q[1] << 16), q[2] | (q[3] << 16), q[4] | (q[5] << 16));
}

// The last, partial group of a string: `rem` (2, 4, 6, 8 or 10) bytes of the 12 are real (the others are zero in o):
// whole 3-byte triples, then the RFC 4648 '=' padding.  Byte stores (the end of a string has no alignment to speak of).
__device__ __forceinline__ void b64_store_tail(char* dst, const uint4 o, unsigned rem)
{
    const unsigned w[4] = {o.x, o.y, o.z, o.w};
    auto chr = [&](unsigned c) { return (char)((w[c >> 2] >> (8 * (c & 3))) & 0xffu); };
    const unsigned full = rem / 3, tail = rem % 3;
    for (unsigned c = 0; c < full * 4; ++c) dst[c] = chr(c);
    if (tail) {
        char* op = dst + full * 4;
        op[0] = chr(full * 4);
        op[1] = chr(full * 4 + 1);                   // low bits already zero (bytes beyond are 0)
        op[2] = tail == 2 ? chr(full * 4 + 2) : '=';
        op[3] = '=';
    }
}

}  // namespace pcmdev
}  // namespace af
