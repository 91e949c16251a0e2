// Host-side strip / tile decoders of the GeoTIFF reader (unet_amd/tiffio.py).  The reference opens rasters through GDAL / rasterio
// (create_tiles_unet.py:252-434, data.py:18-28), which read LZW / PackBits / Deflate files transparently; neither is installed here and a real
// 20000 x 20000 scene is usually compressed.  Deflate is zlib (Python's); the two byte-oriented TIFF codecs are restated here because a pure
// Python loop decodes ~1 MB/s.  No device code: plain C++ behind the C ABI.
#include <stdint.h>
#include <string.h>

#include "unet_hip.h"

// TIFF 6.0 section 13: LZW with MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, "early change" (the code width grows one
// code early).  Returns the number of bytes written to dst, or -1 on a malformed stream / overflow of cap.
extern "C" long long unet_tiff_lzw_decode(const unsigned char* src, long long n, unsigned char* dst, long long cap) {
    if (src == nullptr || dst == nullptr || n < 0 || cap < 0) return -1;
    enum { CLEAR = 256, EOI = 257, MAXC = 4096 };
    static thread_local int32_t prefix[MAXC];
    static thread_local uint8_t suffix[MAXC], first[MAXC];
    static thread_local uint16_t length[MAXC];
    for (int i = 0; i < 256; ++i) { prefix[i] = -1; suffix[i] = (uint8_t)i; first[i] = (uint8_t)i; length[i] = 1; }
    long long bitpos = 0, out = 0;
    const long long nbits = n * 8;
    int width = 9, next = 258, prev = -1;
    for (;;) {
        if (bitpos + width > nbits) break;          // a stream may end without EOI
        const long long byte = bitpos >> 3;
        uint32_t acc = ((uint32_t)src[byte] << 16) | ((byte + 1 < n ? (uint32_t)src[byte + 1] : 0u) << 8) | (byte + 2 < n ? (uint32_t)src[byte + 2] : 0u);
        const int code = (int)((acc >> (24 - (int)(bitpos & 7) - width)) & ((1u << width) - 1));
        bitpos += width;
        if (code == EOI) break;
        if (code == CLEAR) { width = 9; next = 258; prev = -1; continue; }
        int emit = code;
        if (prev < 0) {
            if (code >= 256) return -1;
        } else if (code >= next) {
            if (code != next) return -1;
            emit = prev;                              // KwKwK: the string of prev followed by its own first byte
        }
        const int len = length[emit] + ((prev >= 0 && code >= next) ? 1 : 0);
        if (out + len > cap) return -1;
        unsigned char* p = dst + out + length[emit];
        for (int c = emit; c >= 0; c = prefix[c]) *--p = suffix[c];
        if (prev >= 0 && code >= next) dst[out + len - 1] = first[prev];
        if (prev >= 0 && next < MAXC) {
            prefix[next] = prev;
            suffix[next] = (code >= next) ? first[prev] : first[code];
            first[next] = first[prev];
            length[next] = (uint16_t)(length[prev] + 1);
            ++next;
        }
        out += len;
        prev = code;
        if (next + 1 >= (1 << width) && width < 12) ++width;      // early change
    }
    return out;
}

// PackBits (TIFF 6.0 section 9): n in [0,127]: copy n + 1 literal bytes; n in [-127,-1]: repeat the next byte 1 - n times; -128: no-op
extern "C" long long unet_tiff_packbits_decode(const unsigned char* src, long long n, unsigned char* dst, long long cap) {
    if (src == nullptr || dst == nullptr || n < 0 || cap < 0) return -1;
    long long i = 0, out = 0;
    while (i < n) {
        const int c = (int8_t)src[i++];
        if (c >= 0) {
            const int k = c + 1;
            if (i + k > n || out + k > cap) return -1;
            memcpy(dst + out, src + i, (size_t)k);
            i += k; out += k;
        } else if (c != -128) {
            const int k = 1 - c;
            if (i >= n || out + k > cap) return -1;
            memset(dst + out, src[i++], (size_t)k);
            out += k;
        }
    }
    return out;
}
