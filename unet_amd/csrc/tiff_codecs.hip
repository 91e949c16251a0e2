// Host-side strip / tile decoders of the GeoTIFF reader (unet_amd/tiffio.py).  The reference opens rasters through GDAL / rasterio
// (create_tiles_unet.py:252-434, data.py:18-28), which read LZW / PackBits / Deflate files transparently; neither is installed here and a real
// 20000 x 20000 scene is usually compressed.  Deflate is zlib (Python's); the two byte-oriented TIFF codecs are restated here because a pure
// Python loop decodes ~1 MB/s.  No device code: plain C++ behind the C ABI.
#include <stdint.h>
#include <string.h>

#include "unet_hip.h"

// TIFF 6.0 section 13: LZW with MSB-first codes of 9..12 bits, ClearCode 256, EndOfInformation 257, "early change" (the code width grows one
// code early).  Returns the number of bytes written to dst, or -1 on a malformed stream / overflow of cap.
// Table entries do not store strings: the string of a new code is, by construction, the bytes just written for the previous code plus the
// byte that follows them in the output -- so an entry is (offset into dst, length) and emitting a code is one forward copy out of the
// output already produced (a forward byte copy also gives the KwKwK case, where the last byte of the string is its own first byte).
extern "C" long long unet_tiff_lzw_decode(const unsigned char* src, long long n, unsigned char* dst, long long cap) {
    if (src == nullptr || dst == nullptr || n < 0 || cap < 0) return -1;
    enum { CLEAR = 256, EOI = 257, MAXC = 4096 };
    long long off[MAXC];             // where in dst the string of a code >= 258 was written when the code was created
    int len[MAXC];
    long long bitpos = 0, out = 0, prev_off = 0;
    const long long nbits = n * 8;
    int width = 9, next = 258, prev = -1, prev_len = 0;
    for (;;) {
        if (bitpos + width > nbits) break;          // a stream may end without EOI
        const long long byte = bitpos >> 3;
        uint32_t acc = ((uint32_t)src[byte] << 16) | ((byte + 1 < n ? (uint32_t)src[byte + 1] : 0u) << 8) | (byte + 2 < n ? (uint32_t)src[byte + 2] : 0u);
        const int code = (int)((acc >> (24 - (int)(bitpos & 7) - width)) & ((1u << width) - 1));
        bitpos += width;
        if (code == EOI) break;
        if (code == CLEAR) { width = 9; next = 258; prev = -1; continue; }
        int cur_len;
        if (code < 256) {
            if (out + 1 > cap) return -1;
            dst[out] = (unsigned char)code;
            cur_len = 1;
        } else {
            if (prev < 0 || code > next || (code == next && next >= MAXC)) return -1;      // a code beyond the table
            long long from;
            if (code == next) { from = prev_off; cur_len = prev_len + 1; }      // KwKwK: the previous string followed by its own first byte
            else { from = off[code]; cur_len = len[code]; }
            if (out + cur_len > cap) return -1;
            const unsigned char* q = dst + from;
            unsigned char* p = dst + out;
            if (from + cur_len <= out && cur_len >= 16) memcpy(p, q, (size_t)cur_len);
            else for (int k = 0; k < cur_len; ++k) p[k] = q[k];                 // forward, byte by byte: source and destination may overlap
        }
        if (prev >= 0 && next < MAXC) {            // new entry = previous string + first byte of this one = dst[prev_off, prev_off + prev_len]
            off[next] = prev_off;
            len[next] = prev_len + 1;
            ++next;
        }
        prev_off = out;
        prev_len = cur_len;
        out += cur_len;
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
