// JPEG-in-TIFF strip / tile decoder of the GeoTIFF reader (unet_amd/tiffio.py; C ABI in include/unet_tiff.h).  Host only: g++ links it into
// libunet_tiff.so, it is not part of libunet_hip.so.  The reference opens rasters through GDAL (create_tiles_unet.py:252-434), i.e. through
// libtiff + libjpeg; orthophoto mosaics are commonly delivered as COMPRESS=JPEG GeoTIFFs.
//
// Entropy decoding follows ITU-T T.81 (annex F.2.2: sequential Huffman, annex E.2.4: restart intervals).  The sample arithmetic restates what
// libjpeg(-turbo) computes by default, so that the bytes equal the ones GDAL hands the reference:
//   * inverse DCT: jidctint.c ("islow": 13-bit constants, two passes, the first one keeping 2 extra bits), output range-limited through the
//     library's 10-bit masked table;
//   * chroma upsampling: jdsample.c h2v1 / h2v2 "fancy" triangle filters (plain replication when the component is <= 2 samples wide);
//   * YCbCr -> RGB: jdcolor.c (16-bit fixed-point tables).
#include <stdint.h>
#include <string.h>

#include <vector>

#include "unet_tiff.h"

namespace {

const uint8_t ZIGZAG[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48,
                            41, 34, 27, 20, 13, 6,  7,  14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23,
                            30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct Huff {                       // one DHT table: canonical codes by length (T.81 annex C) + a 9-bit lookup for the short ones
    bool set = false;
    uint8_t vals[256];
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t look_len[512], look_val[512];
};

bool build_huff(Huff& h, const uint8_t* bits /* [16] */, const uint8_t* vals, int nvals) {
    int total = 0;
    for (int i = 0; i < 16; ++i) total += bits[i];
    if (total > 256 || total != nvals) return false;
    memcpy(h.vals, vals, (size_t)total);
    memset(h.look_len, 0, sizeof(h.look_len));
    int code = 0, k = 0;
    for (int l = 1; l <= 16; ++l) {
        h.valptr[l] = k;
        h.mincode[l] = code;
        for (int i = 0; i < bits[l - 1]; ++i, ++k, ++code) {
            if (code >= (1 << l)) return false;
            if (l <= 9) {
                const int first = code << (9 - l), cnt = 1 << (9 - l);
                for (int j = 0; j < cnt; ++j) { h.look_len[first + j] = (uint8_t)l; h.look_val[first + j] = vals[k]; }
            }
        }
        h.maxcode[l] = bits[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h.maxcode[17] = 0x7fffffff;
    h.set = true;
    return true;
}

struct Bits {                       // MSB-first reader over an entropy-coded segment: FF00 -> FF, any other FFxx ends the segment (zeros follow)
    const uint8_t* p;
    const uint8_t* end;
    uint64_t acc = 0;
    int cnt = 0;
    void fill() {
        while (cnt <= 56) {
            uint64_t b = 0;
            if (p < end) {
                b = *p;
                if (b == 0xFF) {
                    if (p + 1 < end && p[1] == 0) p += 2;
                    else b = 0;                      // a marker (or the end of the data): stay in front of it
                } else {
                    ++p;
                }
            }
            acc |= b << (56 - cnt);
            cnt += 8;
        }
    }
    int get(int n) {                // n in [0, 16]
        if (n == 0) return 0;
        if (cnt < n) fill();
        const int v = (int)(acc >> (64 - n));
        acc <<= n;
        cnt -= n;
        return v;
    }
    void reset() { acc = 0; cnt = 0; }
};

inline int decode_sym(Bits& b, const Huff& h) {
    if (b.cnt < 16) b.fill();
    const uint32_t top = (uint32_t)(b.acc >> 55);
    int l = h.look_len[top];
    if (l) {
        b.acc <<= l;
        b.cnt -= l;
        return h.look_val[top];
    }
    for (l = 10; l <= 16; ++l) {
        const int32_t code = (int32_t)(b.acc >> (64 - l));
        if (code <= h.maxcode[l]) {
            b.acc <<= l;
            b.cnt -= l;
            return h.vals[h.valptr[l] + code - h.mincode[l]];
        }
    }
    return -1;
}

inline int extend(int v, int s) { return v < (1 << (s - 1)) ? v + (int)((~0u) << s) + 1 : v; }

struct RangeTable {                 // jdmaster.c prepare_range_limit_table as the inverse DCT indexes it: (x & RANGE_MASK) past the +128 centre
    uint8_t t[1024];
    RangeTable() {
        for (int x = 0; x < 1024; ++x) t[x] = (uint8_t)(x < 128 ? x + 128 : x < 512 ? 255 : x < 896 ? 0 : x - 896);
    }
};
const RangeTable RANGE;
inline uint8_t range_limit(int32_t x) { return RANGE.t[x & 1023]; }

// One 1-D pass of jidctint.c jpeg_idct_islow over EIGHT independent lanes at once (v[i][l]: input i of lane l), so that the compiler turns every
// statement into vector instructions; the integer arithmetic per lane is the library's, statement by statement.  Sums and products are taken
// modulo 2^32 (unsigned) and read back as two's complement before the descale: the same bits as the library's for every stream an encoder
// writes, and defined -- instead of signed overflow -- for the coefficients a corrupt stream can hold (found by the ASan / UBSan fuzz run).
__attribute__((always_inline)) inline void idct_pass8(const int32_t (*v)[8], int32_t (*o)[8], const int sh) {
    enum { CB = 13 };
    typedef uint32_t U;
    const U F0_298 = 2446, F0_390 = 3196, F0_541 = 4433, F0_765 = 6270, F0_899 = 7373, F1_175 = 9633, F1_501 = 12299, F1_847 = 15137,
            F1_961 = 16069, F2_053 = 16819, F2_562 = 20995, F3_072 = 25172;
    const U half = (U)1 << (sh - 1);
    for (int l = 0; l < 8; ++l) {
        U z2 = (U)v[2][l], z3 = (U)v[6][l];
        U z1 = (z2 + z3) * F0_541;
        U tmp2 = z1 - z3 * F1_847;
        U tmp3 = z1 + z2 * F0_765;
        U tmp0 = ((U)v[0][l] + (U)v[4][l]) << CB, tmp1 = ((U)v[0][l] - (U)v[4][l]) << CB;
        const U tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = (U)v[7][l]; tmp1 = (U)v[5][l]; tmp2 = (U)v[3][l]; tmp3 = (U)v[1][l];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        U z4 = tmp1 + tmp3;
        const U z5 = (z3 + z4) * F1_175;
        tmp0 *= F0_298; tmp1 *= F2_053; tmp2 *= F3_072; tmp3 *= F1_501;
        z1 *= (U)0 - F0_899; z2 *= (U)0 - F2_562; z3 *= (U)0 - F1_961; z4 *= (U)0 - F0_390;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        o[0][l] = (int32_t)(tmp10 + tmp3 + half) >> sh; o[7][l] = (int32_t)(tmp10 - tmp3 + half) >> sh;
        o[1][l] = (int32_t)(tmp11 + tmp2 + half) >> sh; o[6][l] = (int32_t)(tmp11 - tmp2 + half) >> sh;
        o[2][l] = (int32_t)(tmp12 + tmp1 + half) >> sh; o[5][l] = (int32_t)(tmp12 - tmp1 + half) >> sh;
        o[3][l] = (int32_t)(tmp13 + tmp0 + half) >> sh; o[4][l] = (int32_t)(tmp13 - tmp0 + half) >> sh;
    }
}

// jidctint.c jpeg_idct_islow: coefficients (natural order) times the quantisation table -> 8 x 8 samples at out (row stride `stride`).
// Pass 1 runs down the columns (lanes = columns) keeping 2 extra bits, pass 2 along the rows (lanes = rows).  The library's zero-coefficient
// shortcuts are exact special cases of the same arithmetic and are not needed here.  (Cloned for AVX2 and resolved when the library is loaded.)
__attribute__((target_clones("avx2", "default"))) void idct_islow(const int16_t* coef, const uint16_t* q, uint8_t* out, int stride) {
    enum { CB = 13, P1 = 2 };
    alignas(32) int32_t a[8][8], w[8][8];
    for (int r = 0; r < 8; ++r)
        for (int k = 0; k < 8; ++k) a[r][k] = (int32_t)coef[8 * r + k] * (int32_t)q[8 * r + k];
    idct_pass8(a, w, CB - P1);                          // w[r][k]: row r of column k
    for (int r = 0; r < 8; ++r)
        for (int k = 0; k < 8; ++k) a[k][r] = w[r][k];  // input k of lane (= row) r
    idct_pass8(a, w, CB + P1 + 3);                      // w[c][r]: column c of row r
    for (int r = 0; r < 8; ++r)
        for (int c = 0; c < 8; ++c) out[(size_t)r * stride + c] = range_limit(w[c][r]);
}

// a block without AC coefficients: both passes reduce to one value, (DC * q0 * 4 + 16) >> 5 (derived from the two descales above)
inline void idct_dc_only(int dc, const uint16_t* q, uint8_t* out, int stride) {
    const uint8_t v = range_limit((int32_t)((uint32_t)((int32_t)dc * (int32_t)q[0]) * 4u + 16u) >> 5);
    for (int r = 0; r < 8; ++r) memset(out + (size_t)r * stride, v, 8);
}

struct Comp {
    int id = 0, h = 1, v = 1, tq = 0, td = 0, ta = 0;
    int cw = 0, ch = 0;             // downsampled_width / _height (samples that carry image data)
    int pw = 0, ph = 0;             // plane size in samples, padded to whole MCUs
    int pred = 0;
    std::vector<uint8_t> plane;
};

struct Decoder {
    uint16_t qt[4][64];
    bool qt_set[4] = {false, false, false, false};
    Huff dc[4], ac[4];
    int W = 0, H = 0, nc = 0, hmax = 1, vmax = 1, ri = 0;
    bool have_frame = false;
    long long cap = -1;             // bytes the caller's buffer holds: a frame header that asks for more is refused before anything is allocated
    Comp comp[4];
};

inline int rd16(const uint8_t* p) { return (p[0] << 8) | p[1]; }

int parse_dqt(Decoder& d, const uint8_t* p, int len) {
    while (len > 0) {
        const int pq = p[0] >> 4, tq = p[0] & 15;
        if (tq > 3 || pq > 1) return -1;
        const int need = 1 + 64 * (pq + 1);
        if (len < need) return -1;
        for (int i = 0; i < 64; ++i) d.qt[tq][ZIGZAG[i]] = (uint16_t)(pq ? rd16(p + 1 + 2 * i) : p[1 + i]);
        d.qt_set[tq] = true;
        p += need; len -= need;
    }
    return 0;
}

int parse_dht(Decoder& d, const uint8_t* p, int len) {
    while (len > 0) {
        if (len < 17) return -1;
        const int tc = p[0] >> 4, th = p[0] & 15;
        if (tc > 1 || th > 3) return -1;
        int total = 0;
        for (int i = 0; i < 16; ++i) total += p[1 + i];
        if (len < 17 + total) return -1;
        if (!build_huff(tc ? d.ac[th] : d.dc[th], p + 1, p + 17, total)) return -1;
        p += 17 + total; len -= 17 + total;
    }
    return 0;
}

int parse_sof(Decoder& d, const uint8_t* p, int len) {
    if (len < 6) return -1;
    if (p[0] != 8) return -2;                          // 12-bit samples
    d.H = rd16(p + 1); d.W = rd16(p + 3); d.nc = p[5];
    if (d.H <= 0 || d.W <= 0 || d.nc < 1 || d.nc > 4 || len < 6 + 3 * d.nc) return -1;
    if (d.cap >= 0 && (long long)d.W * d.H * d.nc > d.cap) return -1;      // (a damaged header of 65535 x 65535 would otherwise cost 4 GB a plane and minutes)
    d.hmax = d.vmax = 1;
    for (int i = 0; i < d.nc; ++i) {
        Comp& c = d.comp[i];
        c.id = p[6 + 3 * i]; c.h = p[7 + 3 * i] >> 4; c.v = p[7 + 3 * i] & 15; c.tq = p[8 + 3 * i];
        if (c.h < 1 || c.h > 4 || c.v < 1 || c.v > 4 || c.tq > 3) return -1;
        if (c.h > d.hmax) d.hmax = c.h;
        if (c.v > d.vmax) d.vmax = c.v;
    }
    const int mcux = (d.W + 8 * d.hmax - 1) / (8 * d.hmax), mcuy = (d.H + 8 * d.vmax - 1) / (8 * d.vmax);
    for (int i = 0; i < d.nc; ++i) {
        Comp& c = d.comp[i];
        c.cw = (d.W * c.h + d.hmax - 1) / d.hmax;
        c.ch = (d.H * c.v + d.vmax - 1) / d.vmax;
        c.pw = mcux * c.h * 8;
        c.ph = mcuy * c.v * 8;
        c.plane.assign((size_t)c.pw * c.ph, 0);
    }
    d.have_frame = true;
    return 0;
}

// one 8 x 8 block of component c at block coordinates (by, bx): F.2.2.1 (DC difference) + F.2.2.2 (AC run / size), then the inverse DCT
int decode_block(Decoder& d, Bits& b, Comp& c, int by, int bx) {
    int16_t blk[64];
    memset(blk, 0, sizeof(blk));
    const Huff& hd = d.dc[c.td];
    const Huff& ha = d.ac[c.ta];
    int s = decode_sym(b, hd);
    if (s < 0 || s > 11) return -1;
    c.pred += s ? extend(b.get(s), s) : 0;
    blk[0] = (int16_t)c.pred;
    bool any_ac = false;
    for (int k = 1; k < 64;) {
        const int rs = decode_sym(b, ha);
        if (rs < 0) return -1;
        const int r = rs >> 4;
        s = rs & 15;
        if (s == 0) {
            if (r != 15) break;                        // end of block
            k += 16;
            continue;
        }
        k += r;
        if (k > 63) return -1;
        blk[ZIGZAG[k]] = (int16_t)extend(b.get(s), s);
        any_ac = true;
        ++k;
    }
    if (by * 8 + 8 <= c.ph && bx * 8 + 8 <= c.pw) {
        uint8_t* o = c.plane.data() + (size_t)by * 8 * c.pw + bx * 8;
        if (any_ac) idct_islow(blk, d.qt[c.tq], o, c.pw);
        else idct_dc_only(blk[0], d.qt[c.tq], o, c.pw);
    }
    return 0;
}

// entropy-coded data of one scan; returns the position behind it (at the next marker) or nullptr
const uint8_t* decode_scan(Decoder& d, const uint8_t* p, const uint8_t* end, const int* sel, int ns) {
    Bits b{p, end};
    for (int i = 0; i < ns; ++i) {
        Comp& c = d.comp[sel[i]];
        if (!d.dc[c.td].set || !d.ac[c.ta].set || !d.qt_set[c.tq]) return nullptr;
        c.pred = 0;
    }
    int mx, my;
    if (ns == 1) {                                     // a scan of one component walks ITS blocks, not padded to whole MCUs (A.2.2)
        const Comp& c = d.comp[sel[0]];
        mx = (c.cw + 7) / 8; my = (c.ch + 7) / 8;
    } else {
        mx = (d.W + 8 * d.hmax - 1) / (8 * d.hmax); my = (d.H + 8 * d.vmax - 1) / (8 * d.vmax);
    }
    long long done = 0;
    int next_rst = 0;
    for (int y = 0; y < my; ++y)
        for (int x = 0; x < mx; ++x, ++done) {
            if (d.ri > 0 && done > 0 && done % d.ri == 0) {                   // E.2.4: byte-align, RSTm, predictions back to zero
                b.reset();
                const uint8_t* q = b.p;
                while (q + 1 < end && !(q[0] == 0xFF && q[1] >= 0xD0 && q[1] <= 0xD7)) ++q;
                if (q + 1 >= end || (q[1] & 7) != next_rst) return nullptr;
                next_rst = (next_rst + 1) & 7;
                b.p = q + 2;
                for (int i = 0; i < ns; ++i) d.comp[sel[i]].pred = 0;
            }
            if (ns == 1) {
                if (decode_block(d, b, d.comp[sel[0]], y, x) < 0) return nullptr;
            } else {
                for (int i = 0; i < ns; ++i) {
                    Comp& c = d.comp[sel[i]];
                    for (int j = 0; j < c.v; ++j)
                        for (int k = 0; k < c.h; ++k)
                            if (decode_block(d, b, c, y * c.v + j, x * c.h + k) < 0) return nullptr;
                }
            }
        }
    const uint8_t* q = b.p;
    while (q + 1 < end && !(q[0] == 0xFF && q[1] != 0 && q[1] != 0xFF)) ++q;
    return q + 1 < end ? q : end;
}

// marker segments of one stream.  tables_only: the JPEGTables tag (no frame expected)
int parse_stream(Decoder& d, const uint8_t* p, long long n, bool tables_only) {
    const uint8_t* end = p + n;
    if (n < 4 || p[0] != 0xFF || p[1] != 0xD8) return -1;
    p += 2;
    bool seen_scan = false;
    while (p + 1 < end) {
        if (p[0] != 0xFF) return -1;
        while (p + 1 < end && p[1] == 0xFF) ++p;      // fill bytes
        if (p + 1 >= end) break;
        const int m = p[1];
        p += 2;
        if (m == 0xD9) return (tables_only || seen_scan) ? 0 : -1;          // EOI
        if (m == 0x01 || (m >= 0xD0 && m <= 0xD7)) continue;
        if (p + 2 > end) return -1;
        const int len = rd16(p);
        if (len < 2 || p + len > end) return -1;
        const uint8_t* s = p + 2;
        const int sl = len - 2;
        p += len;
        int rc = 0;
        if (m == 0xDB) rc = parse_dqt(d, s, sl);
        else if (m == 0xC4) rc = parse_dht(d, s, sl);
        else if (m == 0xDD) { if (sl < 2) return -1; d.ri = rd16(s); }
        else if (m == 0xC0 || m == 0xC1) { if (tables_only || d.have_frame) return -1; rc = parse_sof(d, s, sl); }
        else if ((m >= 0xC2 && m <= 0xCF) && m != 0xC4 && m != 0xC8) return -2;      // progressive / lossless / arithmetic-coded processes (0xCC: DAC)
        else if (m == 0xDA) {
            if (tables_only || !d.have_frame || sl < 1) return -1;
            const int ns = s[0];
            if (ns < 1 || ns > d.nc || sl < 1 + 2 * ns + 3) return -1;
            int sel[4];
            for (int i = 0; i < ns; ++i) {
                int idx = -1;
                for (int j = 0; j < d.nc; ++j) if (d.comp[j].id == s[1 + 2 * i]) idx = j;
                if (idx < 0) return -1;
                sel[i] = idx;
                d.comp[idx].td = s[2 + 2 * i] >> 4;
                d.comp[idx].ta = s[2 + 2 * i] & 15;
                if (d.comp[idx].td > 3 || d.comp[idx].ta > 3) return -1;
            }
            if (s[1 + 2 * ns] != 0 || s[2 + 2 * ns] != 63 || s[3 + 2 * ns] != 0) return -2;          // spectral selection / successive approximation
            if (ns > 1) {
                int blocks = 0;
                for (int i = 0; i < ns; ++i) blocks += d.comp[sel[i]].h * d.comp[sel[i]].v;
                if (blocks > 10) return -1;
            }
            p = decode_scan(d, p, end, sel, ns);
            if (p == nullptr) return -1;
            seen_scan = true;
        }
        if (rc != 0) return rc;
    }
    return (tables_only || seen_scan) ? 0 : -1;      // (a stream may end without EOI)
}

// jdsample.c: one row of h2v1_fancy_upsample (w input samples -> 2 w output samples)
void h2v1_fancy_row(const uint8_t* in, int w, uint8_t* out) {
    int v = in[0];
    *out++ = (uint8_t)v;
    *out++ = (uint8_t)((v * 3 + in[1] + 2) >> 2);
    for (int i = 1; i < w - 1; ++i) {
        v = in[i] * 3;
        *out++ = (uint8_t)((v + in[i - 1] + 1) >> 2);
        *out++ = (uint8_t)((v + in[i + 1] + 2) >> 2);
    }
    v = in[w - 1];
    *out++ = (uint8_t)((v * 3 + in[w - 2] + 1) >> 2);
    *out++ = (uint8_t)v;
}

// jdsample.c: one output row of h2v2_fancy_upsample from the nearer input row `in0` and the further one `in1`
void h2v2_fancy_row(const uint8_t* in0, const uint8_t* in1, int w, uint8_t* out) {
    int thiscol = in0[0] * 3 + in1[0], nextcol = in0[1] * 3 + in1[1], lastcol;
    *out++ = (uint8_t)((thiscol * 4 + 8) >> 4);
    *out++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
    lastcol = thiscol; thiscol = nextcol;
    for (int i = 2; i < w; ++i) {
        nextcol = in0[i] * 3 + in1[i];
        *out++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
        *out++ = (uint8_t)((thiscol * 3 + nextcol + 7) >> 4);
        lastcol = thiscol; thiscol = nextcol;
    }
    *out++ = (uint8_t)((thiscol * 3 + lastcol + 8) >> 4);
    *out++ = (uint8_t)((thiscol * 4 + 7) >> 4);
}

// full-resolution samples of component c: `up` [uh][uw] with uw >= W, uh >= H; returns false for a sampling ratio that is not built
bool upsample(const Decoder& d, const Comp& c, std::vector<uint8_t>& up, int& uw, const uint8_t*& base) {
    if (c.h == d.hmax && c.v == d.vmax) { base = c.plane.data(); uw = c.pw; return true; }
    const bool h2 = c.h * 2 == d.hmax, v1 = c.v == d.vmax, v2 = c.v * 2 == d.vmax;
    if (!h2 || !(v1 || v2)) return false;
    uw = 2 * c.cw;
    const int uh = v1 ? c.ch : 2 * c.ch;
    up.assign((size_t)uw * uh, 0);
    const bool fancy = c.cw > 2;
    for (int y = 0; y < uh; ++y) {
        uint8_t* o = up.data() + (size_t)y * uw;
        const int iy = v1 ? y : y / 2;
        const uint8_t* in0 = c.plane.data() + (size_t)iy * c.pw;
        if (!fancy) {
            for (int x = 0; x < c.cw; ++x) o[2 * x] = o[2 * x + 1] = in0[x];
        } else if (v1) {
            h2v1_fancy_row(in0, c.cw, o);
        } else {
            int other = (y & 1) ? iy + 1 : iy - 1;      // the upper output row of a pair leans on the row above, the lower one on the row below;
            if (other < 0) other = 0;                  // at the edges of the component the row itself stands in (jdmainct.c context rows)
            if (other > c.ch - 1) other = c.ch - 1;
            h2v2_fancy_row(in0, c.plane.data() + (size_t)other * c.pw, c.cw, o);
        }
    }
    base = up.data();
    return true;
}

}  // namespace

extern "C" long long unet_tiff_jpeg_decode(const unsigned char* tables, long long ntables, const unsigned char* src, long long n, int ycbcr_to_rgb,
                                           unsigned char* dst, long long cap, int* dims) {
    if (src == nullptr || dst == nullptr || n < 4 || cap < 0) return -1;
    Decoder d;
    d.cap = cap;
    int rc;
    if (tables != nullptr && ntables > 0 && (rc = parse_stream(d, tables, ntables, true)) != 0) return rc;
    if ((rc = parse_stream(d, src, n, false)) != 0) {
        if (dims != nullptr) { dims[0] = d.H; dims[1] = d.W; dims[2] = d.nc; }
        return rc;
    }
    const long long total = (long long)d.W * d.H * d.nc;
    if (dims != nullptr) { dims[0] = d.H; dims[1] = d.W; dims[2] = d.nc; }
    if (total > cap) return -1;
    std::vector<uint8_t> ups[4];
    const uint8_t* base[4];
    int stride[4];
    for (int i = 0; i < d.nc; ++i)
        if (!upsample(d, d.comp[i], ups[i], stride[i], base[i])) return -2;
    const bool convert = ycbcr_to_rgb != 0 && d.nc == 3;
    // jdcolor.c build_ycc_rgb_table: 16-bit fixed point, FIX(x) = (int)(x * 65536 + 0.5)
    const int32_t F1_402 = 91881, F1_772 = 116130, F0_714 = 46802, F0_344 = 22554, HALF = 1 << 15;
    for (int y = 0; y < d.H; ++y) {
        uint8_t* o = dst + (size_t)y * d.W * d.nc;
        if (convert) {
            const uint8_t *py = base[0] + (size_t)y * stride[0], *pb = base[1] + (size_t)y * stride[1], *pr = base[2] + (size_t)y * stride[2];
            for (int x = 0; x < d.W; ++x) {
                const int32_t Y = py[x], cb = (int32_t)pb[x] - 128, cr = (int32_t)pr[x] - 128;
                const int32_t r = Y + ((F1_402 * cr + HALF) >> 16);
                const int32_t g = Y + ((-F0_344 * cb + HALF - F0_714 * cr) >> 16);
                const int32_t bl = Y + ((F1_772 * cb + HALF) >> 16);
                o[3 * x] = (uint8_t)(r < 0 ? 0 : r > 255 ? 255 : r);
                o[3 * x + 1] = (uint8_t)(g < 0 ? 0 : g > 255 ? 255 : g);
                o[3 * x + 2] = (uint8_t)(bl < 0 ? 0 : bl > 255 ? 255 : bl);
            }
        } else {
            const uint8_t* pc[4];
            for (int c = 0; c < d.nc; ++c) pc[c] = base[c] + (size_t)y * stride[c];
            if (d.nc == 1) memcpy(o, pc[0], (size_t)d.W);
            else if (d.nc == 4) for (int x = 0; x < d.W; ++x) { o[4 * x] = pc[0][x]; o[4 * x + 1] = pc[1][x]; o[4 * x + 2] = pc[2][x]; o[4 * x + 3] = pc[3][x]; }
            else if (d.nc == 3) for (int x = 0; x < d.W; ++x) { o[3 * x] = pc[0][x]; o[3 * x + 1] = pc[1][x]; o[3 * x + 2] = pc[2][x]; }
            else for (int x = 0; x < d.W; ++x) { o[2 * x] = pc[0][x]; o[2 * x + 1] = pc[1][x]; }
        }
    }
    return total;
}
