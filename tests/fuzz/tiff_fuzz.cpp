// Test infrastructure (tests/test_tiff_fuzz_cpu.py compiles this with -fsanitize=address,undefined next to the decoders' sources):
// corrupted strips through the host TIFF decoders -- unet_tiff_jpeg_decode (include/unet_tiff.h), unet_tiff_lzw_decode and
// unet_tiff_packbits_decode (include/unet_hip.h).  A tile folder a user hands to train / predict may hold damaged files: the reader has to
// refuse them with an error code (tiffio.py raises ValueError), never read or write outside its buffers.  The reference reaches the same
// files through GDAL (create_tiles_unet.py:252-434, data.py:18-28), which reports a read error.
//
//   tiff_fuzz <corpus> <iterations per stream> <seed>
// corpus: records of  u8 kind (0 JPEG, 1 JPEG with tables, 2 LZW, 3 PackBits), u32 n_tables, tables, u32 n, stream, u32 decoded size.
// Source and destination buffers are exact-size heap blocks so that the sanitizer sees the first byte past either end.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

extern "C" long long unet_tiff_jpeg_decode(const unsigned char*, long long, const unsigned char*, long long, int, unsigned char*, long long, int*);
extern "C" long long unet_tiff_lzw_decode(const unsigned char*, long long, unsigned char*, long long);
extern "C" long long unet_tiff_packbits_decode(const unsigned char*, long long, unsigned char*, long long);

static uint64_t state = 0x9E3779B97F4A7C15ull;
static uint32_t rnd() { state ^= state << 13; state ^= state >> 7; state ^= state << 17; return (uint32_t)(state >> 16); }

struct Rec { int kind; std::vector<unsigned char> tables, stream; uint32_t size; };

static bool rd(FILE* f, void* p, size_t n) { return fread(p, 1, n, f) == n; }

static void mutate(std::vector<unsigned char>& v) {
    if (v.empty()) return;
    switch (rnd() % 6) {
    case 0: v.resize(rnd() % (v.size() + 1)); break;                                                                   // cut
    case 1: { const int k = 1 + rnd() % 4; for (int i = 0; i < k; ++i) v[rnd() % v.size()] = (unsigned char)rnd(); } break;   // anywhere
    case 2: { const size_t hdr = v.size() < 700 ? v.size() : 700;                                                      // in the headers
              const int k = 1 + rnd() % 3; for (int i = 0; i < k; ++i) v[rnd() % hdr] = (unsigned char)rnd(); } break;
    case 3: { const size_t a = rnd() % v.size(), n = rnd() % 64; for (size_t i = a; i < a + n && i < v.size(); ++i) v[i] = 0xFF; } break;
    case 4: { const size_t a = rnd() % v.size(), n = 1 + rnd() % 32; if (a + n < v.size()) v.erase(v.begin() + a, v.begin() + a + n); } break;
    default: { const size_t a = rnd() % v.size(); v[a] ^= (unsigned char)(1u << (rnd() % 8)); } break;                  // one bit
    }
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 2;
    const int iters = argc > 2 ? atoi(argv[2]) : 1000;
    if (argc > 3) state ^= (uint64_t)atoll(argv[3]) * 0xD1342543DE82EF95ull + 1;
    std::vector<Rec> corpus;
    for (;;) {
        Rec r; unsigned char kind; uint32_t nt, n;
        if (!rd(f, &kind, 1) || !rd(f, &nt, 4)) break;
        r.kind = kind; r.tables.resize(nt);
        if (nt && !rd(f, r.tables.data(), nt)) break;
        if (!rd(f, &n, 4)) break;
        r.stream.resize(n);
        if (n && !rd(f, r.stream.data(), n)) break;
        if (!rd(f, &r.size, 4)) break;
        corpus.push_back(r);
    }
    fclose(f);
    const char* trace = getenv("TIFF_FUZZ_TRACE");      // file that receives each input (as a corpus record) before it is decoded: what hung
    long ok = 0, refused = 0, intact = 0;
    for (const Rec& rec : corpus) {
        for (int it = -1; it < iters; ++it) {                 // it == -1: the stream as written has to decode to its full size
            std::vector<unsigned char> v = rec.stream, t = rec.tables;
            if (it >= 0) {
                if (!t.empty() && rnd() % 3 == 0) mutate(t); else mutate(v);
            }
            unsigned char* src = (unsigned char*)malloc(v.size() ? v.size() : 1);
            memcpy(src, v.data(), v.size());
            unsigned char* tab = t.empty() ? nullptr : (unsigned char*)malloc(t.size());
            if (tab) memcpy(tab, t.data(), t.size());
            const long long cap = (it < 0 || (it & 1)) ? (long long)rec.size : (long long)(rnd() % (rec.size + 1));
            unsigned char* dst = (unsigned char*)malloc(cap ? cap : 1);
            if (trace) {
                FILE* tf = fopen(trace, "wb");
                if (tf) {
                    const unsigned char k = (unsigned char)rec.kind; const uint32_t a = (uint32_t)t.size(), b = (uint32_t)v.size(), c = (uint32_t)cap;
                    fwrite(&k, 1, 1, tf); fwrite(&a, 4, 1, tf); if (a) fwrite(t.data(), 1, a, tf); fwrite(&b, 4, 1, tf); if (b) fwrite(v.data(), 1, b, tf); fwrite(&c, 4, 1, tf);
                    fclose(tf);
                }
            }
            long long r;
            if (rec.kind <= 1) { int dims[3] = {0, 0, 0}; r = unet_tiff_jpeg_decode(tab, (long long)t.size(), src, (long long)v.size(), it & 2, dst, cap, dims); }
            else if (rec.kind == 2) r = unet_tiff_lzw_decode(src, (long long)v.size(), dst, cap);
            else r = unet_tiff_packbits_decode(src, (long long)v.size(), dst, cap);
            if (r > cap) { fprintf(stderr, "kind %d: %lld bytes reported for a %lld-byte buffer\n", rec.kind, r, cap); abort(); }
            if (it < 0) {
                if (r != (long long)rec.size) { fprintf(stderr, "kind %d: the intact stream gave %lld, not %u\n", rec.kind, r, rec.size); return 1; }
                ++intact;
            } else if (r >= 0) ++ok; else ++refused;
            free(src); free(dst); free(tab);
        }
    }
    printf("streams %zu intact %ld decoded %ld refused %ld\n", corpus.size(), intact, ok, refused);
    return 0;
}
