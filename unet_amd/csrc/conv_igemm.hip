// Implicit-GEMM convolution on fp32 MFMA for gfx950 (MI355X).
//
//   y[n, oy, ox, k] = epilogue( sum_{tap} sum_{c} x[n, oy*S + dy_tap, ox*S + dx_tap, c] * W[tap][c][k] )
//
// GEMM view: M = output pixels (32-row MFMA tiles of a TH x TW spatial tile),
// N = produced channels, K = taps x reduction channels.  One workgroup owns a
// BM(=128 pixel) x BN(32/64/128 channel) output tile:
//   * the input HALO tile of the spatial tile is staged ONCE per 16-channel
//     chunk into LDS and re-used by all 9 taps (no im2col, 9x less L2->LDS traffic);
//   * the filter slab of one (tap, chunk) comes from a pre-packed [tap][chunk][cout][16] image (fully coalesced 16-B
//     loads): the 32x32x2 kernel stages it through LDS (double buffered, one barrier per stage), the default 16x16x4
//     kernel loads it straight into MFMA operand registers one stage ahead (one barrier per chunk);
//   * LDS rows are 20 floats (16 + 4 pad) so that the ds_read_b128 operand
//     fetches (4 k-values per lane) are bank-conflict free;
//   * v_mfma_f32_32x32x2_f32: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
//     each lane half reads 4 consecutive channels -> 4 MFMA k-steps per b128 pair.
//
// The same kernel serves forward convs (3x3 s1/s2, 1x1) and input gradients
// (dgrad: flipped taps with transposed packed weights; stride-2 dgrad as 4 output
// parity classes with 1/2/2/4 taps each).
//
// Replaces the ATen conv kernels behind fastai ConvLayer (layers.py) as used by
// XResNet / DynamicUnet built at reference train.py:128-144.

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;   // reduction channels per chunk
constexpr int LDK = 20;  // LDS row length in floats (16 + 4 pad); 24 would make the 16x16x4 operand reads conflict-free, measured no gain

struct TapSet {
    int n;              // number of taps
    int min_dy, min_dx; // halo origin offset (input coords relative to tile origin * S)
    int ext_y, ext_x;   // halo extent beyond (T-1)*S
    int py, px;         // output parity offsets (OS == 2)
    signed char dy[9], dx[9], widx[9];
    // the same tables packed 4 bits per tap for scalar decoding (16x16x4 kernel): dpack nibble t = (dy - min_dy) | (dx - min_dx) << 2,
    // wpack nibble t = widx
    unsigned long long dpack, wpack;
};

struct KArgs {
    const float* x; const float* wp; const float* bias; const float* res; const float* mask;
    float* y; float* colsum; float* colsumsq;
    int x_cs, x_co, res_cs, res_co, mask_cs, mask_co, y_cs, y_co;
    int N, IH, IW, Cin, Cin4;
    int OH, OW, Cout;
    int S, OS, TSH, TSW, tiles_y, tiles_x, ntn;
    int nchunks, coutPad, flags, mtiles;
    int n_base, n_end;   // produced-channel range of this launch (unet_conv_desc.cout_begin / cout_count); n_end <= Cout
    long long wp_stride; // floats between the packed filter images of consecutive batch images (0: one image for all)
    TapSet taps[4];
};

template <int HIT, int NTH>
__device__ __forceinline__ void ld_halo(float4 (&hreg)[HIT], const int (&goff)[HIT], const float* xb, int c0, int Cin4, int tid) {
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int q = (tid + it * NTH) & 3;
        const bool ok = goff[it] >= 0 && (c0 + 4 * q) < Cin4;
        hreg[it] = ok ? *reinterpret_cast<const float4*>(xb + goff[it] + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}
template <int HIT, int NTH>
__device__ __forceinline__ void st_halo(const float4 (&hreg)[HIT], float* dst, int hpix4, int tid) {
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int e = tid + it * NTH;
        if (e < hpix4) *reinterpret_cast<float4*>(dst + (e >> 2) * LDK + (e & 3) * 4) = hreg[it];
    }
}
// Reduction-tail chunk (Cin % 16 != 0): channel c of the chunk is stored at position 4*(c%4) + c/4, so that MFMA
// k-step kk of the b128 operand reads covers channels 4kk..4kk+3 and steps beyond the real channels are skipped.
// The packed filter image uses the same order for that chunk (pack_weights_kernel).
template <int HIT, int NTH>
__device__ __forceinline__ void st_halo_tail(const float4 (&hreg)[HIT], float* dst, int hpix4, int tid) {
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int e = tid + it * NTH;
        if (e < hpix4) {
            float* row = dst + (e >> 2) * LDK + (e & 3);
            row[0] = hreg[it].x; row[4] = hreg[it].y; row[8] = hreg[it].z; row[12] = hreg[it].w;
        }
    }
}
// filter slab staging: one or two float4 per thread, held in named registers (a 2-element
// array here was demoted to scratch by the compiler)
template <int WIT, int NTH, int BN>
__device__ __forceinline__ void ld_w(float4& w0, float4& w1, const float* src, int tid) {
    static_assert(WIT == 1 || WIT == 2, "WIT");
    int e = tid;
    if constexpr ((BN * 4) % NTH != 0) e = e < BN * 4 ? e : BN * 4 - 1;  // clamp: the load stays unconditional
    w0 = *reinterpret_cast<const float4*>(src + e * 4);
    if constexpr (WIT == 2) w1 = *reinterpret_cast<const float4*>(src + (tid + NTH) * 4);
}
template <int WIT, int NTH, int BN>
__device__ __forceinline__ void st_w(const float4& w0, const float4& w1, float* dst, int tid) {
    if ((BN * 4) % NTH == 0 || tid < BN * 4) *reinterpret_cast<float4*>(dst + (tid >> 2) * LDK + (tid & 3) * 4) = w0;
    if constexpr (WIT == 2) {
        const int e = tid + NTH;
        *reinterpret_cast<float4*>(dst + (e >> 2) * LDK + (e & 3) * 4) = w1;
    }
}

template <int TW, int MT, int NT, int WM, int WN, int HIT>
__global__ __launch_bounds__(WM* WN * 64, HIT == 4 ? 3 : 2) void conv_igemm_kernel(const KArgs a) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, TH = BM / TW, NTH = WM * WN * 64;
    constexpr int WIT = (BN * 4 + NTH - 1) / NTH;  // weight float4 items per thread
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const TapSet& ts = a.taps[blockIdx.z];

    int bid = blockIdx.x;
    const int nt = bid % a.ntn; bid /= a.ntn;
    const int mtile = bid;
    const int tx_t = bid % a.tiles_x; bid /= a.tiles_x;
    const int ty_t = bid % a.tiles_y;
    const int img = bid / a.tiles_y;
    const int oy0 = ty_t * TH, ox0 = tx_t * TW;
    const int n0 = a.n_base + nt * BN;

    const int S = a.S;
    const int HH = (TH - 1) * S + ts.ext_y, HW = (TW - 1) * S + ts.ext_x;
    const int HPIX = HH * HW;
    // LDS map: [0,32) ints tap tables | halo[2] | wts[2]   (buffers addressed arithmetically: no runtime-indexed arrays)
    int* s_tapoff = reinterpret_cast<int*>(smem);       // [0..8]  LDS float offset of the tap's window inside the halo tile
    int* s_widx = reinterpret_cast<int*>(smem) + 16;    // [0..8]  packed-weight tap index
    float* lds0 = smem + 32;
    auto halo_buf = [&](int b) -> float* { return lds0 + b * (HPIX * LDK); };
    auto wts_buf = [&](int b) -> float* { return lds0 + 2 * HPIX * LDK + b * (BN * LDK); };
    if (tid < 9) {
        const int t = tid < ts.n ? tid : 0;
        s_tapoff[tid] = ((ts.dy[t] - ts.min_dy) * HW + (ts.dx[t] - ts.min_dx)) * LDK;
        s_widx[tid] = ts.widx[t];
    }

    // ---- per-thread halo item addressing (independent of the chunk) ----
    const float* xb = a.x + (size_t)img * a.IH * a.IW * a.x_cs;
    const int iy0 = oy0 * S + ts.min_dy, ix0 = ox0 * S + ts.min_dx;
    int goff[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int e = tid + it * NTH;
        const int p = e >> 2, q = e & 3;
        const int hy = p / HW, hx = p - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool inb = (e < HPIX * 4) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
        goff[it] = inb ? ((iy * a.IW + ix) * a.x_cs + a.x_co + 4 * q) : -1;
    }

    float4 hreg[HIT];
    float4 wreg0, wreg1 = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_tail = (a.Cin & 15) != 0;

#define LOAD_HALO(chunk_) ld_halo<HIT, NTH>(hreg, goff, xb, (chunk_) * KC, a.Cin4, tid)
#define STORE_HALO(dst_, chunk_) do { if (has_tail && (chunk_) == a.nchunks - 1) st_halo_tail<HIT, NTH>(hreg, (dst_), HPIX * 4, tid); \
                                       else st_halo<HIT, NTH>(hreg, (dst_), HPIX * 4, tid); } while (0)
#define LOAD_W(widx_, chunk_) ld_w<WIT, NTH, BN>(wreg0, wreg1, a.wp + (size_t)img * a.wp_stride + ((size_t)((widx_) * a.nchunks + (chunk_)) * a.coutPad + n0) * KC, tid)
#define STORE_W(dst_) st_w<WIT, NTH, BN>(wreg0, wreg1, (dst_), tid)

    // ---- MFMA operand addressing ----
    int abase[MT], bbase[NT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int pix = (wm * MT + m) * 32 + l31;
        const int ty = pix / TW, tx = pix % TW;
        abase[m] = ((ty * S) * HW + tx * S) * LDK + 4 * h;
    }
#pragma unroll
    for (int n = 0; n < NT; ++n) bbase[n] = ((wn * NT + n) * 32 + l31) * LDK + 4 * h;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;

    // ---- pipeline prologue ----
    const int ntaps = ts.n;
    LOAD_HALO(0);
    LOAD_W(ts.widx[0], 0);
    STORE_HALO(halo_buf(0), 0);
    STORE_W(wts_buf(0));
    __syncthreads();

    int s = 0;
    const int total = a.nchunks * ntaps;
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        int ng = (a.Cin - chunk * KC + 7) >> 3;
        ng = (ng > 2 || (has_tail && chunk == a.nchunks - 1)) ? 2 : ng;   // the tail chunk is stored channel-transposed
        const float* hb = halo_buf(chunk & 1);
        for (int t = 0; t < ntaps; ++t, ++s) {
            const bool has_next = (s + 1) < total;
            if (has_next) {
                const bool wrap = (t + 1 == ntaps);
                LOAD_W(s_widx[wrap ? 0 : t + 1], wrap ? chunk + 1 : chunk);
            }
            const bool halo_next = (chunk + 1 < a.nchunks);
            if (t == 0 && halo_next) LOAD_HALO(chunk + 1);

            const float* wb = wts_buf(s & 1);
            const int tapoff = s_tapoff[t];
            for (int g = 0; g < ng; ++g) {
                float av[MT][4], bv4[NT][4];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float4 f = *reinterpret_cast<const float4*>(hb + abase[m] + tapoff + 8 * g);
                    av[m][0] = f.x; av[m][1] = f.y; av[m][2] = f.z; av[m][3] = f.w;
                }
#pragma unroll
                for (int n = 0; n < NT; ++n) {
                    const float4 f = *reinterpret_cast<const float4*>(wb + bbase[n] + 8 * g);
                    bv4[n][0] = f.x; bv4[n][1] = f.y; bv4[n][2] = f.z; bv4[n][3] = f.w;
                }
#pragma unroll
                for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int n = 0; n < NT; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][kk], bv4[n][kk], acc[m][n], 0, 0, 0);
            }
            if (has_next) STORE_W(wts_buf((s + 1) & 1));
            if (t == ntaps - 1 && halo_next) STORE_HALO(halo_buf((chunk + 1) & 1), chunk + 1);
            __syncthreads();
        }
    }

#undef LOAD_HALO
#undef STORE_HALO
#undef LOAD_W
#undef STORE_W

    // ---- epilogue ----
    // Phased per 32x32 tile (offsets -> residual loads -> mask loads -> math -> stores) so that the
    // optional loads are issued back to back instead of one dependent round trip per element.
    const bool relu = a.flags & UNET_CONV_RELU;
    const int OS = a.OS;
    // per-image bases (64-bit, uniform) + 32-bit in-image pixel index
    const size_t img_pix = (size_t)img * a.OH * a.OW;
    float* yb = a.y + img_pix * a.y_cs + a.y_co;
    const float* resb = a.res ? a.res + img_pix * a.res_cs + a.res_co : nullptr;
    const float* maskb = a.mask ? a.mask + img_pix * a.mask_cs + a.mask_co : nullptr;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int cout = n0 + (wn * NT + n) * 32 + l31;
        const bool cvalid = cout < a.n_end;
        const float bv = (a.bias != nullptr && cvalid) ? a.bias[cout] : 0.f;
        float csum = 0.f, csq = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            int pidx[16];
            bool valid[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int pix = (wm * MT + m) * 32 + row;
                const int ty = pix / TW, tx = pix % TW;
                const int oyt = oy0 + ty, oxt = ox0 + tx;
                const int oy = oyt * OS + ts.py, ox = oxt * OS + ts.px;
                valid[r] = cvalid && oyt < a.TSH && oxt < a.TSW && oy < a.OH && ox < a.OW;
                pidx[r] = valid[r] ? (oy * a.OW + ox) : 0;  // pixel 0 is always addressable
            }
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[m][n][r] + bv;
            if (a.res != nullptr) {
                const int cc = cvalid ? cout : 0;
                float rv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = resb[pidx[r] * a.res_cs + cc];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] += rv[r];
            }
            if (relu) {
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = fmaxf(v[r], 0.f);
            }
            if (a.mask != nullptr) {
                const int cc = cvalid ? cout : 0;
                float mv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) mv[r] = maskb[pidx[r] * a.mask_cs + cc];
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = mv[r] > 0.f ? v[r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if (valid[r]) {
                    yb[pidx[r] * a.y_cs + cout] = v[r];
                    csum += v[r];
                    csq += v[r] * v[r];
                }
            }
        }
        if (a.colsum != nullptr) {
            csum += __shfl_xor(csum, 32);
            csq += __shfl_xor(csq, 32);
            if (h == 0 && cvalid) {
                const size_t row = ((size_t)blockIdx.z * a.mtiles + mtile) * WM + wm;
                a.colsum[row * a.Cout + cout] = csum;
                if (a.colsumsq != nullptr) a.colsumsq[row * a.Cout + cout] = csq;
            }
        }
    }
}

// ---- explicitly scheduled global loads for the 16x16x4 kernel -------------------------------------------------
// The compiler's s_waitcnt insertion merges the wait state of conditional loads conservatively (it drained vmcnt to 0 in
// the middle of the MFMA stream: a full L2 round trip per stage).  The main loop therefore issues its loads as inline asm
// (SGPR base + 32-bit lane offset) and places the vmcnt waits itself; vmcnt counts in issue order, and every path issues a
// fixed number of loads per stage (invalid items load from a clamped, always addressable offset and are zeroed later).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
// wave-uniform pointer -> SGPR pair for the saddr form.  The leading s_nop 4 of every load group covers the "VALU writes
// SGPR -> VMEM reads it" hazard (5 wait states): the compiler's hazard recognizer does not look inside inline asm.
__device__ __forceinline__ u64 sgpr_ptr(const void* p) {
    const u64 b = reinterpret_cast<u64>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
    return ((u64)hi << 32) | lo;
}
// operand-B tiles of one stage: tiles 0/1 at p + {0, TSTR}, tiles 2/3 at p + 2*TSTR + {0, TSTR}; one lane offset
template <int TSTR, int N>
__device__ __forceinline__ void gld_b(v4f (&d)[N], unsigned voff, const char* p) {
    static_assert(N == 2 || N == 4, "operand-B tiles per wave");
    const u64 s0 = sgpr_ptr(p);
    if constexpr (N == 2) {
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %3\n\tglobal_load_dwordx4 %1, %2, %3 offset:%4"
                     : "=&v"(d[0]), "=&v"(d[1]) : "v"(voff), "s"(s0), "n"(TSTR));
    } else {
        const u64 s1 = sgpr_ptr(p + 2 * TSTR);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %5\n\tglobal_load_dwordx4 %1, %4, %5 offset:%7\n\t"
                     "global_load_dwordx4 %2, %4, %6\n\tglobal_load_dwordx4 %3, %4, %6 offset:%7"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(voff), "s"(s0), "s"(s1), "n"(TSTR));
    }
}
// The halo items of one chunk: one base, one lane offset per item.  Executed in EVERY stage with `on` = all ones (fetch) or
// 0 (EXEC is cleared around the loads: nothing is fetched, the registers keep their values).  For the compiler the halo
// registers are thus one unbroken chain of tied asm operands -- no conditional definition, no phi, hence no register copy
// it could schedule between a load and its wait.
template <int N>
__device__ __forceinline__ void gld_halo(v4f (&h)[N], const unsigned (&vo)[N], const void* p, bool fetch) {
    static_assert(N == 4 || N == 10, "halo items per thread");
    const u64 sb = sgpr_ptr(p);
    const u64 on = sgpr_ptr(reinterpret_cast<const void*>(fetch ? ~0ull : 0ull));
    u64 sv;
    if constexpr (N == 4) {
        asm volatile("s_and_saveexec_b64 %[sv], %[on]\n\ts_nop 4\n\t"
                     "global_load_dwordx4 %[h0], %[o0], %[sb]\n\tglobal_load_dwordx4 %[h1], %[o1], %[sb]\n\t"
                     "global_load_dwordx4 %[h2], %[o2], %[sb]\n\tglobal_load_dwordx4 %[h3], %[o3], %[sb]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [h0] "+v"(h[0]), [h1] "+v"(h[1]), [h2] "+v"(h[2]), [h3] "+v"(h[3]), [sv] "=&s"(sv)
                     : [o0] "v"(vo[0]), [o1] "v"(vo[1]), [o2] "v"(vo[2]), [o3] "v"(vo[3]), [sb] "s"(sb), [on] "s"(on)
                     : "scc");   // s_and_saveexec writes SCC
    } else {
        asm volatile("s_and_saveexec_b64 %[sv], %[on]\n\ts_nop 4\n\t"
                     "global_load_dwordx4 %[h0], %[o0], %[sb]\n\tglobal_load_dwordx4 %[h1], %[o1], %[sb]\n\t"
                     "global_load_dwordx4 %[h2], %[o2], %[sb]\n\tglobal_load_dwordx4 %[h3], %[o3], %[sb]\n\t"
                     "global_load_dwordx4 %[h4], %[o4], %[sb]\n\tglobal_load_dwordx4 %[h5], %[o5], %[sb]\n\t"
                     "global_load_dwordx4 %[h6], %[o6], %[sb]\n\tglobal_load_dwordx4 %[h7], %[o7], %[sb]\n\t"
                     "global_load_dwordx4 %[h8], %[o8], %[sb]\n\tglobal_load_dwordx4 %[h9], %[o9], %[sb]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [h0] "+v"(h[0]), [h1] "+v"(h[1]), [h2] "+v"(h[2]), [h3] "+v"(h[3]), [h4] "+v"(h[4]), [h5] "+v"(h[5]),
                       [h6] "+v"(h[6]), [h7] "+v"(h[7]), [h8] "+v"(h[8]), [h9] "+v"(h[9]), [sv] "=&s"(sv)
                     : [o0] "v"(vo[0]), [o1] "v"(vo[1]), [o2] "v"(vo[2]), [o3] "v"(vo[3]), [o4] "v"(vo[4]), [o5] "v"(vo[5]),
                       [o6] "v"(vo[6]), [o7] "v"(vo[7]), [o8] "v"(vo[8]), [o9] "v"(vo[9]), [sb] "s"(sb), [on] "s"(on)
                     : "scc");   // s_and_saveexec writes SCC
    }
}
// s_waitcnt vmcnt(0) that also "defines" every register the outstanding loads write (operand-B tiles and halo items), so
// that no consumer is scheduled above it
template <int NB, int NH>
__device__ __forceinline__ void wait_loads(v4f (&b)[NB], v4f (&h)[NH]) {
    static_assert((NB == 2 || NB == 4) && (NH == 4 || NH == 10), "register groups");
    // ONE statement (a tied operand's input copy, if the compiler ever made one, must not be able to slip in front of the
    // s_waitcnt of a sibling statement); 14 tied operands = 28 of the 30 asm operands allowed
#define UNET_H4 "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3])
#define UNET_H10 UNET_H4, "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), "+v"(h[8]), "+v"(h[9])
    if constexpr (NB == 2 && NH == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), UNET_H4);
    else if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), UNET_H10);
    else if constexpr (NH == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), UNET_H4);
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), UNET_H10);
#undef UNET_H4
#undef UNET_H10
}

// ------------------------------------------------------------------------------------------------
// Same workgroup geometry on v_mfma_f32_16x16x4_f32 (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C: col=l&15,
// row=4*(l>>4)+reg).  Each wave owns (2*MT) x (2*NT) tiles of 16x16; the LDS traffic per MFMA cycle is the
// same as the 32x32x2 form (one ds_read_b128 per operand tile per 16-channel chunk = 4 k-steps), but
//   * output-channel tiles that lie entirely beyond Cout are SKIPPED per wave (uniform branch): produced
//     channel counts are padded to 16 instead of 32/64/128 (Cout = 96 exact, 100 -> 112, 192 exact);
//   * a reduction tail of r < 16 channels costs ceil(r/4) MFMA steps (ds_read_b32 operands) instead of 4.
template <int TW, int MT, int NT, int WM, int WN, int HIT>
__global__ __launch_bounds__(WM* WN * 64, HIT == 4 ? 3 : 2) void conv_igemm16_kernel(const KArgs a) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, TH = BM / TW, NTH = WM * WN * 64;
    constexpr int M16 = 2 * MT, N16 = 2 * NT;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];

    // the wave index is made a scalar so that everything derived from it (tile skipping, operand bases) is wave-uniform
    // control flow (s_cbranch) instead of exec masking
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, kq = lane >> 4;
    const TapSet& ts = a.taps[blockIdx.z];

    // block coordinates (the divisions run on the VALU: readfirstlane returns the results to SGPRs, so that all the
    // pointer arithmetic derived from them is scalar)
    // XCD-aware tile order: workgroups are dealt round-robin to the 8 XCDs in launch order; give every XCD ONE contiguous range
    // of tiles, so that the workgroups resident on an XCD at the same time are spatial neighbours (shared halo columns / rows)
    // and the N-tiles of one pixel tile -- their input reads hit that XCD's L2.  The grid is padded to a multiple of 8.
    const int per_xcd = (int)(gridDim.x >> 3);
    int bid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (bid >= a.mtiles * a.ntn) return;
    const int nt = __builtin_amdgcn_readfirstlane(bid % a.ntn); bid /= a.ntn;
    const int mtile = __builtin_amdgcn_readfirstlane(bid);
    const int tx_t = __builtin_amdgcn_readfirstlane(bid % a.tiles_x); bid /= a.tiles_x;
    const int ty_t = __builtin_amdgcn_readfirstlane(bid % a.tiles_y);
    const int img = __builtin_amdgcn_readfirstlane(bid / a.tiles_y);
    const int oy0 = ty_t * TH, ox0 = tx_t * TW;
    const int n0 = a.n_base + nt * BN;

    const int S = a.S;
    const int HH = (TH - 1) * S + ts.ext_y, HW = (TW - 1) * S + ts.ext_x;
    const int HPIX = HH * HW;
    // LDS map: [0,32) unused | halo[2].  The filter operand never touches LDS in this kernel: the packed image
    // wp[tap][chunk][col][16] is already in MFMA B-operand order (lane (kq, l15) of tile n needs the float4 at
    // [col = tile*16 + l15][4kq..4kq+3]: 64 lanes = one contiguous 1 KiB), so every wave loads its own tiles
    // global -> VGPR one stage ahead (L1/L2 hits: all blocks stream the same slab) and the only barrier left is the
    // one per 16-channel chunk that hands over the halo tile (was: one per tap).
    float* lds0 = smem + 32;
    auto halo_buf = [&](int b) -> float* { return lds0 + b * (HPIX * LDK); };
    // tap tables: 4-bit fields of two scalars, decoded on the SALU (no LDS lookup, no lgkmcnt round trip per stage)
    const unsigned long long dpack = ts.dpack, wpack = ts.wpack;
#define TAP_OFF(t_) ({ const unsigned d_ = (unsigned)(dpack >> (4 * (t_))); (int)(((d_ & 3u) * HW + ((d_ >> 2) & 3u)) * LDK); })
#define TAP_WIDX(t_) ((int)((unsigned)(wpack >> (4 * (t_))) & 15u))

    const float* xb = a.x + (size_t)img * a.IH * a.IW * a.x_cs;
    const int iy0 = oy0 * S + ts.min_dy, ix0 = ox0 * S + ts.min_dx;
    int goff[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int e = tid + it * NTH;
        const int p = e >> 2, q = e & 3;
        const int hy = p / HW, hx = p - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool inb = (e < HPIX * 4) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
        goff[it] = inb ? ((iy * a.IW + ix) * a.x_cs + a.x_co + 4 * q) : -1;
    }

    v4f hreg[HIT];
    const bool has_tail = (a.Cin & 15) != 0;

    // halo items: always HIT loads; an item outside the image / beyond the channels loads offset 0 and is zeroed at the store
#define HALO_OK(it_, c0_) (goff[it_] >= 0 && ((c0_) + 4 * ((tid + (it_) * NTH) & 3)) < a.Cin4)
#define LOAD_HALO(chunk_, on_) do { const int c0_ = (chunk_) * KC; unsigned vo_[HIT]; \
        if (on_) { _Pragma("unroll") for (int it = 0; it < HIT; ++it) vo_[it] = HALO_OK(it, c0_) ? (unsigned)(goff[it] + c0_) * 4u : 0u; } \
        else { _Pragma("unroll") for (int it = 0; it < HIT; ++it) vo_[it] = 0u; } \
        gld_halo<HIT>(hreg, vo_, xb, (on_)); } while (0)
#define STORE_HALO(dst_, chunk_) do { const int c0_ = (chunk_) * KC; const bool tail_ = has_tail && (chunk_) == a.nchunks - 1; \
        _Pragma("unroll") for (int it = 0; it < HIT; ++it) { \
            const int e_ = tid + it * NTH; \
            if (e_ < HPIX * 4) { \
                const v4f v_ = HALO_OK(it, c0_) ? hreg[it] : (v4f){0.f, 0.f, 0.f, 0.f}; \
                if (tail_) { float* row_ = (dst_) + (e_ >> 2) * LDK + (e_ & 3); row_[0] = v_.x; row_[4] = v_.y; row_[8] = v_.z; row_[12] = v_.w; } \
                else *reinterpret_cast<v4f*>((dst_) + (e_ >> 2) * LDK + (e_ & 3) * 4) = v_; \
            } } } while (0)

    // A-operand row bases (floats) inside the halo tile
    int abase[M16];
#pragma unroll
    for (int m = 0; m < M16; ++m) {
        const int pix = (wm * M16 + m) * 16 + l15;
        const int ty = pix / TW, tx = pix % TW;
        abase[m] = ((ty * S) * HW + tx * S) * LDK + 4 * kq;
    }
    // 16-wide output-channel tiles are dealt round-robin to the WN waves (tile n of this wave = block tile n*WN + wn), so the
    // tiles that survive the Cout cut-off are balanced between the waves' MFMA pipes
    // number of this wave's tiles that contain a real channel: tiles n with (n*WN + wn)*16 < Cout - n0
    int nvalid = ((a.n_end - n0 + 15) / 16 - wn + WN - 1) / WN;
    nvalid = (a.n_end - n0 <= wn * 16) ? 0 : (nvalid > N16 ? N16 : nvalid);
    // operand B: uniform slab pointer + one 32-bit per-lane byte offset (tile n is n * WN * 16 columns = n * WN KiB further),
    // loaded unconditionally (the packed image is padded to the block's 128 columns) so every stage issues exactly N16 loads
    const unsigned lane_b = (unsigned)(((wn * 16 + l15) * KC + 4 * kq) * sizeof(float));
    const char* wbase = reinterpret_cast<const char*>(a.wp + (size_t)img * a.wp_stride + (size_t)n0 * KC);
    const size_t slab_b = (size_t)a.coutPad * KC * sizeof(float);
    constexpr int TSTR = WN * 16 * KC * 4;    // bytes between two of this wave's tiles (<= 2 KiB: fits the immediate offset)
    v4f b0[N16], b1[N16];
#define LOAD_B(dst_, widx_, chunk_) gld_b<TSTR, N16>((dst_), lane_b, wbase + (size_t)((widx_) * a.nchunks + (chunk_)) * slab_b)

    f32x4 acc[M16][N16];
#pragma unroll
    for (int m = 0; m < M16; ++m)
#pragma unroll
        for (int n = 0; n < N16; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntaps = ts.n;
#pragma unroll
    for (int it = 0; it < HIT; ++it) hreg[it] = (v4f){0.f, 0.f, 0.f, 0.f};
    LOAD_HALO(0, true);
    LOAD_B(b0, ts.widx[0], 0);
    wait_loads(b0, hreg);
    STORE_HALO(halo_buf(0), 0);
    __syncthreads();

    // Stage = (chunk, tap).  The loop body is written twice (operand B ping-pongs between b0 and b1 without register copies).
    // Loads in flight: the next stage's B tiles and, during the first tap of a chunk, the next chunk's halo items; both are
    // issued at the top of a stage and waited for at its end (behind the stage's 64 MFMAs).
    int t = 0, chunk = 0;
    int ksteps = a.Cin >= KC ? 4 : ((a.Cin + 3) >> 2);   // tail chunk: channel-transposed, step kk = channels 4kk..4kk+3
    const float* hb = halo_buf(0);
    // Every stage ends in ONE wait asm that (re)defines both the B tiles and the halo registers, on every path, so the
    // compiler has no merge point of its own between a load and its wait where it could copy a register that is still in
    // flight; tests/test_isa_cpu.py checks the generated ISA for exactly that.
#define STAGE_BODY(bu_, bl_, FULL_) do { \
        LOAD_HALO(chunk + 1, t == 0 && chunk + 1 < a.nchunks);   /* EXEC-masked off in the other stages */ \
        /* after the last stage: a dummy reload of slab 0 keeps the number of loads per stage fixed */ \
        LOAD_B(bl_, has_next_ ? TAP_WIDX(tn_) : 0, has_next_ ? cn_ : 0); \
        const float* ha_ = hb + TAP_OFF(t); \
        float av_[M16][4]; \
        _Pragma("unroll") for (int m = 0; m < M16; ++m) { \
            const float4 f_ = *reinterpret_cast<const float4*>(ha_ + abase[m]); \
            av_[m][0] = f_.x; av_[m][1] = f_.y; av_[m][2] = f_.z; av_[m][3] = f_.w; \
        } \
        if (FULL_) {   /* full 16-channel chunk: one uniform branch per 16-wide output tile, 16 MFMAs behind each */ \
            _Pragma("unroll") for (int n = 0; n < N16; ++n) { \
                if (n < nvalid) { \
                    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) \
                        _Pragma("unroll") for (int m = 0; m < M16; ++m) \
                            acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[m][kk], (bu_)[n][kk], acc[m][n], 0, 0, 0); \
                } \
            } \
        } else {       /* may be the tail chunk: k-steps beyond the real channels are skipped as well */ \
            _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { \
                if (kk < ksteps) { \
                    _Pragma("unroll") for (int n = 0; n < N16; ++n) { \
                        if (n < nvalid) { \
                            _Pragma("unroll") for (int m = 0; m < M16; ++m) \
                                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av_[m][kk], (bu_)[n][kk], acc[m][n], 0, 0, 0); \
                        } \
                    } \
                } \
            } \
        } \
        wait_loads(bl_, hreg); \
    } while (0)
#define STAGE(bu_, bl_, FULL_) do { \
        int tn_ = t + 1, cn_ = chunk; \
        if (tn_ == ntaps) { tn_ = 0; cn_ = chunk + 1; } \
        const bool has_next_ = cn_ < a.nchunks; \
        STAGE_BODY(bu_, bl_, FULL_); \
        if (tn_ == 0 && has_next_) { \
            STORE_HALO(halo_buf(cn_ & 1), cn_); \
            __syncthreads(); \
            hb = halo_buf(cn_ & 1); \
            const int rem_ = a.Cin - cn_ * KC; \
            ksteps = rem_ >= KC ? 4 : ((rem_ + 3) >> 2); \
        } \
        t = tn_; chunk = cn_; \
    } while (0)

    // The stages of the full 16-channel chunks run first (an even number of them, so that the b0/b1 roles line up), with
    // compile-time 4 k-steps; the remaining stages -- the tail chunk when Cin % 16 != 0 -- run the guarded form.
    const int total = a.nchunks * ntaps;
    const int total_full = ((has_tail ? a.nchunks - 1 : a.nchunks) * ntaps) & ~1;
    for (int s = 0; s < total_full; s += 2) {
        STAGE(b0, b1, true);
        STAGE(b1, b0, true);
    }
    for (int s = total_full; s < total; s += 2) {
        STAGE(b0, b1, false);
        if (s + 1 < total) STAGE(b1, b0, false);
    }
#undef STAGE_BODY
#undef TAP_OFF
#undef TAP_WIDX
#undef STAGE
#undef HALO_OK
#undef LOAD_HALO
#undef STORE_HALO
#undef LOAD_B

    // ---- epilogue (phased as in the 32x32 kernel) ----
    const bool relu = a.flags & UNET_CONV_RELU;
    const int OS = a.OS;
    const size_t img_pix = (size_t)img * a.OH * a.OW;
    float* yb = a.y + img_pix * a.y_cs + a.y_co;
    const float* resb = a.res ? a.res + img_pix * a.res_cs + a.res_co : nullptr;
    const float* maskb = a.mask ? a.mask + img_pix * a.mask_cs + a.mask_co : nullptr;
    // pixel indices of this lane's 4 accumulator rows for every M tile
    int pidx[M16][4];
    bool pval[M16][4];
#pragma unroll
    for (int m = 0; m < M16; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pix = (wm * M16 + m) * 16 + 4 * kq + r;
            const int ty = pix / TW, tx = pix % TW;
            const int oyt = oy0 + ty, oxt = ox0 + tx;
            const int oy = oyt * OS + ts.py, ox = oxt * OS + ts.px;
            pval[m][r] = oyt < a.TSH && oxt < a.TSW && oy < a.OH && ox < a.OW;
            pidx[m][r] = pval[m][r] ? (oy * a.OW + ox) : 0;
        }
#pragma unroll
    for (int n = 0; n < N16; ++n) {
        if (n >= nvalid) continue;
        const int cout = n0 + (n * WN + wn) * 16 + l15;
        const bool cvalid = cout < a.n_end;
        const int cc = cvalid ? cout : 0;
        const float bvv = (a.bias != nullptr && cvalid) ? a.bias[cout] : 0.f;
        float csum = 0.f, csq = 0.f;
        float v[M16][4];
#pragma unroll
        for (int m = 0; m < M16; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[m][r] = acc[m][n][r] + bvv;
        if (resb != nullptr) {
            float rv[M16][4];
#pragma unroll
            for (int m = 0; m < M16; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) rv[m][r] = resb[pidx[m][r] * a.res_cs + cc];
#pragma unroll
            for (int m = 0; m < M16; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m][r] += rv[m][r];
        }
        if (relu) {
#pragma unroll
            for (int m = 0; m < M16; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m][r] = fmaxf(v[m][r], 0.f);
        }
        if (maskb != nullptr) {
            float mv[M16][4];
#pragma unroll
            for (int m = 0; m < M16; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) mv[m][r] = maskb[pidx[m][r] * a.mask_cs + cc];
#pragma unroll
            for (int m = 0; m < M16; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m][r] = mv[m][r] > 0.f ? v[m][r] : 0.f;
        }
#pragma unroll
        for (int m = 0; m < M16; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (cvalid && pval[m][r]) {
                    yb[pidx[m][r] * a.y_cs + cout] = v[m][r];
                    csum += v[m][r];
                    csq += v[m][r] * v[m][r];
                }
            }
        if (a.colsum != nullptr) {
            csum += __shfl_xor(csum, 16);
            csq += __shfl_xor(csq, 16);
            csum += __shfl_xor(csum, 32);
            csq += __shfl_xor(csq, 32);
            if (kq == 0 && cvalid) {
                const size_t row = ((size_t)blockIdx.z * a.mtiles + mtile) * WM + wm;
                a.colsum[row * a.Cout + cout] = csum;
                if (a.colsumsq != nullptr) a.colsumsq[row * a.Cout + cout] = csq;
            }
        }
    }
    // tiles skipped by this wave still owe zeros to the column-sum rows (their channels are >= Cout: nothing to write)
}

// ------------------------------------------------------------------ packing

__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int T,
                                    int mode, int nchunks, int outPad, size_t total) {
    // wp[tap][chunk][o][16]; mode 0: o = cout, reduction r = cin; mode 1: o = cin, reduction r = cout
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int rr = (int)(i & 15);
        size_t j = i >> 4;
        const int o = (int)(j % outPad); j /= outPad;
        const int chunk = (int)(j % nchunks);
        const int tap = (int)(j / nchunks);
        const int red = mode == 0 ? Cin : Cout;
        const bool tail = (red & 15) != 0 && chunk == nchunks - 1;
        const int r = chunk * 16 + (tail ? (4 * (rr & 3) + (rr >> 2)) : rr);
        float v = 0.f;
        if (mode == 0) {
            if (o < Cout && r < Cin) v = w[((size_t)o * Cin + r) * T + tap];
        } else {
            if (o < Cin && r < Cout) v = w[((size_t)r * Cin + o) * T + tap];
        }
        wp[i] = v;
    }
}

// 1x1 "weights" taken from an activation matrix (self-attention: the per-image operands F, G, H are activations):
// element (out o, reduction r) = w[o*so + r*sr]; same packed image as pack_weights_kernel with T = 1.
__global__ void pack_weights_strided_kernel(const float* __restrict__ w, long long so, long long sr, float* __restrict__ wp, int O, int R,
                                            int nchunks, int outPad, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int rr = (int)(i & 15);
        size_t j = i >> 4;
        const int o = (int)(j % outPad);
        const int chunk = (int)(j / outPad);
        const bool tail = (R & 15) != 0 && chunk == nchunks - 1;
        const int r = chunk * 16 + (tail ? (4 * (rr & 3) + (rr >> 2)) : rr);
        wp[i] = (o < O && r < R) ? w[(long long)o * so + (long long)r * sr] : 0.f;
    }
}

// ------------------------------------------------------------------ host side

// MFMA shape of the conv kernels: 16 = v_mfma_f32_16x16x4_f32 with per-tile skipping (default), 32 = v_mfma_f32_32x32x2_f32
static int g_mfma_shape = 16;

struct Plan {
    KArgs k;
    int tw, bm, bn, hit, nparity, mf;
    size_t lds_bytes;
    dim3 grid;
};

int make_plan(const unet_conv_desc* d, Plan* p) {
    UNET_CHECK_ARG(d != nullptr, "conv: null desc");
    UNET_CHECK_ARG(d->x && d->wp && d->y, "conv: null tensor pointer");
    UNET_CHECK_ARG(d->ks == 1 || d->ks == 3, "conv: ks must be 1 or 3 (got %d)", d->ks);
    UNET_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv: stride must be 1 or 2 (got %d)", d->stride);
    UNET_CHECK_ARG(!(d->ks == 1 && d->stride != 1), "conv: 1x1 stride 2 unsupported");
    UNET_CHECK_ARG(d->N > 0 && d->IH > 0 && d->IW > 0 && d->OH > 0 && d->OW > 0 && d->Cin > 0 && d->Cout > 0, "conv: bad dims");
    UNET_CHECK_ARG(unet::slice_ok(d->x_cs, d->x_co, d->Cin), "conv: bad x slice cs=%d co=%d C=%d", d->x_cs, d->x_co, d->Cin);
    UNET_CHECK_ARG(unet::slice_ok(d->y_cs, d->y_co, d->Cout), "conv: bad y slice cs=%d co=%d C=%d", d->y_cs, d->y_co, d->Cout);
    UNET_CHECK_ARG(unet::aligned16(d->x) && unet::aligned16(d->wp), "conv: x/wp must be 16-byte aligned");
    if (d->res) UNET_CHECK_ARG(unet::slice_ok(d->res_cs, d->res_co, d->Cout), "conv: bad res slice");
    if (d->flags & UNET_CONV_MASK) UNET_CHECK_ARG(d->mask && unet::slice_ok(d->mask_cs, d->mask_co, d->Cout), "conv: bad mask slice");
    const int pad = (d->ks - 1) / 2;
    if (d->kind == UNET_CONV_FWD) {
        UNET_CHECK_ARG(d->OH == (d->IH + 2 * pad - d->ks) / d->stride + 1 && d->OW == (d->IW + 2 * pad - d->ks) / d->stride + 1,
                       "conv fwd: output dims %dx%d inconsistent with input %dx%d ks %d stride %d", d->OH, d->OW, d->IH, d->IW, d->ks, d->stride);
    } else if (d->kind == UNET_CONV_DGRAD) {
        // here I* = dims of the forward OUTPUT gradient, O* = dims of the forward INPUT
        UNET_CHECK_ARG(d->IH == (d->OH + 2 * pad - d->ks) / d->stride + 1 && d->IW == (d->OW + 2 * pad - d->ks) / d->stride + 1,
                       "conv dgrad: grad dims %dx%d inconsistent with input dims %dx%d", d->IH, d->IW, d->OH, d->OW);
    } else {
        UNET_CHECK_ARG(false, "conv: bad kind %d", d->kind);
    }
    // the image-local offsets are 32-bit
    UNET_CHECK_ARG((long long)d->IH * d->IW * d->x_cs < (1ll << 31) && (long long)d->OH * d->OW * d->y_cs < (1ll << 31) &&
                       (long long)d->OH * d->OW * (d->res ? d->res_cs : 1) < (1ll << 31) &&
                       (long long)d->OH * d->OW * ((d->flags & UNET_CONV_MASK) ? d->mask_cs : 1) < (1ll << 31),
                   "conv: image too large for 32-bit in-image offsets");

    KArgs& k = p->k;
    memset(&k, 0, sizeof(k));
    k.x = d->x; k.wp = d->wp; k.bias = d->bias; k.res = d->res; k.mask = (d->flags & UNET_CONV_MASK) ? d->mask : nullptr;
    k.y = d->y; k.colsum = d->colsum; k.colsumsq = d->colsumsq;
    k.x_cs = d->x_cs; k.x_co = d->x_co; k.res_cs = d->res_cs; k.res_co = d->res_co;
    k.mask_cs = d->mask_cs; k.mask_co = d->mask_co; k.y_cs = d->y_cs; k.y_co = d->y_co;
    k.N = d->N; k.IH = d->IH; k.IW = d->IW; k.Cin = d->Cin; k.Cin4 = unet::roundup(d->Cin, 4);
    k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout;
    // optional produced-channel range (a wide layer can be issued as several launches with different channel-block widths)
    const int cols = d->cout_count > 0 ? d->cout_count : d->Cout;
    UNET_CHECK_ARG(d->cout_begin >= 0 && (d->cout_begin & 15) == 0 && d->cout_begin + cols <= d->Cout, "conv: bad cout range [%d,+%d) of %d",
                   d->cout_begin, cols, d->Cout);
    k.n_base = d->cout_begin; k.n_end = d->cout_begin + cols;
    UNET_CHECK_ARG(d->wp_img_stride >= 0 && (d->wp_img_stride & 3) == 0, "conv: bad wp_img_stride");
    k.wp_stride = d->wp_img_stride;
    k.flags = d->flags;
    k.nchunks = unet::cdiv(d->Cin, KC);
    k.coutPad = unet::roundup(d->Cout, 128);
    p->nparity = 1;
    k.S = 1; k.OS = 1; k.TSH = d->OH; k.TSW = d->OW;

    const int T = d->ks * d->ks;
    if (d->kind == UNET_CONV_FWD) {
        k.S = d->stride;
        TapSet& t = k.taps[0];
        t.n = T; t.min_dy = -pad; t.min_dx = -pad; t.ext_y = d->ks; t.ext_x = d->ks; t.py = t.px = 0;
        for (int r = 0; r < d->ks; ++r)
            for (int s = 0; s < d->ks; ++s) {
                const int i = r * d->ks + s;
                t.dy[i] = (signed char)(r - pad); t.dx[i] = (signed char)(s - pad); t.widx[i] = (signed char)i;
            }
    } else if (d->stride == 1) {
        TapSet& t = k.taps[0];
        t.n = T; t.min_dy = -pad; t.min_dx = -pad; t.ext_y = d->ks; t.ext_x = d->ks; t.py = t.px = 0;
        for (int r = 0; r < d->ks; ++r)
            for (int s = 0; s < d->ks; ++s) {
                const int i = r * d->ks + s;
                t.dy[i] = (signed char)(pad - r); t.dx[i] = (signed char)(pad - s); t.widx[i] = (signed char)i;
            }
    } else {
        // stride-2 3x3 pad-1 dgrad: 4 output parity classes.  Output row 2*o+py receives
        //   py = 0: r = 1 from grad row o        py = 1: r = 0 from grad row o+1, r = 2 from grad row o
        k.OS = 2; k.TSH = (d->OH + 1) / 2; k.TSW = (d->OW + 1) / 2;
        p->nparity = 4;
        for (int py = 0; py < 2; ++py)
            for (int px = 0; px < 2; ++px) {
                TapSet& t = k.taps[py * 2 + px];
                int rs[2], rdy[2], nr, ss[2], sdx[2], ns;
                if (py == 0) { nr = 1; rs[0] = 1; rdy[0] = 0; } else { nr = 2; rs[0] = 0; rdy[0] = 1; rs[1] = 2; rdy[1] = 0; }
                if (px == 0) { ns = 1; ss[0] = 1; sdx[0] = 0; } else { ns = 2; ss[0] = 0; sdx[0] = 1; ss[1] = 2; sdx[1] = 0; }
                t.n = nr * ns; t.min_dy = 0; t.min_dx = 0; t.ext_y = (py == 0) ? 1 : 2; t.ext_x = (px == 0) ? 1 : 2;
                t.py = py; t.px = px;
                int i = 0;
                for (int a = 0; a < nr; ++a)
                    for (int b = 0; b < ns; ++b, ++i) {
                        t.dy[i] = (signed char)rdy[a]; t.dx[i] = (signed char)sdx[b]; t.widx[i] = (signed char)(rs[a] * 3 + ss[b]);
                    }
            }
    }

    for (int z = 0; z < p->nparity; ++z) {
        TapSet& t = k.taps[z];
        t.dpack = 0; t.wpack = 0;
        for (int i = 0; i < t.n; ++i) {
            const int dyi = t.dy[i] - t.min_dy, dxi = t.dx[i] - t.min_dx;
            UNET_CHECK_ARG(dyi >= 0 && dyi < 4 && dxi >= 0 && dxi < 4 && t.widx[i] >= 0 && t.widx[i] < 16, "conv: tap table out of range");
            t.dpack |= (unsigned long long)(dyi | (dxi << 2)) << (4 * i);
            t.wpack |= (unsigned long long)t.widx[i] << (4 * i);
        }
    }

    p->tw = k.TSW >= 32 ? 32 : (k.TSW >= 16 ? 16 : 8);
    p->bn = cols <= 32 ? 32 : (cols <= 64 ? 64 : 128);
    p->bm = 128;
    p->hit = (k.S == 2) ? 10 : 4;
    p->mf = g_mfma_shape;
    // small problems (deep 16x16 / 32x32 stages): shrink the tile until the grid can fill 256 CUs x 2
    auto blocks = [&](int bm, int bn) {
        const int th_ = bm / p->tw;
        return (long long)d->N * unet::cdiv(k.TSH, th_) * unet::cdiv(k.TSW, p->tw) * unet::cdiv(cols, bn) * p->nparity;
    };
    if (p->bn >= 64 && blocks(128, p->bn) < 400) {
        p->bm = 64;
        if (p->bn == 128 && blocks(64, 128) < 400) p->bn = 64;
    }
    const int th = p->bm / p->tw;
    k.tiles_y = unet::cdiv(k.TSH, th);
    k.tiles_x = unet::cdiv(k.TSW, p->tw);
    k.ntn = unet::cdiv(cols, p->bn);
    UNET_CHECK_ARG(k.n_base + k.ntn * p->bn <= k.coutPad, "conv: cout range leaves the packed filter image");
    // in-image element offsets are 32-bit inside the kernels (the image index is applied in 64 bits)
    UNET_CHECK_ARG((long long)d->IH * d->IW * d->x_cs < (1ll << 31) && (long long)d->OH * d->OW * d->y_cs < (1ll << 31) &&
                   (d->res == nullptr || (long long)d->OH * d->OW * d->res_cs < (1ll << 31)) &&
                   (d->mask == nullptr || (long long)d->OH * d->OW * d->mask_cs < (1ll << 31)),
                   "conv: one image of a tensor exceeds 2^31 elements");
    const long long mtiles_ll = (long long)d->N * k.tiles_y * k.tiles_x;
    UNET_CHECK_ARG(mtiles_ll * k.ntn < (1ll << 31) - 8, "conv: grid too large (%lld pixel tiles x %d channel blocks)", mtiles_ll, k.ntn);
    k.mtiles = (int)mtiles_ll;
    int max_hpix = 0;
    for (int z = 0; z < p->nparity; ++z) {
        const int hh = (th - 1) * k.S + k.taps[z].ext_y, hw = (p->tw - 1) * k.S + k.taps[z].ext_x;
        if (hh * hw > max_hpix) max_hpix = hh * hw;
    }
    UNET_CHECK_ARG(max_hpix * 4 <= p->hit * 256, "conv: halo tile too large (%d pixels)", max_hpix);
    // the 16x16x4 kernel keeps no filter slab in LDS (operand B goes global -> VGPR)
    p->lds_bytes = (size_t)(32 + 2 * max_hpix * LDK + (p->mf == 16 ? 0 : 2 * p->bn * LDK)) * sizeof(float);
    // (the 16x16x4 kernel remaps block ids XCD-aware and needs a multiple of 8; the surplus workgroups exit at once)
    p->grid = dim3((unsigned)unet::roundup((int)((long long)k.mtiles * k.ntn), p->mf == 16 ? 8 : 1), 1, (unsigned)p->nparity);
    UNET_CHECK_ARG((long long)k.mtiles * k.ntn < (1ll << 31), "conv: grid too large");
    return UNET_OK;
}

template <int TW, int MT, int NT, int WM, int WN, int HIT>
int launch_cfg(const Plan& p, hipStream_t st) {
    if (p.mf == 16) {
        auto kern = conv_igemm16_kernel<TW, MT, NT, WM, WN, HIT>;
        static unsigned long long configured = 0;  // per instantiation, one bit per device
        if (unet::first_use_on_device(&configured))
            UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(kern, p.grid, dim3(WM * WN * 64), p.lds_bytes, st, p.k);
        UNET_CHECK_LAUNCH();
        return UNET_OK;
    }
    auto kern = conv_igemm_kernel<TW, MT, NT, WM, WN, HIT>;
    static unsigned long long configured = 0;  // per instantiation, one bit per device
    if (unet::first_use_on_device(&configured))
        UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, p.grid, dim3(WM * WN * 64), p.lds_bytes, st, p.k);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <int TW, int HIT>
int launch_bn(const Plan& p, hipStream_t st) {
    if (p.bm == 64) {
        if (p.bn == 64) return launch_cfg<TW, 1, 1, 2, 2, HIT>(p, st);
        return launch_cfg<TW, 1, 2, 2, 2, HIT>(p, st);
    }
    switch (p.bn) {
        case 32: return launch_cfg<TW, 1, 1, 4, 1, HIT>(p, st);
        case 64: return launch_cfg<TW, 2, 1, 2, 2, HIT>(p, st);
        default: return launch_cfg<TW, 2, 2, 2, 2, HIT>(p, st);
    }
}

template <int HIT>
int launch_tw(const Plan& p, hipStream_t st) {
    switch (p.tw) {
        case 32: return launch_bn<32, HIT>(p, st);
        case 16: return launch_bn<16, HIT>(p, st);
        default: return launch_bn<8, HIT>(p, st);
    }
}

}  // namespace

extern "C" int unet_conv2d_colsum_rows(const unet_conv_desc* d) {
    Plan p;
    int rc = make_plan(d, &p);
    if (rc != UNET_OK) return rc;
    const int wm = (p.bn == 32 && p.bm == 128) ? 4 : 2;
    return p.nparity * p.k.mtiles * wm;
}

extern "C" int unet_set_mfma_shape(int shape) {
    UNET_CHECK_ARG(shape == 16 || shape == 32, "mfma shape must be 16 or 32");
    g_mfma_shape = shape;
    return UNET_OK;
}

extern "C" int unet_conv2d_variant(const unet_conv_desc* d) {
    Plan p;
    int rc = make_plan(d, &p);
    if (rc != UNET_OK) return rc;
    return p.tw * 10000 + p.bn * 10 + (p.hit == 10 ? 1 : 0) + (p.bm == 64 ? 5 : 0);
}

extern "C" int unet_conv2d(const unet_conv_desc* d, void* stream) {
    Plan p;
    int rc = make_plan(d, &p);
    if (rc != UNET_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    return (p.hit == 10) ? launch_tw<10>(p, st) : launch_tw<4>(p, st);
}

extern "C" size_t unet_pack_weights_size(int Cout, int Cin, int ks, int mode) {
    const int T = ks * ks;
    const int red = mode == 0 ? Cin : Cout, out = mode == 0 ? Cout : Cin;
    return (size_t)T * unet::cdiv(red, KC) * unet::roundup(out, 128) * KC;
}

extern "C" int unet_pack_weights_strided(const float* w, long long so, long long sr, float* wp, int O, int R, void* stream) {
    UNET_CHECK_ARG(w && wp && O > 0 && R > 0, "pack_weights_strided: bad args");
    const int nchunks = unet::cdiv(R, KC), outPad = unet::roundup(O, 128);
    const size_t total = (size_t)nchunks * outPad * KC;
    hipLaunchKernelGGL(pack_weights_strided_kernel, dim3(unet::ew_grid((long long)total, 256)), dim3(256), 0, (hipStream_t)stream, w, so, sr,
                       wp, O, R, nchunks, outPad, total);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_pack_weights(const float* w, float* wp, int Cout, int Cin, int ks, int mode, void* stream) {
    UNET_CHECK_ARG(w && wp, "pack_weights: null pointer");
    UNET_CHECK_ARG((ks == 1 || ks == 3) && (mode == 0 || mode == 1) && Cout > 0 && Cin > 0, "pack_weights: bad args");
    const int T = ks * ks;
    const int red = mode == 0 ? Cin : Cout, out = mode == 0 ? Cout : Cin;
    const int nchunks = unet::cdiv(red, KC), outPad = unet::roundup(out, 128);
    const size_t total = (size_t)T * nchunks * outPad * KC;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(unet::ew_grid((long long)total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wp, Cout, Cin, T, mode, nchunks, outPad, total);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
