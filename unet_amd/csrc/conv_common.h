// Shared by the fp32 (conv_igemm.hip) and bf16 (conv_bf16.hip) implicit-GEMM convolution kernels of libunet_hip.so: tap tables,
// kernel arguments, the host-side launch plan, and the inline-asm global-load helpers of the direct-operand main loop.
// Both kernels move the SAME bytes per stage -- a reduction chunk is 64 bytes per pixel (16 fp32 or 32 bf16 channels), an LDS halo row
// is 64 + 16 pad bytes, a filter tile is 16 columns x 64 bytes = 1 KiB in MFMA operand order -- so the geometry code is common.
#pragma once
#include "common.h"

namespace unetconv {

constexpr int LDK = 20;  // LDS halo row length in dwords: 64 bytes of channels + 16 bytes pad (conflict-free ds_read_b128 lane groups)

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
    int fold;            // bf16 kernel: the reduction tail runs tap-folded (bf16_fold_tail below)
    int sliver;          // fp32 16x16x4 kernel: the last 1..4 output channels run on v_mfma_f32_4x4x1 (f32_sliver below)
    const float* wsl;    // its filter image [tap][chunk16][4][16]
    long long wp_stride; // floats between the packed filter images of consecutive batch images (0: one image for all)
    int cps;             // split-K: reduction chunks per split (blockIdx.y = split index); 0 = the whole reduction in one workgroup
    long long slab;      // split-K: elements between the partial-sum slabs of consecutive splits (y then points at slab 0)
    TapSet taps[4];
};

// ---- the fp32 packed filter image --------------------------------------------------------------------------------
// wp[tap][chunk16][outPad][16] (a reduction tail stored channel-transposed, see pack_weights_kernel), and for an output width that
// leaves 1..4 channels beyond a multiple of 16 (100 = 6 * 16 + 4) a SLIVER image behind it: wsl[tap][chunk16][4][16] holds the
// filters of those channels, position p of a 16-float row = reduction channel p of the chunk (tail chunk: the same transposition
// as the main image, channel 4 (p % 4) + p / 4).  conv_igemm16_kernel<.,2,2,2,2,4> multiplies them with v_mfma_f32_4x4x1_16B_f32
// (64 pixels x 4 channels per instruction at the FLOP rate of the 16x16x4 form) instead of paying a seventh 16-wide tile for 4 of
// its 16 columns.
__host__ __device__ inline bool f32_sliver(int out) { return out >= 16 && (out & 15) != 0 && (out & 15) <= 4; }
__host__ __device__ inline size_t f32_image_elems(int red, int out, int T) {
    const int nchunks = (red + 15) / 16, outPad = (out + 127) / 128 * 128;
    return (size_t)T * nchunks * outPad * 16 + (f32_sliver(out) ? (size_t)T * nchunks * 64 : 0);
}
// element i of the sliver image (i counted from its start); same (w, Cout, Cin, T, mode) convention as bf16_image_value
__device__ inline float f32_sliver_value(const float* __restrict__ w, int Cout, int Cin, int T, int mode, int nchunks, size_t i) {
    const int pos = (int)(i & 15), j = (int)((i >> 4) & 3);
    const int chunk = (int)((i >> 6) % nchunks), tap = (int)((i >> 6) / nchunks);
    const int red = mode == 0 ? Cin : Cout, out = mode == 0 ? Cout : Cin;
    const bool tail = (red & 15) != 0 && chunk == nchunks - 1;
    const int r = chunk * 16 + (tail ? (4 * (pos & 3) + (pos >> 2)) : pos);
    const int o = (out & ~15) + j;
    if (r >= red || o >= out) return 0.f;
    return mode == 0 ? w[((size_t)o * Cin + r) * T + tap] : w[((size_t)r * Cin + o) * T + tap];
}

// ---- the bf16 packed filter image ------------------------------------------------------------------------------
// wp[tap][chunk32][outPad][32], and for a 3x3 filter whose reduction leaves a tail of 1..8 channels (100 = 3 * 32 + 4) three more
// slabs fold[j][outPad][32]: k-slot (kq, c) of fold slab j = tail channel c of filter tap 4 j + kq.  One 16x16x32 MFMA then
// multiplies the tail channels of FOUR taps (each 8-channel lane group reads its own tap's pixel): the tail chunk costs 3 stages
// instead of 9 (30 instead of 36 for 100 channels).  Only single-tap-set launches use it (everything but the stride-2 input
// gradient, which reads the unfolded tail chunk that is still part of the image).
__host__ __device__ inline bool bf16_fold_tail(int red, int T) { return T == 9 && (red & 31) != 0 && (red & 31) <= 8; }
__host__ __device__ inline size_t bf16_image_elems(int red, int out_pad, int T) {
    return (size_t)(T * ((red + 31) / 32) + (bf16_fold_tail(red, T) ? 3 : 0)) * out_pad * 32;
}
// mode 2 = mode 0 with the image columns in pixel-shuffle order: column q = ij * (Cout / 4) + c holds filter 4 c + ij
__host__ __device__ inline int ps_filter_of(int q, int Cout) { const int nf = Cout >> 2; return 4 * (q % nf) + q / nf; }
// element i of the image of the fp32 master parameter w[Cout][Cin][T]; mode 0: out = cout, reduction = cin; mode 1: the reverse
__device__ inline float bf16_image_value(const float* __restrict__ w, int Cout, int Cin, int T, int mode, int nchunks, int outPad, size_t i) {
    const int rr = (int)(i & 31);
    size_t j = i >> 5;
    const int o = (int)(j % outPad);
    const int slab = (int)(j / outPad);
    const int red = mode == 1 ? Cout : Cin;
    int tap, r;
    if (slab < T * nchunks) {
        tap = slab / nchunks;
        r = (slab % nchunks) * 32 + rr;
    } else {
        tap = 4 * (slab - T * nchunks) + (rr >> 3);
        r = (nchunks - 1) * 32 + (rr & 7);
        if (tap >= T) return 0.f;
    }
    if (r >= red) return 0.f;
    if (mode == 2) return o < Cout ? w[((size_t)ps_filter_of(o, Cout) * Cin + r) * T + tap] : 0.f;
    if (mode == 0) return o < Cout ? w[((size_t)o * Cin + r) * T + tap] : 0.f;
    return o < Cin ? w[((size_t)r * Cin + o) * T + tap] : 0.f;
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
// The same with a lane offset of its own for the LAST tile: the kernels that share an odd last output tile between the two waves of a
// pixel row (see `nsplit` in conv_igemm16_kernel) address that tile out of the regular stride.
template <int TSTR, int N>
__device__ __forceinline__ void gld_bl(v4f (&d)[N], unsigned voff, unsigned voff_last, const char* p) {
    static_assert(N == 2 || N == 4, "operand-B tiles per wave");
    const u64 s0 = sgpr_ptr(p);
    if constexpr (N == 2) {
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %4\n\tglobal_load_dwordx4 %1, %3, %4 offset:%5"
                     : "=&v"(d[0]), "=&v"(d[1]) : "v"(voff), "v"(voff_last), "s"(s0), "n"(TSTR));
    } else {
        const u64 s1 = sgpr_ptr(p + 2 * TSTR);
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %6\n\tglobal_load_dwordx4 %1, %4, %6 offset:%8\n\t"
                     "global_load_dwordx4 %2, %4, %7\n\tglobal_load_dwordx4 %3, %5, %7 offset:%8"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(voff), "v"(voff_last), "s"(s0), "s"(s1), "n"(TSTR));
    }
}
// The halo items of one chunk: one base, one lane offset per item.  Executed in EVERY stage with `on` = all ones (fetch) or
// 0 (EXEC is cleared around the loads: nothing is fetched, the registers keep their values).  For the compiler the halo
// registers are thus one unbroken chain of tied asm operands -- no conditional definition, no phi, hence no register copy
// it could schedule between a load and its wait.
template <int N>
__device__ __forceinline__ void gld_halo(v4f (&h)[N], const unsigned (&vo)[N], const void* p, bool fetch) {
    static_assert(N == 4 || N == 6 || N == 10, "halo items per thread");
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
    } else if constexpr (N == 6) {
        asm volatile("s_and_saveexec_b64 %[sv], %[on]\n\ts_nop 4\n\t"
                     "global_load_dwordx4 %[h0], %[o0], %[sb]\n\tglobal_load_dwordx4 %[h1], %[o1], %[sb]\n\t"
                     "global_load_dwordx4 %[h2], %[o2], %[sb]\n\tglobal_load_dwordx4 %[h3], %[o3], %[sb]\n\t"
                     "global_load_dwordx4 %[h4], %[o4], %[sb]\n\tglobal_load_dwordx4 %[h5], %[o5], %[sb]\n\t"
                     "s_mov_b64 exec, %[sv]"
                     : [h0] "+v"(h[0]), [h1] "+v"(h[1]), [h2] "+v"(h[2]), [h3] "+v"(h[3]), [h4] "+v"(h[4]), [h5] "+v"(h[5]), [sv] "=&s"(sv)
                     : [o0] "v"(vo[0]), [o1] "v"(vo[1]), [o2] "v"(vo[2]), [o3] "v"(vo[3]), [o4] "v"(vo[4]), [o5] "v"(vo[5]), [sb] "s"(sb), [on] "s"(on)
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
// The four halo items of a chunk plus ONE item of the chunk's sliver filters (base of its own; lanes selected by `msl`, which is zero
// outside sliver launches).  Same contract as gld_halo: executed in every stage, EXEC-masked.
__device__ __forceinline__ void gld_halo4_sl(v4f (&h)[4], const unsigned (&vo)[4], const void* p, bool fetch, v4f& sl, unsigned vosl,
                                             const void* psl, u64 msl) {
    const u64 sb = sgpr_ptr(p), sbl = sgpr_ptr(psl);
    const u64 on = sgpr_ptr(reinterpret_cast<const void*>(fetch ? ~0ull : 0ull));
    const u64 ms = sgpr_ptr(reinterpret_cast<const void*>(msl));
    u64 sv;
    asm volatile("s_and_saveexec_b64 %[sv], %[on]\n\ts_nop 4\n\t"
                 "global_load_dwordx4 %[h0], %[o0], %[sb]\n\tglobal_load_dwordx4 %[h1], %[o1], %[sb]\n\t"
                 "global_load_dwordx4 %[h2], %[o2], %[sb]\n\tglobal_load_dwordx4 %[h3], %[o3], %[sb]\n\t"
                 "s_and_b64 exec, exec, %[ms]\n\ts_nop 0\n\t"
                 "global_load_dwordx4 %[sl], %[osl], %[sbl]\n\t"
                 "s_mov_b64 exec, %[sv]"
                 : [h0] "+v"(h[0]), [h1] "+v"(h[1]), [h2] "+v"(h[2]), [h3] "+v"(h[3]), [sl] "+v"(sl), [sv] "=&s"(sv)
                 : [o0] "v"(vo[0]), [o1] "v"(vo[1]), [o2] "v"(vo[2]), [o3] "v"(vo[3]), [osl] "v"(vosl), [sb] "s"(sb), [sbl] "s"(sbl),
                   [on] "s"(on), [ms] "s"(ms)
                 : "scc");
}
__device__ __forceinline__ void wait_loads_sl(v4f (&b)[4], v4f (&h)[4], v4f& sl) {
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(sl));
}
// s_waitcnt vmcnt(0) that also "defines" every register the outstanding loads write (operand-B tiles and halo items), so
// that no consumer is scheduled above it
template <int NB, int NH>
__device__ __forceinline__ void wait_loads(v4f (&b)[NB], v4f (&h)[NH]) {
    static_assert((NB == 2 || NB == 4) && (NH == 4 || NH == 6 || NH == 10), "register groups");
    // ONE statement (a tied operand's input copy, if the compiler ever made one, must not be able to slip in front of the
    // s_waitcnt of a sibling statement); 14 tied operands = 28 of the 30 asm operands allowed
#define UNET_H4 "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3])
#define UNET_H6 UNET_H4, "+v"(h[4]), "+v"(h[5])
#define UNET_H10 UNET_H4, "+v"(h[4]), "+v"(h[5]), "+v"(h[6]), "+v"(h[7]), "+v"(h[8]), "+v"(h[9])
    if constexpr (NB == 4 && NH == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), UNET_H6);
    else if constexpr (NB == 2 && NH == 6) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), UNET_H6);
    else if constexpr (NB == 2 && NH == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), UNET_H4);
    else if constexpr (NB == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), UNET_H10);
    else if constexpr (NH == 4) asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), UNET_H4);
    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), UNET_H10);
#undef UNET_H4
#undef UNET_H6
#undef UNET_H10
}

struct Plan {
    KArgs k;
    int tw, bm, bn, hit, nparity, mf, max_hpix;
    size_t lds_bytes;
    dim3 grid;
    // split-K (small grids with a long reduction): `splits` workgroups share one output tile, each reduces cps chunks into an fp32
    // slab [split][pixel][cp] of the caller's workspace; splitk_reduce_kernel adds the slabs in split order and applies the epilogue
    int splits, cp;
    size_t ws_floats;
    unet_tuning tune;        // the switches of this launch (a copy: the descriptor's or the defaults)
};


// one element of an activation tensor of either storage type (fp32 | bf16 bit pattern)
__device__ __forceinline__ float ld_act(const float* p) { return *p; }
__device__ __forceinline__ float ld_act(const unsigned short* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void st_act(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_act(unsigned short* p, float v) { *p = __builtin_bit_cast(unsigned short, (__bf16)v); }

// kc: reduction channels per chunk (16 fp32 / 32 bf16 = 64 bytes); vec: channels per 16-byte access (4 fp32 / 8 bf16): channel
// strides, offsets and the zero-padded channel count of a slice are multiples of vec; mf: MFMA shape of the fp32 kernels (16 | 32)
// big_tile: allow the 256-pixel workgroup tile (bf16 kernel: the math is 16x cheaper, so halving the filter-operand loads per MFMA pays)
// splitk: unet_tuning.conv_splitk of this plan (0: never split; callers that must not split pass 0)
// plan_batch: unet_tuning.plan_batch (0: the descriptor's N decides tile sizes / splits; n: as if the batch were n images)
static inline int make_plan(const unet_conv_desc* d, Plan* p, int kc, int vec, int mf, int big_tile, int splitk, int plan_batch = 0) {
    UNET_CHECK_ARG(d != nullptr, "conv: null desc");
    UNET_CHECK_ARG(d->x && d->wp && d->y, "conv: null tensor pointer");
    UNET_CHECK_ARG(d->ks == 1 || d->ks == 3, "conv: ks must be 1 or 3 (got %d)", d->ks);
    UNET_CHECK_ARG(d->stride == 1 || d->stride == 2, "conv: stride must be 1 or 2 (got %d)", d->stride);
    UNET_CHECK_ARG(!(d->ks == 1 && d->stride != 1), "conv: 1x1 stride 2 unsupported");
    UNET_CHECK_ARG(d->N > 0 && d->IH > 0 && d->IW > 0 && d->OH > 0 && d->OW > 0 && d->Cin > 0 && d->Cout > 0, "conv: bad dims");
    UNET_CHECK_ARG(unet::slice_ok_v(d->x_cs, d->x_co, d->Cin, vec), "conv: bad x slice cs=%d co=%d C=%d", d->x_cs, d->x_co, d->Cin);
    UNET_CHECK_ARG(unet::slice_ok_v(d->y_cs, d->y_co, d->Cout, d->y_f32 ? 4 : vec), "conv: bad y slice cs=%d co=%d C=%d", d->y_cs, d->y_co, d->Cout);
    UNET_CHECK_ARG(unet::aligned16(d->x) && unet::aligned16(d->wp), "conv: x/wp must be 16-byte aligned");
    if (d->res) UNET_CHECK_ARG(unet::slice_ok_v(d->res_cs, d->res_co, d->Cout, vec), "conv: bad res slice");
    if (d->flags & UNET_CONV_MASK) UNET_CHECK_ARG(d->mask && unet::slice_ok_v(d->mask_cs, d->mask_co, d->Cout, vec), "conv: bad mask slice");
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
    k.N = d->N; k.IH = d->IH; k.IW = d->IW; k.Cin = d->Cin; k.Cin4 = unet::roundup(d->Cin, vec);
    k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout;
    // optional produced-channel range (a wide layer can be issued as several launches with different channel-block widths)
    const int cols = d->cout_count > 0 ? d->cout_count : d->Cout;
    UNET_CHECK_ARG(d->cout_begin >= 0 && (d->cout_begin & 15) == 0 && d->cout_begin + cols <= d->Cout, "conv: bad cout range [%d,+%d) of %d",
                   d->cout_begin, cols, d->Cout);
    k.n_base = d->cout_begin; k.n_end = d->cout_begin + cols;
    UNET_CHECK_ARG(d->wp_img_stride >= 0 && (d->wp_img_stride & 3) == 0, "conv: bad wp_img_stride");
    k.wp_stride = d->wp_img_stride;
    k.flags = d->flags;
    k.nchunks = unet::cdiv(d->Cin, kc);
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
    p->mf = mf;
    // small problems (deep 16x16 / 32x32 stages): shrink the tile until the grid can fill 256 CUs x 2
    auto blocks = [&](int bm, int bn) {
        const int th_ = bm / p->tw;
        return (long long)(plan_batch > 0 ? plan_batch : d->N) * unet::cdiv(k.TSH, th_) * unet::cdiv(k.TSW, p->tw) * unet::cdiv(cols, bn) * p->nparity;
    };
    // Split-K first: a grid that cannot fill the chip with full-size tiles although the reduction is long (deep low-resolution stages,
    // small batches: BASELINE configs[0], predict at batch 1).  Instead of shrinking the tile -- fewer MACs per operand byte and still one
    // long serial reduction per workgroup -- `splits` workgroups per output tile each take a contiguous range of reduction chunks.
    // Partial sums meet in fixed order in the reduce kernel: deterministic, and the accumulation chain of an output element becomes
    // `splits` chains of K / splits products (the fp32 MFMA sums one k-ordered chain: its rounding error grows like sqrt(K)).
    p->splits = 1; p->cp = 0; p->ws_floats = 0;
    bool split = false;
    // fp32 on a small grid: narrower channel blocks (64, then 32) of the 256-pixel kernel put a workgroup on every CU.  Preferred to a split
    // reduction of the generic kernel (isolated launches, scripts/conv_f32_bias.py: 16 x 16 512 -> 512 0.207 -> 0.153 ms, 1024 -> 512 0.393 -> 0.299,
    // 32 x 32 128 -> 128 0.070 -> 0.045) and to the 64-pixel tile (32 x 32 256 -> 256 0.171 -> 0.147)
    int nb_first = 0;
    if (kc == 16 && big_tile && big_tile != 2 && d->ks == 3 && k.S == 1 && p->nparity == 1 && mf == 16 && d->colsum == nullptr && d->colsumsq == nullptr &&
        (p->tw == 32 || p->tw == 16) && p->bn == 128 && blocks(256, 128) < 256)
        nb_first = blocks(256, 64) >= 256 ? 64 : (blocks(256, 32) >= 256 ? 32 : 0);
    if (nb_first) p->bn = nb_first;
    if (!nb_first && splitk && mf == 16 && p->nparity == 1 && d->colsum == nullptr && d->colsumsq == nullptr && k.nchunks >= 8 && blocks(128, p->bn) < (splitk > 1 ? splitk : (kc == 32 ? 400 : 256))) {      // (bf16: measured +1.2 % on the step at 400; fp32 indifferent)
        // at least two chunks per split; when even the deepest split of full-size tiles leaves most CUs idle (a handful of pixel tiles:
        // 8 x 8 stages at batch 2), the tile shrinks as well
        const int smax = k.nchunks / 2 < 32 ? k.nchunks / 2 : 32;
        if (p->bn >= 64 && blocks(128, p->bn) * smax < 384) {
            p->bm = 64;
            if (p->bn == 128 && blocks(64, 128) * smax < 384) p->bn = 64;
        }
        const long long b = blocks(p->bm, p->bn);
        int sp = (int)((384 + b - 1) / b);
        if (sp > smax) sp = smax;
        if (sp >= 2) {
            split = true;
            k.cps = unet::cdiv(k.nchunks, sp);
            p->splits = unet::cdiv(k.nchunks, k.cps);
            p->cp = unet::roundup(cols, 4);
            k.slab = (long long)d->N * d->OH * d->OW * p->cp;
            p->ws_floats = (size_t)p->splits * k.slab;
        } else {
            p->bm = 128;
            p->bn = cols <= 32 ? 32 : (cols <= 64 ? 64 : 128);
        }
    }
    // bf16: the 256-pixel x 128-channel tile (conv_bf16_t256_kernel) for 3x3 / stride-1 launches from 64 blocks up -- on the deep 32 x 32
    // stages (a quarter of the chip's workgroup slots) it still beats the generic 128- / 64-pixel tiles by 1.2-1.7x, 512 -> 512: 109 -> 65 us,
    // with or without a split reduction on top (scripts/conv_mid_ab.py).  big_tile == 2: the order of round 3's first half (shrink first).
    // fp32 (kc == 16): the same kernel in its float form (a reduction tail runs transposed with its spare MFMA steps skipped; an output width of
    // 16 n + 1..4 takes a whole channel tile there instead of the 4-channel sliver of conv_igemm16_kernel); not for launches that emit column sums
    const bool f32_fit = kc != 16 || (mf == 16 && d->colsum == nullptr && d->colsumsq == nullptr);
    const bool big_ok = big_tile && f32_fit && d->ks == 3 && p->bm == 128 && (p->bn == 128 || big_tile != 2) && (p->tw == 32 || (p->tw == 16 && big_tile != 2)) && k.S == 1 && p->nparity == 1 &&
                        blocks(256, p->bn) >= (big_tile >= 3 ? 64 * (big_tile - 2) : (big_tile == 2 ? 512 : (kc == 16 ? 256 : 64))) &&        // (fp32 is MFMA-bound either way: it wants every CU busy)
                        (long long)d->IH * d->IW * d->x_cs * (kc == 16 ? 4 : 2) < (1ll << 31) - 65536 &&        // (bytes of ONE image at the storage width: the buffer descriptor's num_records, and the OOB offset 0x80000000 must stay outside it)
                        (long long)d->OH * d->OW * d->y_cs * 4 < (1ll << 31) - 65536;       // (its halo items and result stores go through buffer descriptors: one image within 2 GiB)
    if (!split && !(big_ok && big_tile != 2) && p->bn >= 64 && blocks(128, p->bn) < 400) {
        p->bm = 64;
        if (p->bn == 128 && blocks(64, 128) < 400) p->bn = 64;
    }
    if (big_ok && p->bm == 128) {
        p->bm = 256;          // 8 x 32 pixel patch per workgroup, each wave 128 pixels x 64 channels
        p->hit = 6;
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
    p->max_hpix = max_hpix;
    // the 16x16x4 kernel keeps no filter slab in LDS (operand B goes global -> VGPR)
    p->lds_bytes = (size_t)(32 + 2 * max_hpix * LDK + (p->mf == 16 ? 0 : 2 * p->bn * LDK)) * sizeof(float);
    // fp32 sliver (kc == 16 only): the 128 x 128 tile of the 16x16x4 kernel, single tap set, the launch that produces the last channels,
    // one filter image for all batch images, no column sums
    k.sliver = 0; k.wsl = nullptr;
    if (kc == 16 && p->mf == 16 && p->bm == 128 && p->bn == 128 && p->hit == 4 && p->nparity == 1 && f32_sliver(d->Cout) && k.n_end == d->Cout &&
        d->wp_img_stride == 0 && d->colsum == nullptr && d->colsumsq == nullptr && p->splits == 1) {
        k.sliver = 1;
        k.wsl = d->wp + (size_t)T * k.nchunks * k.coutPad * 16;
        p->lds_bytes += (size_t)2 * 9 * 64 * sizeof(float);      // two chunk buffers at the kernel's fixed stride of 9 taps (a 1x1 filter uses one tap of each)
    }
    // (the 16x16x4 kernel remaps block ids XCD-aware and needs a multiple of 8; the surplus workgroups exit at once)
    p->grid = dim3((unsigned)unet::roundup((int)((long long)k.mtiles * k.ntn), p->mf == 16 ? 8 : 1), (unsigned)p->splits, (unsigned)p->nparity);
    UNET_CHECK_ARG((long long)k.mtiles * k.ntn < (1ll << 31), "conv: grid too large");
    return UNET_OK;
}


// Split-K launches: the kernels write plain partial sums (no bias / residual / activation / mask) into the workspace slabs; called by
// both storage types after make_plan.  Returns false when the plan splits but the caller's workspace is missing or too small
// (the planner then has to be re-run without splitting).
static inline bool splitk_redirect(const unet_conv_desc* d, Plan* p) {
    if (p->splits <= 1) return true;
    if (d->splitk_ws == nullptr || d->splitk_ws_floats < p->ws_floats) return false;
    KArgs& k = p->k;
    k.bias = nullptr; k.res = nullptr; k.mask = nullptr; k.flags = 0;
    k.y = d->splitk_ws - k.n_base;          // the kernels address channel c of a pixel as y[pixel * y_cs + y_co + c]: slab column 0 = n_base
    k.y_cs = p->cp; k.y_co = 0;
    return true;
}
// epilogue of a split launch (elementwise.hip): y = act(sum_s slab[s] + bias + res) masked, in split order
int splitk_reduce(const unet_conv_desc* d, const Plan& p, hipStream_t st);

// bf16-storage kernels (conv_bf16.hip), reached through unet_conv2d / unet_conv2d_variant with desc.dtype == UNET_BF16
int conv2d_bf16(const unet_conv_desc* d, hipStream_t st);
int conv2d_bf16_variant(const unet_conv_desc* d);
int plan_bf16_public(const unet_conv_desc* d, Plan* p);
// 1x1 convolutions with a reduction of at most 8 channels (conv_igemm.hip: conv1x1_smallk_kernel), both storage types
int conv2d_t256_f32(const Plan& p, hipStream_t st);       // conv_bf16.hip: conv_bf16_t256_kernel<.., float>
bool conv_gemm1x1_applies(const unet_conv_desc* d);      // conv1x1.hip: 1x1 / stride-1 convs of whole reduction chunks on the flat-pixel GEMM kernel
int conv_gemm1x1(const unet_conv_desc* d, hipStream_t st);
int conv_gemm1x1_ps_check(const unet_conv_desc* d);      // validation of a unet_conv_desc.pixel_shuffle descriptor (UNET_OK: conv_gemm1x1 takes it)
bool conv_smallk_applies(const unet_conv_desc* d);
int conv_smallk_bf16(const unet_conv_desc* d, hipStream_t st);
// 3x3 forward convolutions of at most 8 input channels (the stem's first conv; conv_igemm.hip: conv3x3_smallcin_kernel), both storage types
// 1x1 forward convolutions with at most 16 produced channels (the segmentation head; conv1x1.hip: conv1x1_head_kernel), both storage types
bool conv_head1x1_applies(const unet_conv_desc* d);
int conv_head1x1(const unet_conv_desc* d, hipStream_t st);
bool conv_smallcin_applies(const unet_conv_desc* d);
int conv_smallcin_bf16(const unet_conv_desc* d, hipStream_t st);

}  // namespace unetconv
