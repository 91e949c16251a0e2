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

#include "conv_common.h"

namespace {

using namespace unetconv;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 16;   // reduction channels per chunk


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

// ------------------------------------------------------------------------------------------------
// Same workgroup geometry on v_mfma_f32_16x16x4_f32 (A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C: col=l&15,
// row=4*(l>>4)+reg).  Each wave owns (2*MT) x (2*NT) tiles of 16x16; the LDS traffic per MFMA cycle is the
// same as the 32x32x2 form (one ds_read_b128 per operand tile per 16-channel chunk = 4 k-steps), but
//   * output-channel tiles that lie entirely beyond Cout are SKIPPED per wave (uniform branch): produced
//     channel counts are padded to 16 instead of 32/64/128 (Cout = 96 exact, 100 -> 112, 192 exact);
//   * a reduction tail of r < 16 channels costs ceil(r/4) MFMA steps (ds_read_b32 operands) instead of 4.
template <int TW, int MT, int NT, int WM, int WN, int HIT, bool SLV = false>
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
    // split-K: this workgroup reduces chunks [kc0, kc1) only and writes its partial sums to slab blockIdx.y
    const int kc0 = a.cps ? (int)blockIdx.y * a.cps : 0;
    const int kc1 = a.cps ? (kc0 + a.cps < a.nchunks ? kc0 + a.cps : a.nchunks) : a.nchunks;

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
    // Sliver (conv_common.h, f32_sliver): in the 128 x 128 tile the block that ends the output range computes its last 1..4
    // channels with v_mfma_f32_4x4x1_16B_f32 instead of a whole 16-wide tile: lane l = 4 blk + i supplies A = x[pixel l][k],
    // lane 4 blk + j supplies B = w[k][channel j], and lane 4 blk + j receives D[r] = pixel 4 blk + r, channel j (16 blocks of 4 x 4).
    // The wave with fewer full tiles (wn = 1) multiplies it: 16 instructions of 8 cycles per stage next to 48 of 32.
    // (SLV: an instantiation of its own, so that launches without a sliver run exactly the code they ran before)
    constexpr bool SL = SLV;
    static_assert(!SLV || (MT == 2 && NT == 2 && WM == 2 && WN == 2 && HIT == 4), "sliver: the 128 x 128 tile only");
    const bool sliver = SL && a.sliver != 0 && nt == a.ntn - 1;
    const bool slw = sliver && wn == 1;
    v4f slreg = {0.f, 0.f, 0.f, 0.f};                        // this thread's item of the chunk's sliver filters
    const int sl_items = ts.n * 16;                          // float4 items per chunk: [tap][4 channels][16 k]
    const u64 sl_mask = sliver ? __ballot(tid < sl_items) : 0ull;
    // byte offset of this thread's item inside the sliver image at chunk 0: tap tid >> 4 (its filter tap), float4 (tid & 15) of the [4][16] block
    const unsigned vosl0 = (unsigned)((TAP_WIDX((tid >> 4) < ts.n ? (tid >> 4) : 0) * a.nchunks * 64 + (tid & 15) * 4) * 4);
    float* sl0 = lds0 + 2 * HPIX * LDK;
    auto sl_buf = [&](int b) -> float* { return sl0 + b * (9 * 64); };

    // halo items: always HIT loads; an item outside the image / beyond the channels loads offset 0 and is zeroed at the store
#define HALO_OK(it_, c0_) (goff[it_] >= 0 && ((c0_) + 4 * ((tid + (it_) * NTH) & 3)) < a.Cin4)
#define LOAD_HALO(chunk_, on_) do { const int c0_ = (chunk_) * KC; unsigned vo_[HIT]; \
        if (on_) { _Pragma("unroll") for (int it = 0; it < HIT; ++it) vo_[it] = HALO_OK(it, c0_) ? (unsigned)(goff[it] + c0_) * 4u : 0u; } \
        else { _Pragma("unroll") for (int it = 0; it < HIT; ++it) vo_[it] = 0u; } \
        if constexpr (SL) { \
            gld_halo4_sl(hreg, vo_, xb, (on_), slreg, vosl0 + (unsigned)(chunk_) * 256u, a.wsl != nullptr ? (const void*)a.wsl : (const void*)a.wp, sl_mask); \
        } else gld_halo<HIT>(hreg, vo_, xb, (on_)); } while (0)
#define WAIT_LOADS(b_) do { if constexpr (SL) wait_loads_sl(b_, hreg, slreg); else wait_loads(b_, hreg); } while (0)
#define STORE_HALO(dst_, chunk_) do { const int c0_ = (chunk_) * KC; const bool tail_ = has_tail && (chunk_) == a.nchunks - 1; \
        _Pragma("unroll") for (int it = 0; it < HIT; ++it) { \
            const int e_ = tid + it * NTH; \
            if (e_ < HPIX * 4) { \
                const v4f v_ = HALO_OK(it, c0_) ? hreg[it] : (v4f){0.f, 0.f, 0.f, 0.f}; \
                if (tail_) { float* row_ = (dst_) + (e_ >> 2) * LDK + (e_ & 3); row_[0] = v_.x; row_[4] = v_.y; row_[8] = v_.z; row_[12] = v_.w; } \
                else *reinterpret_cast<v4f*>((dst_) + (e_ >> 2) * LDK + (e_ & 3) * 4) = v_; \
            } } \
        if (SL && sliver && tid < sl_items) *reinterpret_cast<v4f*>(sl_buf((chunk_) & 1) + tid * 4) = slreg; } while (0)

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
    const int ntile_blk = sliver ? (a.n_end - n0) / 16 : (a.n_end - n0 + 15) / 16;        // 16-wide tiles of this block (sliver: full ones only)
    int nvalid = (ntile_blk - wn + WN - 1) / WN;
    nvalid = (ntile_blk <= wn) ? 0 : (nvalid > N16 ? N16 : nvalid);
    // sliver operands: this lane's pixel (wave pixel index = lane) inside the halo tile, its channel's filter row, two accumulators
    // (even / odd k: two independent chains of 8)
    int sl_abase = 0;
    {
        const int pix = wm * (M16 * 16) + lane;
        sl_abase = (((pix / TW) * S) * HW + (pix % TW) * S) * LDK;
    }
    f32x4 sacc0 = {0.f, 0.f, 0.f, 0.f}, sacc1 = {0.f, 0.f, 0.f, 0.f};
#define SLIVER_STAGE() do { if (SL && slw) { \
        const float* sa_ = hb + TAP_OFF(t) + sl_abase; \
        const float* sb_ = sl_buf(chunk & 1) + t * 64 + (lane & 3) * 16; \
        _Pragma("unroll") for (int q_ = 0; q_ < 4; ++q_) { \
            const float4 xa_ = *reinterpret_cast<const float4*>(sa_ + 4 * q_); \
            const float4 wb_ = *reinterpret_cast<const float4*>(sb_ + 4 * q_); \
            sacc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa_.x, wb_.x, sacc0, 0, 0, 0); \
            sacc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa_.y, wb_.y, sacc1, 0, 0, 0); \
            sacc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa_.z, wb_.z, sacc0, 0, 0, 0); \
            sacc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(xa_.w, wb_.w, sacc1, 0, 0, 0); \
        } } } while (0)
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
    LOAD_HALO(kc0, true);
    LOAD_B(b0, ts.widx[0], kc0);
    WAIT_LOADS(b0);
    STORE_HALO(halo_buf(kc0 & 1), kc0);
    __syncthreads();

    // Stage = (chunk, tap).  The loop body is written twice (operand B ping-pongs between b0 and b1 without register copies).
    // Loads in flight: the next stage's B tiles and, during the first tap of a chunk, the next chunk's halo items; both are
    // issued at the top of a stage and waited for at its end (behind the stage's 64 MFMAs).
    int t = 0, chunk = kc0;
    int ksteps = (a.Cin - kc0 * KC) >= KC ? 4 : ((a.Cin - kc0 * KC + 3) >> 2);   // tail chunk: channel-transposed, step kk = channels 4kk..4kk+3
    const float* hb = halo_buf(kc0 & 1);
    // Every stage ends in ONE wait asm that (re)defines both the B tiles and the halo registers, on every path, so the
    // compiler has no merge point of its own between a load and its wait where it could copy a register that is still in
    // flight; tests/test_isa_cpu.py checks the generated ISA for exactly that.
#define STAGE_BODY(bu_, bl_, FULL_) do { \
        LOAD_HALO(chunk + 1, t == 0 && chunk + 1 < kc1);   /* EXEC-masked off in the other stages */ \
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
        SLIVER_STAGE(); \
        WAIT_LOADS(bl_); \
    } while (0)
#define STAGE(bu_, bl_, FULL_) do { \
        int tn_ = t + 1, cn_ = chunk; \
        if (tn_ == ntaps) { tn_ = 0; cn_ = chunk + 1; } \
        const bool has_next_ = cn_ < kc1; \
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
    const int total = (kc1 - kc0) * ntaps;
    const int total_full = (((has_tail && kc1 == a.nchunks) ? kc1 - kc0 - 1 : kc1 - kc0) * ntaps) & ~1;
    for (int s = 0; s < total_full; s += 2) {
        STAGE(b0, b1, true);
        STAGE(b1, b0, true);
    }
    for (int s = total_full; s < total; s += 2) {
        STAGE(b0, b1, false);
        if (s + 1 < total) STAGE(b1, b0, false);
    }
#undef STAGE_BODY
#undef SLIVER_STAGE
#undef WAIT_LOADS
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
    float* yb = a.y + (a.cps ? (size_t)blockIdx.y * a.slab : (size_t)0) + img_pix * a.y_cs + a.y_co;
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
    // the bias of this lane's channel in every channel tile, fetched once and first: a load between the result stores of two tiles makes
    // the compiler wait for the older stores as well (one in-order counter).  (Measured against the per-tile load on one box: equal within
    // 0.3 % here -- three workgroups per CU cover the round trips; it mattered in the bf16 kernel.)
    float bvn[N16];
#pragma unroll
    for (int n = 0; n < N16; ++n) {
        const int cout = n0 + (n * WN + wn) * 16 + l15;
        bvn[n] = (a.bias != nullptr && n < nvalid && cout < a.n_end) ? a.bias[cout] : 0.f;
    }
#pragma unroll
    for (int n = 0; n < N16; ++n) {
        if (n >= nvalid) continue;
        const int cout = n0 + (n * WN + wn) * 16 + l15;
        const bool cvalid = cout < a.n_end;
        const int cc = cvalid ? cout : 0;
        const float bvv = bvn[n];
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
    if (SL && slw) {
        // sliver results: lane l = 4 blk + j holds pixels 4 blk + r (r = 0..3) of this wave's 64, channel j beyond the full tiles
        const int cout = n0 + ntile_blk * 16 + (lane & 3);
        const bool cvalid = cout < a.n_end;
        const int cc = cvalid ? cout : 0;
        const float bvv = (a.bias != nullptr && cvalid) ? a.bias[cout] : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int pix = wm * (M16 * 16) + (lane & ~3) + r;
            const int ty = pix / TW, tx = pix % TW;
            const int oyt = oy0 + ty, oxt = ox0 + tx;
            const int oy = oyt * OS + ts.py, ox = oxt * OS + ts.px;
            const bool pv = oyt < a.TSH && oxt < a.TSW && oy < a.OH && ox < a.OW;
            const int pi = pv ? (oy * a.OW + ox) : 0;
            float v = sacc0[r] + sacc1[r] + bvv;
            if (resb != nullptr) v += resb[pi * a.res_cs + cc];
            if (relu) v = fmaxf(v, 0.f);
            if (maskb != nullptr) v = maskb[pi * a.mask_cs + cc] > 0.f ? v : 0.f;
            if (cvalid && pv) yb[pi * a.y_cs + cout] = v;
        }
    }
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
        const int red = mode == 1 ? Cout : Cin;
        const bool tail = (red & 15) != 0 && chunk == nchunks - 1;
        const int r = chunk * 16 + (tail ? (4 * (rr & 3) + (rr >> 2)) : rr);
        float v = 0.f;
        if (mode == 2) {
            if (o < Cout && r < Cin) v = w[((size_t)unetconv::ps_filter_of(o, Cout) * Cin + r) * T + tap];
        } else if (mode == 0) {
            if (o < Cout && r < Cin) v = w[((size_t)o * Cin + r) * T + tap];
        } else {
            if (o < Cin && r < Cout) v = w[((size_t)r * Cin + o) * T + tap];
        }
        wp[i] = v;
    }
    // the sliver image behind it (conv_common.h, f32_sliver)
    const int out_ = mode == 1 ? Cin : Cout;
    if (unetconv::f32_sliver(out_)) {
        const size_t nsl = (size_t)T * nchunks * 64;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nsl; i += (size_t)gridDim.x * blockDim.x)
            wp[total + i] = unetconv::f32_sliver_value(w, Cout, Cin, T, mode, nchunks, i);
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
    // the sliver image behind it, as pack_weights_kernel writes it (conv_common.h, f32_sliver): a launch with wp_img_stride == 0 at such
    // a width reads it whatever packer made the image
    if (unetconv::f32_sliver(O)) {
        const size_t nsl = (size_t)nchunks * 64;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nsl; i += (size_t)gridDim.x * blockDim.x) {
            const int pos = (int)(i & 15), j = (int)((i >> 4) & 3), chunk = (int)(i >> 6);
            const bool tail = (R & 15) != 0 && chunk == nchunks - 1;
            const int r = chunk * 16 + (tail ? (4 * (pos & 3) + (pos >> 2)) : pos);
            const int o = (O & ~15) + j;
            wp[total + i] = (o < O && r < R) ? w[(long long)o * so + (long long)r * sr] : 0.f;
        }
    }
}

// ------------------------------------------------------------------ host side

}  // namespace

// The defaults of unet_tuning: constants, with environment overrides read ONCE when the library is loaded (A/B runs of whole programs).
// Nothing writes them afterwards: the library has no mutable process state.
namespace unetconv {
static int env_int(const char* name, int dflt) {
    const char* e = getenv(name);
    return (e != nullptr && e[0] != 0) ? atoi(e) : dflt;
}
const unet_tuning& tuning_defaults() {
    static const unet_tuning t = [] {
        unet_tuning v;
        v.conv_splitk = env_int("UNET_CONV_SPLITK", 1);
        v.mfma_shape = 16;
        v.f32_big_tile = 1;
        v.bf16_big_tile = 1;
        v.t256_tiles_per_wg = 0;
        v.t256_sliver = env_int("UNET_T256_SLIVER", 1);
        v.conv1x1_gemm = env_int("UNET_CONV1X1_GEMM", 0);
        v.wgrad_mfma_shape = 32;
        v.wgrad_bf16_k4 = 1;
        v.wgrad_1x1 = 1;
        v.wgrad_narrow = 1;
        v.plan_batch = 0;
        v.wgrad_wgs = env_int("UNET_WGRAD_WGS", 0);
        v.conv_smallcin = env_int("UNET_CONV_SMALLCIN", 1);
        v.conv_head1x1 = env_int("UNET_CONV_HEAD1X1", 1);
        return v;
    }();
    return t;
}
}  // namespace unetconv

extern "C" void unet_tuning_default(unet_tuning* t) {
    if (t != nullptr) *t = unetconv::tuning_defaults();
}

namespace {

// splitk < 0: the tuning's own value; 0: a plan that must not split (column-sum launches, a caller without a workspace)
static int make_plan(const unet_conv_desc* d, Plan* p, int splitk = -1) {
    UNET_CHECK_ARG(d != nullptr, "conv: null desc");
    const unet_tuning t = unetconv::tuning_of(d->tuning);
    UNET_CHECK_ARG(t.mfma_shape == 16 || t.mfma_shape == 32, "conv: unet_tuning.mfma_shape must be 16 or 32 (start from unet_tuning_default())");
    const int rc = unetconv::make_plan(d, p, KC, 4, t.mfma_shape, t.f32_big_tile ? 1 : 0, splitk < 0 ? t.conv_splitk : splitk, t.plan_batch);
    p->tune = t;
    return rc;
}


template <int TW, int MT, int NT, int WM, int WN, int HIT>
int launch_cfg(const Plan& p, hipStream_t st) {
    if constexpr (MT == 2 && NT == 2 && WM == 2 && WN == 2 && HIT == 4) {
        if (p.mf == 16 && p.k.sliver) {
            auto kern = conv_igemm16_kernel<TW, MT, NT, WM, WN, HIT, true>;
            static unsigned long long configured = 0;  // per instantiation, one bit per device
            if (unet::first_use_on_device(&configured))
                UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            hipLaunchKernelGGL(kern, p.grid, dim3(WM * WN * 64), p.lds_bytes, st, p.k);
            UNET_CHECK_LAUNCH();
            return UNET_OK;
        }
    }
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
    // the plan of the launch that WILL carry the column-sum pointers (they may still be null in this query): never split
    int rc = make_plan(d, &p, 0);
    if (rc != UNET_OK) return rc;
    const int wm = (p.bn == 32 && p.bm == 128) ? 4 : 2;
    return p.nparity * p.k.mtiles * wm;
}

static int make_plan_ws(const unet_conv_desc* d, Plan* p);

extern "C" int unet_conv2d_variant(const unet_conv_desc* d) {
    if (d != nullptr && d->pixel_shuffle) { const int rc = unetconv::conv_gemm1x1_ps_check(d); return rc == UNET_OK ? 8 : rc; }
    if (d != nullptr && d->dtype == UNET_BF16) return unetconv::conv2d_bf16_variant(d);
    Plan p;
    int rc = make_plan_ws(d, &p);
    if (rc != UNET_OK) return rc;
    if (unetconv::conv_smallk_applies(d)) return 9;          // conv1x1_smallk_kernel
    if (unetconv::conv_smallcin_applies(d)) return 10;       // conv3x3_smallcin_kernel
    if (unetconv::conv_head1x1_applies(d)) return 11;        // conv1x1_head_kernel
    if (unetconv::conv_gemm1x1_applies(d)) return 8;         // conv1x1_gemm_kernel
    const bool large = p.bm == 256 && p.bn == 128 && p.tw == 32 && (long long)p.k.mtiles * p.k.ntn >= 512;
    return p.tw * 10000 + p.bn * 10 + (p.hit == 10 ? 1 : 0) + (p.bm == 64 ? 5 : 0) + (p.bm == 256 ? (large ? 7 : 6) : 0) + (p.splits > 1 ? 1000000 * p.splits : 0);
}

// ------------------------------------------------------------------ split-K epilogue
// y[p][c] = mask(act(sum_s slab[s][p][c] + bias[c] + res[p][c])): the slabs are added in split order (deterministic), four channels per
// thread (16-byte slab reads; 16-byte fp32 / 8-byte bf16 stores).  HBM-bound: splits x the output once in, the output once out.
namespace {
template <typename T>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int splits, long long slab, long long P, int cols,
                                                            int cp, const float* __restrict__ bias, const T* __restrict__ res, int res_cs,
                                                            int res_co, const T* __restrict__ mask, int mask_cs, int mask_co,
                                                            T* __restrict__ y, int y_cs, int y_co, int relu) {
    const int q4 = cp >> 2;
    const long long total = P * q4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long p = i / q4;
        const int c = 4 * (int)(i - p * q4);
        float4 v = *reinterpret_cast<const float4*>(ws + p * cp + c);
        for (int s = 1; s < splits; ++s) {
            const float4 u = *reinterpret_cast<const float4*>(ws + (size_t)s * slab + p * cp + c);
            v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
        }
        float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool live = c + r < cols;
            float t = live ? o[r] : 0.f;
            if (live && bias != nullptr) t += bias[c + r];
            if (live && res != nullptr) t += unetconv::ld_act(res + (size_t)p * res_cs + res_co + c + r);
            if (relu) t = fmaxf(t, 0.f);
            if (live && mask != nullptr) t = unetconv::ld_act(mask + (size_t)p * mask_cs + mask_co + c + r) > 0.f ? t : 0.f;
            o[r] = t;
        }
        T* yo = y + (size_t)p * y_cs + y_co + c;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (c + r < cols) unetconv::st_act(yo + r, o[r]);
    }
}
}  // namespace

int unetconv::splitk_reduce(const unet_conv_desc* d, const Plan& p, hipStream_t st) {
    const long long P = (long long)d->N * d->OH * d->OW;
    const int cols = p.k.n_end - p.k.n_base, nb = p.k.n_base;
    const int relu = (d->flags & UNET_CONV_RELU) ? 1 : 0;
    const float* bias = d->bias ? d->bias + nb : nullptr;
    const void* mask = (d->flags & UNET_CONV_MASK) ? (const void*)d->mask : nullptr;
    const int grid = unet::ew_grid(P * (p.cp >> 2), 256);
    if (d->dtype == UNET_BF16 && !d->y_f32) {
        typedef unsigned short T;
        hipLaunchKernelGGL((splitk_reduce_kernel<T>), dim3(grid), dim3(256), 0, st, d->splitk_ws, p.splits, p.k.slab, P, cols, p.cp, bias,
                           (const T*)d->res, d->res_cs, d->res_co + nb, (const T*)mask, d->mask_cs, d->mask_co + nb, (T*)d->y, d->y_cs,
                           d->y_co + nb, relu);
    } else {
        // (bf16 storage with fp32 logits: residual / mask operands of such a launch would be bf16 -- the head has neither)
        UNET_CHECK_ARG(d->dtype == UNET_F32 || (d->res == nullptr && mask == nullptr), "split-K: fp32 output with bf16 residual / mask");
        hipLaunchKernelGGL((splitk_reduce_kernel<float>), dim3(grid), dim3(256), 0, st, d->splitk_ws, p.splits, p.k.slab, P, cols, p.cp, bias,
                           (const float*)d->res, d->res_cs, d->res_co + nb, (const float*)mask, d->mask_cs, d->mask_co + nb, (float*)d->y,
                           d->y_cs, d->y_co + nb, relu);
    }
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

// ------------------------------------------------------------------------------------------------
// 1x1 convolution with a reduction of at most 8 channels and a wide output: the input gradient of the segmentation head (5 -> 100
// at 16 x 512 x 512).  On the MFMA kernels this launch is one 16- / 32-channel chunk of zeros around 5 real channels in front of an
// epilogue that writes 0.87 GB in 8-byte pieces (0.75 ms bf16, 1.1 ms fp32: 1.2 TB/s); it is an HBM-bound elementwise product.  Here a
// thread owns one pixel x one 16-byte group of output channels: the pixel's (up to 8) inputs in registers, the filter in LDS, FMAs in
// ascending channel order, residual / mask / result as 16-byte vectors, consecutive threads on consecutive groups of the same pixel.
namespace {
template <typename T> struct SmallK;
template <> struct SmallK<float> {
    static constexpr int VEC = 4;
    __device__ static float w(const float* wp, int o, int r) { return wp[o * 16 + (r >> 2) + 4 * (r & 3)]; }       // one (tail) chunk, channel-transposed
    __device__ static void ld(const float* p, float (&v)[VEC]) { const float4 f = *reinterpret_cast<const float4*>(p); v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w; }
    __device__ static void st(float* p, const float (&v)[VEC]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct SmallK<unsigned short> {
    static constexpr int VEC = 8;
    __device__ static float w(const unsigned short* wp, int o, int r) { return __uint_as_float((unsigned)wp[o * 32 + r] << 16); }
    __device__ static void ld(const unsigned short* p, float (&v)[VEC]) {
        const uint4 u = *reinterpret_cast<const uint4*>(p);
        const unsigned q[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(q[i] << 16); v[2 * i + 1] = __uint_as_float(q[i] & 0xffff0000u); }
    }
    __device__ static void st(unsigned short* p, const float (&v)[VEC]) {
        typedef __bf16 bf16x8_ __attribute__((ext_vector_type(8)));
        const bf16x8_ h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3], (__bf16)v[4], (__bf16)v[5], (__bf16)v[6], (__bf16)v[7]};
        *reinterpret_cast<uint4*>(p) = __builtin_bit_cast(uint4, h);
    }
};

template <typename T>
__global__ __launch_bounds__(256) void conv1x1_smallk_kernel(const T* __restrict__ x, int x_cs, int x_co, const T* __restrict__ wp, int K,
                                                             const float* __restrict__ bias, const T* __restrict__ res, int res_cs, int res_co,
                                                             const T* __restrict__ mask, int mask_cs, int mask_co, T* __restrict__ y, int y_cs,
                                                             int y_co, long long P, int cols, int groups, int relu) {
    constexpr int VEC = SmallK<T>::VEC, XV = 8 / VEC;
    __shared__ float wl[8 * 512];                       // [k][column], columns padded to the vector width
    const int colsv = groups * VEC;
    for (int i = threadIdx.x; i < 8 * colsv; i += 256) {
        const int k = i / colsv, c = i - k * colsv;
        wl[i] = (k < K && c < cols) ? SmallK<T>::w(wp, c, k) : 0.f;
    }
    __syncthreads();
    const int ppb = 256 / groups;                       // pixels per workgroup pass
    const int pl = (int)threadIdx.x / groups, gi = (int)threadIdx.x - pl * groups;
    if (pl >= ppb) return;
    const int c = gi * VEC;
    float bv[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) bv[j] = (bias != nullptr && c + j < cols) ? bias[c + j] : 0.f;
    for (long long p = (long long)blockIdx.x * ppb + pl; p < P; p += (long long)gridDim.x * ppb) {
        float xv[8];
#pragma unroll
        for (int h = 0; h < XV; ++h) {
            float t[VEC];
            if (h * VEC < K) SmallK<T>::ld(x + (size_t)p * x_cs + x_co + h * VEC, t);
            else {
#pragma unroll
                for (int j = 0; j < VEC; ++j) t[j] = 0.f;
            }
#pragma unroll
            for (int j = 0; j < VEC; ++j) xv[h * VEC + j] = t[j];
        }
        float o[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (k < K) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) o[j] = fmaf(xv[k], wl[k * colsv + c + j], o[j]);
            }
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] += bv[j];
        if (res != nullptr) {
            float t[VEC];
            SmallK<T>::ld(res + (size_t)p * res_cs + res_co + c, t);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] += t[j];
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = fmaxf(o[j], 0.f);
        }
        if (mask != nullptr) {
            float t[VEC];
            SmallK<T>::ld(mask + (size_t)p * mask_cs + mask_co + c, t);
#pragma unroll
            for (int j = 0; j < VEC; ++j) o[j] = t[j] > 0.f ? o[j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) o[j] = c + j < cols ? o[j] : 0.f;
        T* yo = y + (size_t)p * y_cs + y_co + c;
        if (VEC == 8 || c + VEC <= cols) SmallK<T>::st(yo, o);      // (bf16 slices own their 8-channel padding: zeros; fp32 padding lanes are left alone)
        else {
#pragma unroll
            for (int j = 0; j < VEC; ++j)
                if (c + j < cols) unetconv::st_act(yo + j, o[j]);
        }
    }
}

// the launches this kernel takes (both storage types; everything else about the descriptor was validated by the planner)
bool smallk_applies(const unet_conv_desc* d) {
    const int vec = d->dtype == UNET_BF16 ? 8 : 4;
    return d->ks == 1 && d->stride == 1 && d->Cin <= 8 && d->Cout >= 16 && unet::roundup(d->Cout, vec) <= 512 && d->colsum == nullptr &&
           d->colsumsq == nullptr && d->cout_begin == 0 && (d->cout_count == 0 || d->cout_count == d->Cout) && d->wp_img_stride == 0 &&
           !(d->dtype == UNET_BF16 && d->y_f32) && unet::roundup(d->Cout, vec) <= d->y_cs - d->y_co &&
           (d->res == nullptr || unet::roundup(d->Cout, vec) <= d->res_cs - d->res_co) &&
           (!(d->flags & UNET_CONV_MASK) || unet::roundup(d->Cout, vec) <= d->mask_cs - d->mask_co);
}

template <typename T>
int launch_smallk(const unet_conv_desc* d, hipStream_t st) {
    const long long P = (long long)d->N * d->OH * d->OW;
    const int groups = unet::cdiv(d->Cout, SmallK<T>::VEC), ppb = 256 / groups;
    long long blocks = (P + ppb - 1) / ppb;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((conv1x1_smallk_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, (const T*)d->x, d->x_cs, d->x_co, (const T*)d->wp, d->Cin,
                       d->bias, (const T*)d->res, d->res_cs, d->res_co, (d->flags & UNET_CONV_MASK) ? (const T*)d->mask : (const T*)nullptr, d->mask_cs,
                       d->mask_co, (T*)d->y, d->y_cs, d->y_co, P, d->Cout, groups, (d->flags & UNET_CONV_RELU) ? 1 : 0);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

// ---- 3x3 convolution with at most 8 INPUT channels: the stem's first conv (train.py:131-135: Conv2d(n_in, 32, 3, stride 2) on the 3 / 4 / 8-band
// tile) -- SURVEY 8a row A1: "conv0 is HBM-bound (C_in 4)".  On the MFMA kernels its 36-wide reduction pads to nine taps x a 16- / 32-channel
// chunk (4 x / 8 x the products) in front of small 32-pixel tiles: 0.51 ms fp32 / 0.28 ms bf16 at 16 x 512^2 for 0.4 / 0.13 GB of traffic
// (profiles/r05_z: 0.5-0.8 TB/s).  Here it is what it is, an HBM-bound stencil: a thread owns one output pixel x 8 output channels, reads the
// nine input pixels as 16-byte vectors (zero outside the image), takes the filter from an fp32 LDS copy [tap][cin][cout] of the SAME packed
// image the MFMA kernels read (BatchNorm fold included) and accumulates in ascending (tap, channel) order; consecutive threads own consecutive
// channel groups of one pixel (one coalesced row per pixel).
template <typename T> struct SmallCin;
template <> struct SmallCin<float> {
    static constexpr int VEC = 4;
    // fp32 image, one (tail) chunk, channel-transposed: [tap][outPad][16], position (r >> 2) + 4 (r & 3) holds reduction channel r
    __device__ static float w(const float* wp, int outPad, int tap, int o, int r) { return wp[((size_t)tap * outPad + o) * 16 + (r >> 2) + 4 * (r & 3)]; }
};
template <> struct SmallCin<unsigned short> {
    static constexpr int VEC = 8;
    // bf16 image, one chunk: [tap][outPad][32] (the folded-tail slabs behind it are not read)
    __device__ static float w(const unsigned short* wp, int outPad, int tap, int o, int r) { return __uint_as_float((unsigned)wp[((size_t)tap * outPad + o) * 32 + r] << 16); }
};

typedef float f32x2 __attribute__((ext_vector_type(2)));

template <typename T, int PX, int NX>
__global__ __launch_bounds__(256) void conv3x3_smallcin_kernel(const T* __restrict__ x, int x_cs, int x_co, const T* __restrict__ wp, int Cin, int outPad,
                                                               const float* __restrict__ bias, T* __restrict__ y, int y_cs, int y_co, int IH, int IW,
                                                               int OH, int OW, int stride, long long P, int Cout, int groups, int relu) {
    // PX pixels per thread, 64 / groups apart... (ppb apart): every filter value read from LDS feeds PX FMAs -- at one pixel per thread the
    // two 16-byte LDS reads per 8 FMAs bound the kernel (LDS 128 B / clk / CU), at four the VALU does.
    constexpr int VEC = SmallK<T>::VEC;                        // NX 16-byte input vectors per pixel (fp32 with 5..8 input channels: two)
    __shared__ float wl[9 * 8 * 64];                           // [tap][cin][cout padded to groups * 8]
    const int cp = groups * 8;
    for (int i = threadIdx.x; i < 9 * 8 * cp; i += 256) {
        const int o = i % cp, r = (i / cp) & 7, t = i / (8 * cp);
        wl[i] = (r < Cin && o < Cout) ? SmallCin<T>::w(wp, outPad, t, o, r) : 0.f;
    }
    __syncthreads();
    const int ppb = 256 / groups;
    const int pl = (int)threadIdx.x / groups, gi = (int)threadIdx.x - pl * groups;
    if (pl >= ppb) return;
    // fp32: a thread owns channels [4 gi, 4 gi + 4) and [4 groups + 4 gi, ...): each of its two 16-byte stores continues its neighbour's
    const int c0 = VEC == 8 ? gi * 8 : gi * 4, c1 = VEC == 8 ? gi * 8 + 4 : groups * 4 + gi * 4;
    float bv[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bv[j] = (bias != nullptr && c0 + j < Cout) ? bias[c0 + j] : 0.f;
        bv[4 + j] = (bias != nullptr && c1 + j < Cout) ? bias[c1 + j] : 0.f;
    }
    // 32-bit pixel arithmetic (the launcher checks the sizes): 64-bit divisions cost more instructions than the nine taps
    for (unsigned pb = blockIdx.x * (unsigned)(ppb * PX); pb < (unsigned)P; pb += gridDim.x * (unsigned)(ppb * PX)) {
        f32x2 o[PX][4];
        const T* xp[PX];
        int iy0[PX], ix0[PX];
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            unsigned p = pb + (unsigned)(q * ppb + pl);
            if (p >= (unsigned)P) p = (unsigned)P - 1;         // computed, not stored
            const unsigned row = p / (unsigned)OW, img = row / (unsigned)OH;
            const int ox = (int)(p - row * (unsigned)OW), oy = (int)(row - img * (unsigned)OH);
            iy0[q] = oy * stride - 1;
            ix0[q] = ox * stride - 1;
            xp[q] = x + (size_t)img * IH * IW * x_cs + x_co + (iy0[q] * IW + ix0[q]) * x_cs;
#pragma unroll
            for (int j = 0; j < 4; ++j) o[q][j] = f32x2{0.f, 0.f};
        }
        // one tap at a time, the next tap's pixels in flight while this one is multiplied (a fully unrolled tap loop keeps 9 x PX vectors and
        // 72 filter vectors live: 256 VGPRs, one wave per SIMD)
        float cur[PX][NX * VEC], nxt[PX][NX * VEC];
        auto load = [&](int t, float (&d)[PX][NX * VEC]) {
            const int ky = t / 3, kx = t - 3 * ky;
#pragma unroll
            for (int q = 0; q < PX; ++q) {
                const int iy = iy0[q] + ky, ix = ix0[q] + kx;
                const bool in = (unsigned)iy < (unsigned)IH && (unsigned)ix < (unsigned)IW;
#pragma unroll
                for (int h = 0; h < NX; ++h) {
                    float v[VEC];
                    SmallK<T>::ld(in ? xp[q] + (ky * IW + kx) * x_cs + h * VEC : x + x_co, v);      // branch-free: a padding tap reads pixel 0 and drops it
#pragma unroll
                    for (int j = 0; j < VEC; ++j) d[q][h * VEC + j] = in ? v[j] : 0.f;
                }
            }
        };
        load(0, cur);
#pragma unroll 1
        for (int t = 0; t < 9; ++t) {
            if (t < 8) load(t + 1, nxt);
            const float* wt = wl + t * 8 * cp;
#pragma unroll
            for (int r = 0; r < NX * VEC; ++r) {
                if (r < Cin) {
                    // v_pk_fma_f32: two channels per lane and instruction (each half is the same fused multiply-add the scalar form does)
                    const f32x2 wa = *reinterpret_cast<const f32x2*>(wt + r * cp + c0), wb = *reinterpret_cast<const f32x2*>(wt + r * cp + c0 + 2);
                    const f32x2 wc = *reinterpret_cast<const f32x2*>(wt + r * cp + c1), wd = *reinterpret_cast<const f32x2*>(wt + r * cp + c1 + 2);
#pragma unroll
                    for (int q = 0; q < PX; ++q) {
                        const f32x2 a = {cur[q][r], cur[q][r]};
                        o[q][0] = __builtin_elementwise_fma(a, wa, o[q][0]);
                        o[q][1] = __builtin_elementwise_fma(a, wb, o[q][1]);
                        o[q][2] = __builtin_elementwise_fma(a, wc, o[q][2]);
                        o[q][3] = __builtin_elementwise_fma(a, wd, o[q][3]);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < PX; ++q)
#pragma unroll
                for (int r = 0; r < NX * VEC; ++r) cur[q][r] = nxt[q][r];
        }
#pragma unroll
        for (int q = 0; q < PX; ++q) {
            const unsigned p = pb + (unsigned)(q * ppb + pl);
            if (p >= (unsigned)P) break;
            float e[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                e[j] = o[q][j >> 1][j & 1] + bv[j];
                if (relu) e[j] = fmaxf(e[j], 0.f);
                if ((j < 4 ? c0 + j : c1 + j - 4) >= Cout) e[j] = 0.f;
            }
            T* yo = y + (size_t)p * y_cs + y_co;
            if constexpr (VEC == 8) SmallK<T>::st(yo + c0, e);
            else {
                const float lo[4] = {e[0], e[1], e[2], e[3]}, hi[4] = {e[4], e[5], e[6], e[7]};
                if (c0 < Cout) SmallK<T>::st(yo + c0, lo);
                if (c1 < Cout) SmallK<T>::st(yo + c1, hi);
            }
        }
    }
}

// the launches it takes: a plain forward 3x3 conv (stride 1 | 2, padding 1) of <= 8 input channels held in ONE chunk of the packed image,
// 16..32 output channels in whole vectors, no residual / mask / column sums / channel range.  Measured alone at 16 x 4 x 512^2 -> 32, stride 2
// (scripts/ab_conv_smallcin.py): fp32 86 us against 271 us on conv_igemm16_kernel (bit-identical results: the same ascending (tap, channel)
// chain of fused multiply-adds per output), bf16 93 against 180 us.  64 outputs lose to the MFMA kernels (8 -> 64: 273 / 219 us against
// 150 / 55 us), and so does a single bf16 tile (17 against 13 us): bf16 launches below 2^18 output pixels stay where they were -- counted with
// unet_tuning.plan_batch when that is set, so that a batch-invariant plan picks ONE kernel for every batch size.
bool smallcin_applies(const unet_conv_desc* d) {
    const int vec = d->dtype == UNET_BF16 ? 8 : 4;
    const int plan_n = unetconv::tuning_of(d->tuning).plan_batch > 0 ? unetconv::tuning_of(d->tuning).plan_batch : d->N;
    return d->ks == 3 && (d->stride == 1 || d->stride == 2) && d->kind == UNET_CONV_FWD && d->Cin <= 8 && d->Cout >= 16 && d->Cout <= 32 &&
           d->Cout % vec == 0 && d->res == nullptr && !(d->flags & UNET_CONV_MASK) && d->colsum == nullptr && d->colsumsq == nullptr &&
           d->cout_begin == 0 && (d->cout_count == 0 || d->cout_count == d->Cout) && d->wp_img_stride == 0 && !d->pixel_shuffle &&
           !(d->dtype == UNET_BF16 && d->y_f32) && d->Cout <= d->y_cs - d->y_co && unet::roundup(d->Cin, vec) <= d->x_cs - d->x_co &&
           (d->dtype != UNET_BF16 || (long long)plan_n * d->OH * d->OW >= (1ll << 18)) &&
           (long long)d->N * d->OH * d->OW < (1ll << 31) - 256 * 12 * 1024 && (long long)d->IH * d->IW * d->x_cs < (1ll << 31);
}

template <typename T, int PX>
int launch_smallcin_px(const unet_conv_desc* d, hipStream_t st) {
    const long long P = (long long)d->N * d->OH * d->OW;
    const int groups = unet::cdiv(d->Cout, 8), ppb = 256 / groups * PX;
    long long blocks = (P + ppb - 1) / ppb;
    if (blocks > 256 * 12) blocks = 256 * 12;
    auto kern = conv3x3_smallcin_kernel<T, PX, 1>;
    if (d->Cin > SmallK<T>::VEC) kern = conv3x3_smallcin_kernel<T, PX, 8 / SmallK<T>::VEC>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, (const T*)d->x, d->x_cs, d->x_co, (const T*)d->wp, d->Cin,
                       unet::roundup(d->Cout, 128), d->bias, (T*)d->y, d->y_cs, d->y_co, d->IH, d->IW, d->OH, d->OW, d->stride, P, d->Cout, groups,
                       (d->flags & UNET_CONV_RELU) ? 1 : 0);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
// four pixels per thread once that still leaves >= 4 workgroups per CU, else one (a single 512^2 tile: 256 workgroups instead of 64)
template <typename T>
int launch_smallcin(const unet_conv_desc* d, hipStream_t st) {
    const long long P = (long long)d->N * d->OH * d->OW;
    const int ppb4 = 256 / unet::cdiv(d->Cout, 8) * 4;
    return (P + ppb4 - 1) / ppb4 >= 1024 ? launch_smallcin_px<T, 4>(d, st) : launch_smallcin_px<T, 1>(d, st);
}
}  // namespace

namespace unetconv {
bool conv_smallk_applies(const unet_conv_desc* d) { return smallk_applies(d); }
int conv_smallk_bf16(const unet_conv_desc* d, hipStream_t st) { return launch_smallk<unsigned short>(d, st); }
bool conv_smallcin_applies(const unet_conv_desc* d) { return tuning_of(d->tuning).conv_smallcin != 0 && smallcin_applies(d); }
int conv_smallcin_bf16(const unet_conv_desc* d, hipStream_t st) { return launch_smallcin<unsigned short>(d, st); }
}

// plan with split-K when the caller brought a workspace for it, else the plain plan
static int make_plan_ws(const unet_conv_desc* d, Plan* p) {
    int rc = make_plan(d, p);
    if (rc != UNET_OK) return rc;
    if (!unetconv::splitk_redirect(d, p)) rc = make_plan(d, p, 0);
    return rc;
}

extern "C" int unet_conv2d(const unet_conv_desc* d, void* stream) {
    if (d != nullptr && d->pixel_shuffle) {          // 1x1 conv + activation + PixelShuffle(2) store: conv1x1_gemm_kernel only (both storage types)
        const int rc = unetconv::conv_gemm1x1_ps_check(d);
        return rc == UNET_OK ? unetconv::conv_gemm1x1(d, (hipStream_t)stream) : rc;
    }
    if (d != nullptr && d->dtype == UNET_BF16) return unetconv::conv2d_bf16(d, (hipStream_t)stream);
    UNET_CHECK_ARG(d == nullptr || d->dtype == UNET_F32, "conv: unknown dtype %d", d->dtype);
    Plan p;
    int rc = make_plan_ws(d, &p);
    if (rc != UNET_OK) return rc;
    hipStream_t st = (hipStream_t)stream;
    if (smallk_applies(d)) return launch_smallk<float>(d, st);
    if (unetconv::conv_smallcin_applies(d)) return launch_smallcin<float>(d, st);
    if (unetconv::conv_head1x1_applies(d)) return unetconv::conv_head1x1(d, st);
    if (unetconv::conv_gemm1x1_applies(d)) return unetconv::conv_gemm1x1(d, st);
    if (p.hit == 6) rc = unetconv::conv2d_t256_f32(p, st);          // the 256-pixel tile (conv_bf16.hip: conv_bf16_t256_kernel<.., float>)
    else rc = (p.hit == 10) ? launch_tw<10>(p, st) : launch_tw<4>(p, st);
    if (rc != UNET_OK || p.splits <= 1) return rc;
    return unetconv::splitk_reduce(d, p, st);
}

extern "C" size_t unet_conv2d_splitk_workspace(const unet_conv_desc* d) {
    Plan p;
    if (d == nullptr || d->pixel_shuffle) return 0;
    const int rc = d->dtype == UNET_BF16 ? unetconv::plan_bf16_public(d, &p) : make_plan(d, &p);
    return rc == UNET_OK ? p.ws_floats : 0;
}

extern "C" size_t unet_pack_weights_size(int Cout, int Cin, int ks, int mode) {
    const int T = ks * ks;
    const int red = mode == 1 ? Cout : Cin, out = mode == 1 ? Cin : Cout;
    return unetconv::f32_image_elems(red, out, T);
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
    UNET_CHECK_ARG((ks == 1 || ks == 3) && (mode == 0 || mode == 1 || (mode == 2 && ks == 1 && Cout % 64 == 0)) && Cout > 0 && Cin > 0, "pack_weights: bad args");
    const int T = ks * ks;
    const int red = mode == 1 ? Cout : Cin, out = mode == 1 ? Cin : Cout;
    const int nchunks = unet::cdiv(red, KC), outPad = unet::roundup(out, 128);
    const size_t total = (size_t)T * nchunks * outPad * KC;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(unet::ew_grid((long long)total, 256)), dim3(256), 0, (hipStream_t)stream,
                       w, wp, Cout, Cin, T, mode, nchunks, outPad, total);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
