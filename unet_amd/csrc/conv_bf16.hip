// Implicit-GEMM convolution with bf16 storage and fp32 accumulation for gfx950 (MI355X): v_mfma_f32_16x16x32_bf16.
//
// The bf16-storage variant of BASELINE.json configs[1]: activations, activation gradients and the packed filter image are bf16,
// every product is accumulated in fp32 by the matrix core, master weights / optimizer state stay fp32 (conv_igemm.hip is the fp32
// parity path).  Same workgroup geometry, halo staging, direct global -> VGPR filter operand, asm-scheduled loads and XCD-aware
// tile order as conv_igemm16_kernel -- a reduction chunk is again 64 bytes per pixel, now 32 channels, and ONE MFMA consumes
// what four v_mfma_f32_16x16x4_f32 did -- with two differences:
//   * operand roles are swapped: A = filter tile (16 output channels x 32 k), B = pixel tile (32 k x 16 pixels), so that the
//     accumulator of lane (pixel l&15, quad l>>4) holds FOUR CONSECUTIVE CHANNELS of one pixel: the epilogue moves 8-byte
//     (bf16) / 16-byte (fp32 logits) vectors per lane instead of scalars;
//   * a reduction tail (Cin % 32 != 0) is zero padded (the math is 16x cheaper than fp32; the kernel is load bound).
//
// lane l: A[i = l&15][k = 8(l>>4) + j] = W[cout i][channel 8(l>>4) + j]   (packed image wp[tap][chunk][cout][32], 1 KiB per tile)
//         B[k = 8(l>>4) + j][col = l&15] = halo[pixel l&15 shifted by tap][channel 8(l>>4) + j]   (one ds_read_b128)
//         D[row = 4(l>>4) + r][col = l&15] -> channel 4(l>>4) + r of pixel l&15
//
// Replaces the same ATen conv kernels as conv_igemm.hip (fastai ConvLayer convs of the model built at reference train.py:128-144).

#include "conv_common.h"

namespace {

using namespace unetconv;

constexpr int KCB = 32;   // reduction channels per chunk (64 bytes of bf16)
// LDS halo row length in dwords: 64 bytes of channels + 32 bytes pad.  With 96-byte rows the 16 lanes of every ds_read_b128 lane group
// ({0-3,12-15,20-27}, ...: pixel rows r, 16-byte slot (6 r + kq) mod 16) hit 16 different slots -- conflict free; the 80-byte rows of
// the fp32 kernel are 2-way conflicted on 3 of 16 slots (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = 0.5), which that MFMA-bound
// kernel hides and this one (27 % MFMA busy) does not.
constexpr int LDKB = 24;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

__device__ __forceinline__ f32x4 ld_bf16x4(const u16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ void st_bf16x4(u16* p, f32x4 v) {
    const bf16x4 h = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};       // v_cvt_pk_bf16_f32: round to nearest even
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, h);
}

template <int TW, int MT, int NT, int WM, int WN, int HIT>
__global__ __launch_bounds__(WM* WN * 64, HIT == 4 ? 3 : 2) void conv_bf16_kernel(const KArgs a, const int y_f32) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32, TH = BM / TW, NTH = WM * WN * 64;
    constexpr int M16 = 2 * MT, N16 = 2 * NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l15 = lane & 15, kq = lane >> 4;
    const TapSet& ts = a.taps[blockIdx.z];

    // XCD-aware tile order (see conv_igemm16_kernel): every XCD owns one contiguous range of tiles
    const int per_xcd = (int)(gridDim.x >> 3);
    int bid = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (bid >= a.mtiles * a.ntn) return;
    const int nt = __builtin_amdgcn_readfirstlane(bid % a.ntn); bid /= a.ntn;
    const int tx_t = __builtin_amdgcn_readfirstlane(bid % a.tiles_x); bid /= a.tiles_x;
    const int ty_t = __builtin_amdgcn_readfirstlane(bid % a.tiles_y);
    const int img = __builtin_amdgcn_readfirstlane(bid / a.tiles_y);
    const int oy0 = ty_t * TH, ox0 = tx_t * TW;
    const int n0 = a.n_base + nt * BN;
    // split-K: this workgroup reduces chunks [kc0, kc1) only and writes fp32 partial sums to slab blockIdx.y (conv_common.h, Plan)
    const int kc0 = a.cps ? (int)blockIdx.y * a.cps : 0;
    const int kc1 = a.cps ? (kc0 + a.cps < a.nchunks ? kc0 + a.cps : a.nchunks) : a.nchunks;

    const int S = a.S;
    const int HH = (TH - 1) * S + ts.ext_y, HW = (TW - 1) * S + ts.ext_x;
    const int HPIX = HH * HW;
    float* lds0 = smem + 32;
    auto halo_buf = [&](int b) -> float* { return lds0 + b * (HPIX * LDKB); };
    const unsigned long long dpack = ts.dpack, wpack = ts.wpack;
#define TAP_OFF(t_) ({ const unsigned d_ = (unsigned)(dpack >> (4 * (t_))); (int)(((d_ & 3u) * HW + ((d_ >> 2) & 3u)) * LDKB); })
#define TAP_WIDX(t_) ((int)((unsigned)(wpack >> (4 * (t_))) & 15u))

    // halo items: 16 bytes = 8 channels; 4 items per pixel and chunk.  goff = bf16 ELEMENT offset inside the image
    const char* xb = reinterpret_cast<const char*>(a.x) + (size_t)img * a.IH * a.IW * a.x_cs * 2;
    const int iy0 = oy0 * S + ts.min_dy, ix0 = ox0 * S + ts.min_dx;
    int goff[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) {
        const int e = tid + it * NTH;
        const int p = e >> 2, q = e & 3;
        const int hy = p / HW, hx = p - hy * HW;
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool inb = (e < HPIX * 4) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW;
        goff[it] = inb ? ((iy * a.IW + ix) * a.x_cs + a.x_co + 8 * q) : -1;
    }

    v4f hreg[HIT];
#define HALO_OK(it_, c0_) (goff[it_] >= 0 && ((c0_) + 8 * ((tid + (it_) * NTH) & 3)) < a.Cin4)
#define LOAD_HALO(chunk_, on_) do { const int c0_ = (chunk_) * KCB; unsigned vo_[HIT]; \
        if (on_) { _Pragma("unroll") for (int it = 0; it < HIT; ++it) vo_[it] = HALO_OK(it, c0_) ? (unsigned)(goff[it] + c0_) * 2u : 0u; } \
        else { _Pragma("unroll") for (int it = 0; it < HIT; ++it) vo_[it] = 0u; } \
        gld_halo<HIT>(hreg, vo_, xb, (on_)); } while (0)
#define STORE_HALO(dst_, chunk_) do { const int c0_ = (chunk_) * KCB; \
        _Pragma("unroll") for (int it = 0; it < HIT; ++it) { \
            const int e_ = tid + it * NTH; \
            if (e_ < HPIX * 4) { \
                const v4f v_ = HALO_OK(it, c0_) ? hreg[it] : (v4f){0.f, 0.f, 0.f, 0.f}; \
                *reinterpret_cast<v4f*>((dst_) + (e_ >> 2) * LDKB + (e_ & 3) * 4) = v_; \
            } } } while (0)

    // 16-wide output-channel tiles dealt round-robin to the WN waves; tiles entirely beyond the produced range are skipped
    int nvalid = ((a.n_end - n0 + 15) / 16 - wn + WN - 1) / WN;
    nvalid = (a.n_end - n0 <= wn * 16) ? 0 : (nvalid > N16 ? N16 : nvalid);
    // An ODD number of live tiles (Cout = 100: 7 of the block's 8) would leave the wn = 0 waves one tile more than the wn = 1 waves.  The
    // odd last tile is SHARED instead: both waves of a pixel row load it (wave wn = 1 reads its last tile one KiB lower), and each
    // multiplies it with the first half of ITS pixel tiles -- wave wn = 1 numbers its pixel tiles rotated by half, so the two halves
    // are covered by the same code.  Every output element keeps its k-ordered accumulation chain: results do not change.
    // Measured (scripts/conv_sweep.py, 16 x 512^2): the step is MFMA-throughput bound per CU, not per wave -- the balance alone is
    // worth ~2 %; what the 100-channel layers lose against 96 / 128 channels is the padding of 6.25 tiles to 7.
    int nsplit = -1;
    if constexpr (WN == 2 && (M16 & 1) == 0) {
        if ((a.n_end - n0 + 15) / 16 == 2 * N16 - 1) { nsplit = N16 - 1; nvalid = N16; }
    }
    const int nfull = nsplit >= 0 ? nsplit : nvalid;
    // the wn = 1 wave numbers its pixel tiles rotated by half, so that "pixel tiles 0..M16/2-1" of the shared tile are the other half
    const int mrot = (nsplit >= 0 && wn != 0) ? M16 / 2 : 0;
    // pixel-operand row bases (dwords) inside the halo tile
    int pbase[M16];
#pragma unroll
    for (int m = 0; m < M16; ++m) {
        const int pix = (wm * M16 + (m ^ mrot)) * 16 + l15;
        const int ty = pix / TW, tx = pix % TW;
        pbase[m] = ((ty * S) * HW + tx * S) * LDKB + 4 * kq;
    }
#define TILE_COL(n_) (((n_) == nsplit ? (n_) * WN : (n_) * WN + wn) * 16)
    const unsigned lane_b = (unsigned)((wn * 16 + l15) * 64 + 16 * kq);
    const unsigned lane_b_last = lane_b - (unsigned)((nsplit >= 0 ? wn : 0) * 16 * 64);     // the shared tile: one tile (1 KiB) lower for wn = 1
    const char* wbase = reinterpret_cast<const char*>(a.wp) + ((size_t)img * a.wp_stride + (size_t)n0 * KCB) * 2;
    const size_t slab_b = (size_t)a.coutPad * KCB * 2;
    constexpr int TSTR = WN * 16 * 64;     // bytes between two of this wave's filter tiles
    v4f b0[N16], b1[N16];
#define LOAD_B(dst_, slab_) gld_bl<TSTR, N16>((dst_), lane_b, lane_b_last, wbase + (size_t)(slab_) * slab_b)

    f32x4 acc[M16][N16];
#pragma unroll
    for (int m = 0; m < M16; ++m)
#pragma unroll
        for (int n = 0; n < N16; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int ntaps = ts.n;
#pragma unroll
    for (int it = 0; it < HIT; ++it) hreg[it] = (v4f){0.f, 0.f, 0.f, 0.f};
    // reduction tail folded over the taps (conv_common.h, bf16_fold_tail): the last chunk runs ntail stages of four taps each
    const int fold = a.fold, last = a.nchunks - 1, ntail = (ntaps + 3) >> 2;
    const int fold_slab0 = 9 * a.nchunks;
    LOAD_HALO(kc0, true);
    LOAD_B(b0, (fold && last == 0) ? fold_slab0 : ts.widx[0] * a.nchunks + kc0);      // (a split launch never folds)
    wait_loads(b0, hreg);
    STORE_HALO(halo_buf(kc0 & 1), kc0);
    __syncthreads();

    int t = 0, chunk = kc0;
    const float* hb = halo_buf(kc0 & 1);
    // stage = (chunk, tap): the next stage's filter tiles (and, during the first tap of a chunk, the next chunk's halo items) are
    // issued at the top and waited for behind the stage's MFMAs; ONE wait asm per stage redefines every register a load writes
    constexpr int MH = M16 / 2;
    constexpr bool HALF_LOADS = M16 > 4;
#define STAGE_BODY(bu_, bl_) do { \
        LOAD_HALO(chunk + 1, t == 0 && chunk + 1 < kc1); \
        LOAD_B(bl_, has_next_ ? ((fold && cn_ == last) ? fold_slab0 + tn_ : TAP_WIDX(tn_) * a.nchunks + cn_) : 0); \
        int aoff_; \
        if (fold && chunk == last) {   /* lane group kq reads channel group 0 of the tail chunk at ITS tap, 4 t + kq */ \
            int tl_ = 4 * t + kq; \
            tl_ = tl_ < ntaps ? tl_ : 0;        /* (beyond the taps the fold slab holds zeros) */ \
            aoff_ = TAP_OFF(tl_) - 4 * kq; \
        } else aoff_ = TAP_OFF(t); \
        const float* ha_ = hb + aoff_; \
        /* pixel tiles in two halves (the shared odd filter tile takes the first only); the 8-tile wave also reads them from LDS per half: \
           16 operand registers live instead of 32 */ \
        bf16x8 pv_[HALF_LOADS ? MH : M16]; \
        if (!HALF_LOADS) { \
            _Pragma("unroll") for (int m = 0; m < M16; ++m) pv_[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f*>(ha_ + pbase[m])); \
        } \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) { \
            if (HALF_LOADS) { \
                _Pragma("unroll") for (int m = 0; m < MH; ++m) \
                    pv_[m] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const v4f*>(ha_ + pbase[h * MH + m])); \
            } \
            _Pragma("unroll") for (int n = 0; n < N16; ++n) { \
                if (n < nfull || (h == 0 && n == nsplit)) { \
                    const bf16x8 wv_ = __builtin_bit_cast(bf16x8, (bu_)[n]); \
                    _Pragma("unroll") for (int m = 0; m < MH; ++m) \
                        acc[h * MH + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv_, pv_[HALF_LOADS ? m : h * MH + m], acc[h * MH + m][n], 0, 0, 0); \
                } \
            } \
        } \
        wait_loads(bl_, hreg); \
    } while (0)
#define STAGE(bu_, bl_) do { \
        int tn_ = t + 1, cn_ = chunk; \
        if (tn_ == ((fold && chunk == last) ? ntail : ntaps)) { tn_ = 0; cn_ = chunk + 1; } \
        const bool has_next_ = cn_ < kc1; \
        STAGE_BODY(bu_, bl_); \
        if (tn_ == 0 && has_next_) { \
            STORE_HALO(halo_buf(cn_ & 1), cn_); \
            __syncthreads(); \
            hb = halo_buf(cn_ & 1); \
        } \
        t = tn_; chunk = cn_; \
    } while (0)

    const int total = fold ? last * ntaps + ntail : (kc1 - kc0) * ntaps;
    for (int s = 0; s + 1 < total; s += 2) {
        STAGE(b0, b1);
        STAGE(b1, b0);
    }
    if (total & 1) STAGE(b0, b1);
#undef STAGE_BODY
#undef STAGE
#undef TAP_OFF
#undef TAP_WIDX
#undef HALO_OK
#undef LOAD_HALO
#undef STORE_HALO
#undef LOAD_B

    // ---- epilogue: lane (pixel l15 of every pixel tile) x (4 consecutive channels 4kq.. of every channel tile) ----
    const bool relu = a.flags & UNET_CONV_RELU;
    const int OS = a.OS;
    const size_t img_pix = (size_t)img * a.OH * a.OW;
    const u16* resb = a.res ? reinterpret_cast<const u16*>(a.res) + img_pix * a.res_cs + a.res_co : nullptr;
    const u16* maskb = a.mask ? reinterpret_cast<const u16*>(a.mask) + img_pix * a.mask_cs + a.mask_co : nullptr;
    int pidx[M16];
    bool pval[M16];
#pragma unroll
    for (int m = 0; m < M16; ++m) {
        const int pix = (wm * M16 + (m ^ mrot)) * 16 + l15;
        const int ty = pix / TW, tx = pix % TW;
        const int oyt = oy0 + ty, oxt = ox0 + tx;
        const int oy = oyt * OS + ts.py, ox = oxt * OS + ts.px;
        pval[m] = oyt < a.TSH && oxt < a.TSW && oy < a.OH && ox < a.OW;
        pidx[m] = pval[m] ? (oy * a.OW + ox) : 0;
    }
#pragma unroll
    for (int n = 0; n < N16; ++n) {
        if (n >= nvalid) continue;
        const int c4 = n0 + TILE_COL(n) + 4 * kq;
        const int m_hi = (n == nsplit) ? M16 / 2 : M16;   // this wave's share of tile n
        const bool cvalid = c4 < a.n_end;       // channels c4..c4+3 beyond Cout (Cout % 4 != 0) are zero filters: zeros land in pad lanes
        const int cc = cvalid ? c4 : 0;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr && cvalid) {
#pragma unroll
            for (int r = 0; r < 4; ++r) bv[r] = (c4 + r < a.Cout) ? a.bias[c4 + r] : 0.f;
        }
        f32x4 v[M16];
#pragma unroll
        for (int m = 0; m < M16; ++m) v[m] = acc[m][n] + bv;
        if (resb != nullptr) {
            f32x4 rv[M16];
#pragma unroll
            for (int m = 0; m < M16; ++m) rv[m] = ld_bf16x4(resb + (size_t)pidx[m] * a.res_cs + cc);
#pragma unroll
            for (int m = 0; m < M16; ++m) v[m] += rv[m];
        }
        if (relu) {
#pragma unroll
            for (int m = 0; m < M16; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m][r] = fmaxf(v[m][r], 0.f);
        }
        if (maskb != nullptr) {
            f32x4 mv[M16];
#pragma unroll
            for (int m = 0; m < M16; ++m) mv[m] = ld_bf16x4(maskb + (size_t)pidx[m] * a.mask_cs + cc);
#pragma unroll
            for (int m = 0; m < M16; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[m][r] = mv[m][r] > 0.f ? v[m][r] : 0.f;
        }
        if (y_f32) {
            float* yb = a.y + (a.cps ? (size_t)blockIdx.y * a.slab : (size_t)0) + img_pix * a.y_cs + a.y_co;
#pragma unroll
            for (int m = 0; m < M16; ++m)
                if (cvalid && pval[m] && m < m_hi) *reinterpret_cast<f32x4*>(yb + (size_t)pidx[m] * a.y_cs + c4) = v[m];
        } else {
            u16* yb = reinterpret_cast<u16*>(a.y) + img_pix * a.y_cs + a.y_co;
#pragma unroll
            for (int m = 0; m < M16; ++m)
                if (cvalid && pval[m] && m < m_hi) st_bf16x4(yb + (size_t)pidx[m] * a.y_cs + c4, v[m]);
        }
    }
#undef TILE_COL
}

// ---- the 256-pixel tile of the large 3x3 layers: a main loop without bookkeeping --------------------------------------------------
// In-kernel stamps of conv_bf16_kernel<32,4,2,2,2,6> (scripts/stamps_conv_bf16.py, 128 -> 128 at 16 x 512^2): 77 % of a workgroup's life
// is the main loop, 2070 clocks per stage for the 2 x 32 MFMAs (1024 clocks) of the two waves of a SIMD, and only 10 % of that inside the
// vmcnt waits and barriers.  The ISA of one stage: 32 MFMAs next to ~130 SALU, ~50 VALU, 35 branches and 44 s_waitcnt -- tap-table
// decoding, 64-bit slab addresses, readfirstlanes, per-tile validity branches, LDS addresses.  A wave issues one instruction per ~4
// clocks in order, so each wave spends ~1000 clocks per stage issuing bookkeeping during which it issues no MFMA: the loop is
// instruction-issue bound.  This kernel is the same decomposition (2 x 2 waves, each 128 pixels x 64 channels, halo tile of 10 x 34
// pixels x 32 channels double buffered in LDS, filter tiles global -> VGPR one stage ahead, identical accumulation chains per output
// element up to the tap order of the input gradient) specialised to what the planner sends here -- 3x3 filter, stride 1, 32-pixel-wide
// tile -- so that everything the generic loop computes per stage is a literal:
//   * the nine taps are unrolled: LDS operand address = ONE lane register + an immediate (tap, pixel tile), no address arithmetic;
//   * the filter slab pointer advances by one 64-bit scalar add per stage (taps in LDS order; the input gradient walks the slabs
//     backwards: KArgs.sliver = 1);
//   * halo items are buffer loads (out-of-image and channel-tail items read as zero through the descriptor's range check: no
//     clamping, no select before the LDS store), issued in the first tap of a chunk only;
//   * only the third / fourth channel tile of a wave can be absent or shared: two wave-uniform branches per half stage.
constexpr int T256_ROWB = 4 * LDKB;          // bytes per halo pixel in LDS (64 of channels + 32 pad)
typedef int v4i __attribute__((ext_vector_type(4)));

// NL filter tiles of one stage at p + lane offset + {0, 2048, 4096, 6144}; the last one through `vs` when it is the shared tile (SH)
template <int NL, int SH>
__device__ __forceinline__ void gld_bn(v4f (&d)[4], unsigned v0, unsigned v2, unsigned vs, const char* p) {
    static_assert(NL >= 1 && NL <= 4, "filter tiles per wave");
    if constexpr (NL == 4)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %4, %7\n\tglobal_load_dwordx4 %1, %4, %7 offset:2048\n\t"
                     "global_load_dwordx4 %2, %5, %7\n\tglobal_load_dwordx4 %3, %6, %7 offset:2048"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]), "=&v"(d[3]) : "v"(v0), "v"(v2), "v"(SH ? vs : v2), "s"(p));
    else if constexpr (NL == 3)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %3, %5\n\tglobal_load_dwordx4 %1, %3, %5 offset:2048\n\t"
                     "global_load_dwordx4 %2, %4, %5"
                     : "=&v"(d[0]), "=&v"(d[1]), "=&v"(d[2]) : "v"(v0), "v"(SH ? vs : v2), "s"(p));
    else if constexpr (NL == 2)
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %2, %4\n\tglobal_load_dwordx4 %1, %3, %4 offset:2048"
                     : "=&v"(d[0]), "=&v"(d[1]) : "v"(v0), "v"(SH ? vs : v0), "s"(p));
    else
        asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=&v"(d[0]) : "v"(SH ? vs : v0), "s"(p));
}
// six halo items through a buffer descriptor; out-of-range offsets read as zero
__device__ __forceinline__ void gld_halo6_buf(v4f (&h)[6], const unsigned (&vo)[6], v4i rs, int soff) {
    asm volatile("s_nop 4\n\t"
                 "buffer_load_dwordx4 %[h0], %[o0], %[rs], %[so] offen\n\tbuffer_load_dwordx4 %[h1], %[o1], %[rs], %[so] offen\n\t"
                 "buffer_load_dwordx4 %[h2], %[o2], %[rs], %[so] offen\n\tbuffer_load_dwordx4 %[h3], %[o3], %[rs], %[so] offen\n\t"
                 "buffer_load_dwordx4 %[h4], %[o4], %[rs], %[so] offen\n\tbuffer_load_dwordx4 %[h5], %[o5], %[rs], %[so] offen"
                 : [h0] "+v"(h[0]), [h1] "+v"(h[1]), [h2] "+v"(h[2]), [h3] "+v"(h[3]), [h4] "+v"(h[4]), [h5] "+v"(h[5])
                 : [o0] "v"(vo[0]), [o1] "v"(vo[1]), [o2] "v"(vo[2]), [o3] "v"(vo[3]), [o4] "v"(vo[4]), [o5] "v"(vo[5]), [rs] "s"(rs), [so] "s"(soff));
}

// s_waitcnt vmcnt(N) that "defines" the filter-tile set (and, with H, the halo registers) it completes: all but the N youngest loads have
// landed; no consumer of those registers can be scheduled above it
template <int N, bool H>
__device__ __forceinline__ void wait_cnt(v4f (&x)[4], v4f (&h)[6]) {
    if constexpr (H)
        asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]),
                     "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "+v"(h[4]), "+v"(h[5]) : [n] "n"(N));
    else
        asm volatile("s_waitcnt vmcnt(%[n])" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]) : [n] "n"(N));
}

#ifdef UNET_STAMPS
// diagnostic build only (-DUNET_STAMPS; scripts/stamps_conv_bf16.py): wave 0 of every workgroup records shader-clock stamps around its
// prologue, main loop and epilogue and the clocks it spent inside the per-stage vmcnt waits and the chunk barriers
__device__ unsigned long long* g_stamps = nullptr;
#define STAMP() __builtin_amdgcn_s_memtime()
#define STAMP_WAIT(x_) do { const unsigned long long w0_ = STAMP(); x_; const unsigned long long d_ = STAMP() - w0_; stw_ += d_; stwt_[stt_] += d_; } while (0)
#define STAMP_BAR(x_) do { const unsigned long long w0_ = STAMP(); x_; stb_ += STAMP() - w0_; } while (0)
#else
#define STAMP_WAIT(x_) x_
#define STAMP_BAR(x_) x_
#endif

// NTOT = 16-wide channel tiles of the workgroup (5..8), dealt round-robin to the two waves of a pixel row: NF = NTOT / 2 full tiles per
// wave, and an odd last tile SHARED -- each wave multiplies it with the first half of ITS pixel tiles, wave wn = 1 numbers its halves
// swapped (`r`), so the code of the two waves is the same and the stage body has no branch at all.
// A workgroup walks `tpw` consecutive output tiles: the first halo tile and filter tiles of tile i + 1 are requested BEFORE the epilogue
// of tile i, so the HBM latency of a tile's prologue (14 % of a one-tile workgroup's life in the stamps) hides behind the stores.
// TW = 32 (an 8 x 32 pixel patch) or 16 (16 x 16: the stages whose output is 16..31 pixels wide).
// T = unsigned short (bf16 storage: a chunk is 32 channels, one v_mfma_f32_16x16x32_bf16 per tile pair and stage) or float (the fp32 parity
// path, same bytes everywhere: a chunk is 16 channels = 64 bytes per pixel, the filter tile 16 columns x 64 B, four v_mfma_f32_16x16x4_f32 per
// tile pair and stage with the operand roles of the bf16 form -- A = filter, B = pixels -- so that a lane again holds four consecutive
// channels of one pixel and results move as 16-byte vectors; a reduction tail (Cin % 16) runs as a channel-transposed last chunk, an output
// width of 16 n + 1..4 as the 4-channel sliver below).
// SLV (fp32 storage, odd NTOT): the last channel tile of the block holds only 1..4 real channels (the 100-wide final ResBlock: 6 x 16 + 4).
// It takes the place of the shared odd tile -- same filter loads (lane column l15 & 3 of that tile), same half of the pixel tiles per wave --
// but multiplies with v_mfma_f32_4x4x1_16B_f32: A = filter (lane 4 b + i: output channel i, reduction channel of the lane's k-slot), B = the
// pixel operand the 16x16x4 tiles use (lane = pixel l15, reduction channel 4 kq + kk), D: lane (l15, kq) gets the four output channels of ITS
// pixel, summed over the reduction channels of k-slot class kq.  Two passes instead of eight per pixel tile and MFMA step; the four kq-class
// partial sums of a pixel meet in the epilogue ((kq0 + kq1) + (kq2 + kq3): two lane exchanges, the same order in every lane).
template <int NTOT, int TW, typename T, bool SLV = false>
__global__ __launch_bounds__(256, 2) void conv_bf16_t256_kernel(const KArgs a, const int y_f32_arg, const int tpw, const int bn) {
    static_assert(!SLV || ((NTOT & 1) && sizeof(T) == 4), "the 4-channel sliver replaces the shared odd tile of the fp32 form");
    constexpr int TH = 256 / TW, M16 = 8, N16 = 4, MH = 4, HIT = 6, NTH = 256;
    constexpr int EB = (int)sizeof(T), VEC = 16 / EB, KCT = 64 / EB;       // bytes per element, channels per 16-byte item / per chunk
    const int y_f32 = EB == 4 ? 1 : y_f32_arg;
    constexpr int T256_HW = TW + 2, T256_HPIX = (TH + 2) * T256_HW, T256_BUFB = T256_HPIX * T256_ROWB;
    static_assert((TW == 32 || TW == 16) && T256_HPIX * 4 <= HIT * NTH, "pixel patch of the 256-pixel tile");
    constexpr int NF = NTOT / 2, NL = NF + (NTOT & 1);            // full tiles per wave, filter tiles a wave loads per stage
    constexpr int NST = NF * 4 + (NTOT & 1) * 2;                  // 16-byte result stores a wave issues per tile (bf16 output)
    static_assert(NTOT >= 1 && NTOT <= 8, "channel tiles of the (32- / 64- / 128-wide) block");
    extern __shared__ __attribute__((aligned(16))) char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l15 = lane & 15, kq = lane >> 4;
    const TapSet& ts = a.taps[0];
#ifdef UNET_STAMPS
    const unsigned long long st0_ = STAMP();
    unsigned long long stw_ = 0, stb_ = 0, stl_ = 0, ste_ = 0, stwt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st1_ = 0;
    int stt_ = 9;          // tap of the stage whose wait is being stamped (9: folded tail)
#endif

    // XCD-aware order (see conv_igemm16_kernel): every XCD owns one contiguous range of workgroups, a workgroup `tpw` consecutive tiles
    const int per_xcd = (int)(gridDim.x >> 3);
    const int wg = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    const int ntiles = a.mtiles * a.ntn;
    int tile = wg * tpw;
    const int tend = tile + tpw < ntiles ? tile + tpw : ntiles;
    if (tile >= tend) return;
    const int kc0 = a.cps ? (int)blockIdx.y * a.cps : 0;
    const int kc1 = a.cps ? (kc0 + a.cps < a.nchunks ? kc0 + a.cps : a.nchunks) : a.nchunks;
    const int rev = a.sliver;                                  // taps in LDS order: the input gradient reads the filter slabs backwards
    const int fold = a.fold, last = a.nchunks - 1;
    // fp32: a reduction tail (Cin % 16 != 0: the 100-channel layers) is the last chunk, stored CHANNEL-TRANSPOSED in the filter image and in LDS
    // (k-slot (kq, kk) = channel 4 kk + kq) so that MFMA step kk covers channels 4 kk .. 4 kk + 3 and the steps beyond the real channels are
    // skipped (as in conv_igemm16_kernel); it runs in a section of its own behind the full chunks
    const bool ftail = EB == 4 && (a.Cin & 15) != 0 && kc1 == a.nchunks;
    const int kend = (fold || ftail) ? last : kc1;             // chunks [kc0, kend) run nine plain stages; a folded tail chunk three
    const bool first_fold = fold && kc0 == last;

    // ---- per-tile state: output origin, channel block, filter slabs, halo items ----
    // halo items are byte offsets inside the image; out-of-image items lie far beyond the descriptor's range (they read as zero)
    constexpr unsigned OOB = 0x80000000u;
    int oy0, ox0, n0, img;
    const char *wbase, *w0, *wfold;
    v4i rs;
    unsigned goff[HIT];
    const size_t slab_b = (size_t)a.coutPad * KCT * EB;
    const long long tap_step = (rev ? -1ll : 1ll) * (long long)a.nchunks * (long long)slab_b;
#define T256_SETUP(tile_) do { \
        int id_ = (tile_); \
        const int nt_ = __builtin_amdgcn_readfirstlane(id_ % a.ntn); id_ /= a.ntn; \
        const int tx_ = __builtin_amdgcn_readfirstlane(id_ % a.tiles_x); id_ /= a.tiles_x; \
        const int ty_ = __builtin_amdgcn_readfirstlane(id_ % a.tiles_y); \
        img = __builtin_amdgcn_readfirstlane(id_ / a.tiles_y); \
        oy0 = ty_ * TH; ox0 = tx_ * TW; n0 = a.n_base + nt_ * bn; \
        wbase = reinterpret_cast<const char*>(sgpr_ptr(reinterpret_cast<const char*>(a.wp) + ((size_t)img * a.wp_stride + (size_t)n0 * KCT) * EB)); \
        w0 = wbase + (rev ? 8ll * a.nchunks * (long long)slab_b : 0ll);       /* slab of LDS tap 0, chunk 0 */ \
        wfold = wbase + (size_t)9 * a.nchunks * slab_b; \
        const u64 xb_ = sgpr_ptr(reinterpret_cast<const char*>(a.x) + (size_t)img * a.IH * a.IW * a.x_cs * EB); \
        rs[0] = (int)(unsigned)xb_; rs[1] = (int)(unsigned)(xb_ >> 32); \
        rs[2] = __builtin_amdgcn_readfirstlane(a.IH * a.IW * a.x_cs * EB); rs[3] = 0x00020000; \
        const int iy0_ = oy0 + ts.min_dy, ix0_ = ox0 + ts.min_dx; \
        _Pragma("unroll") for (int it = 0; it < HIT; ++it) { \
            const int e = tid + it * NTH; \
            const int p = e >> 2, q = e & 3; \
            const int hy = p / T256_HW, hx = p - hy * T256_HW; \
            const int iy = iy0_ + hy, ix = ix0_ + hx; \
            const bool inb = (e < T256_HPIX * 4) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW; \
            goff[it] = inb ? (unsigned)((iy * a.IW + ix) * a.x_cs + a.x_co + VEC * q) * (unsigned)EB : OOB; \
        } } while (0)
    // channels of the tail chunk beyond the (8-padded) input width belong to a neighbouring slice: this thread's channel group (tid & 3)
    const bool tail_cut = last * KCT + VEC * (tid & 3) >= a.Cin4;
    v4f hreg[HIT];
#pragma unroll
    for (int it = 0; it < HIT; ++it) hreg[it] = (v4f){0.f, 0.f, 0.f, 0.f};
#define T256_LOAD_HALO(chunk_, on_) do { unsigned vo_[HIT]; const bool cut_ = !(on_) || (tail_cut && (chunk_) == last); \
        _Pragma("unroll") for (int it = 0; it < HIT; ++it) vo_[it] = cut_ ? OOB : goff[it]; \
        gld_halo6_buf(hreg, vo_, rs, (on_) ? (chunk_) * 64 : 0); } while (0)
    const unsigned vst = (unsigned)((tid >> 2) * T256_ROWB + (tid & 3) * 16);
    const bool st5 = tid + 5 * NTH < T256_HPIX * 4;
#define T256_STORE_HALO(buf_) do { char* d_ = lds + (buf_) * T256_BUFB + vst; \
        _Pragma("unroll") for (int it = 0; it < HIT; ++it) \
            if (it < 5 || st5) *reinterpret_cast<v4f*>(d_ + it * (NTH / 4) * T256_ROWB) = hreg[it]; } while (0)

#define T256_STORE_HALO_T(buf_) do { char* d_ = lds + (buf_) * T256_BUFB + (tid >> 2) * T256_ROWB + (tid & 3) * 4; \
        _Pragma("unroll") for (int it = 0; it < HIT; ++it) \
            if (it < 5 || st5) { float* r_ = reinterpret_cast<float*>(d_ + it * (NTH / 4) * T256_ROWB); \
                r_[0] = hreg[it][0]; r_[4] = hreg[it][1]; r_[8] = hreg[it][2]; r_[12] = hreg[it][3]; } } while (0)
    // ---- channel tiles of this wave ----
    const int r = (NTOT & 1) ? wn : 0;                  // pixel-tile halves swapped (accumulator half h holds pixel tiles 4 (h ^ r) ..)
#define TILE_COL(n_) (((n_) == NF ? (n_) * 2 : (n_) * 2 + wn) * 16)
    const unsigned voff = (unsigned)((wn * 16 + l15) * 64 + 16 * kq);
    const unsigned voff2 = voff + 4096u;
    const unsigned voffs = SLV ? (unsigned)((l15 & 3) * 64 + 16 * kq) + (NF >= 2 ? 4096u : 0u)        // sliver: column l15 & 3 of the last tile
                               : (NF >= 2 ? voff2 : voff) - (unsigned)(((NTOT & 1) ? wn : 0) * 1024);      // the shared tile: one tile (1 KiB) lower for wn = 1
    v4f b0[N16], b1[N16];

    f32x4 acc[M16][N16];
#pragma unroll
    for (int m = 0; m < M16; ++m)
#pragma unroll
        for (int n = 0; n < N16; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // LDS operand address of pixel tile 0, tap 0 in buffer 0 (bytes); pixel tile m: + MOFF(m), tap t: + TOFF(t) -- immediates
    const unsigned vbase = (unsigned)(((wm * 128 / TW) * T256_HW + l15) * T256_ROWB + 16 * kq);
    constexpr int HALF_B = (64 / TW) * T256_HW * T256_ROWB;              // pixel tiles 4..7 lie 64 / TW patch rows below tiles 0..3
    const unsigned hsw = (unsigned)(r * HALF_B);
#define MOFF(m_) (((((m_) * 16) / TW) * T256_HW + ((m_) * 16) % TW) * T256_ROWB)
#define TOFF(t_) ((((t_) / 3) * T256_HW + (t_) % 3) * T256_ROWB)
    // folded tail: lane group kq reads channel group 0 of the tail chunk at ITS tap 4 j + kq (tap index in filter order); computed where
    // it is used (three registers that the main loop does not have to carry)
    auto vfold = [&](int j) -> unsigned {
        int tl = 4 * j + kq;
        tl = tl < 9 ? tl : 0;                   // (beyond the taps the fold slab holds zeros)
        const int pos = rev ? 8 - tl : tl;
        return (unsigned)(((pos / 3) * T256_HW + pos % 3) * T256_ROWB) - 16u * (unsigned)kq;
    };

    // One stage: 2 halves x (4 pixel tiles from LDS, up to 4 x 4 MFMAs).  (Reading the pixel tiles half a stage ahead into a second
    // register set -- 234 instead of 216 VGPRs -- was built and measured against this form on one box: equal within 1 %; the partner wave
    // of the SIMD already covers the LDS round trip.)
#define T256_MFMA(bu_, addr_) do { \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) { \
            v4f pv_[MH]; \
            const unsigned ah_ = (addr_) + (h == 0 ? hsw : HALF_B - hsw); \
            _Pragma("unroll") for (int m = 0; m < MH; ++m) \
                pv_[m] = *reinterpret_cast<const v4f*>(lds + ah_ + MOFF(m)); \
            _Pragma("unroll") for (int n = 0; n < NL; ++n) { \
                if (n < NF || h == 0) { \
                    if constexpr (EB == 2) { \
                        const bf16x8 wv_ = __builtin_bit_cast(bf16x8, (bu_)[n]); \
                        _Pragma("unroll") for (int m = 0; m < MH; ++m) \
                            acc[h * MH + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv_, __builtin_bit_cast(bf16x8, pv_[m]), acc[h * MH + m][n], 0, 0, 0); \
                    } else if (SLV && n == NF) { \
                        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) \
                            _Pragma("unroll") for (int m = 0; m < MH; ++m) \
                                acc[h * MH + m][n] = __builtin_amdgcn_mfma_f32_4x4x1f32((bu_)[n][kk], pv_[m][kk], acc[h * MH + m][n], 0, 0, 0); \
                    } else { \
                        _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) \
                            _Pragma("unroll") for (int m = 0; m < MH; ++m) \
                                acc[h * MH + m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32((bu_)[n][kk], pv_[m][kk], acc[h * MH + m][n], 0, 0, 0); \
                    } \
                } \
            } \
        } } while (0)

    // ---- first tile: halo tile of the first chunk, filter tiles of the first stage ----
    T256_SETUP(tile);
    T256_LOAD_HALO(kc0, true);
    gld_bn<NL, NTOT & 1>(b0, voff, voff2, voffs, first_fold ? wfold : w0 + (size_t)kc0 * slab_b);

    for (;;) {
        // ONE wait statement for the first loads of every tile, here: a second one on another path (after the epilogue, say) makes the
        // compiler merge the two definitions of the load registers with copies placed BEFORE the wait -- of registers still in flight
        // (caught by unet_amd/isa_check.py)
        wait_cnt<0, true>(b0, hreg);
        __syncthreads();                    // (every wave has left the LDS buffers of the previous tile)
        if (ftail && kc0 == last) T256_STORE_HALO_T(kc0 & 1); else T256_STORE_HALO(kc0 & 1);
        __syncthreads();
#ifdef UNET_STAMPS
        const unsigned long long sl0_ = STAMP();
        if (st1_ == 0) st1_ = sl0_;
#endif

        const char* wcur = w0 + (size_t)kc0 * slab_b;
        unsigned vcur = vbase + (unsigned)((kc0 & 1) * T256_BUFB);
        for (int chunk = kc0; chunk < kend; ++chunk) {
            const bool more = chunk + 1 < kend;
            const bool next_any = more || fold || ftail;     // another halo tile is needed
#pragma unroll
            for (int t = 0; t < 9; ++t) {
#ifdef UNET_STAMPS
                stt_ = t;
#endif
                // the NEXT stage's filter tiles (after the last stage: a dummy fetch of the first slab keeps the load / wait pattern fixed)
                if (t < 8) wcur += tap_step;
                else wcur = (more || ftail) ? w0 + (size_t)(chunk + 1) * slab_b : (fold ? wfold : wbase);
                if (t & 1) gld_bn<NL, NTOT & 1>(b0, voff, voff2, voffs, wcur); else gld_bn<NL, NTOT & 1>(b1, voff, voff2, voffs, wcur);
                if (t == 0) T256_LOAD_HALO(chunk + 1, next_any);        // (nothing follows: out-of-range offsets, the load count stays the same)
                // The wait for the next stage's filter tiles (sched_barrier: it names only the registers being loaded, nothing else keeps
                // it behind the stage's MFMAs).  vmcnt counts in issue order and the halo items were issued BEHIND the tiles of tap 1: tap 0
                // ends with vmcnt(6) and leaves them in flight until the end of tap 1 -- two stages for the HBM latency instead of one
                // (stamps by tap: the wait of tap 1 is the longest, ~530 clocks per chunk, the others 50-150).  Every chunk issues the
                // same loads (after the last one with out-of-range halo offsets), so the count is a literal.
                if (t & 1) { T256_MFMA(b1, vcur + TOFF(t)); __builtin_amdgcn_sched_barrier(0); if (t == 1) STAMP_WAIT((wait_cnt<0, true>(b0, hreg))); else STAMP_WAIT((wait_cnt<0, false>(b0, hreg))); }
                else { T256_MFMA(b0, vcur + TOFF(t)); __builtin_amdgcn_sched_barrier(0); if (t == 0) STAMP_WAIT((wait_cnt<HIT, false>(b1, hreg))); else STAMP_WAIT((wait_cnt<0, false>(b1, hreg))); }
            }
            // the ninth stage loaded b1: the next chunk starts with b0 again
#pragma unroll
            for (int n = 0; n < N16; ++n) b0[n] = b1[n];
            if (next_any) {
                if (ftail && !more) T256_STORE_HALO_T((chunk + 1) & 1); else T256_STORE_HALO((chunk + 1) & 1);
                STAMP_BAR(__syncthreads());
                vcur = vbase + (unsigned)(((chunk + 1) & 1) * T256_BUFB);
            }
        }
        if constexpr (EB == 4) {
            if (ftail) {        // nine stages over the transposed tail chunk, `ks` of the four MFMA steps each; b0 holds its first filter tiles
                const int ks = ((a.Cin - last * KCT) + 3) >> 2;
                const char* wt = w0 + (size_t)last * slab_b;
#define T256_MFMA_T(bu_, addr_) do { \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) { \
            v4f pv_[MH]; \
            const unsigned ah_ = (addr_) + (h == 0 ? hsw : HALF_B - hsw); \
            _Pragma("unroll") for (int m = 0; m < MH; ++m) pv_[m] = *reinterpret_cast<const v4f*>(lds + ah_ + MOFF(m)); \
            _Pragma("unroll") for (int n = 0; n < NL; ++n) { \
                if (n < NF || h == 0) { \
                    _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) { \
                        if (kk < ks) { \
                            _Pragma("unroll") for (int m = 0; m < MH; ++m) { \
                                if (SLV && n == NF) acc[h * MH + m][n] = __builtin_amdgcn_mfma_f32_4x4x1f32((bu_)[n][kk], pv_[m][kk], acc[h * MH + m][n], 0, 0, 0); \
                                else acc[h * MH + m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32((bu_)[n][kk], pv_[m][kk], acc[h * MH + m][n], 0, 0, 0); \
                            } \
                        } \
                    } \
                } \
            } \
        } } while (0)
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    wt = t < 8 ? wt + tap_step : wbase;          // (after the last stage: a dummy fetch, as in the main loop)
                    if (t & 1) gld_bn<NL, NTOT & 1>(b0, voff, voff2, voffs, wt); else gld_bn<NL, NTOT & 1>(b1, voff, voff2, voffs, wt);
                    if (t & 1) { T256_MFMA_T(b1, vcur + TOFF(t)); __builtin_amdgcn_sched_barrier(0); wait_cnt<0, false>(b0, hreg); }
                    else { T256_MFMA_T(b0, vcur + TOFF(t)); __builtin_amdgcn_sched_barrier(0); wait_cnt<0, false>(b1, hreg); }
                }
#undef T256_MFMA_T
            }
        }
        if (fold) {
#ifdef UNET_STAMPS
            stt_ = 9;
#endif
            gld_bn<NL, NTOT & 1>(b1, voff, voff2, voffs, wfold + slab_b);
            T256_MFMA(b0, vcur + vfold(0));
            __builtin_amdgcn_sched_barrier(0);
            STAMP_WAIT((wait_cnt<0, false>(b1, hreg)));
            gld_bn<NL, NTOT & 1>(b0, voff, voff2, voffs, wfold + 2 * slab_b);
            T256_MFMA(b1, vcur + vfold(1));
            __builtin_amdgcn_sched_barrier(0);
            STAMP_WAIT((wait_cnt<0, false>(b0, hreg)));
            T256_MFMA(b0, vcur + vfold(2));
        }
#ifdef UNET_STAMPS
        const unsigned long long se0_ = STAMP();
        stl_ += se0_ - sl0_;
#endif

        // ---- the next tile's first loads go out before this tile's epilogue (after the last tile: out-of-range halo offsets and the
        //      first slab again, so that the load / wait pattern is the same) ----
        const int oy0c = oy0, ox0c = ox0, n0c = n0, imgc = img;
        ++tile;
        const bool more_tiles = tile < tend;
        T256_SETUP(more_tiles ? tile : tile - 1);
        T256_LOAD_HALO(kc0, more_tiles);
        gld_bn<NL, NTOT & 1>(b0, voff, voff2, voffs, first_fold ? wfold : w0 + (size_t)kc0 * slab_b);

        // ---- epilogue: lane (pixel l15 of every pixel tile) x (4 consecutive channels 4kq.. of every channel tile), half the pixel
        //      tiles at a time (the next tile's 40 load registers are live here).  bf16 results move as 16-byte vectors: the lanes of
        //      16-lane rows kq and kq ^ 1 exchange halves of two pixel tiles (v_permlane16_swap_b32: odd rows of the first operand <->
        //      even rows of the second), after which lane (l15, kq) holds channels 8 (kq >> 1) .. + 7 of pixel tile 2 j + (kq & 1); the
        //      residual and the mask are fetched in that form and swapped back.  Half as many memory instructions, each 16 bytes a lane.
        const bool relu = a.flags & UNET_CONV_RELU;
        const size_t img_pix = (size_t)imgc * a.OH * a.OW;
        const T* resb = a.res ? reinterpret_cast<const T*>(a.res) + img_pix * a.res_cs + a.res_co : nullptr;
        const T* maskb = a.mask ? reinterpret_cast<const T*>(a.mask) + img_pix * a.mask_cs + a.mask_co : nullptr;
        auto ld4 = [](const T* p_) -> f32x4 { if constexpr (EB == 2) return ld_bf16x4(p_); else return *reinterpret_cast<const f32x4*>(p_); };
        auto unpack = [](unsigned lo, unsigned hi) -> f32x4 {
            return (f32x4){__uint_as_float(lo << 16), __uint_as_float(lo & 0xffff0000u), __uint_as_float(hi << 16), __uint_as_float(hi & 0xffff0000u)};
        };
        if constexpr (EB == 4) {
            // (two passes like the bf16 form below: operands applied in place, then the stores of a pixel tile over its channel tiles in order;
            // the pixel indices are computed again for the second pass -- eight more live registers made <8, 32, float> spill)
            auto pix_of = [&](int mm, int& pidx_, bool& pval_) {
                const int pix = (wm * M16 + (mm ^ (r * MH))) * 16 + l15;
                const int oy = oy0c + pix / TW, ox = ox0c + pix % TW;
                pval_ = oy < a.OH && ox < a.OW;
                pidx_ = pval_ ? (oy * a.OW + ox) : 0;
            };
            // the sliver: lane (l15, kq) holds, for each of the four pixel tiles of this wave's half, the partial sums of ITS pixel over the
            // reduction channels of k-slot class kq.  The four classes meet ((kq0 + kq1) + (kq2 + kq3), the same in every lane), then lane row kq
            // keeps pixel tile kq: one 16-byte vector of channels 96..99 per lane, with an epilogue of its own (first: 16 accumulator registers die here)
            f32x4 sv = {0.f, 0.f, 0.f, 0.f};
            int spidx = 0;
            bool spval = false;
            const int sc4 = n0c + TILE_COL(NF);
            if constexpr (SLV) {
#pragma unroll
                for (int m = 0; m < MH; ++m) {
                    f32x4 v = acc[m][NF];
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], 16);
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] += __shfl_xor(v[q], 32);
                    if (kq == m) sv = v;
                }
                pix_of(kq, spidx, spval);
                spval = spval && sc4 < a.n_end;
                if (a.bias != nullptr) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv[q] += (sc4 + q < a.Cout) ? a.bias[sc4 + q] : 0.f;
                }
                if (resb != nullptr) sv += ld4(resb + (size_t)spidx * a.res_cs + (sc4 < a.n_end ? sc4 : 0));
                if (relu) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv[q] = fmaxf(sv[q], 0.f);
                }
                if (maskb != nullptr) {
                    const f32x4 mv = ld4(maskb + (size_t)spidx * a.mask_cs + (sc4 < a.n_end ? sc4 : 0));
#pragma unroll
                    for (int q = 0; q < 4; ++q) sv[q] = mv[q] > 0.f ? sv[q] : 0.f;
                }
            }
            if (resb != nullptr || maskb != nullptr || a.bias != nullptr || relu) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int pidx[MH];
                bool pval[MH];
#pragma unroll
                for (int m = 0; m < MH; ++m) pix_of(h * MH + m, pidx[m], pval[m]);
#pragma unroll
                for (int n = 0; n < NL; ++n) {
                    if (n == NF && (h != 0 || SLV)) continue;          // the shared tile: this wave's first half only (the sliver: below)
                    const int c4 = n0c + TILE_COL(n) + 4 * kq;
                    const bool cvalid = c4 < a.n_end;
                    const int cc = cvalid ? c4 : 0;
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias != nullptr && cvalid) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) bv[q] = (c4 + q < a.Cout) ? a.bias[c4 + q] : 0.f;
                    }
#pragma unroll
                    for (int m = 0; m < MH; ++m) acc[h * MH + m][n] += bv;
                    if (resb != nullptr) {
#pragma unroll
                        for (int m = 0; m < MH; ++m) acc[h * MH + m][n] += ld4(resb + (size_t)pidx[m] * a.res_cs + cc);
                    }
                    if (relu) {
#pragma unroll
                        for (int m = 0; m < MH; ++m)
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[h * MH + m][n][q] = fmaxf(acc[h * MH + m][n][q], 0.f);
                    }
                    if (maskb != nullptr) {
#pragma unroll
                        for (int m = 0; m < MH; ++m) {
                            const f32x4 mv = ld4(maskb + (size_t)pidx[m] * a.mask_cs + cc);
#pragma unroll
                            for (int q = 0; q < 4; ++q) acc[h * MH + m][n][q] = mv[q] > 0.f ? acc[h * MH + m][n][q] : 0.f;
                        }
                    }
                }
            }
            }
            float* yb = a.y + (a.cps ? (size_t)blockIdx.y * a.slab : (size_t)0) + img_pix * a.y_cs + a.y_co;
#pragma unroll
            for (int m = 0; m < M16; ++m) {
                int pidx;
                bool pval;
                pix_of(m, pidx, pval);
#pragma unroll
                for (int n = 0; n < NL; ++n) {
                    if (n == NF && (m >= MH || SLV)) continue;
                    const int c4 = n0c + TILE_COL(n) + 4 * kq;
                    if (c4 < a.n_end && pval) *reinterpret_cast<f32x4*>(yb + (size_t)pidx * a.y_cs + c4) = acc[m][n];
                }
                if (SLV && m == MH - 1 && spval) *reinterpret_cast<f32x4*>(yb + (size_t)spidx * a.y_cs + sc4) = sv;      // (behind its pixels' other pieces)
            }
        } else if (y_f32) {          // bf16 operands, fp32 results (a cold path): one pass
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                int pidx[MH];
                bool pval[MH];
#pragma unroll
                for (int m = 0; m < MH; ++m) {
                    const int pix = (wm * M16 + ((h * MH + m) ^ (r * MH))) * 16 + l15;
                    const int oy = oy0c + pix / TW, ox = ox0c + pix % TW;
                    pval[m] = oy < a.OH && ox < a.OW;
                    pidx[m] = pval[m] ? (oy * a.OW + ox) : 0;
                }
#pragma unroll
                for (int n = 0; n < NL; ++n) {
                    if (n == NF && h != 0) continue;          // the shared tile: this wave's first half only
                    const int c4 = n0c + TILE_COL(n) + 4 * kq;
                    const bool cvalid = c4 < a.n_end;
                    const int cc = cvalid ? c4 : 0;
                    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
                    if (a.bias != nullptr && cvalid) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) bv[q] = (c4 + q < a.Cout) ? a.bias[c4 + q] : 0.f;
                    }
                    f32x4 v[MH];
#pragma unroll
                    for (int m = 0; m < MH; ++m) v[m] = acc[h * MH + m][n] + bv;
                    if (resb != nullptr) {
#pragma unroll
                        for (int m = 0; m < MH; ++m) v[m] += ld4(resb + (size_t)pidx[m] * a.res_cs + cc);
                    }
                    if (relu) {
#pragma unroll
                        for (int m = 0; m < MH; ++m)
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[m][q] = fmaxf(v[m][q], 0.f);
                    }
                    if (maskb != nullptr) {
#pragma unroll
                        for (int m = 0; m < MH; ++m) {
                            const f32x4 mv = ld4(maskb + (size_t)pidx[m] * a.mask_cs + cc);
#pragma unroll
                            for (int q = 0; q < 4; ++q) v[m][q] = mv[q] > 0.f ? v[m][q] : 0.f;
                        }
                    }
                    float* yb = a.y + (a.cps ? (size_t)blockIdx.y * a.slab : (size_t)0) + img_pix * a.y_cs + a.y_co;
#pragma unroll
                    for (int m = 0; m < MH; ++m)
                        if (cvalid && pval[m]) *reinterpret_cast<f32x4*>(yb + (size_t)pidx[m] * a.y_cs + c4) = v[m];
                }
            }
        } else {
            // the bias of this wave's channel tiles, fetched once and first: a load between two result stores would make the compiler wait for
            // the older store as well (one in-order counter) -- that wait, per channel tile, was most of the epilogue
            f32x4 bvn[NL];
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                const int c4 = n0c + TILE_COL(n) + 4 * kq;
                bvn[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (a.bias != nullptr) {
                    if (c4 + 3 < a.Cout) bvn[n] = *reinterpret_cast<const f32x4*>(a.bias + c4);
                    else {
#pragma unroll
                        for (int q = 0; q < 4; ++q) bvn[n][q] = (c4 + q < a.Cout) ? a.bias[c4 + q] : 0.f;
                    }
                }
            }
            // Two passes.  First every operand of the epilogue (residual, mask) is fetched and applied in place in the accumulators; then all
            // results leave back to back, for one pixel-tile pair the channel tiles in order.  A store covers 32 bytes of a pixel; with a
            // residual fetch (a full memory round trip) between the stores of neighbouring channel tiles the pieces of one 128-byte line
            // reached the L2 microseconds apart and many lines went out to HBM half written, twice: WRITE_SIZE of the 100-channel launches
            // at 512^2 was 1.71 GB for 0.84 GB of results, 0.97 GB without residual (scripts/conv_write_probe.py).
            // the pixel this lane moves for pixel-tile pair j of half h: tile 2 j + (kq & 1)
            int pidx2[2][MH / 2];
            bool pval2[2][MH / 2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int j = 0; j < MH / 2; ++j) {
                    const int pix = (wm * M16 + ((h * MH + 2 * j + (kq & 1)) ^ (r * MH))) * 16 + l15;
                    const int oy = oy0c + pix / TW, ox = ox0c + pix % TW;
                    pval2[h][j] = oy < a.OH && ox < a.OW;
                    pidx2[h][j] = pval2[h][j] ? (oy * a.OW + ox) : 0;
                }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int n = 0; n < NL; ++n) {
                if (n == NF && h != 0) continue;          // the shared tile: this wave's first half only
                const int c8 = n0c + TILE_COL(n) + 8 * (kq >> 1);          // the 8 channels this lane moves
                const int cc8 = c8 < a.n_end ? c8 : 0;
#pragma unroll
                for (int m = 0; m < MH; ++m) acc[h * MH + m][n] += bvn[n];
                if (resb != nullptr) {
                    uint4 rr[MH / 2];
#pragma unroll
                    for (int j = 0; j < MH / 2; ++j) rr[j] = *reinterpret_cast<const uint4*>(resb + (size_t)pidx2[h][j] * a.res_cs + cc8);
#pragma unroll
                    for (int j = 0; j < MH / 2; ++j) {
                        const auto s0 = __builtin_amdgcn_permlane16_swap(rr[j].x, rr[j].z, false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(rr[j].y, rr[j].w, false, false);
                        acc[h * MH + 2 * j][n] += unpack(s0[0], s1[0]);
                        acc[h * MH + 2 * j + 1][n] += unpack(s0[1], s1[1]);
                    }
                }
                if (relu) {
#pragma unroll
                    for (int m = 0; m < MH; ++m)
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[h * MH + m][n][q] = fmaxf(acc[h * MH + m][n][q], 0.f);
                }
                if (maskb != nullptr) {
                    uint4 rr[MH / 2];
#pragma unroll
                    for (int j = 0; j < MH / 2; ++j) rr[j] = *reinterpret_cast<const uint4*>(maskb + (size_t)pidx2[h][j] * a.mask_cs + cc8);
#pragma unroll
                    for (int j = 0; j < MH / 2; ++j) {
                        const auto s0 = __builtin_amdgcn_permlane16_swap(rr[j].x, rr[j].z, false, false);
                        const auto s1 = __builtin_amdgcn_permlane16_swap(rr[j].y, rr[j].w, false, false);
                        const f32x4 m0 = unpack(s0[0], s1[0]), m1 = unpack(s0[1], s1[1]);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc[h * MH + 2 * j][n][q] = m0[q] > 0.f ? acc[h * MH + 2 * j][n][q] : 0.f;
                            acc[h * MH + 2 * j + 1][n][q] = m1[q] > 0.f ? acc[h * MH + 2 * j + 1][n][q] : 0.f;
                        }
                    }
                }
            }
            }
            const __amdgpu_buffer_rsrc_t rsy = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<u16*>(a.y) + img_pix * a.y_cs + a.y_co, 0,
                                                                                  __builtin_amdgcn_readfirstlane(a.OH * a.OW * a.y_cs * 2), 0x00020000);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int j = 0; j < MH / 2; ++j) {
#pragma unroll
                for (int n = 0; n < NL; ++n) {
                    if (n == NF && h != 0) continue;
                    const int c8 = n0c + TILE_COL(n) + 8 * (kq >> 1);
                    const bool cvalid8 = c8 < a.n_end;
                    const f32x4 va = acc[h * MH + 2 * j][n], vb = acc[h * MH + 2 * j + 1][n];
                    const bf16x4 x_ = {(__bf16)va[0], (__bf16)va[1], (__bf16)va[2], (__bf16)va[3]};
                    const bf16x4 y_ = {(__bf16)vb[0], (__bf16)vb[1], (__bf16)vb[2], (__bf16)vb[3]};
                    const uint2 xu = __builtin_bit_cast(uint2, x_), yu = __builtin_bit_cast(uint2, y_);
                    const auto s0 = __builtin_amdgcn_permlane16_swap(xu.x, yu.x, false, false);
                    const auto s1 = __builtin_amdgcn_permlane16_swap(xu.y, yu.y, false, false);
                    typedef unsigned v4u __attribute__((ext_vector_type(4)));
                    const unsigned vo = (cvalid8 && pval2[h][j]) ? (unsigned)(pidx2[h][j] * a.y_cs + c8) * 2u : OOB;
                    __builtin_amdgcn_raw_buffer_store_b128((v4u){s0[0], s1[0], s0[1], s1[1]}, rsy, (int)vo, 0, 0);
                }
            }
            }
        }
#ifdef UNET_STAMPS
        ste_ += STAMP() - se0_;
#endif
        if (!more_tiles) break;
#pragma unroll
        for (int m = 0; m < M16; ++m)
#pragma unroll
            for (int n = 0; n < N16; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    wait_cnt<0, true>(b0, hreg);             // (the dummy loads issued after the last tile)
#undef T256_MFMA
#undef T256_LOAD_HALO
#undef T256_STORE_HALO
#undef T256_STORE_HALO_T
#undef T256_SETUP
#undef MOFF
#undef TOFF
#undef TILE_COL
#ifdef UNET_STAMPS
    if (g_stamps != nullptr && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the stores of this wave have left
        unsigned long long* o = g_stamps + (size_t)blockIdx.x * 24;
#pragma unroll
        for (int i = 0; i < 10; ++i) o[8 + i] = stwt_[i];
        o[0] = st0_; o[1] = st1_; o[2] = stl_; o[3] = STAMP(); o[4] = stw_; o[5] = stb_; o[6] = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        o[7] = 1; o[18] = ste_; o[19] = (unsigned long long)(tend - wg * tpw);
    }
#endif
}

#ifdef UNET_STAMPS
extern "C" int unet_debug_set_stamps(unsigned long long* buf) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &buf, sizeof(buf)) == hipSuccess ? 0 : -3;
}
#endif


template <int NTOT, int TW, typename T, bool SLV = false>
int launch_t256n(const Plan& p, int y_f32, hipStream_t st) {
    auto kern = conv_bf16_t256_kernel<NTOT, TW, T, SLV>;
    static unsigned long long configured = 0;
    if (unet::first_use_on_device(&configured))
        UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    // consecutive tiles per workgroup: as many as leave >= 4 workgroups for each of the 512 slots of the chip (2 per CU)
    const int ntiles = p.k.mtiles * p.k.ntn;
    int tpw = p.tune.t256_tiles_per_wg > 0 ? p.tune.t256_tiles_per_wg : ntiles / 2048;      // (unet_tuning.t256_tiles_per_wg forces a count)
    tpw = tpw < 1 ? 1 : (tpw > 16 ? 16 : tpw);
    dim3 grid = p.grid;
    grid.x = (unsigned)unet::roundup(unet::cdiv(ntiles, tpw), 8);
    hipLaunchKernelGGL(kern, grid, dim3(256), (size_t)2 * (256 / TW + 2) * (TW + 2) * T256_ROWB, st, p.k, y_f32, tpw, p.bn);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

// the channel-tile count of a block is a template parameter: a launch whose last channel block is narrower than the others (Cout = 228 in
// 128-wide blocks: 8 + 7 tiles) is issued as two launches over disjoint channel ranges
template <typename T>
int launch_t256(const Plan& p, int y_f32, hipStream_t st) {
    const int cols = p.k.n_end - p.k.n_base, nblk = p.k.ntn, full = p.bn / 16;
    const int last_tiles = (cols - (nblk - 1) * p.bn + 15) / 16;
    auto one = [&](const Plan& q, int tiles) {
        if constexpr (sizeof(T) == 4) {
            // fp32: a 7-tile block whose last tile holds 1..4 real channels (100 = 6 x 16 + 4: the final ResBlock pair) multiplies them as a sliver
            const int last_w = (q.k.n_end - q.k.n_base) - (q.k.ntn - 1) * q.bn - (tiles - 1) * 16;
            if (q.tune.t256_sliver && tiles == 7 && q.tw == 32 && last_w >= 1 && last_w <= 4) return launch_t256n<7, 32, T, true>(q, y_f32, st);
        }
        if (q.tw == 32) {
            switch (tiles) {
                case 8: return launch_t256n<8, 32, T>(q, y_f32, st);
                case 7: return launch_t256n<7, 32, T>(q, y_f32, st);
                case 6: return launch_t256n<6, 32, T>(q, y_f32, st);
                case 5: return launch_t256n<5, 32, T>(q, y_f32, st);
                case 4: return launch_t256n<4, 32, T>(q, y_f32, st);
                case 3: return launch_t256n<3, 32, T>(q, y_f32, st);
                case 2: return launch_t256n<2, 32, T>(q, y_f32, st);
                case 1: return launch_t256n<1, 32, T>(q, y_f32, st);
            }
        } else if (q.tw == 16) {
            switch (tiles) {
                case 8: return launch_t256n<8, 16, T>(q, y_f32, st);
                case 7: return launch_t256n<7, 16, T>(q, y_f32, st);
                case 6: return launch_t256n<6, 16, T>(q, y_f32, st);
                case 5: return launch_t256n<5, 16, T>(q, y_f32, st);
                case 4: return launch_t256n<4, 16, T>(q, y_f32, st);
                case 3: return launch_t256n<3, 16, T>(q, y_f32, st);
                case 2: return launch_t256n<2, 16, T>(q, y_f32, st);
                case 1: return launch_t256n<1, 16, T>(q, y_f32, st);
            }
        }
        unet::set_error("conv bf16: %d channel tiles / tile width %d in the 256-pixel tile", tiles, q.tw);
        return (int)UNET_E_UNSUPPORTED;
    };
    if (nblk == 1 || last_tiles == full) return one(p, nblk == 1 ? last_tiles : full);
    Plan q = p;                                       // the full blocks
    q.k.ntn = nblk - 1; q.k.n_end = p.k.n_base + (nblk - 1) * p.bn;
    int rc = one(q, full);
    if (rc != UNET_OK) return rc;
    q = p;                                            // the narrow last block
    q.k.ntn = 1; q.k.n_base = p.k.n_base + (nblk - 1) * p.bn;
    return one(q, last_tiles);
}

// packed filter image (bf16; layout: conv_common.h, bf16_image_value) from the fp32 master parameter [Cout,Cin,ks,ks]
//   mode 0: o = cout, reduction r = cin;  mode 1 (input gradient): o = cin, reduction r = cout
__global__ void pack_weights_bf16_kernel(const float* __restrict__ w, u16* __restrict__ wp, int Cout, int Cin, int T, int mode, int nchunks,
                                         int outPad, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
        wp[i] = __builtin_bit_cast(u16, (__bf16)bf16_image_value(w, Cout, Cin, T, mode, nchunks, outPad, i));
}

// 1x1 "weights" that are bf16 activations (self-attention operands): element (out o, reduction r) = w[o * so + r * sr]; the image
// unet_pack_weights_bf16 makes for ks = 1, mode 0: wp[chunk32][outPad][32], zero padded
__global__ void pack_weights_strided_bf16_kernel(const u16* __restrict__ w, long long so, long long sr, u16* __restrict__ wp, int O, int R,
                                                 int outPad, size_t total) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int rr = (int)(i & 31);
        const size_t j = i >> 5;
        const int o = (int)(j % outPad), r = (int)(j / outPad) * 32 + rr;
        wp[i] = (o < O && r < R) ? w[(long long)o * so + (long long)r * sr] : (u16)0;
    }
}

template <int TW, int MT, int NT, int WM, int WN, int HIT>
int launch_cfg(const Plan& p, int y_f32, hipStream_t st) {
    auto kern = conv_bf16_kernel<TW, MT, NT, WM, WN, HIT>;
    static unsigned long long configured = 0;  // per instantiation, one bit per device
    if (unet::first_use_on_device(&configured))
        UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(kern, p.grid, dim3(WM * WN * 64), p.lds_bytes, st, p.k, y_f32);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <int TW, int HIT>
int launch_bn(const Plan& p, int y_f32, hipStream_t st) {
    if (p.bm == 256) {          // (dispatched to conv_bf16_t256_kernel by conv2d_bf16)
        unet::set_error("conv bf16: inconsistent plan for the 256-pixel tile");
        return UNET_E_UNSUPPORTED;
    }
    if constexpr (HIT == 6) {
        unet::set_error("conv bf16: inconsistent plan (halo item count 6 without the 256-pixel tile)");
        return UNET_E_UNSUPPORTED;
    } else
    if (p.bm == 64) {
        if (p.bn == 64) return launch_cfg<TW, 1, 1, 2, 2, HIT>(p, y_f32, st);
        return launch_cfg<TW, 1, 2, 2, 2, HIT>(p, y_f32, st);
    }
    switch (p.bn) {
        case 32: return launch_cfg<TW, 1, 1, 4, 1, HIT>(p, y_f32, st);
        case 64: return launch_cfg<TW, 2, 1, 2, 2, HIT>(p, y_f32, st);
        default: return launch_cfg<TW, 2, 2, 2, 2, HIT>(p, y_f32, st);
    }
}

template <int HIT>
int launch_tw(const Plan& p, int y_f32, hipStream_t st) {
    switch (p.tw) {
        case 32: return launch_bn<32, HIT>(p, y_f32, st);
        case 16: return launch_bn<16, HIT>(p, y_f32, st);
        default: return launch_bn<8, HIT>(p, y_f32, st);
    }
}

// splitk < 0: the tuning's own value; 0: a plan that must not split
int plan_bf16(const unet_conv_desc* d, Plan* p, int splitk = -1) {
    UNET_CHECK_ARG(d != nullptr, "conv: null desc");
    const unet_tuning t = unetconv::tuning_of(d->tuning);
    int rc = unetconv::make_plan(d, p, KCB, 8, 16, t.bf16_big_tile, splitk < 0 ? t.conv_splitk : splitk, t.plan_batch);
    p->tune = t;
    if (rc != UNET_OK) return rc;
    UNET_CHECK_ARG(d->colsum == nullptr && d->colsumsq == nullptr, "conv bf16: column sums are not available in the bf16 kernel");
    p->lds_bytes = (size_t)(32 + 2 * p->max_hpix * LDKB) * sizeof(float);
    if (p->hit == 6) p->k.sliver = p->k.taps[0].dy[0] > 0 ? 1 : 0;      // 256-pixel tile: 1 = the taps of the input gradient (filter slabs backwards)
    p->k.fold = (p->nparity == 1 && p->splits == 1 && bf16_fold_tail(d->Cin, d->ks * d->ks)) ? 1 : 0;
    UNET_CHECK_ARG(d->Cout % 4 == 0 || d->y_co + unet::roundup(d->Cout, 4) <= d->y_cs, "conv bf16: the output slice must own its 4-channel padding");
    UNET_CHECK_ARG(unet::aligned16(d->y) && (!d->res || unet::aligned16(d->res)) && (!d->mask || unet::aligned16(d->mask)),
                   "conv bf16: y/res/mask must be 16-byte aligned");
    return UNET_OK;
}

}  // namespace

namespace unetconv {

// plan with split-K when the caller brought a workspace for it, else the plain plan
static int plan_bf16_ws(const unet_conv_desc* d, Plan* p) {
    int rc = plan_bf16(d, p);
    if (rc != UNET_OK) return rc;
    if (!splitk_redirect(d, p)) rc = plan_bf16(d, p, 0);
    return rc;
}

int plan_bf16_public(const unet_conv_desc* d, Plan* p) { return plan_bf16(d, p); }

int conv2d_bf16(const unet_conv_desc* d, hipStream_t st) {
    Plan p;
    int rc = plan_bf16_ws(d, &p);
    if (rc != UNET_OK) return rc;
    if (conv_smallk_applies(d)) return conv_smallk_bf16(d, st);
    if (conv_smallcin_applies(d)) return conv_smallcin_bf16(d, st);
    if (conv_head1x1_applies(d)) return conv_head1x1(d, st);
    if (conv_gemm1x1_applies(d)) return conv_gemm1x1(d, st);
    const int y_f32 = p.splits > 1 ? 1 : d->y_f32;        // partial sums are fp32 slabs
    if (p.hit == 6) rc = launch_t256<unsigned short>(p, y_f32, st);
    else rc = (p.hit == 10) ? launch_tw<10>(p, y_f32, st) : launch_tw<4>(p, y_f32, st);
    if (rc != UNET_OK || p.splits <= 1) return rc;
    return splitk_reduce(d, p, st);
}

// the fp32 launches the planner put on the 256-pixel tile (conv_igemm.hip: unet_conv2d)
int conv2d_t256_f32(const Plan& p0, hipStream_t st) {
    Plan p = p0;
    p.k.sliver = p.k.taps[0].dy[0] > 0 ? 1 : 0;        // (this kernel reads the field as: the taps of the input gradient, filter slabs backwards)
    p.k.fold = 0;
    return launch_t256<float>(p, 1, st);
}

int conv2d_bf16_variant(const unet_conv_desc* d) {
    Plan p;
    int rc = plan_bf16_ws(d, &p);
    if (rc != UNET_OK) return rc;
    if (conv_smallk_applies(d)) return 9;          // conv1x1_smallk_kernel
    if (conv_smallcin_applies(d)) return 10;       // conv3x3_smallcin_kernel
    if (conv_head1x1_applies(d)) return 11;        // conv1x1_head_kernel
    if (conv_gemm1x1_applies(d)) return 8;         // conv1x1_gemm_kernel
    // 256-pixel tile: ...7 = the large layers (128-wide blocks, 32-pixel patches, >= 512 blocks: the launches bench.py's roofline follows), ...6 = its
    // narrow-block / 16-pixel-patch / small-grid launches
    const bool large = p.bm == 256 && p.bn == 128 && p.tw == 32 && (long long)p.k.mtiles * p.k.ntn >= 512;
    return p.tw * 10000 + p.bn * 10 + (p.hit == 10 ? 1 : 0) + (p.bm == 64 ? 5 : 0) + (p.bm == 256 ? (large ? 7 : 6) : 0) + (p.splits > 1 ? 1000000 * p.splits : 0);
}

}  // namespace unetconv

extern "C" size_t unet_pack_weights_size_bf16(int Cout, int Cin, int ks, int mode) {
    const int T = ks * ks;
    const int red = mode == 1 ? Cout : Cin, out = mode == 1 ? Cin : Cout;
    return bf16_image_elems(red, unet::roundup(out, 128), T);
}

extern "C" int unet_pack_weights_strided_bf16(const unet_bf16* w, long long so, long long sr, unet_bf16* wp, int O, int R, void* stream) {
    UNET_CHECK_ARG(w && wp && O > 0 && R > 0, "pack_weights_strided_bf16: bad args");
    const int outPad = unet::roundup(O, 128);
    const size_t total = bf16_image_elems(R, outPad, 1);
    hipLaunchKernelGGL(pack_weights_strided_bf16_kernel, dim3(unet::ew_grid((long long)total, 256)), dim3(256), 0, (hipStream_t)stream, w, so, sr, wp,
                       O, R, outPad, total);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_pack_weights_bf16(const float* w, unet_bf16* wp, int Cout, int Cin, int ks, int mode, void* stream) {
    UNET_CHECK_ARG(w && wp, "pack_weights_bf16: null pointer");
    UNET_CHECK_ARG((ks == 1 || ks == 3) && (mode == 0 || mode == 1 || (mode == 2 && ks == 1 && Cout % 64 == 0)) && Cout > 0 && Cin > 0, "pack_weights_bf16: bad args");
    const int T = ks * ks;
    const int red = mode == 1 ? Cout : Cin, out = mode == 1 ? Cin : Cout;
    const int nchunks = unet::cdiv(red, KCB), outPad = unet::roundup(out, 128);
    const size_t total = bf16_image_elems(red, outPad, T);
    hipLaunchKernelGGL(pack_weights_bf16_kernel, dim3(unet::ew_grid((long long)total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, Cout,
                       Cin, T, mode, nchunks, outPad, total);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
