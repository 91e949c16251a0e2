// Weight gradient of the 3x3 / 1x1 convolutions on fp32 MFMA for gfx950.
//
//   dW[k][c][r][s] = sum_{n,oy,ox} dy[n,oy,ox,k] * x[n, oy*S + r - pad, ox*S + s - pad, c]
//
// GEMM view: M = 64 output channels (k), N = 64 input channels (c), reduction =
// pixels; one accumulator tile per filter tap, all taps kept live (each wave owns
// 32 k x 32 c x T taps = T*16 accumulator VGPRs).  A workgroup walks a contiguous
// range of pixel tiles (split-K over pixels); per tile it stages the dy tile
// [PT pixels][64 k] and the x HALO tile [halo pixels][64 c] into LDS once and all
// taps read shifted windows of the same halo image.  Both MFMA operands are read
// as 32 consecutive floats per half-wave (NHWC rows), i.e. conflict-free ds_read_b32:
//   v_mfma_f32_32x32x2_f32: A[i=l&31][kk=l>>5] = dy[pixel 2*step+kk][k0+i]
//                           B[kk=l>>5][j=l&31] = x[pixel 2*step+kk shifted by tap][c0+j]
// Partials are written [split][tap][k][c] (coalesced) and reduced deterministically
// by a second kernel into the torch layout [Cout][Cin][ks][ks].
//
// Replaces the autograd weight-gradient ATen kernels for fastai ConvLayer convs
// (reference train.py:247-250 -> loss.backward()).

#include <type_traits>
#include <utility>

#include "common.h"

namespace {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{})
template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }


typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BK = 64;  // output channels per block
constexpr int BC = 64;  // input channels per block

struct WArgs {
    const float* x; const float* dy; float* part;
    float* bpart;   // optional [splits][Cout] partial column sums of dy (bias gradient), written by the cblk == 0 workgroups
    int x_cs, x_co, dy_cs, dy_co;
    int N, IH, IW, Cin, Cin4, OH, OW, Cout, Cout4;
    int tiles_y, tiles_x, total_tiles, tiles_per_block;
    int kt, ct;  // number of k / c block tiles
    int cw, nnb; // narrow-output kernel: input-channel chunk width, column blocks per chunk
    int xcd_map; // bf16 kernel: grid.y is a multiple of 8 and the channel blocks of ONE pixel split are dealt to ONE XCD
    int csub, ksub;   // wgrad_kernel (fp32): Cin <= 32 / Cout <= 32 -- the waves whose channel half would be empty take the other HALF OF THE PIXELS
                      // of every tile instead and write a partial image of their own (the reduce kernel sums splits x sub-splits slices)
};

template <int PTW, int S, int KS>
__global__ __launch_bounds__(256, 2) void wgrad_kernel(const WArgs a) {
    constexpr int PT = (S == 1) ? 64 : 32;      // pixels per tile
    constexpr int PTH = PT / PTW;
    constexpr int PAD = (KS - 1) / 2;
    constexpr int T = KS * KS;
    constexpr int HH = (PTH - 1) * S + KS, HW = (PTW - 1) * S + KS, HPIX = HH * HW;
    constexpr int DIT = (PT * 16 + 255) / 256;     // dy float4 items per thread
    constexpr int XIT = (HPIX * 16 + 255) / 256;   // x  float4 items per thread

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dyT = smem;                 // [PT][64]
    float* xh = smem + PT * BK;        // [HPIX][64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // wave = (output-channel half, input-channel half) of the 64 x 64 block.  Narrow layers (the stem: 32 -> 32 pads BOTH to 64, a quarter of
    // the MFMAs useful: 29 TFLOP/s at 16 x 256^2, profiles/r05_a_layer_bench_f32.txt): a half that holds no channels becomes a half of the
    // PIXELS instead (a.csub / a.ksub) -- the wave multiplies the step groups of its pixel sub-split for channel half 0 and writes a partial
    // image of its own.
    const int wk_raw = __builtin_amdgcn_readfirstlane(wave >> 1), wc_raw = __builtin_amdgcn_readfirstlane(wave & 1);
    const int wk = a.ksub ? 0 : wk_raw, wc = a.csub ? 0 : wc_raw;
    const int nsub = (a.ksub ? 2 : 1) * (a.csub ? 2 : 1);
    const int sub = (a.ksub ? wk_raw : 0) * (a.csub ? 2 : 1) + (a.csub ? wc_raw : 0);
    const int l31 = lane & 31, h = lane >> 5;
    const int kblk = blockIdx.x / a.ct, cblk = blockIdx.x % a.ct;
    const int k0 = kblk * BK, c0 = cblk * BC;
    const int split = blockIdx.y;
    const int tile_begin = split * a.tiles_per_block;
    int tile_end = tile_begin + a.tiles_per_block;
    if (tile_end > a.total_tiles) tile_end = a.total_tiles;

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // One pixel tile: global -> registers (load_tile) -> LDS (store_tile).  The NEXT tile is fetched while the current one is
    // multiplied and stored behind it: the loads of a tile are a full L2 / HBM round trip that only the co-resident workgroup used to
    // cover (MFMA pipe busy 81 %).
    float4 rd[DIT], rx[XIT];
    // tile position (uniform) and ONE item of it (compile-time index): the in-loop prefetch issues its DIT + XIT loads one by one
    // between groups of MFMA steps instead of as a block in front of them -- measured by switching the in-loop loads off, the block
    // cost 10-12 % (256 -> 256 at 16 x 128^2: 2.64 ms, 2.33 without), although the data has a whole tile's MFMAs to arrive: a wave
    // issues in order, and thirteen address computations + vector-memory issues in a row are thirteen gaps in its MFMA stream.
    struct TPos { int oy0, ox0; const float* dyb; const float* xb; };
    auto tile_pos = [&](int tile) {
        int b = tile;
        TPos t;
        const int tx = b % a.tiles_x; b /= a.tiles_x;
        const int ty = b % a.tiles_y;
        const int img = b / a.tiles_y;
        t.oy0 = ty * PTH; t.ox0 = tx * PTW;
        t.dyb = a.dy + (size_t)img * a.OH * a.OW * a.dy_cs;
        t.xb = a.x + (size_t)img * a.IH * a.IW * a.x_cs;
        return t;
    };
    auto load_item = [&](const TPos& t, auto idx) {
        constexpr int I = decltype(idx)::value;
        if constexpr (I < DIT) {
            const int e = tid + I * 256;
            const int p = e >> 4, q = e & 15;
            const int oy = t.oy0 + p / PTW, ox = t.ox0 + p % PTW;
            const bool ok = (e < PT * 16) && oy < a.OH && ox < a.OW && (k0 + 4 * q) < a.Cout4;
            // (unconditional load from a clamped address + select: a conditional load is an exec-mask branch around it)
            const float4 v = *reinterpret_cast<const float4*>(t.dyb + (ok ? ((size_t)oy * a.OW + ox) * a.dy_cs + a.dy_co + k0 + 4 * q : (size_t)0));
            rd[I] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        } else if constexpr (I < DIT + XIT) {
            constexpr int J = I - DIT;
            const int e = tid + J * 256;
            const int p = e >> 4, q = e & 15;
            const int iy = t.oy0 * S - PAD + p / HW, ix = t.ox0 * S - PAD + p % HW;
            const bool ok = (e < HPIX * 16) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW && (c0 + 4 * q) < a.Cin4;
            const float4 v = *reinterpret_cast<const float4*>(t.xb + (ok ? ((size_t)iy * a.IW + ix) * a.x_cs + a.x_co + c0 + 4 * q : (size_t)0));
            rx[J] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto load_tile = [&](int tile) {          // all items at once (the first tile of the workgroup)
        const TPos t = tile_pos(tile);
        static_for<DIT + XIT>([&](auto i) { load_item(t, i); });
    };
    // bias gradient = column sums of dy, by the workgroups of the first input-channel block: every thread adds the channel quad of ITS dy
    // items as it stores them (item it of thread tid is always quad tid & 15), the 16 threads of a quad meet once at the end.  (64 threads
    // walking the staged tile, one dependent LDS read per pixel in front of the tile barrier, cost 1-2 % of a 3x3 launch.)
    const bool do_bias = a.bpart != nullptr && cblk == 0;
    float4 bq4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int e = tid + it * 256;
            if (e < PT * 16) *reinterpret_cast<float4*>(dyT + e * 4) = rd[it];
            if (do_bias) { bq4.x += rd[it].x; bq4.y += rd[it].y; bq4.z += rd[it].z; bq4.w += rd[it].w; }     // (items beyond the tile are zero)
        }
#pragma unroll
        for (int j = 0; j < XIT; ++j) {
            const int e = tid + j * 256;
            if (e < HPIX * 16) *reinterpret_cast<float4*>(xh + e * 4) = rx[j];
        }
    };

    // per-lane operand bases (floats)
    const float* abase = dyT + h * BK + wk * 32 + l31;                  // + (2*step) * BK
    const float* bbase = xh + (h * S) * BC + wc * 32 + l31;             // + window offset

    if (tile_begin < tile_end) {
        load_tile(tile_begin);
        store_tile();
    }
    __syncthreads();
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const bool has_next = tile + 1 < tile_end;
        const TPos nx = tile_pos(has_next ? tile + 1 : tile);
        // PT / 2 MFMA steps in groups of GS; in front of group g the prefetch items g * IPG .. are issued (scheduling barriers keep the
        // groups apart: a full unroll would otherwise hoist operand reads until the register file is full)
        constexpr int NG = 8, GS = (PT / 2) / NG, NIT = DIT + XIT, IPG = (NIT + NG - 1) / NG;
        static_assert((PT / 2) % NG == 0, "step groups");
        static_for<NG>([&](auto g) {
            constexpr int G = decltype(g)::value;
            if (has_next) static_for<IPG>([&](auto kk) { load_item(nx, std::integral_constant<int, G * IPG + decltype(kk)::value>{}); });
            if (nsub == 1 || G * nsub / NG == sub) {          // (wave-uniform: the step groups of this wave's pixel sub-split)
#pragma unroll
            for (int sg = 0; sg < GS; ++sg) {
                const int step = G * GS + sg;
                // pixel 2*step + h ; 2*step is even and PTW is even, so px = (2*step % PTW) + h, py = 2*step / PTW
                const int py = (2 * step) / PTW, px = (2 * step) % PTW;
                const float av = abase[(2 * step) * BK];
#pragma unroll
                for (int r = 0; r < KS; ++r)
#pragma unroll
                    for (int s = 0; s < KS; ++s) {
                        const float bv = bbase[((py * S + r) * HW + px * S + s) * BC];
                        acc[r * KS + s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[r * KS + s], 0, 0, 0);
                    }
            }
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        __syncthreads();                  // every wave is done reading this tile
        if (has_next) store_tile();
        __syncthreads();
    }

    if (do_bias) {          // (the staged tiles are dead: [16 item rows][64 channels] floats, summed in row order)
        reinterpret_cast<float4*>(smem)[tid] = bq4;
        __syncthreads();
        if (tid < BK && k0 + tid < a.Cout) {
            float t = 0.f;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) t += smem[rr * 64 + tid];
            a.bpart[(size_t)split * a.Cout + k0 + tid] = t;
        }
    }
    // ---- write partials: part[split][tap][k][c] ----
    const size_t KC_ = (size_t)a.Cout * a.Cin;
    float* pb = a.part + ((size_t)split * nsub + sub) * T * KC_;
    const int c = c0 + wc * 32 + l31;
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + wk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (k < a.Cout && c < a.Cin) pb[(size_t)t * KC_ + (size_t)k * a.Cin + c] = acc[t][r];
        }
}

// Same decomposition on v_mfma_f32_16x16x4_f32: each wave owns 2x2 tiles of 16x16 per tap and skips the tiles that lie
// entirely beyond Cout / Cin, so channel counts are padded to 16 instead of 64 (100x100 costs 49 of 64 tiles, 96x96 36).
//   A[i=l&15][kk=l>>4] = dy[pixel 4*step+kk][k0 + 16*mt + i],  B[kk][j=l&15] = x[pixel shifted by tap][c0 + 16*nt + j]
template <int PTW, int S, int KS>
__global__ __launch_bounds__(256, 2) void wgrad16_kernel(const WArgs a) {
    constexpr int PT = (S == 1) ? 64 : 32;
    constexpr int PTH = PT / PTW;
    constexpr int PAD = (KS - 1) / 2;
    constexpr int T = KS * KS;
    constexpr int HH = (PTH - 1) * S + KS, HW = (PTW - 1) * S + KS, HPIX = HH * HW;
    constexpr int DIT = (PT * 16 + 255) / 256;
    constexpr int XIT = (HPIX * 16 + 255) / 256;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // LDS row stride (floats): the 4 k-lanes (kq) of a 16x16x4 operand read 4 different pixel rows; rows must start
    // 16 banks apart inside a 32-lane group: 80 (== 16 mod 32) for unit pixel stride, 72 for stride-2 inputs (2*72 == 16 mod 32)
    constexpr int LD = (S == 1) ? 80 : 72;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dyT = smem;
    float* xh = smem + PT * LD;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, kq = lane >> 4;
    const int kblk = blockIdx.x / a.ct, cblk = blockIdx.x % a.ct;
    const int k0 = kblk * BK, c0 = cblk * BC;
    const int split = blockIdx.y;
    const int tile_begin = split * a.tiles_per_block;
    int tile_end = tile_begin + a.tiles_per_block;
    if (tile_end > a.total_tiles) tile_end = a.total_tiles;
    // which of this wave's 2x2 16-tiles hold real channels (uniform per wave)
    const bool mv0 = k0 + wk * 32 < a.Cout, mv1 = k0 + wk * 32 + 16 < a.Cout;
    const bool nv0 = c0 + wc * 32 < a.Cin, nv1 = c0 + wc * 32 + 16 < a.Cin;

    f32x4 acc[2][2][T];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[i][j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto stage_tile = [&](int tile) {
        int b = tile;
        const int tx = b % a.tiles_x; b /= a.tiles_x;
        const int ty = b % a.tiles_y;
        const int img = b / a.tiles_y;
        const int oy0 = ty * PTH, ox0 = tx * PTW;
        const float* dyb = a.dy + (size_t)img * a.OH * a.OW * a.dy_cs;
        const float* xb = a.x + (size_t)img * a.IH * a.IW * a.x_cs;
        {
            float4 r[DIT];
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const int e = tid + it * 256;
                const int p = e >> 4, q = e & 15;
                const int oy = oy0 + p / PTW, ox = ox0 + p % PTW;
                const bool ok = (e < PT * 16) && oy < a.OH && ox < a.OW && (k0 + 4 * q) < a.Cout4;
                r[it] = ok ? *reinterpret_cast<const float4*>(dyb + ((size_t)oy * a.OW + ox) * a.dy_cs + a.dy_co + k0 + 4 * q)
                           : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int it = 0; it < DIT; ++it) {
                const int e = tid + it * 256;
                if (e < PT * 16) *reinterpret_cast<float4*>(dyT + (e >> 4) * LD + (e & 15) * 4) = r[it];
            }
        }
        const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
        constexpr int XB = 4;
#pragma unroll
        for (int base = 0; base < XIT; base += XB) {
            float4 r[XB];
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int e = tid + (base + j) * 256;
                const int p = e >> 4, q = e & 15;
                const int iy = iy0 + p / HW, ix = ix0 + p % HW;
                const bool ok = (e < HPIX * 16) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW && (c0 + 4 * q) < a.Cin4;
                r[j] = ok ? *reinterpret_cast<const float4*>(xb + ((size_t)iy * a.IW + ix) * a.x_cs + a.x_co + c0 + 4 * q)
                          : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int j = 0; j < XB; ++j) {
                const int e = tid + (base + j) * 256;
                if (e < HPIX * 16) *reinterpret_cast<float4*>(xh + (e >> 4) * LD + (e & 15) * 4) = r[j];
            }
        }
    };

    // lane kq handles pixel 4*step + kq of the tile: px = (4*step % PTW) + kq, py = 4*step / PTW (PTW is a multiple of 4)
    const float* abase = dyT + kq * LD + wk * 32 + l15;
    const float* bbase = xh + (kq * S) * LD + wc * 32 + l15;

    // one pixel tile: 16 steps x T taps x up to 4 MFMAs; M1/N1 = second 16-row / 16-column tile of this wave is real
    auto compute = [&](auto m1_tag, auto n1_tag) {
        constexpr bool M1 = decltype(m1_tag)::value, N1 = decltype(n1_tag)::value;
#pragma unroll
        for (int step = 0; step < PT / 4; ++step) {
            const int py = (4 * step) / PTW, px = (4 * step) % PTW;
            const float a0 = abase[(4 * step) * LD];
            const float a1 = M1 ? abase[(4 * step) * LD + 16] : 0.f;
#pragma unroll
            for (int r = 0; r < KS; ++r)
#pragma unroll
                for (int s = 0; s < KS; ++s) {
                    const int off = ((py * S + r) * HW + px * S + s) * LD;
                    const int t = r * KS + s;
                    const float b0 = bbase[off];
                    acc[0][0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b0, acc[0][0][t], 0, 0, 0);
                    if constexpr (M1) acc[1][0][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b0, acc[1][0][t], 0, 0, 0);
                    if constexpr (N1) {
                        const float b1 = bbase[off + 16];
                        acc[0][1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, b1, acc[0][1][t], 0, 0, 0);
                        if constexpr (M1) acc[1][1][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, b1, acc[1][1][t], 0, 0, 0);
                    }
                }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    using T1 = std::true_type;
    using T0 = std::false_type;

    const bool do_bias = a.bpart != nullptr && cblk == 0 && tid < BK;
    float bsum = 0.f;
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        stage_tile(tile);
        __syncthreads();
        if (do_bias) {
#pragma unroll 8
            for (int p = 0; p < PT; ++p) bsum += dyT[p * LD + tid];
        }
#ifdef UNET_WGRAD16_NOSKIP
        compute(T1{}, T1{});
#else
        if (mv0 && nv0) {            // (a wave whose first tile is empty has nothing to do)
            if (mv1 && nv1) compute(T1{}, T1{});
            else if (mv1) compute(T1{}, T0{});
            else if (nv1) compute(T0{}, T1{});
            else compute(T0{}, T0{});
        }
#endif
        __syncthreads();
    }

    if (do_bias && k0 + tid < a.Cout) a.bpart[(size_t)split * a.Cout + k0 + tid] = bsum;
    const size_t KC_ = (size_t)a.Cout * a.Cin;
    float* pb = a.part + (size_t)split * T * KC_;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 32 + 16 * j + l15;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = k0 + wk * 32 + 16 * i + 4 * kq + r;
                    if (k < a.Cout && c < a.Cin) pb[(size_t)t * KC_ + (size_t)k * a.Cin + c] = acc[i][j][t][r];
                }
        }
}

// ------------------------------------------------------------------------------------------------
// Narrow-output weight gradient (3x3, stride 1, 80 < Cout <= 112, image width >= 32): the 100->100 pair of the final ResBlock
// and the 96-wide last UnetBlock, where the 64x64-tiled kernel pads both channel counts to 128 (1.64x / 1.78x the MFMAs).
// GEMM view with the filter taps FLATTENED into the N dimension:
//     dW[k][n] = sum_p dy[p][k] * X[p][n],   n = t * CW + c  (tap t, channel c of a <= 112-wide input-channel chunk)
// so that M = Cout pads to 16 and N = 9 * CW pads to 16 ONCE (100x100: 7 x 57 tiles of 16x16 instead of 8 x 8 x 9 / 4).
// A wave owns all KT output-channel tiles x NTW column tiles (35 or 42 f32x4 accumulators), a workgroup of 4 waves
// 4 * NTW consecutive column tiles; grid = (chunks x column blocks, pixel splits).  Pixel tile = one image row of 32 pixels,
// tiles walk down a column strip; dy tile [32][112] and a ring of 3 input rows [3 x 34][112] in LDS (60 KB: two workgroups per CU).  Row stride 112 = 16 mod 32 makes the
// two pixel rows of a 32-lane ds_read_b32 group hit disjoint banks.
//   v_mfma_f32_16x16x4_f32: A[i=l&15][kk=l>>4] = dy[pixel 4*step+kk][16*mt + i]
//                           B[kk=l>>4][j=l&15] = x[pixel 4*step+kk shifted by tap(n)][c(n)],  n = 16*tile + j
// SLV (round 5; Cout = 16 KT + 1..4: the 100 -> 100 pair, 36.9 % of the step's FLOPs): the last 1..4 output channels no longer pay a
// whole 16-row tile per column tile (100 -> 112 rows: 10.7 % of the issued MFMAs multiplied zeros).  They run as a 4-ROW SLIVER on
// v_mfma_f32_4x4x1_16B_f32 (16 blocks of 4 x 4, one k each, 8 clocks instead of 32): block b = lane / 4 = 4 kk + cg is (pixel class kk,
// column group cg), so the B operand IS the register the 16x16x4 tiles of that column tile use (lane 16 kk + 4 cg + j = pixel kk, column
// 4 cg + j), and A is dy[pixel 4*step + kk][16 KT + (lane & 3)].  D: lane l, VGPR r = row 16 KT + r, column l & 15, summed over the pixels
// of class kk = l >> 4; the four classes meet in the epilogue, (kk0 + kk1) + (kk2 + kk3), two lane exchanges.
template <int KT, int NTW, bool SLV = false>
__global__ __launch_bounds__(256, 2) void wgrad_flat_kernel(const WArgs a) {
    constexpr int PT = 32, HW = 34, HPIX = 3 * HW, LD = 112, Q = LD / 4;
    constexpr int DIT = (PT * Q + 255) / 256, XIT = (HPIX * Q + 255) / 256;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dyT = smem;              // [PT][LD]
    float* xh = smem + PT * LD;     // [HPIX][LD]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kk = lane >> 4;
    // XCD-aware block mapping: workgroups are dealt round-robin to the 8 XCDs in launch order, and the column blocks / channel
    // chunks of ONE pixel split stream the same x and dy rows -- put them on the same XCD (ids 8 apart) so that all but the
    // first read hit that XCD's L2.  grid = (cols, splits rounded up to 8).
    const int cols = gridDim.x;
    const int lin = blockIdx.x + cols * blockIdx.y;
    const int xcd = lin & 7, qq = lin >> 3;
    const int bx = qq % cols, split = xcd + 8 * (qq / cols);
    const int chunk = bx / a.nnb, nb = bx % a.nnb;
    const int c0 = chunk * a.cw;
    const int CW = (a.Cin - c0) < a.cw ? (a.Cin - c0) : a.cw;     // channels of this chunk
    const int NTOT = 9 * CW;                                        // real columns
    const int tile_begin = split * a.tiles_per_block;
    int tile_end = tile_begin + a.tiles_per_block;
    if (tile_end > a.total_tiles) tile_end = a.total_tiles;
    if (tile_begin >= a.total_tiles) return;      // padding of the split count to a multiple of 8 (uniform per workgroup)

    // per-lane operand bases (floats): column n -> (tap row r, tap column s, channel c).  The halo tile is a RING of three
    // image rows (row iy lives in slot (iy + 3) % 3): pixel tiles walk DOWN a 32-pixel column strip, so a new tile only
    // fetches the one input row it does not share with its predecessor; the slot of tap row r changes per tile.
    int bfix[NTW], brow[NTW];
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n = ((nb * 4 + wave) * NTW + q) * 16 + l15;
        const int nn = n < NTOT ? n : 0;          // columns beyond 9*CW compute garbage that is never stored
        const int t = nn / CW, c = nn - t * CW;
        const int r = t / 3, s_ = t - 3 * r;
        bfix[q] = (s_ + kk) * LD + c;
        brow[q] = r;
    }
    const int abase = kk * LD + l15;
    const int sbase = kk * LD + 16 * KT + (l15 & 3);          // sliver: row (l & 3) of the last 4 output channels, pixel class kk

    f32x4 acc[KT][NTW];
    f32x4 sacc[SLV ? NTW : 1];
#pragma unroll
    for (int m = 0; m < KT; ++m)
#pragma unroll
        for (int q = 0; q < NTW; ++q) acc[m][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < (SLV ? NTW : 1); ++q) sacc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // Staging (global -> registers -> LDS); pixels outside the image are zero, channels beyond the chunk are never read
    // back.  tile = (img * tiles_x + tx) * OH + oy.  The 35-accumulator form has 32 registers to spare: it fetches the next
    // tile's dy rows and its one new input row BEFORE the MFMA loop of the current tile and stores them after it.
    constexpr bool PF = (KT * NTW + (SLV ? NTW : 0) <= 36);
    constexpr int RIT = (HW * Q + 255) / 256;     // float4 items per thread per input row
    const int cw4 = (CW + 3) & ~3;
    struct TilePos { int oy, ox0; const float* dyb; const float* xb; };
    auto tile_pos = [&](int tile) {
        int b = tile;
        TilePos t;
        t.oy = b % a.OH; b /= a.OH;
        const int tx = b % a.tiles_x;
        const int img = b / a.tiles_x;
        t.ox0 = tx * PT;
        t.dyb = a.dy + ((size_t)img * a.OH + t.oy) * a.OW * a.dy_cs + a.dy_co;
        t.xb = a.x + (size_t)img * a.IH * a.IW * a.x_cs + a.x_co + c0;
        return t;
    };
    auto load_dy = [&](const TilePos& t, float4 (&r)[DIT]) {
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int e = tid + it * 256;
            const int p = e / Q, q = e - p * Q;
            const bool ok = (e < PT * Q) && (t.ox0 + p) < a.OW && 4 * q < a.Cout4;
            r[it] = ok ? *reinterpret_cast<const float4*>(t.dyb + (size_t)(t.ox0 + p) * a.dy_cs + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_dy = [&](const float4 (&r)[DIT]) {
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int e = tid + it * 256;
            if (e < PT * Q) *reinterpret_cast<float4*>(dyT + e * 4) = r[it];
        }
    };
    auto load_row = [&](const TilePos& t, int hy, float4 (&r)[RIT]) {      // input row oy - 1 + hy
        const int iy = t.oy - 1 + hy;
#pragma unroll
        for (int j = 0; j < RIT; ++j) {
            const int e = tid + j * 256;
            const int hx = e / Q, q = e - hx * Q;
            const int ix = t.ox0 - 1 + hx;
            const bool ok = (e < HW * Q) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW && 4 * q < cw4;
            r[j] = ok ? *reinterpret_cast<const float4*>(t.xb + ((size_t)iy * a.IW + ix) * a.x_cs + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_row = [&](const TilePos& t, int hy, const float4 (&r)[RIT]) {
        float* dst = xh + ((t.oy + 2 + hy) % 3) * (HW * LD);               // ring slot of row iy = (iy + 3) % 3
#pragma unroll
        for (int j = 0; j < RIT; ++j) {
            const int e = tid + j * 256;
            if (e < HW * Q) *reinterpret_cast<float4*>(dst + e * 4) = r[j];
        }
    };
    // bias gradient = column sums of dy, by the first workgroup column: every thread adds the channel quads of ITS dy items, as they pass
    // through its registers, to sums of its own in LDS behind the tiles (item it of thread tid is always the same quad; the 42-accumulator
    // form has no registers left for them); all items meet once at the end.  (One thread per output channel walking the staged tile --
    // 32 dependent LDS reads per tile in front of the barrier -- cost 6.5 % of a 100 -> 100 launch.)
    const bool do_bias = a.bpart != nullptr && bx == 0;
    float4* bred = reinterpret_cast<float4*>(smem + (PT + HPIX) * LD) + tid;          // [DIT][256] float4
    if (do_bias) {
#pragma unroll
        for (int it = 0; it < DIT; ++it) bred[it * 256] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    auto bias_add = [&](const float4 (&r)[DIT]) {
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            float4 v = bred[it * 256];
            v.x += r[it].x; v.y += r[it].y; v.z += r[it].z; v.w += r[it].w;
            bred[it * 256] = v;
        }
    };
    auto stage_rows = [&](const TilePos& t, int hy0) {
        float4 rd[DIT];
        load_dy(t, rd);
        if (do_bias) bias_add(rd);
        store_dy(rd);
        for (int hy = hy0; hy < 3; ++hy) {
            float4 rx[RIT];
            load_row(t, hy, rx);
            store_row(t, hy, rx);
        }
    };

    // Prefetch inside a column strip: the next tile is ONE ROW further down, so every item's address is the previous tile's plus a
    // row stride and its column / channel validity does not change.  The per-item offsets are set up once per strip; the in-loop
    // prefetch then costs a few adds per item instead of re-deriving (pixel, channel quad) from the thread index -- measured by
    // switching the in-loop loads off: 7.53 -> 6.60 ms per 100->100 launch, i.e. their ADDRESS ARITHMETIC, not their bytes, cost 12 %.
    // Element offsets inside the image row block (dy: row oy, x: row iy).  An item that is never valid (a channel quad beyond the real
    // channels: 3 of 28 for 100 channels; a pixel beyond the row) keeps the offset of the NEAREST VALID item -- the last real quad of its
    // own pixel, the last pixel of the row -- with the bits inverted (negative = select zero).  Round 2 sent all of them to ONE address
    // (offset 0 of the row / the first pixel of the image): 11 % of all lanes of every load instruction hammered a single cache line and
    // the fabric read traffic of the 100 -> 100 launch rose from 8.1 to 10.0 GB (profiles/r02_b vs r02_a; the 96-channel instantiation,
    // which has no invalid quads, did not move).  The clamped address is one a neighbouring lane loads anyway: no request of its own.
    int dyo[DIT], xo[RIT];
    auto strip_setup = [&](const TilePos& t) {
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const int e = tid + it * 256;
            const int p = e / Q, q = e - p * Q;
            const bool ok = (e < PT * Q) && (t.ox0 + p) < a.OW && 4 * q < a.Cout4;
            const int pc = (t.ox0 + p) < a.OW ? t.ox0 + p : a.OW - 1, qc = 4 * q < a.Cout4 ? 4 * q : a.Cout4 - 4;
            const int off = pc * a.dy_cs + qc;
            dyo[it] = ok ? off : ~off;
        }
#pragma unroll
        for (int j = 0; j < RIT; ++j) {
            const int e = tid + j * 256;
            const int hx = e / Q, q = e - hx * Q;
            const int ix = t.ox0 - 1 + hx;
            const bool ok = (e < HW * Q) && ix >= 0 && ix < a.IW && 4 * q < cw4;
            const int ixc = ix < 0 ? 0 : (ix < a.IW ? ix : a.IW - 1), qc = 4 * q < cw4 ? 4 * q : cw4 - 4;
            const int off = ixc * a.x_cs + qc;
            xo[j] = ok ? off : ~off;
        }
    };
    // the tile one row below `t` (same strip): dy row t.oy + 1 ... = nx.oy, input row nx.oy + 1
    auto prefetch_rolling = [&](const TilePos& nx, float4 (&rdy)[DIT], float4 (&rxr)[RIT]) {
        const float* dyrow = nx.dyb;                                            // already points at image row nx.oy
        const int iy = nx.oy + 1;
        const bool rowok = iy < a.IH;                                           // (iy >= 0 always: nx.oy >= 1)
        const float* xrow = nx.xb + (size_t)(rowok ? iy : a.IH - 1) * a.IW * a.x_cs;       // below the image: the last row's lines, zero selected
        // unconditional loads from a clamped address + select: a conditional load is a branch around it (8 exec-mask branches per tile)
#pragma unroll
        for (int it = 0; it < DIT; ++it) {
            const bool ok = dyo[it] >= 0;
            const float4 v = *reinterpret_cast<const float4*>(dyrow + (ok ? dyo[it] : ~dyo[it]));
            rdy[it] = ok ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int j = 0; j < RIT; ++j) {
            const bool ok = xo[j] >= 0;
            const float4 v = *reinterpret_cast<const float4*>(xrow + (ok ? xo[j] : ~xo[j]));
            rxr[j] = (ok && rowok) ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };

    if (tile_begin < tile_end) {
        const TilePos t0 = tile_pos(tile_begin);
        stage_rows(t0, 0);
        if (PF) strip_setup(t0);
    }
    __syncthreads();
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const int oy = tile % a.OH;
        const bool has_next = tile + 1 < tile_end;
        const bool rolling = has_next && oy + 1 < a.OH;      // the next tile is one row further down the same column strip
        TilePos nx;
        float4 pd[DIT], px[RIT];
        if (has_next) nx = tile_pos(tile + 1);
        if (PF && rolling) {
            prefetch_rolling(nx, pd, px);
            if (do_bias) bias_add(pd);
        }
        int bq[NTW];
#pragma unroll
        for (int q = 0; q < NTW; ++q) bq[q] = bfix[q] + ((oy + 2 + brow[q]) % 3) * (HW * LD);   // tap row r reads input row oy - 1 + r
        // (a full unroll hoists all 96 operand loads and spills; the 42-accumulator form only has room for one step's operands)
        auto mma_step = [&](int step) {
            float av[KT], bv[NTW];
            float as_ = 0.f;
#pragma unroll
            for (int m = 0; m < KT; ++m) av[m] = dyT[abase + step * 4 * LD + m * 16];
            if constexpr (SLV) as_ = dyT[sbase + step * 4 * LD];
#pragma unroll
            for (int q = 0; q < NTW; ++q) bv[q] = xh[bq[q] + step * 4 * LD];
#pragma unroll
            for (int m = 0; m < KT; ++m)
#pragma unroll
                for (int q = 0; q < NTW; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[q], acc[m][q], 0, 0, 0);
            if constexpr (SLV) {
#pragma unroll
                for (int q = 0; q < NTW; ++q) sacc[q] = __builtin_amdgcn_mfma_f32_4x4x1f32(as_, bv[q], sacc[q], 0, 0, 0);
            }
        };
        if constexpr (!PF) {
#pragma unroll 1
            for (int step = 0; step < PT / 4; ++step) mma_step(step);
        } else {
#pragma unroll 2
            for (int step = 0; step < PT / 4; ++step) mma_step(step);
        }
        if (has_next) {
            __syncthreads();                 // everyone is done reading this tile (the new row replaces row oy - 1)
            if (PF && rolling) { store_dy(pd); store_row(nx, 2, px); }
            else {
                stage_rows(nx, rolling ? 2 : 0);
                if (PF) strip_setup(nx);          // a new column strip begins
            }
            __syncthreads();
        }
    }

    if (do_bias) {          // channel c adds the item sums of its quad in item order
        __syncthreads();
        const float* red = smem + (PT + HPIX) * LD;
        if (tid < a.Cout) {
            float t = 0.f;
            for (int e = tid >> 2; e < PT * Q; e += Q) t += red[e * 4 + (tid & 3)];
            a.bpart[(size_t)split * a.Cout + tid] = t;
        }
    }
    // ---- write partials: part[split][tap][k][c] ----
    const size_t KC_ = (size_t)a.Cout * a.Cin;
    float* pb = a.part + (size_t)split * 9 * KC_;
#pragma unroll
    for (int q = 0; q < NTW; ++q) {
        const int n = ((nb * 4 + wave) * NTW + q) * 16 + l15;
        if (n >= NTOT) continue;
        const int t = n / CW, c = n - t * CW;
        float* pt = pb + (size_t)t * KC_ + c0 + c;
#pragma unroll
        for (int m = 0; m < KT; ++m)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = 16 * m + 4 * kk + r;
                if (k < a.Cout) pt[(size_t)k * a.Cin] = acc[m][q][r];
            }
    }
    if constexpr (SLV) {
#pragma unroll
        for (int q = 0; q < NTW; ++q) {
            f32x4 v = sacc[q];
#pragma unroll
            for (int r = 0; r < 4; ++r) {          // the four pixel classes of a column: (kk0 + kk1) + (kk2 + kk3), the same order in every lane
                v[r] += __shfl_xor(v[r], 16);
                v[r] += __shfl_xor(v[r], 32);
            }
            const int n = ((nb * 4 + wave) * NTW + q) * 16 + l15;
            if (n >= NTOT || kk != 0) continue;
            const int t = n / CW, c = n - t * CW;
            float* pt = pb + (size_t)t * KC_ + c0 + c;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (16 * KT + r < a.Cout) pt[(size_t)(16 * KT + r) * a.Cin] = v[r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16-storage weight gradient: x and dy are bf16 NHWC, products are accumulated in fp32 (v_mfma_f32_16x16x32_bf16), partials and
// dW are fp32.  The reduction runs over PIXELS, which are the strided dimension of NHWC: both MFMA operands are K-major reads of a
// row-major [pixel][channel] LDS image -- exactly what ds_read_b64_tr_b16 delivers (a 4-pixel x 16-channel block, column-major,
// per 16-lane group).  MFMA k index -> pixel: lane group g reads pixels {4g..4g+3} and {16+4g..16+4g+3} of the 32-pixel tile
// (the same map for both operands), so that the two groups of a 32-lane half touch 8 CONSECUTIVE pixel rows: with a row stride
// of 160 bytes (stride-2 inputs: 144) they fall into 8 disjoint 32-byte bank windows -- conflict-free.
//   A[row = k][kk] = dy[pixel(kk)][k0 + 16 mt + row]     B[kk][col = c] = x[pixel(kk) shifted by tap][c0 + 16 nt + col]
//   D[row = 4(l>>4) + r][col = l&15] -> dW partial [tap][k][c]
// Workgroup = 64 output x 64 input channels x all taps (each wave 2 x 2 tiles of 16 x 16 per tap: 144 accumulator VGPRs for 3x3),
// split-K over pixel tiles, second-stage sum by wgrad_reduce_kernel as in the fp32 path.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16;

__device__ __forceinline__ bf16x8 ld_tr_pair(const char* p0, const char* p1) {
    typedef __attribute__((address_space(3))) s16x4* lds_p;
    const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p0));
    const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(p1));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

template <int PTW, int S, int KS>
__global__ __launch_bounds__(256, 2) void wgrad_bf16_kernel(const WArgs a) {
    // pixel tile = PT pixels = KB MFMA k-blocks of 32 (a PTH x PTW patch); the next tile is fetched global -> registers while the
    // current one is multiplied, and stored to LDS behind it (one barrier pair per KB * T * 4 MFMAs per wave)
    constexpr int KB = (S == 1) ? 4 : 1, PT = 32 * KB, PTH = PT / PTW, PAD = (KS - 1) / 2, T = KS * KS;     // (stride 2: the halo of 128 pixels would not fit the prefetch registers)
    constexpr int HH = (PTH - 1) * S + KS, HW = (PTW - 1) * S + KS, HPIX = HH * HW;
    constexpr int RSD = 160, RSX = (S == 1) ? 160 : 144;         // LDS row strides in bytes (64 channels = 128 bytes + pad)
    constexpr int DIT = PT * 8 / 256;                            // 16-byte dy items per thread
    constexpr int XIT = (HPIX * 8 + 255) / 256;                  // 16-byte x items per thread
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smemb[];
    char* dyT = smemb;                      // [PT][RSD]
    char* xh = smemb + PT * RSD;            // [HPIX][RSX]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave >> 1, wc = wave & 1;
    const int l15 = lane & 15, g = lane >> 4, q = l15 >> 2, pp = l15 & 3;
    // The kt x ct channel blocks of one pixel split stream the same x and dy rows.  Workgroups are dealt round-robin to the 8 XCDs in
    // launch order, so with the plain (x = block, y = split) numbering every XCD's L2 fetched every slab for itself: 0.82 GB per launch
    // for 0.27 GB of operands at 4.3 TB/s -- the kernel ran at the L2-miss bandwidth, not at the MFMA rate (profiles/r02_b_bf16).  With
    // xcd_map the blocks of split s all land on XCD s % 8 (linear ids 8 apart): one fetch per slab and XCD.
    int bx = blockIdx.x, split = blockIdx.y;
    if (a.xcd_map) {
        const int cols = gridDim.x, lin = blockIdx.x + cols * blockIdx.y;
        const int qq = lin >> 3;
        bx = qq % cols;
        split = (lin & 7) + 8 * (qq / cols);
    }
    const int kblk = bx / a.ct, cblk = bx % a.ct;
    const int k0 = kblk * BK, c0 = cblk * BC;
    const int tile_begin = split * a.tiles_per_block;
    int tile_end = tile_begin + a.tiles_per_block;
    if (tile_end > a.total_tiles) tile_end = a.total_tiles;
    if (tile_begin >= a.total_tiles) return;          // padding of the split count to a multiple of 8 (uniform per workgroup)

    f32x4 acc[2][2][T];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < T; ++t) acc[i][j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const u16* xg = reinterpret_cast<const u16*>(a.x);
    const u16* dyg = reinterpret_cast<const u16*>(a.dy);
    uint4 rd[DIT], rx[XIT];
    auto load_tile = [&](int tile) {
        int b = tile;
        const int tx = b % a.tiles_x; b /= a.tiles_x;
        const int ty = b % a.tiles_y;
        const int img = b / a.tiles_y;
        const int oy0 = ty * PTH, ox0 = tx * PTW;
        const u16* dyb = dyg + (size_t)img * a.OH * a.OW * a.dy_cs;
        const u16* xb = xg + (size_t)img * a.IH * a.IW * a.x_cs;
#pragma unroll
        for (int j = 0; j < DIT; ++j) {
            const int e = tid + j * 256;
            const int p = e >> 3, qq = e & 7;
            const int oy = oy0 + p / PTW, ox = ox0 + p % PTW;
            const bool ok = oy < a.OH && ox < a.OW && (k0 + 8 * qq) < a.Cout4;
            rd[j] = ok ? *reinterpret_cast<const uint4*>(dyb + ((size_t)oy * a.OW + ox) * a.dy_cs + a.dy_co + k0 + 8 * qq) : make_uint4(0u, 0u, 0u, 0u);
        }
        const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
#pragma unroll
        for (int j = 0; j < XIT; ++j) {
            const int e = tid + j * 256;
            const int p = e >> 3, qq = e & 7;
            const int iy = iy0 + p / HW, ix = ix0 + p % HW;
            const bool ok = (e < HPIX * 8) && iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW && (c0 + 8 * qq) < a.Cin4;
            rx[j] = ok ? *reinterpret_cast<const uint4*>(xb + ((size_t)iy * a.IW + ix) * a.x_cs + a.x_co + c0 + 8 * qq) : make_uint4(0u, 0u, 0u, 0u);
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < DIT; ++j) {
            const int e = tid + j * 256;
            *reinterpret_cast<uint4*>(dyT + (e >> 3) * RSD + (e & 7) * 16) = rd[j];
        }
#pragma unroll
        for (int j = 0; j < XIT; ++j) {
            const int e = tid + j * 256;
            if (e < HPIX * 8) *reinterpret_cast<uint4*>(xh + (e >> 3) * RSX + (e & 7) * 16) = rx[j];
        }
    };

    // per-lane transposed-read addresses inside k-block 0: this lane supplies pixel row P1 = 4g + q (first read) / P2 = 16 + 4g + q
    // (second), columns 4pp..4pp+3; k-block kb adds 32 pixels
    const int P1 = 4 * g + q, P2 = 16 + 4 * g + q;
    const char* a1 = dyT + P1 * RSD + (wk * 32 + 4 * pp) * 2;
    const char* a2 = dyT + P2 * RSD + (wk * 32 + 4 * pp) * 2;
    int xr1[KB], xr2[KB];
#pragma unroll
    for (int kb = 0; kb < KB; ++kb) {
        const int p1 = 32 * kb + P1, p2 = 32 * kb + P2;
        xr1[kb] = ((p1 / PTW) * S * HW + (p1 % PTW) * S) * RSX + (wc * 32 + 4 * pp) * 2;
        xr2[kb] = ((p2 / PTW) * S * HW + (p2 % PTW) * S) * RSX + (wc * 32 + 4 * pp) * 2;
    }
    // which of this wave's 2 x 2 tiles hold real channels (wave-uniform)
    const bool mv1 = k0 + wk * 32 + 16 < a.Cout, nv1 = c0 + wc * 32 + 16 < a.Cin;
    const bool mv0 = k0 + wk * 32 < a.Cout, nv0 = c0 + wc * 32 < a.Cin;

    // bias gradient = column sums of dy: every thread adds up the 8 channels of its own dy items as they pass through its registers, the 32
    // threads of a channel group meet once at the end (see wgrad_bf16_k4_kernel: a serial walk of the staged tile cost a 1x1 tile's MFMAs)
    const bool do_bias = a.bpart != nullptr && cblk == 0;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto bias_add = [&]() {
#pragma unroll
        for (int j = 0; j < DIT; ++j) {
            const unsigned w[4] = {rd[j].x, rd[j].y, rd[j].z, rd[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { bs[2 * i] += __uint_as_float(w[i] << 16); bs[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
        }
    };
    if (tile_begin < tile_end) {
        load_tile(tile_begin);
        if (do_bias) bias_add();
        store_tile();
    }
    __syncthreads();
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const bool has_next = tile + 1 < tile_end;
        if (has_next) load_tile(tile + 1);
        if (mv0 && nv0) {
#pragma unroll
            for (int kb = 0; kb < KB; ++kb) {
                const bf16x8 av0 = ld_tr_pair(a1 + kb * 32 * RSD, a2 + kb * 32 * RSD);
                const bf16x8 av1 = ld_tr_pair(a1 + kb * 32 * RSD + 32, a2 + kb * 32 * RSD + 32);
#pragma unroll
                for (int r = 0; r < KS; ++r)
#pragma unroll
                    for (int s_ = 0; s_ < KS; ++s_) {
                        const int t = r * KS + s_;
                        const int off = (r * HW + s_) * RSX;
                        const bf16x8 bv0 = ld_tr_pair(xh + xr1[kb] + off, xh + xr2[kb] + off);
                        acc[0][0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av0, bv0, acc[0][0][t], 0, 0, 0);
                        if (mv1) acc[1][0][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av1, bv0, acc[1][0][t], 0, 0, 0);
                        if (nv1) {
                            const bf16x8 bv1 = ld_tr_pair(xh + xr1[kb] + off + 32, xh + xr2[kb] + off + 32);
                            acc[0][1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av0, bv1, acc[0][1][t], 0, 0, 0);
                            if (mv1) acc[1][1][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av1, bv1, acc[1][1][t], 0, 0, 0);
                        }
                    }
            }
        }
        __syncthreads();                  // every wave is done reading this tile
        if (has_next) {
            if (do_bias) bias_add();
            store_tile();
        }
        __syncthreads();
    }

    if (do_bias) {          // (the images are dead: [32 rows of the item map][64 channels] floats, summed in row order)
        float* red = reinterpret_cast<float*>(smemb);
#pragma unroll
        for (int i = 0; i < 8; ++i) red[(tid >> 3) * 64 + (tid & 7) * 8 + i] = bs[i];
        __syncthreads();
        if (tid < BK && k0 + tid < a.Cout) {
            float t = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) t += red[rr * 64 + tid];
            a.bpart[(size_t)split * a.Cout + k0 + tid] = t;
        }
    }
    const size_t KC_ = (size_t)a.Cout * a.Cin;
    float* pb = a.part + (size_t)split * T * KC_;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 32 + 16 * j + l15;
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = k0 + wk * 32 + 16 * i + 4 * g + r;
                    if (k < a.Cout && c < a.Cin) pb[(size_t)t * KC_ + (size_t)k * a.Cin + c] = acc[i][j][t][r];
                }
        }
}

// ------------------------------------------------------------------------------------------------
// The same bf16 weight gradient for the shapes that carry the step (3x3 and 1x1, stride 1, 32-pixel-wide tiles), with the two things the
// ISA of wgrad_bf16_kernel<32,1,3> showed to bind it taken out (per 128-pixel tile and wave: 144 MFMAs next to 179 LDS and ~730
// scalar / vector / branch instructions -- ~4200 issue clocks against 2304 MFMA clocks, two waves per SIMD):
//   * wave decomposition KV x 1 instead of 2 x 2 tiles: a wave owns ONE 16-wide input-channel tile and all (up to four) 16-wide
//     output-channel tiles of the block.  The dy fragments of a k-block serve all nine taps (8 transposed reads per 36 MFMAs), each x fragment four
//     MFMAs: 26 LDS reads per 36 MFMAs instead of 40 -- the 2 x 2 form needs 139 B/clk/CU of LDS bandwidth at full MFMA rate, more
//     than the array delivers;
//   * no address arithmetic in the loops: every LDS operand address is one lane register + an immediate; the global loads are
//     buffer loads whose per-lane offsets are computed ONCE (tile-invariant part, the two column-edge flags in the free low bits),
//     a tile adds its origin (4 VALU per 16-byte item; rows outside the image leave the descriptor's range and read as zero);
//   * the output-channel block is KV x 16 channels wide, KV = 1..4 chosen per launch so that Cout is padded least (96 = 2 x 48,
//     128 = 2 x 64, 100 -> 2 x 64): one branch-free body per instantiation (two bodies in one kernel made the allocator spill).
// Same split-K over pixel tiles and the same partial layout as wgrad_bf16_kernel: a drop-in for <32, 1, 3>.
template <int KV, int KS>       // KV: 16-wide output-channel tiles per block (a.kt blocks of 16 KV channels: the launcher picks the KV that pads Cout least); KS: 3 | 1
__global__ __launch_bounds__(256, 2) void wgrad_bf16_k4_kernel(const WArgs a) {
    constexpr int PTW = 32, PTH = 4, PT = 128, KB = 4, T = KS * KS, PAD = (KS - 1) / 2, HH = PTH + 2 * PAD, HW = PTW + 2 * PAD, HPIX = HH * HW;
    constexpr int RS = 160;                                      // LDS row stride in bytes (64 channels = 128 bytes + pad), both images
    constexpr int DIT = PT * 8 / 256, XIT = (HPIX * 8 + 255) / 256;    // 16-byte items per thread: 4 of dy, 7 of x
    constexpr unsigned OOB = 0x80000000u;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) char smemb[];
    char* dyT = smemb;                      // [PT][RS]
    char* xh = smemb + PT * RS;             // [HPIX][RS]

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, g = lane >> 4, q = l15 >> 2, pp = l15 & 3;
    int bx = blockIdx.x, split = blockIdx.y;
    if (a.xcd_map) {                        // the channel blocks of one pixel split on one XCD (see wgrad_bf16_kernel)
        const int cols = gridDim.x, lin = blockIdx.x + cols * blockIdx.y;
        const int qq = lin >> 3;
        bx = qq % cols;
        split = (lin & 7) + 8 * (qq / cols);
    }
    const int kblk = bx / a.ct, cblk = bx % a.ct;
    const int k0 = kblk * (16 * KV), c0 = cblk * BC;
    const int tile_begin = split * a.tiles_per_block;
    int tile_end = tile_begin + a.tiles_per_block;
    if (tile_end > a.total_tiles) tile_end = a.total_tiles;
    if (tile_begin >= a.total_tiles) return;          // padding of the split count to a multiple of 8 (uniform per workgroup)
    const bool cvw = c0 + 16 * wave < a.Cin;                                         // this wave's input-channel tile holds real channels

    f32x4 acc[KV][T];
#pragma unroll
    for (int i = 0; i < KV; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[i][t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // tile-invariant byte offsets of this thread's items relative to the tile origin; bit 0: the item lies in the tile's first halo
    // column (outside the image for the first tile of a row), bit 1: at or beyond the image's last column in the LAST tile of a row
    // (kept in LDS behind the two images, [item][thread]: 11 registers that the accumulators need)
    unsigned* rel = reinterpret_cast<unsigned*>(smemb + PT * RS + HPIX * RS) + tid;
#define RELD(j_) rel[(j_) * 256]
#define RELX(j_) rel[(DIT + (j_)) * 256]
    const int last_ox0 = (a.tiles_x - 1) * PTW;
#pragma unroll
    for (int j = 0; j < DIT; ++j) {
        const int e = tid + j * 256;
        const int p = e >> 3, qq = e & 7;
        const bool chan = 8 * qq < 16 * KV && (k0 + 8 * qq) < a.Cout4;
        const unsigned off = (unsigned)(((p / PTW) * a.OW + p % PTW) * a.dy_cs + a.dy_co + k0 + 8 * qq) * 2u;
        RELD(j) = chan ? (off | (last_ox0 + p % PTW >= a.OW ? 2u : 0u)) : OOB;
    }
#pragma unroll
    for (int j = 0; j < XIT; ++j) {
        const int e = tid + j * 256;
        const int p = e >> 3, qq = e & 7;
        const int hx = p % HW;
        const bool chan = (e < HPIX * 8) && (c0 + 8 * qq) < a.Cin4;
        const unsigned off = (unsigned)(((p / HW) * a.IW + hx) * a.x_cs + a.x_co + c0 + 8 * qq) * 2u;
        RELX(j) = chan ? (off | (hx < PAD ? 1u : 0u) | (last_ox0 - PAD + hx >= a.IW ? 2u : 0u)) : OOB;
    }
    uint4 rd[DIT], rx[XIT];
    const unsigned dy_img_b = (unsigned)(a.OH * a.OW * a.dy_cs) * 2u, x_img_b = (unsigned)(a.IH * a.IW * a.x_cs) * 2u;
    auto load_tile = [&](int tile) {
        int b = tile;
        const int tx = __builtin_amdgcn_readfirstlane(b % a.tiles_x); b /= a.tiles_x;
        const int ty = __builtin_amdgcn_readfirstlane(b % a.tiles_y);
        const int img = __builtin_amdgcn_readfirstlane(b / a.tiles_y);
        const int oy0 = ty * PTH, ox0 = tx * PTW;
        const unsigned edge = (tx == 0 ? 1u : 0u) | (tx == a.tiles_x - 1 ? 2u : 0u);
        u16* dyb = const_cast<u16*>(reinterpret_cast<const u16*>(a.dy)) + (size_t)img * a.OH * a.OW * a.dy_cs;
        u16* xb = const_cast<u16*>(reinterpret_cast<const u16*>(a.x)) + (size_t)img * a.IH * a.IW * a.x_cs;
        const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(dyb, 0, (int)dy_img_b, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(xb, 0, (int)x_img_b, 0x00020000);
        const unsigned based = (unsigned)((oy0 * a.OW + ox0) * a.dy_cs) * 2u;
        const unsigned basex = (unsigned)(((oy0 - PAD) * a.IW + ox0 - PAD) * a.x_cs) * 2u;      // (first tile row / column: wraps far out of range)
#pragma unroll
        for (int j = 0; j < DIT; ++j) {
            const unsigned rl = RELD(j);
            const unsigned vo = (rl & (edge & 2u)) ? OOB : ((rl & ~15u) + based) | (rl & OOB);
            rd[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsd, (int)vo, 0, 0));
        }
#pragma unroll
        for (int j = 0; j < XIT; ++j) {
            const unsigned rl = RELX(j);
            const unsigned vo = (rl & edge) ? OOB : ((rl & ~15u) + basex) | (rl & OOB);
            rx[j] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rsx, (int)vo, 0, 0));
        }
    };
    char* std_ = dyT + (tid >> 3) * RS + (tid & 7) * 16;
    char* stx_ = xh + (tid >> 3) * RS + (tid & 7) * 16;
    const bool xlast = tid + (XIT - 1) * 256 < HPIX * 8;
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < DIT; ++j) *reinterpret_cast<uint4*>(std_ + j * 32 * RS) = rd[j];
#pragma unroll
        for (int j = 0; j < XIT; ++j)
            if (j < XIT - 1 || xlast) *reinterpret_cast<uint4*>(stx_ + j * 32 * RS) = rx[j];
    };

    // per-lane transposed-read addresses inside k-block 0: this lane supplies pixel row P1 = 4g + q (first read) / P2 = 16 + 4g + q
    // (second), columns 4pp..4pp+3 of its tile; k-block kb = tile row kb: + kb * 32 rows of dy, + kb * HW rows of x (immediates)
    const int P1 = 4 * g + q, P2 = 16 + 4 * g + q;
    const char* a1 = dyT + P1 * RS + (4 * pp) * 2;
    const char* a2 = dyT + P2 * RS + (4 * pp) * 2;
    const char* x1 = xh + P1 * RS + (wave * 16 + 4 * pp) * 2;
    const char* x2 = xh + P2 * RS + (wave * 16 + 4 * pp) * 2;

    // bias gradient = column sums of dy, by the workgroups of the first input-channel block: every thread adds up the 8 channels of ITS
    // 16-byte dy items as they pass through its registers (4 pixels per tile), the 32 threads of a channel group meet once at the end.
    // (A serial walk of the staged tile by 64 threads -- 128 dependent LDS reads per tile -- cost more than the tile's MFMAs in the
    // 1x1 launches: 96 -> 384 at 256 x 256 took 1.0 ms in the step against 0.29 ms without the bias.)
    const bool do_bias = a.bpart != nullptr && cblk == 0;
    float bs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto bias_add = [&]() {
#pragma unroll
        for (int j = 0; j < DIT; ++j) {
            const unsigned w[4] = {rd[j].x, rd[j].y, rd[j].z, rd[j].w};
#pragma unroll
            for (int i = 0; i < 4; ++i) { bs[2 * i] += __uint_as_float(w[i] << 16); bs[2 * i + 1] += __uint_as_float(w[i] & 0xffff0000u); }
        }
    };
    load_tile(tile_begin);
    if (do_bias) bias_add();
    store_tile();
    __syncthreads();
#define K4_BODY(KV_) do { \
        _Pragma("nounroll") for (int kb = 0; kb < KB; ++kb) { \
            bf16x8 av_[KV_]; \
            _Pragma("unroll") for (int i = 0; i < KV_; ++i) av_[i] = ld_tr_pair(a1 + kb * 32 * RS + 32 * i, a2 + kb * 32 * RS + 32 * i); \
            _Pragma("unroll") for (int r = 0; r < KS; ++r) \
                _Pragma("unroll") for (int s_ = 0; s_ < KS; ++s_) { \
                    const bf16x8 bv_ = ld_tr_pair(x1 + (kb * HW + r * HW + s_) * RS, x2 + (kb * HW + r * HW + s_) * RS); \
                    _Pragma("unroll") for (int i = 0; i < KV_; ++i) \
                        acc[i][r * KS + s_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av_[i], bv_, acc[i][r * KS + s_], 0, 0, 0); \
                } \
        } } while (0)
    for (int tile = tile_begin; tile < tile_end; ++tile) {
        const bool has_next = tile + 1 < tile_end;
        if (has_next) load_tile(tile + 1);
        if (cvw) {
            K4_BODY(KV);
        }
        __syncthreads();                  // every wave is done reading this tile
        if (has_next) {
            if (do_bias) bias_add();
            store_tile();
        }
        __syncthreads();
    }
#undef K4_BODY
#undef RELD
#undef RELX

    if (do_bias) {          // (the images are dead: [32 pixel rows of the item map][64 channels] floats, summed in row order)
        float* red = reinterpret_cast<float*>(smemb);
#pragma unroll
        for (int i = 0; i < 8; ++i) red[(tid >> 3) * 64 + (tid & 7) * 8 + i] = bs[i];
        __syncthreads();
        if (tid < 16 * KV && k0 + tid < a.Cout) {
            float t = 0.f;
#pragma unroll 8
            for (int rr = 0; rr < 32; ++rr) t += red[rr * 64 + tid];
            a.bpart[(size_t)split * a.Cout + k0 + tid] = t;
        }
    }
    const size_t KC_ = (size_t)a.Cout * a.Cin;
    float* pb = a.part + (size_t)split * T * KC_;
    const int c = c0 + wave * 16 + l15;
#pragma unroll
    for (int i = 0; i < KV; ++i)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = k0 + 16 * i + 4 * g + r;
                if (k < a.Cout && c < a.Cin) pb[(size_t)t * KC_ + (size_t)k * a.Cin + c] = acc[i][t][r];
            }
}

// ------------------------------------------------------------------------------------------------
// 1x1 weight gradient = a plain GEMM dW[k][c] = sum_p dy[p][k] x[p][c] over FLAT pixels (PixelShuffle_ICNR convs,
// identity-path convs, head, self-attention products).  The 64x64-tiled kernel stages 512 B per pixel for 8 kFLOP
// (L2->LDS bound, ~46 TFLOP/s); here a workgroup owns 128 x 128 channels (each wave 64 x 64 = 2x2 MFMA tiles, 64
// accumulator VGPRs), i.e. twice the FLOPs per staged byte, with the next pixel tile prefetched into registers.
__global__ __launch_bounds__(256, 2) void wgrad1x1_kernel(const WArgs a, long long P) {
    constexpr int PT = 64, LDW = 128, IT = PT * 32 / 256;     // 8 float4 per thread per operand
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* dyT = smem;                 // [PT][128]
    float* xT = smem + PT * LDW;       // [PT][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wk = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, h = lane >> 5;
    const int kblk = blockIdx.x / a.ct, cblk = blockIdx.x % a.ct;
    const int k0 = kblk * 128, c0 = cblk * 128;
    const int split = blockIdx.y;
    const long long tile_begin = (long long)split * a.tiles_per_block;
    long long tile_end = tile_begin + a.tiles_per_block;
    if (tile_end > a.total_tiles) tile_end = a.total_tiles;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float4 rd[IT], rx[IT];
    auto load_tile = [&](long long tile) {
        const long long p0 = tile * PT;
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int e = tid + it * 256;
            const int p = e >> 5, q = e & 31;
            const long long pp = p0 + p;
            const bool okp = pp < P;
            rd[it] = (okp && (k0 + 4 * q) < a.Cout4) ? *reinterpret_cast<const float4*>(a.dy + (size_t)pp * a.dy_cs + a.dy_co + k0 + 4 * q)
                                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            rx[it] = (okp && (c0 + 4 * q) < a.Cin4) ? *reinterpret_cast<const float4*>(a.x + (size_t)pp * a.x_cs + a.x_co + c0 + 4 * q)
                                                     : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    // bias gradient = column sums of dy (first input-channel block): every thread adds the channel quad of its dy items (always quad
    // tid & 31) as it stores them; the 8 threads of a quad meet once at the end (see wgrad_kernel)
    const bool do_bias = a.bpart != nullptr && cblk == 0;
    float4 bq4 = make_float4(0.f, 0.f, 0.f, 0.f);
    auto store_tile = [&]() {
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int e = tid + it * 256;
            *reinterpret_cast<float4*>(dyT + e * 4) = rd[it];
            *reinterpret_cast<float4*>(xT + e * 4) = rx[it];
            if (do_bias) { bq4.x += rd[it].x; bq4.y += rd[it].y; bq4.z += rd[it].z; bq4.w += rd[it].w; }
        }
    };

    const float* abase = dyT + h * LDW + wk * 64 + l31;
    const float* bbase = xT + h * LDW + wc * 64 + l31;

    if (tile_begin < tile_end) {
        load_tile(tile_begin);
        store_tile();
    }
    __syncthreads();
    for (long long tile = tile_begin; tile < tile_end; ++tile) {
        const bool has_next = tile + 1 < tile_end;
        if (has_next) load_tile(tile + 1);
#pragma unroll
        for (int step = 0; step < PT / 2; ++step) {
            const float a0 = abase[(2 * step) * LDW], a1 = abase[(2 * step) * LDW + 32];
            const float b0 = bbase[(2 * step) * LDW], b1 = bbase[(2 * step) * LDW + 32];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
        if (has_next) store_tile();
        __syncthreads();
    }

    if (do_bias) {          // (the staged tiles are dead: [8 item rows][128 channels] floats, summed in row order)
        reinterpret_cast<float4*>(smem)[tid] = bq4;
        __syncthreads();
        if (tid < 128 && k0 + tid < a.Cout) {
            float t = 0.f;
#pragma unroll
            for (int rr = 0; rr < 8; ++rr) t += smem[rr * 128 + tid];
            a.bpart[(size_t)split * a.Cout + k0 + tid] = t;
        }
    }
    const size_t KC_ = (size_t)a.Cout * a.Cin;
    float* pb = a.part + (size_t)split * KC_;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + wc * 64 + 32 * j + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = k0 + wk * 64 + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (k < a.Cout && c < a.Cin) pb[(size_t)k * a.Cin + c] = acc[i][j][r];
            }
        }
}

// ------------------------------------------------------------------------------------------------
// 1x1 weight gradient with a handful of output channels (the classification / regression head: 100 -> n_classes <= 16): a
// 128 x 128 MFMA block would be 96 % padding and is HBM-bound anyway (x is read once, 1.7 GB at 16 x 512^2 x 100).  Plain
// FMA: a thread owns 4 input channels (one float4 of x per pixel) x all K output channels; the PS pixel lanes of a workgroup
// and the workgroups write separate partial rows [row][k][c] (no atomics, no LDS), summed by wgrad_reduce_kernel.
template <int K4>   // ceil(Cout / 4): 1..4
__global__ __launch_bounds__(256) void wgrad1x1_small_kernel(const WArgs a, long long P, int PS, long long pix_per_block) {
    extern __shared__ __attribute__((aligned(16))) float smem[];      // [PS][KK * Cin4 + KK]: lane partials [k][c] + bias sums [k]
    const int C4 = a.Cin4 >> 2;
    const int tid = threadIdx.x;
    const int q = tid % C4, lane = tid / C4;
    const bool active = lane < PS;
    const long long p_begin = (long long)blockIdx.x * pix_per_block;
    long long p_end = p_begin + pix_per_block;
    if (p_end > P) p_end = P;
    float acc[4 * K4][4];
    float bsum[4 * K4];
#pragma unroll
    for (int k = 0; k < 4 * K4; ++k) {
        bsum[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[k][j] = 0.f;
    }
    if (active) {
        for (long long p = p_begin + lane; p < p_end; p += PS) {
            const float4 xv = *reinterpret_cast<const float4*>(a.x + (size_t)p * a.x_cs + a.x_co + 4 * q);
            const float* dyp = a.dy + (size_t)p * a.dy_cs + a.dy_co;
#pragma unroll
            for (int k4 = 0; k4 < K4; ++k4) {
                const float4 d = (4 * k4 < a.Cout4) ? *reinterpret_cast<const float4*>(dyp + 4 * k4) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float dv[4] = {d.x, d.y, d.z, d.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    acc[4 * k4 + i][0] += dv[i] * xv.x; acc[4 * k4 + i][1] += dv[i] * xv.y;
                    acc[4 * k4 + i][2] += dv[i] * xv.z; acc[4 * k4 + i][3] += dv[i] * xv.w;
                    bsum[4 * k4 + i] += dv[i];
                }
            }
        }
    }
    // sum the PS pixel lanes of this workgroup in a fixed order (deterministic) through LDS: one partial row per workgroup
    constexpr int KK = 4 * K4;
    const int rowlen = C4 * 4;                      // floats per k of one lane
    const int lstride = KK * rowlen + KK;           // floats per lane
    if (active) {
        float* mine = smem + (size_t)lane * lstride;
#pragma unroll
        for (int k = 0; k < KK; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[k * rowlen + 4 * q + j] = acc[k][j];
        if (q == 0) {
#pragma unroll
            for (int k = 0; k < KK; ++k) mine[KK * rowlen + k] = bsum[k];
        }
    }
    __syncthreads();
    const size_t KCf = (size_t)a.Cout * a.Cin;
    float* pb = a.part + (size_t)blockIdx.x * KCf;
    for (int e = tid; e < a.Cout * a.Cin; e += 256) {
        const int k = e / a.Cin, c = e - k * a.Cin;
        float sum = 0.f;
        for (int l = 0; l < PS; ++l) sum += smem[(size_t)l * lstride + k * rowlen + c];
        pb[e] = sum;
    }
    if (a.bpart != nullptr && tid < a.Cout) {
        float sum = 0.f;
        for (int l = 0; l < PS; ++l) sum += smem[(size_t)l * lstride + KK * rowlen + tid];
        a.bpart[(size_t)blockIdx.x * a.Cout + tid] = sum;
    }
}

// second stage for FEW outputs and MANY partial rows (the head): 4 row lanes x 64 elements per workgroup, fixed summation order
__global__ __launch_bounds__(256) void wgrad_reduce_rows_kernel(const float* __restrict__ part, float* __restrict__ dw, int rows, int T,
                                                                size_t KC_, int accumulate) {
    __shared__ float sm[4][64];
    const size_t total = (size_t)T * KC_;
    const int el = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const size_t e = (size_t)blockIdx.x * 64 + el;
    float s0 = 0.f, s1 = 0.f;
    if (e < total) {
        int r = rl;
        for (; r + 4 < rows; r += 8) { s0 += part[(size_t)r * total + e]; s1 += part[(size_t)(r + 4) * total + e]; }
        for (; r < rows; r += 4) s0 += part[(size_t)r * total + e];
    }
    sm[rl][el] = s0 + s1;
    __syncthreads();
    if (rl == 0 && e < total) {
        const float sum = (sm[0][el] + sm[1][el]) + (sm[2][el] + sm[3][el]);
        const size_t t = e / KC_, i = e - t * KC_;
        float* o = dw + i * T + t;
        *o = accumulate ? (*o + sum) : sum;
    }
}

// dw[(k*Cin + c)*T + t] (=|+=) sum_split part[split][t][k][c]
// one thread per (t, k*Cin+c): reads are coalesced along c for every split, 8 independent loads in flight
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int T,
                                                           size_t KC_, int accumulate) {
    const size_t total = (size_t)T * KC_;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
        const size_t t = e / KC_, i = e - t * KC_;
        const float* p = part + e;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sp = 0;
        for (; sp + 4 <= splits; sp += 4) {
            s0 += p[(size_t)(sp + 0) * total];
            s1 += p[(size_t)(sp + 1) * total];
            s2 += p[(size_t)(sp + 2) * total];
            s3 += p[(size_t)(sp + 3) * total];
        }
        for (; sp < splits; ++sp) s0 += p[(size_t)sp * total];
        const float s = (s0 + s1) + (s2 + s3);
        float* o = dw + i * T + t;
        *o = accumulate ? (*o + s) : s;
    }
}

// The same sum for many splits: Q threads per element, thread q takes splits q, q + Q, .. (four independent chains), the Q partial sums meet
// in LDS in a fixed order.  One thread per element walked up to 512 splits as 128 dependent steps on a grid of 144 workgroups (64 x 64 x 9
// elements): 18.5 us a launch at 1.5 TB/s, 56 launches a step.
template <int Q>
__global__ __launch_bounds__(256) void wgrad_reduce_q_kernel(const float* __restrict__ part, float* __restrict__ dw, int splits, int T, size_t KC_,
                                                             int accumulate) {
    constexpr int E = 256 / Q;
    __shared__ float ps[Q][E];
    const int el = threadIdx.x % E, q = threadIdx.x / E;
    const size_t total = (size_t)T * KC_;
    const size_t e = (size_t)blockIdx.x * E + el;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (e < total) {
        const float* p = part + e;
        int sp = q;
        for (; sp + 3 * Q < splits; sp += 4 * Q) {
            s0 += p[(size_t)sp * total];
            s1 += p[(size_t)(sp + Q) * total];
            s2 += p[(size_t)(sp + 2 * Q) * total];
            s3 += p[(size_t)(sp + 3 * Q) * total];
        }
        for (; sp < splits; sp += Q) s0 += p[(size_t)sp * total];
    }
    ps[q][el] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (q == 0 && e < total) {
        float s = ps[0][el];
#pragma unroll
        for (int j = 1; j < Q; ++j) s += ps[j][el];
        const size_t t = e / KC_, i = e - t * KC_;
        float* o = dw + i * T + t;
        *o = accumulate ? (*o + s) : s;
    }
}

// dbias[c] = sum over the pixel splits of bpart[split][c], in double.  64 channels per workgroup, four threads per channel take every
// fourth split and meet in LDS in a fixed order (one thread per channel walking up to 256 splits was a 22 us dependent chain, 18 times a step).
__global__ __launch_bounds__(256) void wgrad_bias_reduce_kernel(const float* __restrict__ bpart, float* __restrict__ dbias, int splits, int Cout) {
    __shared__ double part[4][64];
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double s = 0.0;
    if (c < Cout)
        for (int sp = q; sp < splits; sp += 4) s += (double)bpart[(size_t)sp * Cout + c];
    part[q][cl] = s;
    __syncthreads();
    if (q == 0 && c < Cout) dbias[c] = (float)(((part[0][cl] + part[1][cl]) + part[2][cl]) + part[3][cl]);
}

// unet_tuning.wgrad_narrow: the narrow-output kernel (wgrad_flat_kernel) for 80 < Cout <= 112 (0 falls back to the 64x64-tiled kernel);
// .wgrad_1x1: the 128x128-tiled GEMM kernel for 1x1 weight gradients
struct WPlan {
    unet_tuning tune;
    WArgs k;
    int ptw, splits, T, narrow, gemm1x1, small1x1, ps, bf16;
    int nsub;        // partial images per workgroup (wgrad_kernel's pixel sub-splits of narrow layers): the reduce kernels sum splits * nsub slices
    long long pix_per_block;
    size_t lds_bytes, lds_bytes16;
};

int make_wplan(const unet_wgrad_desc* d, WPlan* p) {
    UNET_CHECK_ARG(d != nullptr, "wgrad: null desc");
    p->tune = unetconv::tuning_of(d->tuning);
    UNET_CHECK_ARG(p->tune.wgrad_mfma_shape == 16 || p->tune.wgrad_mfma_shape == 32, "wgrad: unet_tuning.wgrad_mfma_shape must be 16 or 32 (start from unet_tuning_default())");
    UNET_CHECK_ARG(d->x && d->dy && d->dw, "wgrad: null tensor pointer");
    UNET_CHECK_ARG(d->ks == 1 || d->ks == 3, "wgrad: ks must be 1 or 3");
    UNET_CHECK_ARG(d->stride == 1 || d->stride == 2, "wgrad: stride must be 1 or 2");
    UNET_CHECK_ARG(!(d->ks == 1 && d->stride != 1), "wgrad: 1x1 stride 2 unsupported");
    UNET_CHECK_ARG(d->N > 0 && d->IH > 0 && d->IW > 0 && d->OH > 0 && d->OW > 0 && d->Cin > 0 && d->Cout > 0, "wgrad: bad dims");
    UNET_CHECK_ARG((long long)d->N * d->OH * d->OW < (1ll << 31), "wgrad: more than 2^31 output pixels");
    const int pad = (d->ks - 1) / 2;
    UNET_CHECK_ARG(d->OH == (d->IH + 2 * pad - d->ks) / d->stride + 1 && d->OW == (d->IW + 2 * pad - d->ks) / d->stride + 1,
                   "wgrad: output dims inconsistent");
    UNET_CHECK_ARG(d->dtype == UNET_F32 || d->dtype == UNET_BF16, "wgrad: unknown dtype %d", d->dtype);
    const int vec = d->dtype == UNET_BF16 ? 8 : 4;
    p->bf16 = d->dtype == UNET_BF16;
    UNET_CHECK_ARG(unet::slice_ok_v(d->x_cs, d->x_co, d->Cin, vec), "wgrad: bad x slice");
    UNET_CHECK_ARG(unet::slice_ok_v(d->dy_cs, d->dy_co, d->Cout, vec), "wgrad: bad dy slice");
    UNET_CHECK_ARG(unet::aligned16(d->x) && unet::aligned16(d->dy), "wgrad: x/dy must be 16-byte aligned");
    WArgs& k = p->k;
    memset(&k, 0, sizeof(k));
    k.x = d->x; k.dy = d->dy; k.part = d->workspace;
    k.x_cs = d->x_cs; k.x_co = d->x_co; k.dy_cs = d->dy_cs; k.dy_co = d->dy_co;
    k.N = d->N; k.IH = d->IH; k.IW = d->IW; k.Cin = d->Cin; k.Cin4 = unet::roundup(d->Cin, vec);
    k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout; k.Cout4 = unet::roundup(d->Cout, vec);
    p->T = d->ks * d->ks;
    p->nsub = 1;
    p->ptw = d->OW >= 32 ? 32 : (d->OW >= 16 ? 16 : 8);
    const int pt = p->bf16 ? (d->stride == 1 ? 128 : 32) : (d->stride == 1 ? 64 : 32);   // bf16: four (stride 2: one) MFMA k-blocks of 32 pixels
    const int pth = pt / p->ptw;
    k.tiles_y = unet::cdiv(d->OH, pth);
    k.tiles_x = unet::cdiv(d->OW, p->ptw);
    k.total_tiles = d->N * k.tiles_y * k.tiles_x;
    k.kt = unet::cdiv(d->Cout, BK);
    k.ct = unet::cdiv(d->Cin, BC);
    // narrow-output specialisation (taps flattened into the column dimension): only where the 64x64-tiled kernel pads
    p->narrow = (!p->bf16 && p->tune.wgrad_narrow && d->ks == 3 && d->stride == 1 && d->Cout > 80 && d->Cout <= 112 && d->OW >= 32) ? 1 : 0;
    int cols = k.kt * k.ct;
    if (p->narrow) {
        p->ptw = 32;
        k.tiles_y = d->OH;
        k.tiles_x = unet::cdiv(d->OW, 32);
        k.total_tiles = d->N * k.tiles_y * k.tiles_x;
        const int nch = unet::cdiv(d->Cin, 112);
        k.cw = unet::roundup(unet::cdiv(d->Cin, nch), 4);
        const int ntw = d->Cout > 96 ? 5 : 7;                     // column tiles per wave (KT = 7 / 6 output-channel tiles)
        k.nnb = unet::cdiv(unet::cdiv(9 * k.cw, 16), 4 * ntw);
        cols = nch * k.nnb;
    }
    // heads: 1x1 with <= 16 output channels and <= 1024 input channels -> FMA kernel, one partial row per (workgroup, pixel lane)
    p->small1x1 = (!p->bf16 && d->ks == 1 && d->Cout <= 16 && k.Cin4 <= 512 && p->tune.wgrad_1x1) ? 1 : 0;
    p->ps = 1; p->pix_per_block = 0;
    if (p->small1x1) {
        const long long P = (long long)d->N * d->OH * d->OW;
        p->ps = 256 / (k.Cin4 / 4);
        {
            const int kk = 4 * ((d->Cout + 3) / 4);
            const int per_lane = (kk * k.Cin4 + kk) * (int)sizeof(float);
            if (p->ps > 65536 / per_lane) p->ps = 65536 / per_lane;
            if (p->ps < 1) p->ps = 1;
        }
        long long blocks = P / ((long long)p->ps * 64);            // >= 64 pixels per thread
        if (blocks > 1024) blocks = 1024;
        if (blocks < 1) blocks = 1;
        p->pix_per_block = (P + blocks - 1) / blocks;
        p->splits = (int)((P + p->pix_per_block - 1) / p->pix_per_block);       // one partial row per workgroup
        p->gemm1x1 = 0;
        p->lds_bytes = p->lds_bytes16 = 0;
        return UNET_OK;
    }
    p->gemm1x1 = (!p->bf16 && d->ks == 1 && p->tune.wgrad_1x1) ? 1 : 0;
    // ... except where its 128 x 128 blocks are mostly padding or the whole launch is a few GFLOP: there the 64 x 64-blocked general kernel wins
    // (scripts/ab_wgrad1x1.py: SelfAttention's 48 x 4096 products 54 -> 39 us, the identity-path convs 64 -> 128 at 64^2 35 -> 25 us,
    // 256 -> 512 at 16^2 32 -> 24 us; from 4.3 GFLOP up the GEMM kernel is 1.4-1.8 x faster).  unet_tuning.wgrad_1x1 = 2 forces the GEMM kernel.
    if (p->gemm1x1 && p->tune.wgrad_1x1 != 2 &&
        ((d->Cin <= 64 || d->Cout <= 64) || 2.0 * d->N * d->OH * d->OW * (double)d->Cin * d->Cout < 3.0e9))
        p->gemm1x1 = 0;
    if (p->gemm1x1) {       // flat 64-pixel tiles, 128 x 128 channel blocks
        k.kt = unet::cdiv(d->Cout, 128);
        k.ct = unet::cdiv(d->Cin, 128);
        k.total_tiles = unet::cdiv((long long)d->N * d->OH * d->OW, 64);
        k.tiles_y = k.tiles_x = 1;
        cols = k.kt * k.ct;
    }
    // aim for ~512 workgroups (256 CUs x 2 resident); at least 4 tiles per block.
    // bf16 storage (round 5): a workgroup's pixel tiles cost 8 x fewer MFMA clocks than in fp32, but its partial filter image -- T x 64 x 64
    // floats, written once and read once by the reduce kernel -- costs the same: at 512 workgroups the 19 GFLOP encoder layers spend ~4 us
    // multiplying and ~40 us moving 2 x 75 MB of partials (profiles/r05_a_layer_bench_bf16.txt: 64 -> 64 at 16 x 128^2 73 us, MFMA roof 8).
    // One workgroup per CU halves the partials and doubles the reduction length of each.  Measured alone on the chip
    // (scripts/ab_wgrad_wgs.py, profiles/r05_b_wgrad_wgs.txt), 256 against 512 workgroups: 64 -> 64 at 128^2 54.6 / 73.3 us, 128 -> 128 at 64^2
    // 42.1 / 49.9, 512 -> 512 at 16^2 56.7 / 66.3, 384 -> 384 at 64^2 188.8 / 238.3; from 256 -> 256 at 128^2 (309 GFLOP) up the longer
    // reductions win with two workgroups per CU (277.7 / 325.4), and so do the 1x1 launches (one tap: a ninth of the partial bytes).
    int target = 512;
    if (p->tune.wgrad_wgs > 0) target = p->tune.wgrad_wgs;
    else if (p->bf16 && d->ks == 3 && d->stride == 1 && (long long)k.total_tiles * cols <= 20000) target = 256;
    int want = target / cols;
    if ((p->narrow || p->bf16) && want >= 8) want &= ~7;      // the XCD-aware mapping pads the split count to a multiple of 8: stay within 512
    if (want < 1) want = 1;
    int tpb = unet::cdiv(k.total_tiles, want);
    if (tpb < 4) tpb = 4;
    if (tpb > k.total_tiles) tpb = k.total_tiles;
    k.tiles_per_block = tpb;
    p->splits = unet::cdiv(k.total_tiles, tpb);
    // fp32 wgrad_kernel on layers of <= 32 input and / or output channels: empty channel halves become pixel sub-splits (see the kernel)
    if (!p->bf16 && !p->narrow && !p->gemm1x1 && p->tune.wgrad_mfma_shape == 32 && p->tune.wgrad_narrow != 3) {
        k.csub = d->Cin <= 32 ? 1 : 0;
        k.ksub = d->Cout <= 32 ? 1 : 0;
        p->nsub = (1 + k.csub) * (1 + k.ksub);
    }
    const int hh = (pth - 1) * d->stride + d->ks, hw = (p->ptw - 1) * d->stride + d->ks;
    p->lds_bytes = (size_t)(pt * BK + hh * hw * BC) * sizeof(float);
    p->lds_bytes16 = (size_t)(pt + hh * hw) * (d->stride == 1 ? 80 : 72) * sizeof(float);
    if (p->bf16) p->lds_bytes = (size_t)pt * 160 + (size_t)hh * hw * (d->stride == 1 ? 160 : 144);
    return UNET_OK;
}

// unet_tuning.wgrad_mfma_shape = 16: the 16x16x4 wgrad form measured slower (register pressure): opt-in only

template <int PTW, int S, int KS>
int launch_w16(const WPlan& p, hipStream_t st) {
    auto kern = wgrad16_kernel<PTW, S, KS>;
    static unsigned long long configured = 0;   // one bit per device
    if (unet::first_use_on_device(&configured)) {
        UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(kern, dim3(p.k.kt * p.k.ct, p.splits), dim3(256), p.lds_bytes16, st, p.k);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <int PTW, int S, int KS>
int launch_w(const WPlan& p, hipStream_t st) {
    if (p.tune.wgrad_mfma_shape == 16) return launch_w16<PTW, S, KS>(p, st);
    auto kern = wgrad_kernel<PTW, S, KS>;
    static unsigned long long configured = 0;   // one bit per device
    if (unet::first_use_on_device(&configured)) {
        UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    hipLaunchKernelGGL(kern, dim3(p.k.kt * p.k.ct, p.splits), dim3(256), p.lds_bytes, st, p.k);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

// unet_tuning.wgrad_bf16_k4: 3x3 / stride 1 / 32-wide tiles on wgrad_bf16_k4_kernel (0: wgrad_bf16_kernel)

template <int KV, int KS>
int launch_wb_k4n(const WPlan& p, hipStream_t st) {
    auto kern = wgrad_bf16_k4_kernel<KV, KS>;
    static unsigned long long configured = 0;   // one bit per device
    if (unet::first_use_on_device(&configured))
        UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    WArgs k = p.k;
    k.kt = unet::cdiv(p.k.Cout, 16 * KV);
    k.xcd_map = p.splits >= 8 ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(k.kt * p.k.ct, k.xcd_map ? unet::roundup(p.splits, 8) : p.splits), dim3(256), p.lds_bytes + 11 * 256 * sizeof(unsigned),
                       st, k);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <int KS>
int launch_wb_k4(const WPlan& p, hipStream_t st) {
    // tiles per block: 3 or 4, whichever pads the 16-wide tiles of Cout less (4 on a tie: fewer reads of x per MFMA; narrower blocks
    // only when Cout itself is that narrow -- at KV = 1 every x fragment would serve a single MFMA)
    const int tiles = unet::cdiv(p.k.Cout, 16);
    const int kv = tiles <= 2 ? tiles : (unet::roundup(tiles, 3) < unet::roundup(tiles, 4) ? 3 : 4);
    switch (kv) {
        case 1: return launch_wb_k4n<1, KS>(p, st);
        case 2: return launch_wb_k4n<2, KS>(p, st);
        case 3: return launch_wb_k4n<3, KS>(p, st);
        default: return launch_wb_k4n<4, KS>(p, st);
    }
}

template <int PTW, int S, int KS>
int launch_wb(const WPlan& p, hipStream_t st) {
    if constexpr (PTW == 32 && S == 1) {
        // (its in-image byte offsets are 32-bit with bit 31 = out of range: one image of either tensor within 2 GiB)
        if (p.tune.wgrad_bf16_k4 && (long long)p.k.IH * p.k.IW * p.k.x_cs * 2 < (1ll << 31) - 65536 && (long long)p.k.OH * p.k.OW * p.k.dy_cs * 2 < (1ll << 31) - 65536)
            return launch_wb_k4<KS>(p, st);
    }
    auto kern = wgrad_bf16_kernel<PTW, S, KS>;
    static unsigned long long configured = 0;   // one bit per device
    if (unet::first_use_on_device(&configured)) {
        UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    }
    WArgs k = p.k;
    k.xcd_map = p.splits >= 8 ? 1 : 0;           // fewer splits than XCDs: plain numbering
    hipLaunchKernelGGL(kern, dim3(p.k.kt * p.k.ct, k.xcd_map ? unet::roundup(p.splits, 8) : p.splits), dim3(256), p.lds_bytes, st, k);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <int PTW>
int launch_wb_ptw(const WPlan& p, int ks, int stride, hipStream_t st) {
    if (ks == 1) return launch_wb<PTW, 1, 1>(p, st);
    if (stride == 1) return launch_wb<PTW, 1, 3>(p, st);
    return launch_wb<PTW, 2, 3>(p, st);
}

template <int PTW>
int launch_w_ptw(const WPlan& p, int ks, int stride, hipStream_t st) {
    if (ks == 1) return launch_w<PTW, 1, 1>(p, st);
    if (stride == 1) return launch_w<PTW, 1, 3>(p, st);
    return launch_w<PTW, 2, 3>(p, st);
}

}  // namespace

extern "C" size_t unet_conv2d_wgrad_workspace(const unet_wgrad_desc* d) {
    WPlan p;
    if (make_wplan(d, &p) != UNET_OK) return 0;
    // split-K partials of dW followed by the [splits][Cout] partial column sums of dy (bias gradient)
    return (size_t)p.splits * p.nsub * p.T * d->Cout * d->Cin + (size_t)p.splits * d->Cout;
}

extern "C" int unet_conv2d_wgrad(const unet_wgrad_desc* d, void* stream) {
    WPlan p;
    int rc = make_wplan(d, &p);
    if (rc != UNET_OK) return rc;
    const size_t npart = (size_t)p.splits * p.nsub * p.T * d->Cout * d->Cin;
    const size_t need = npart + (size_t)p.splits * d->Cout;
    UNET_CHECK_ARG(d->workspace != nullptr && d->workspace_floats >= need, "wgrad: workspace too small (%zu < %zu floats)",
                   d->workspace_floats, need);
    p.k.bpart = d->dbias != nullptr ? d->workspace + npart : nullptr;
    hipStream_t st = (hipStream_t)stream;
    if (p.bf16) {
        switch (p.ptw) {
            case 32: rc = launch_wb_ptw<32>(p, d->ks, d->stride, st); break;
            case 16: rc = launch_wb_ptw<16>(p, d->ks, d->stride, st); break;
            default: rc = launch_wb_ptw<8>(p, d->ks, d->stride, st); break;
        }
    } else if (p.small1x1) {
        const long long P = (long long)d->N * d->OH * d->OW;
        const dim3 grid((unsigned)p.splits);
        const int k4 = (d->Cout + 3) / 4;
        const size_t lds = (size_t)p.ps * (4 * k4 * p.k.Cin4 + 4 * k4) * sizeof(float);
        UNET_CHECK_ARG(lds <= 64 * 1024, "wgrad: head kernel needs %zu bytes of LDS", lds);
        switch (k4) {
            case 1: hipLaunchKernelGGL(wgrad1x1_small_kernel<1>, grid, dim3(256), lds, st, p.k, P, p.ps, p.pix_per_block); break;
            case 2: hipLaunchKernelGGL(wgrad1x1_small_kernel<2>, grid, dim3(256), lds, st, p.k, P, p.ps, p.pix_per_block); break;
            case 3: hipLaunchKernelGGL(wgrad1x1_small_kernel<3>, grid, dim3(256), lds, st, p.k, P, p.ps, p.pix_per_block); break;
            default: hipLaunchKernelGGL(wgrad1x1_small_kernel<4>, grid, dim3(256), lds, st, p.k, P, p.ps, p.pix_per_block); break;
        }
        UNET_CHECK_LAUNCH();
        rc = UNET_OK;
    } else if (p.gemm1x1) {
        static unsigned long long configured = 0;   // one bit per device
        if (unet::first_use_on_device(&configured)) {
            UNET_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad1x1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        hipLaunchKernelGGL(wgrad1x1_kernel, dim3(p.k.kt * p.k.ct, p.splits), dim3(256), (size_t)2 * 64 * 128 * sizeof(float), st, p.k,
                           (long long)d->N * d->OH * d->OW);
        UNET_CHECK_LAUNCH();
        rc = UNET_OK;
    } else if (p.narrow) {
        const size_t lds = (size_t)(32 + 3 * 34) * 112 * sizeof(float) + (size_t)4 * 256 * 16;      // the two tiles + the bias item sums [DIT = 4][256] float4
        const dim3 grid(unet::cdiv(d->Cin, p.k.cw) * p.k.nnb, unet::roundup(p.splits, 8));
        static unsigned long long configured = 0;   // one bit per device
        if (unet::first_use_on_device(&configured)) {
            UNET_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_flat_kernel<7, 5>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            UNET_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_flat_kernel<6, 5, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            UNET_CHECK_HIP(hipFuncSetAttribute((const void*)wgrad_flat_kernel<6, 7>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        }
        // 97..100 output channels (the 100 -> 100 pair): six 16-row tiles + a 4-row sliver; unet_tuning.wgrad_narrow = 2 keeps the seven-tile form (A/B)
        if (d->Cout > 96 && d->Cout <= 100 && p.tune.wgrad_narrow != 2) hipLaunchKernelGGL((wgrad_flat_kernel<6, 5, true>), grid, dim3(256), lds, st, p.k);
        else if (d->Cout > 96) hipLaunchKernelGGL((wgrad_flat_kernel<7, 5>), grid, dim3(256), lds, st, p.k);
        else hipLaunchKernelGGL((wgrad_flat_kernel<6, 7>), grid, dim3(256), lds, st, p.k);
        UNET_CHECK_LAUNCH();
        rc = UNET_OK;
    } else
    switch (p.ptw) {
        case 32: rc = launch_w_ptw<32>(p, d->ks, d->stride, st); break;
        case 16: rc = launch_w_ptw<16>(p, d->ks, d->stride, st); break;
        default: rc = launch_w_ptw<8>(p, d->ks, d->stride, st); break;
    }
    if (rc != UNET_OK) return rc;
    const size_t KC_ = (size_t)d->Cout * d->Cin;
    const int slices = p.splits * p.nsub;          // partial images to sum
    if (p.small1x1)
        hipLaunchKernelGGL(wgrad_reduce_rows_kernel, dim3(unet::cdiv((int)(KC_ * p.T), 64)), dim3(256), 0, st, d->workspace, d->dw,
                           p.splits, p.T, KC_, d->accumulate);
    else if (slices >= 32 && KC_ * p.T < ((size_t)1 << 31) / 8)
        hipLaunchKernelGGL(wgrad_reduce_q_kernel<8>, dim3((unsigned)unet::cdiv((long long)(KC_ * p.T), 32LL)), dim3(256), 0, st, d->workspace, d->dw,
                           slices, p.T, KC_, d->accumulate);
    else if (slices >= 8 && KC_ * p.T < ((size_t)1 << 31) / 8)
        hipLaunchKernelGGL(wgrad_reduce_q_kernel<4>, dim3((unsigned)unet::cdiv((long long)(KC_ * p.T), 64LL)), dim3(256), 0, st, d->workspace, d->dw,
                           slices, p.T, KC_, d->accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(unet::ew_grid((long long)KC_ * p.T, 256)), dim3(256), 0, st, d->workspace, d->dw,
                           slices, p.T, KC_, d->accumulate);
    UNET_CHECK_LAUNCH();
    if (d->dbias != nullptr) {
        hipLaunchKernelGGL(wgrad_bias_reduce_kernel, dim3(unet::cdiv(d->Cout, 64)), dim3(256), 0, st, p.k.bpart, d->dbias, p.splits,
                           d->Cout);
        UNET_CHECK_LAUNCH();
    }
    return UNET_OK;
}
