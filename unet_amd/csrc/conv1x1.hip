// conv1x1_gemm_kernel: 1x1 / stride-1 convolutions whose reduction is whole 64-byte chunks -- the PixelShuffle_ICNR convs of the decoder
// (reference train.py:141, blur=True: ConvLayer(ni, 4 nf, ks=1) in front of every PixelShuffle), forward and input gradient, and the
// identity-path convs of the encoder ResBlocks -- as what they are: a GEMM over FLAT pixels, Y[p][co] = sum_ci X[p][ci] W[co][ci].
//
// The implicit-GEMM conv kernels treat them as a one-tap filter: a halo tile of the pixel patch goes global -> registers -> LDS -> registers
// behind one barrier per 64-byte chunk, with nine-tap bookkeeping around a single stage (fp32: 4.2 ms per cfg2 step on
// conv_igemm16_kernel, bf16: 1.5 ms on conv_bf16_kernel at 1.4-2.4 TB/s).  Without a halo there is nothing to share through LDS: the MFMA
// B operand of lane (l15, kq) -- pixel l15 of a 16-pixel tile, channels VEC kq .. of the chunk -- IS a 16-byte piece of the NHWC row of that
// pixel, so every wave fetches its operands global -> VGPR directly, one chunk ahead (two register sets, compiler-scheduled plain loads: all
// of them unconditional, the counts per stage fixed), no LDS, no barrier.
//   workgroup = 4 waves = 2 (pixel halves) x 2 (channel halves): 128 pixels x 128 channels; wave = 4 x 4 tiles of 16 x 16
//   filter image = the packed 1x1 image of unet_pack_weights[_bf16]: wp[chunk][outPad][64 B], tile n of a chunk = 1 KiB in operand order
//   channel tiles dealt round-robin to the two waves of a pixel half (a 96-wide block: 3 + 3 tiles, not 4 + 2)
//   XCD-aware order: the channel blocks of one pixel tile are consecutive workgroups of one XCD (their pixel rows hit that XCD's L2)
// Roofline: bf16 storage is HBM-bound (96 -> 384 at 16 x 256^2: 1.0 GB for 77 GFLOP), fp32 MFMA-bound (the same layer: 2.0 GB, 0.49 ms
// of fp32 MFMA time).  Algorithmic bytes per pixel: EB (Cin + Cout [+ residual / mask]); FLOP per pixel: 2 Cin Cout.
#include "conv_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct G1Args {
    const char* x; const char* wp; const float* bias; const char* res; const char* mask; char* y;
    int x_cs, x_co, res_cs, res_co, mask_cs, mask_co, y_cs, y_co;
    long long P;
    int nchunks, coutPad, n_base, n_end, Cout, relu, y_f32, ntn;
    long long ntiles;          // pixel tiles x channel blocks
    // pixel-shuffle store (unet_conv_desc.pixel_shuffle): produced channel q = ij * nf + c of input pixel (img, h, w) goes to channel c of
    // output pixel (img, 2 h + (ij >> 1), 2 w + (ij & 1)) of a [N, 2 H, 2 W] tensor; bias is indexed in the filter's own order 4 c + ij
    int ps, nf, H, W;
    // unet_conv_desc.ps_tail: quads [0, tail_q) of tail + pixel * tail_cs + tail_co go to channel tail_at of the same OUTPUT pixel of y's buffer
    const char* tail; int tail_cs, tail_co, tail_q, tail_at;
};

// STAGED = false: the pixel operand of a wave goes global -> VGPR directly (128-pixel workgroup tile, 4 x 4 tiles per wave).
// STAGED = true (unet_tuning.conv1x1_gemm = 2): a 256-pixel workgroup tile whose 64-byte pixel chunk is fetched ONCE per workgroup with coalesced
// 16-byte items (4 per thread), staged through a double-buffered LDS image (rows of 64 + 32 bytes: conflict-free ds_read_b128 groups, as in the
// 256-pixel 3x3 kernel) and read by both channel-half waves; 8 x 4 tiles per wave, one barrier per chunk.
template <typename T, bool STAGED>
__global__ __launch_bounds__(256, 2) void conv1x1_gemm_kernel(const G1Args a) {
    constexpr int EB = (int)sizeof(T), KCB = 64;                  // bytes per element / per reduction chunk of one pixel
    constexpr int MT = STAGED ? 8 : 4, NT = 4, TPIX = MT * 32, ROWB = 96, BUFB = TPIX * ROWB;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1, l15 = lane & 15, kq = lane >> 4;
    const long long per_xcd = (long long)(gridDim.x >> 3);
    const long long wg = (long long)(blockIdx.x & 7) * per_xcd + (long long)(blockIdx.x >> 3);
    if (wg >= a.ntiles) return;
    const int nb = (int)(wg % a.ntn);
    const long long pt = wg / a.ntn;
    const long long p0 = pt * TPIX + wm * (TPIX / 2);
    const int c0 = a.n_base + nb * 128;                             // first produced channel of the block

    // operand addresses: pixel m * 16 + l15 of this wave's 64 (clamped to the last pixel: rows beyond P are computed and not stored)
    const char* xp[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        long long p = p0 + m * 16 + l15;
        p = p < a.P ? p : a.P - 1;
        xp[m] = a.x + ((size_t)p * a.x_cs + a.x_co) * EB + 16 * kq;
    }
    // filter tile n of this wave = channels c0 + (2 n + wn) * 16 ..: lane (l15, kq) reads 16 bytes at [column][16 kq]
    const char* wl = a.wp + ((size_t)(c0 + wn * 16 + l15) * KCB + 16 * kq);
    const size_t slab = (size_t)a.coutPad * KCB;
    int nv = 0;                                                      // tiles of this wave that hold produced channels
#pragma unroll
    for (int n = 0; n < NT; ++n) nv += (c0 + (2 * n + wn) * 16 < a.n_end) ? 1 : 0;
    nv = __builtin_amdgcn_readfirstlane(nv);

    f32x4 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto mma = [&](const f32x4 (&xs)[MT], const f32x4 (&ws)[NT]) {
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (n < nv) {
                if constexpr (EB == 2) {
                    const bf16x8 wv = __builtin_bit_cast(bf16x8, ws[n]);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wv, __builtin_bit_cast(bf16x8, xs[m]), acc[m][n], 0, 0, 0);
                } else {
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
                        for (int m = 0; m < MT; ++m) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(ws[n][kk], xs[m][kk], acc[m][n], 0, 0, 0);
                }
            }
        }
    };
    const int last = a.nchunks - 1;
    if constexpr (!STAGED) {
        f32x4 xa[MT], wa[NT], xb[MT], wb[NT];
        auto load = [&](f32x4 (&xs)[MT], f32x4 (&ws)[NT], int c) {
#pragma unroll
            for (int m = 0; m < MT; ++m) xs[m] = *reinterpret_cast<const f32x4*>(xp[m] + (size_t)c * KCB);
#pragma unroll
            for (int n = 0; n < NT; ++n) ws[n] = *reinterpret_cast<const f32x4*>(wl + (size_t)c * slab + (size_t)n * 2048);
        };
        // two chunks per trip, the next chunk's operands in flight behind the current chunk's MFMAs; beyond the last chunk the loads repeat it
        // (unconditional: the compiler's wait counts stay exact), the MFMAs are skipped
        load(xa, wa, 0);
        for (int c = 0; c < a.nchunks; c += 2) {
            load(xb, wb, c + 1 < last ? c + 1 : last);
            mma(xa, wa);
            load(xa, wa, c + 2 < last ? c + 2 : last);
            if (c + 1 < a.nchunks) mma(xb, wb);
        }
    } else {
        // staging items of this thread: item e = tid + 256 it covers piece (e & 3) of pixel (e >> 2) of the 256-pixel tile
        const char* sp[4];
        unsigned so[4];
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int e = tid + 256 * it, pe = e >> 2, q = e & 3;
            long long p = pt * TPIX + pe;
            p = p < a.P ? p : a.P - 1;
            sp[it] = a.x + ((size_t)p * a.x_cs + a.x_co) * EB + 16 * q;
            so[it] = (unsigned)(pe * ROWB + 16 * q);
        }
        const unsigned rbase = (unsigned)((wm * (TPIX / 2) + l15) * ROWB + 16 * kq);          // pixel tile m of this wave: + m * 16 * ROWB
        f32x4 st[4], wa[NT], wb[NT];
        auto gload = [&](f32x4 (&ws)[NT], int c) {
#pragma unroll
            for (int it = 0; it < 4; ++it) st[it] = *reinterpret_cast<const f32x4*>(sp[it] + (size_t)c * KCB);
#pragma unroll
            for (int n = 0; n < NT; ++n) ws[n] = *reinterpret_cast<const f32x4*>(wl + (size_t)c * slab + (size_t)n * 2048);
        };
        auto sstore = [&](int buf) {
#pragma unroll
            for (int it = 0; it < 4; ++it) *reinterpret_cast<f32x4*>(lds + buf * BUFB + so[it]) = st[it];
        };
        auto stage = [&](const f32x4 (&ws)[NT], int buf) {
            f32x4 xs[MT];
#pragma unroll
            for (int m = 0; m < MT; ++m) xs[m] = *reinterpret_cast<const f32x4*>(lds + buf * BUFB + rbase + m * 16 * ROWB);
            mma(xs, ws);
        };
        gload(wa, 0);
        sstore(0);
        __syncthreads();
        for (int c = 0; c < a.nchunks; c += 2) {
            gload(wb, c + 1 < last ? c + 1 : last);          // chunk c + 1: pixel items + filter tiles in flight behind the MFMAs of chunk c
            stage(wa, 0);
            sstore(1);
            __syncthreads();
            gload(wa, c + 2 < last ? c + 2 : last);
            if (c + 1 < a.nchunks) stage(wb, 1);
            sstore(0);
            __syncthreads();
        }
    }

    // ---- epilogue: lane (l15, kq) holds channels 4 kq .. 4 kq + 3 of tile n for pixel l15 of pixel tile m
    const T* resb = reinterpret_cast<const T*>(a.res);
    const T* maskb = reinterpret_cast<const T*>(a.mask);
    auto ld4 = [](const T* p_) -> f32x4 {
        if constexpr (EB == 2) {
            const uint2 u = *reinterpret_cast<const uint2*>(p_);
            return (f32x4){__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
        } else return *reinterpret_cast<const f32x4*>(p_);
    };
    // stores: pixel tile by pixel tile, its channel tiles back to back (the pieces of one pixel's 128-byte lines reach the L2 together)
    f32x4 bv[NT];
    int c4s[NT];
#pragma unroll
    for (int n = 0; n < NT; ++n) {
        const int c4 = c0 + (2 * n + wn) * 16 + 4 * kq;
        c4s[n] = c4;
        bv[n] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (a.bias != nullptr && n < nv && c4 < a.n_end) {
            if (a.ps) {
                const int ij = c4 / a.nf, c = c4 - ij * a.nf;
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[n][q] = a.bias[4 * (c + q) + ij];
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) bv[n][q] = (c4 + q < a.Cout) ? a.bias[c4 + q] : 0.f;
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const long long p = p0 + m * 16 + l15;
        if (p >= a.P) continue;
        long long pimg = 0;
        int ph = 0, pw = 0;
        if (a.ps) {
            pimg = p / ((long long)a.H * a.W);
            const int rem = (int)(p - pimg * a.H * a.W);
            ph = rem / a.W; pw = rem - ph * a.W;
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) {
            if (n >= nv) continue;
            const int c4 = c4s[n];
            if (c4 >= a.n_end) continue;
            f32x4 v = acc[m][n] + bv[n];
            if (resb != nullptr) v += ld4(resb + (size_t)p * a.res_cs + a.res_co + c4);
            if (a.relu) {
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = fmaxf(v[q], 0.f);
            }
            if (maskb != nullptr) {
                const f32x4 mv = ld4(maskb + (size_t)p * a.mask_cs + a.mask_co + c4);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = mv[q] > 0.f ? v[q] : 0.f;
            }
            size_t yo = (size_t)p * a.y_cs + a.y_co + c4;
            if (a.ps) {
                const int ij = c4 / a.nf, c = c4 - ij * a.nf;
                yo = ((size_t)(pimg * 2 * a.H + 2 * ph + (ij >> 1)) * (2 * a.W) + 2 * pw + (ij & 1)) * a.y_cs + a.y_co + c;
            }
            if (EB == 4 || a.y_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.y) + yo) = v;
            else {
                const bf16x4 o = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(a.y) + yo) = __builtin_bit_cast(uint2, o);
            }
            if (a.ps && a.tail != nullptr && c4 % a.nf == a.nf - 4) {      // the last quad of this output pixel's shuffled channels: append the tail
                const size_t op = (yo - (size_t)a.y_co - (size_t)(a.nf - 4)) / (size_t)a.y_cs;
                const char* src = a.tail + ((size_t)op * a.tail_cs + a.tail_co) * EB;
                char* dst = a.y + ((size_t)op * a.y_cs + a.tail_at) * EB;
                for (int t = 0; t < a.tail_q; ++t) {
                    if constexpr (EB == 4) *reinterpret_cast<f32x4*>(dst + 16 * t) = *reinterpret_cast<const f32x4*>(src + 16 * t);
                    else *reinterpret_cast<uint2*>(dst + 8 * t) = *reinterpret_cast<const uint2*>(src + 8 * t);
                }
            }
        }
    }
}


// ---- conv1x1_head_kernel: a 1x1 convolution with at most 16 OUTPUT channels -- the segmentation head (reference train.py:141-144: the last layer
// of DynamicUnet, ConvLayer(100, n_classes, ks=1, act_cls=None)) --, forward.  SURVEY 8a: 2 x 100 x 5 FLOP against 400 B (fp32) per pixel: a
// stream over the widest activation of the network (16 x 512^2 x 100: 1.7 GB fp32, 0.87 GB bf16) that the tiled implicit-GEMM kernels read at
// 3.3-3.4 TB/s through their LDS halo (conv_igemm16_kernel<32,1,1,4,1,4> 508 us, conv_bf16_kernel<32,1,1,4,1,4> 262 us in the step; this
// kernel alone: 432 / 203 us = 4.2 / 5.0 TB/s).  Here a
// wave owns 64 consecutive pixels and nothing else: the MFMA B operand of lane (l15, kq) IS the 16-byte piece [16 kq, 16 kq + 16) of chunk c of
// pixel l15 (global -> VGPR, every load of a trip issued before the first MFMA), the filter (<= 16 x 128 values) lives in registers for the whole
// kernel, no LDS, no barrier.  Accumulation per output element = the chain of the implicit-GEMM kernels (chunks ascending; fp32: the four
// k-steps of a chunk ascending, a reduction tail as ONE transposed step), so the logits are the same bits.
struct HeadArgs {
    const char* x; const char* wp; const float* bias; char* y;
    int x_cs, x_co, y_cs, y_co;
    long long P;
    int Cin, Cout, coutPad, relu, y_f32;
};

template <typename T, int NCH, int MT>
__global__ __launch_bounds__(256) void conv1x1_head_kernel(const HeadArgs a) {
    constexpr int EB = (int)sizeof(T), KC = 64 / EB, VEC = 16 / EB;               // 64-byte chunks of KC channels; MT pixel tiles of 16 per wave and trip
    const int lane = threadIdx.x & 63, l15 = lane & 15, kq = lane >> 4;
    const int nch = (a.Cin + KC - 1) / KC, tail = a.Cin % KC;                       // (tail: fp32 only -- bf16 tails are zero padded chunks)
    const int cin_v = (a.Cin + VEC - 1) / VEC * VEC;                                // channels that exist in the slice (pad lanes are zeros)
    // filter: lane (cout l15, kq) keeps its 16 bytes of every chunk
    f32x4 w[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        w[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (c < nch) w[c] = *reinterpret_cast<const f32x4*>(a.wp + ((size_t)(c * a.coutPad + l15) * KC) * EB + 16 * kq);
    }
    float bv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = (a.bias != nullptr && 4 * kq + r < a.Cout) ? a.bias[4 * kq + r] : 0.f;
    const long long nwaves = (long long)gridDim.x * 4, wave0 = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    constexpr int WP = 16 * MT;
    auto ldx = [](const char* q) { return *reinterpret_cast<const f32x4*>(q); };
    for (long long g = wave0; g * WP < a.P; g += nwaves) {
        f32x4 xs[MT][NCH];
        const char* xp[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            long long p = g * WP + m * 16 + l15;
            p = p < a.P ? p : a.P - 1;                       // computed, not stored
            xp[m] = a.x + ((size_t)p * a.x_cs + a.x_co) * EB;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                xs[m][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (EB == 4) {
                    if (c < nch && !(tail != 0 && c == nch - 1)) xs[m][c] = ldx(xp[m] + (size_t)(c * KC + 4 * kq) * EB);
                    else if (c == nch - 1 && tail != 0) {   // transposed tail: k-slot kq of step kk = channel kq + 4 kk of the chunk
#pragma unroll
                        for (int kk = 0; kk < 4; ++kk)
                            if (kq + 4 * kk < tail) xs[m][c][kk] = *reinterpret_cast<const float*>(xp[m] + (size_t)(c * KC + kq + 4 * kk) * EB);
                    }
                } else {
                    if (c < nch && c * KC + VEC * kq < cin_v) xs[m][c] = ldx(xp[m] + (size_t)(c * KC + VEC * kq) * EB);
                }
            }
        }
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            if (c < nch) {
                if constexpr (EB == 2) {
#pragma unroll
                    for (int m = 0; m < MT; ++m)
                        acc[m] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w[c]), __builtin_bit_cast(bf16x8, xs[m][c]), acc[m], 0, 0, 0);
                } else {
                    const int steps = (tail != 0 && c == nch - 1) ? (tail + 3) / 4 : 4;
#pragma unroll
                    for (int kk = 0; kk < 4; ++kk)
                        if (kk < steps) {
#pragma unroll
                            for (int m = 0; m < MT; ++m) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[c][kk], xs[m][c][kk], acc[m], 0, 0, 0);
                        }
                }
            }
        }
        // lane (pixel l15, kq): channels 4 kq .. 4 kq + 3 (only produced channels are written: pad lanes of a slice belong to its owner)
        if (4 * kq < a.Cout) {
            const int nr = a.Cout - 4 * kq;                  // valid channels of this lane's quad (>= 4: the whole vector)
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const long long p = g * WP + m * 16 + l15;
                if (p >= a.P) continue;
                f32x4 v = acc[m];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] += bv[r];
                    if (a.relu) v[r] = fmaxf(v[r], 0.f);
                }
                if (EB == 4 || a.y_f32) {
                    float* yo = reinterpret_cast<float*>(a.y) + (size_t)p * a.y_cs + a.y_co + 4 * kq;
                    if (nr >= 4) *reinterpret_cast<f32x4*>(yo) = v;
                    else {
#pragma unroll
                        for (int r = 0; r < 3; ++r) if (r < nr) yo[r] = v[r];
                    }
                } else {
                    __bf16* yo = reinterpret_cast<__bf16*>(a.y) + (size_t)p * a.y_cs + a.y_co + 4 * kq;
                    if (nr >= 4) *reinterpret_cast<bf16x4*>(yo) = (bf16x4){(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
                    else {
#pragma unroll
                        for (int r = 0; r < 3; ++r) if (r < nr) yo[r] = (__bf16)v[r];
                    }
                }
            }
        }
    }
}
}  // namespace

namespace unetconv {

// the launches this kernel takes (the descriptor was validated by the planner): one filter image for the batch, whole 64-byte reduction
// chunks, a grid that fills the chip (smaller problems keep the generic kernels and their split reduction)
// Measured (scripts/ab_conv1x1.py: the ten PixelShuffle-conv launches of a cfg2 step, isolated, batch 16; profiles/r04_d_conv1x1.log,
// r04_o_conv1x1_staged.log).  The DIRECT form (operands global -> VGPR) loses to the implicit-GEMM kernels as a replacement on plain 1x1 convs --
// fp32 5.8 vs 4.1 ms, bf16 1.77 vs 1.38: two waves fetch every pixel row through the L1 and one chunk of prefetch distance leaves the waves
// waiting (PMC: MFMA busy 0.45-0.55 vs 0.68-0.77, wait / active 21 vs 5).  The STAGED form (pixel chunk once per workgroup through LDS, 256-pixel
// tile) is level with them in fp32 (4.22 vs 4.08: it loses where 256-pixel tiles leave a half-empty last round of workgroups) and ahead in bf16
// storage (1.28 vs 1.38 ms; the fused upsample 503 vs 543 us).  unet_tuning.conv1x1_gemm: 0 = auto (default): staged for bf16 storage, the
// implicit-GEMM kernels for fp32; 1 = direct, 2 = staged (both storage types), -1 = never; pixel_shuffle descriptors always run here (staged).
static bool shape_ok(const unet_conv_desc* d) {
    if (d->ks != 1 || d->stride != 1 || d->wp_img_stride != 0 || d->colsum != nullptr || d->colsumsq != nullptr) return false;
    const int kct = d->dtype == UNET_BF16 ? 32 : 16;
    if (d->Cin < kct || d->Cin % kct != 0) return false;
    if (d->cout_begin % 128 != 0) return false;
    const int cols = d->cout_count ? d->cout_count : d->Cout;
    if (cols % 4 != 0 && d->cout_begin + cols != d->Cout) return false;      // (a channel range ends on a 4-channel vector unless it ends the tensor)
    const int pb = tuning_of(d->tuning).plan_batch;          // (the grid test is a plan decision: unet_tuning.plan_batch makes it batch-invariant)
    const long long P = (long long)(pb > 0 ? pb : d->N) * d->OH * d->OW;
    return ((P + 127) / 128) * ((cols + 127) / 128) >= 256;
}
static int form_of(const unet_conv_desc* d) {          // 0: not this kernel, 1: direct, 2: staged
    const int mode = tuning_of(d->tuning).conv1x1_gemm;
    if (d->pixel_shuffle) return mode == 1 ? 1 : 2;
    if (mode < 0) return 0;
    if (mode == 0) return d->dtype == UNET_BF16 ? 2 : 0;
    return mode == 1 ? 1 : 2;
}
bool conv_gemm1x1_applies(const unet_conv_desc* d) { return form_of(d) != 0 && shape_ok(d); }

// unet_conv_desc.pixel_shuffle: validated here (the planner of the implicit-GEMM kernels never sees these descriptors); only this kernel
// stores that way, so a descriptor it does not take is UNSUPPORTED (callers ask unet_conv2d_variant first and keep the two-pass form)
int conv_gemm1x1_ps_check(const unet_conv_desc* d) {
    UNET_CHECK_ARG(d->x && d->wp && d->y && unet::aligned16(d->x) && unet::aligned16(d->wp) && unet::aligned16(d->y), "conv pixel_shuffle: null / unaligned tensor pointer");
    UNET_CHECK_ARG(d->dtype == UNET_F32 || d->dtype == UNET_BF16, "conv: unknown dtype %d", d->dtype);
    const int vec = d->dtype == UNET_BF16 ? 8 : 4;
    UNET_CHECK_ARG(d->ks == 1 && d->stride == 1 && d->kind == UNET_CONV_FWD && d->N > 0 && d->IH > 0 && d->IW > 0 && d->OH == d->IH && d->OW == d->IW && d->Cin > 0 && d->Cout > 0,
                   "conv pixel_shuffle: a 1x1 / stride-1 forward convolution");
    UNET_CHECK_ARG(d->Cout % 64 == 0, "conv pixel_shuffle: Cout = 4 nf with nf a multiple of 16 (got %d)", d->Cout);
    UNET_CHECK_ARG(unet::slice_ok_v(d->x_cs, d->x_co, d->Cin, vec), "conv pixel_shuffle: bad x slice");
    UNET_CHECK_ARG(unet::slice_ok_v(d->y_cs, d->y_co, d->Cout / 4, 4), "conv pixel_shuffle: bad y slice (Cout / 4 channels of a [N, 2 OH, 2 OW] tensor)");
    UNET_CHECK_ARG((d->flags & ~(UNET_CONV_RELU | UNET_CONV_MASK)) == 0, "conv pixel_shuffle: unknown flag bits 0x%x", d->flags);
    UNET_CHECK_ARG(d->bias == nullptr || (((uintptr_t)d->bias) & 3) == 0, "conv pixel_shuffle: bias must be a float pointer");
    UNET_CHECK_ARG(d->res == nullptr && !(d->flags & UNET_CONV_MASK) && d->colsum == nullptr && d->colsumsq == nullptr && d->cout_begin == 0 &&
                   d->cout_count == 0 && d->wp_img_stride == 0, "conv pixel_shuffle: no residual / mask / column sums / channel range / per-image filters");
    UNET_CHECK_ARG((long long)d->N * d->OH * d->OW * 4 < (1ll << 31), "conv pixel_shuffle: more than 2^31 output pixels");
    if (d->ps_tail != nullptr) {
        const int tq = unet::roundup(d->ps_tail_c, 4);
        UNET_CHECK_ARG(unet::aligned16(d->ps_tail) && d->ps_tail_c > 0 && d->ps_tail_cs > 0 && d->ps_tail_cs % 4 == 0 && d->ps_tail_co >= 0 && d->ps_tail_co % 4 == 0 &&
                       d->ps_tail_co + tq <= d->ps_tail_cs, "conv pixel_shuffle: bad ps_tail slice (whole quads: cs, co multiples of 4, co + roundup(c, 4) <= cs)");
        UNET_CHECK_ARG(d->ps_tail_at % 4 == 0 && d->ps_tail_at >= d->y_co + d->Cout / 4 && d->ps_tail_at + tq <= d->y_cs && !(d->dtype == UNET_BF16 && d->y_f32),
                       "conv pixel_shuffle: ps_tail_at must be a multiple of 4 behind the shuffled channels with ps_tail_at + roundup(ps_tail_c, 4) <= y_cs");
    }
    if (!shape_ok(d)) {
        unet::set_error("conv pixel_shuffle: only conv1x1_gemm_kernel stores pixel-shuffled (whole reduction chunks, >= 256 blocks of 128 x 128); "
                        "ask unet_conv2d_variant first and keep conv + unet_shuffle_blur otherwise");
        return UNET_E_UNSUPPORTED;
    }
    return UNET_OK;
}

int conv_gemm1x1(const unet_conv_desc* d, hipStream_t st) {
    G1Args a;
    const bool bf = d->dtype == UNET_BF16;
    const int kct = bf ? 32 : 16;
    a.x = (const char*)d->x; a.wp = (const char*)d->wp; a.bias = d->bias; a.res = (const char*)d->res;
    a.mask = (d->flags & UNET_CONV_MASK) ? (const char*)d->mask : nullptr; a.y = (char*)d->y;
    a.x_cs = d->x_cs; a.x_co = d->x_co; a.res_cs = d->res_cs; a.res_co = d->res_co; a.mask_cs = d->mask_cs; a.mask_co = d->mask_co;
    a.y_cs = d->y_cs; a.y_co = d->y_co;
    a.P = (long long)d->N * d->OH * d->OW;
    a.nchunks = d->Cin / kct;
    a.coutPad = unet::roundup(d->Cout, 128);
    a.n_base = d->cout_begin;
    a.n_end = d->cout_begin + (d->cout_count ? d->cout_count : d->Cout);
    a.Cout = d->Cout;
    a.relu = (d->flags & UNET_CONV_RELU) ? 1 : 0;
    a.y_f32 = bf ? d->y_f32 : 1;
    a.tail = d->pixel_shuffle ? (const char*)d->ps_tail : nullptr;
    a.tail_cs = d->ps_tail_cs; a.tail_co = d->ps_tail_co; a.tail_q = unet::roundup(d->ps_tail_c, 4) / 4; a.tail_at = d->ps_tail_at;
    const bool staged = form_of(d) == 2;
    const int tpix = staged ? 256 : 128;
    a.ntn = unet::cdiv(a.n_end - a.n_base, 128);
    a.ntiles = ((a.P + tpix - 1) / tpix) * a.ntn;
    a.ps = d->pixel_shuffle ? 1 : 0; a.nf = d->Cout / 4; a.H = d->OH; a.W = d->OW;
    UNET_CHECK_ARG(a.ntiles < (1ll << 31) - 8, "conv 1x1: grid too large");
    const unsigned grid = (unsigned)((a.ntiles + 7) / 8 * 8);
    if (staged) {
        if (bf) hipLaunchKernelGGL((conv1x1_gemm_kernel<unsigned short, true>), dim3(grid), dim3(256), 2 * 256 * 96, st, a);
        else hipLaunchKernelGGL((conv1x1_gemm_kernel<float, true>), dim3(grid), dim3(256), 2 * 256 * 96, st, a);
    } else {
        if (bf) hipLaunchKernelGGL((conv1x1_gemm_kernel<unsigned short, false>), dim3(grid), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv1x1_gemm_kernel<float, false>), dim3(grid), dim3(256), 0, st, a);
    }
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}


// the head kernel's launches: 1x1 / stride 1, 9..128 reduction channels in ONE filter image, at most 16 produced channels, bias / ReLU only
bool conv_head1x1_applies(const unet_conv_desc* d) {
    if (tuning_of(d->tuning).conv_head1x1 == 0) return false;
    const int vec = d->dtype == UNET_BF16 ? 8 : 4;
    return d->ks == 1 && d->stride == 1 && d->Cin > 8 && d->Cin <= 128 && d->Cout <= 16 && d->res == nullptr && !(d->flags & UNET_CONV_MASK) &&
           d->colsum == nullptr && d->colsumsq == nullptr && d->cout_begin == 0 && (d->cout_count == 0 || d->cout_count == d->Cout) &&
           d->wp_img_stride == 0 && !d->pixel_shuffle && unet::roundup(d->Cin, vec) <= d->x_cs - d->x_co && d->Cout <= d->y_cs - d->y_co &&
           d->y_co % 4 == 0 && d->y_cs % 4 == 0 && d->x_co % vec == 0 && d->x_cs % vec == 0;
}

int conv_head1x1(const unet_conv_desc* d, hipStream_t st) {
    HeadArgs a;
    const bool bf = d->dtype == UNET_BF16;
    a.x = (const char*)d->x; a.wp = (const char*)d->wp; a.bias = d->bias; a.y = (char*)d->y;
    a.x_cs = d->x_cs; a.x_co = d->x_co; a.y_cs = d->y_cs; a.y_co = d->y_co;
    a.P = (long long)d->N * d->OH * d->OW;
    a.Cin = d->Cin; a.Cout = d->Cout; a.coutPad = unet::roundup(d->Cout, 128);
    a.relu = (d->flags & UNET_CONV_RELU) ? 1 : 0;
    a.y_f32 = bf ? d->y_f32 : 1;
    // pixel tiles per wave and trip: 4 with bf16 storage (16 loads in flight per lane), 2 in fp32 (14 loads; 28 cost half the occupancy) --
    // measured at 16 x 512^2 x 100 -> 5 (scripts/ab_conv_head.py): bf16 203 us (4) / 217 (2) against 257 on conv_bf16_kernel, fp32 475 (4) /
    // 432 (2) against 485 on conv_igemm16_kernel; nontemporal loads cost 20 % in every form.  unet_tuning.conv_head1x1 = 2 swaps the two (A/B).
    const bool swap = tuning_of(d->tuning).conv_head1x1 == 2;
    const int mt = (bf != swap) ? 4 : 2;
    long long blocks = (a.P + 64 * mt - 1) / (64 * mt);
    if (blocks > 256 * 16) blocks = 256 * 16;
    void (*kern)(const HeadArgs);
    if (bf) kern = mt == 4 ? conv1x1_head_kernel<u16, 4, 4> : conv1x1_head_kernel<u16, 4, 2>;
    else kern = mt == 4 ? conv1x1_head_kernel<float, 8, 4> : conv1x1_head_kernel<float, 8, 2>;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, st, a);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

}  // namespace unetconv
