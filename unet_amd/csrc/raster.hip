// Sliding-window prediction over a raster that stays resident in HBM as INTEGERS (BASELINE.json configs[4]).
//
// The reference does this in two steps on the host: create_tiles_unet.split_raster (create_tiles_unet.py:252-434: nodata -> 0,
// slidingwindow windows, emptiness filter, one GeoTIFF per window) and predict.save_predictions (predict.py:191-222: learn.predict per
// tile file; :257-334: sum of softmax probabilities + hit counter -> divide -> argmax).  Here the raster is uploaded once, and
//   unet_raster_nodata_zero      create_tiles_unet.py:344-352  numpy_image[:, (numpy_image == nodata).any(axis=0)] = 0
//   unet_window_nonzero          create_tiles_unet.py:379      np.sum(crop != 0) of every window (the max_empty filter)
//   unet_window_gather           data.py:18-28 + utils.py:248-249,288-289 + IntToFloatTensor: window -> int32 -> float32 -> /255 [/255],
//                                written straight into the NHWC input buffers of the network (no NCHW fp32 tile is ever made)
//   unet_mosaic_accumulate_windows  predict.py:193-203 softmax + predict.py:284-292 placement of a whole BATCH of windows in one launch,
//                                contributions to one mosaic pixel added in window order (deterministic: no atomics)
//   unet_mosaic_finalize_rows    predict.py:306-334 divide by the hit counter, argmax (classification) / -9999 fill (regression)
// All of it is HBM-bound byte / float traffic: one pass over the data each, coalesced along x.
#include "common.h"

using namespace unet;

namespace {

template <typename S> __device__ __forceinline__ int as_i32(S v) { return (int)v; }                 // data.py:24: cast through int32
template <> __device__ __forceinline__ int as_i32<float>(float v) { return (int)v; }                // truncation toward zero, as numpy astype
template <typename S> __device__ __forceinline__ bool eq_nodata(S v, double nd) { return (double)v == nd; }

__device__ __forceinline__ void st_act(float* p, float v) { *p = v; }
__device__ __forceinline__ void st_act(unet_bf16* p, float v) { *p = __builtin_bit_cast(unet_bf16, (__bf16)v); }

template <typename S>
__global__ __launch_bounds__(256) void nodata_zero_kernel(S* __restrict__ r, int Cb, long long HW, double nodata) {
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < HW; p += (long long)gridDim.x * blockDim.x) {
        bool bad = false;
        for (int c = 0; c < Cb; ++c) bad |= eq_nodata(r[(size_t)c * HW + p], nodata);
        if (bad)
            for (int c = 0; c < Cb; ++c) r[(size_t)c * HW + p] = (S)0;
    }
}

// one workgroup per (window, slice of rows); integer atomics: the count does not depend on the order
template <typename S>
__global__ __launch_bounds__(256) void window_nonzero_kernel(const S* __restrict__ r, int Cb, long long band_stride, int row_stride,
                                                             const int* __restrict__ win, int th, int tw, int slices,
                                                             unsigned long long* __restrict__ counts) {
    const int j = blockIdx.x / slices, sl = blockIdx.x % slices;
    const int y0 = win[4 * j], x0 = win[4 * j + 1];
    const long long total = (long long)Cb * th * tw;
    unsigned int n = 0;
    for (long long i = (long long)sl * 256 + threadIdx.x; i < total; i += (long long)slices * 256) {
        const int x = (int)(i % tw);
        const long long t = i / tw;
        const int y = (int)(t % th), c = (int)(t / th);
        n += r[(size_t)c * band_stride + (size_t)(y0 + y) * row_stride + x0 + x] != (S)0;
    }
    for (int o = 32; o > 0; o >>= 1) n += __shfl_down(n, o);
    if ((threadIdx.x & 63) == 0 && n) atomicAdd(counts + j, (unsigned long long)n);
}

// x[j][y][x][co + c] = float(int32(src[c][y0 + y][x0 + x])) / 255 [/ 255]; lanes above the band count are left alone
template <typename S, typename T>
__global__ __launch_bounds__(256) void window_gather_kernel(const S* __restrict__ r, int Cb, long long src_stride, long long band_stride,
                                                            int row_stride, const int* __restrict__ win, int n, int th, int tw,
                                                            int div2, T* __restrict__ x, int x_cs, int x_co) {
    const long long per = (long long)th * tw, total = per * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i / per);
        const long long p = i - (long long)j * per;
        const int y = (int)(p / tw), xx = (int)(p - (long long)y * tw);
        const int y0 = win[4 * j], x0 = win[4 * j + 1], src = win[4 * j + 2];
        const S* s = r + (size_t)src * src_stride + (size_t)(y0 + y) * row_stride + x0 + xx;
        T* o = x + (size_t)i * x_cs + x_co;
        for (int c = 0; c < Cb; ++c) {
            float v = (float)as_i32(s[(size_t)c * band_stride]);
            if (div2) v = __fdiv_rn(v, 255.0f);
            st_act(o + c, __fdiv_rn(v, 255.0f));
        }
    }
}

constexpr int MAXWIN = 64;      // windows per launch of the batched accumulate
constexpr int MAXC = 64;      // class count bound shared with the loss kernels (CE_MAXC)

// mode 0: softmax over the C logits (classification)   1: raw values (regression, predict.py:195-197)
// Thread (j, y, x) owns mosaic pixel (Y, X) of window j only when no earlier window of this launch covers it; it then adds the
// contributions of windows j, j+1, ... that cover (Y, X) in that order: every mosaic pixel is updated by one thread, in window order.
__global__ __launch_bounds__(256) void mosaic_acc_windows_kernel(const float* __restrict__ z, int z_cs, int z_co, int C, int th, int tw,
                                                                 const int* __restrict__ win, int n, int oy, int ox, int mode,
                                                                 float* __restrict__ mosaic, int32_t* __restrict__ count, int MH, int MW,
                                                                 int row_lo, int row_hi) {
    __shared__ int sy[MAXWIN], sx[MAXWIN];
    if (threadIdx.x < n) {
        sy[threadIdx.x] = win[4 * threadIdx.x] - oy;
        sx[threadIdx.x] = win[4 * threadIdx.x + 1] - ox;
    }
    __syncthreads();
    const long long per = (long long)th * tw, total = per * n, plane = (long long)MH * MW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i / per);
        const long long p = i - (long long)j * per;
        const int ty = (int)(p / tw), tx = (int)(p - (long long)ty * tw);
        const int Y = sy[j] + ty, X = sx[j] + tx;
        if (Y < row_lo || Y >= row_hi || Y < 0 || Y >= MH || X < 0 || X >= MW) continue;
        bool first = true;
        for (int k = 0; k < j; ++k) first &= !(Y >= sy[k] && Y < sy[k] + th && X >= sx[k] && X < sx[k] + tw);
        if (!first) continue;
        const size_t m = (size_t)Y * MW + X;
        int hits = 0;
        for (int k = j; k < n; ++k) {
            if (!(Y >= sy[k] && Y < sy[k] + th && X >= sx[k] && X < sx[k] + tw)) continue;
            const float* zp = z + ((size_t)k * per + (size_t)(Y - sy[k]) * tw + (X - sx[k])) * z_cs + z_co;
            if (mode == 0) {        // exactly the arithmetic of softmax_argmax_kernel (elementwise.hip): a tile's probabilities are the same numbers
                float mx = zp[0];
                for (int c = 1; c < C; ++c) mx = fmaxf(mx, zp[c]);
                float s = 0.f;
                for (int c = 0; c < C; ++c) s += expf(zp[c] - mx);
                for (int c = 0; c < C; ++c) {
                    float* q = mosaic + (size_t)c * plane + m;
                    *q = __fadd_rn(*q, expf(zp[c] - mx) / s);
                }
            } else {
                for (int c = 0; c < C; ++c) {
                    float* q = mosaic + (size_t)c * plane + m;
                    *q = __fadd_rn(*q, zp[c]);
                }
            }
            ++hits;
        }
        count[m] += hits;
    }
}

// rows [row0, row0 + nrows) of a [C][MH][MW] mosaic: mean over the hits, argmax (first maximum, as numpy), optional fill where nothing
// was placed (regression: predict.py:312-315)
__global__ __launch_bounds__(256) void mosaic_fin_rows_kernel(float* __restrict__ mosaic, const int32_t* __restrict__ count, int C, int MH,
                                                              int MW, int row0, int nrows, uint8_t* __restrict__ amax, int has_fill,
                                                              float fill) {
    const long long total = (long long)nrows * MW, plane = (long long)MH * MW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const size_t m = (size_t)row0 * MW + i;
        const int cnt = count[m];
        float best = -INFINITY;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            float v = mosaic[(size_t)c * plane + m];
            if (cnt > 0) { v = v / (float)cnt; mosaic[(size_t)c * plane + m] = v; }
            else if (has_fill) { v = fill; mosaic[(size_t)c * plane + m] = v; }
            if (v > best) { best = v; bi = c; }
        }
        if (amax) amax[i] = (uint8_t)bi;
    }
}


// ---- training feed (reference train.py:345 -> data.py:18-28 open_npy, utils.py:239-295 the batch transform, IntToFloatTensor) ----
// A batch arrives as the INTEGERS of its tile files (pinned staging -> one asynchronous copy); value scaling, the int64 widening of the
// mask and the default flip augmentation happen here.  Image j is mirrored along x when bit j of hflip is set, along y for vflip
// (utils.py:239-291 applies the pipeline to the first ceil(B * n_transform_imgs) - B images: the host decides the bits).
template <typename S>
__global__ __launch_bounds__(256) void tiles_stage_kernel(const S* __restrict__ src, int n, int Cb, int H, int W, int div2,
                                                          unsigned long long hflip, unsigned long long vflip, float* __restrict__ dst) {
    const long long plane = (long long)H * W, per = plane * Cb, total = per * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i / per);
        const long long r = i - (long long)j * per;
        const int c = (int)(r / plane);
        const long long p = r - (long long)c * plane;
        int y = (int)(p / W), x = (int)(p - (long long)y * W);
        if ((vflip >> j) & 1ull) y = H - 1 - y;
        if ((hflip >> j) & 1ull) x = W - 1 - x;
        float v = (float)as_i32(src[(size_t)j * per + (size_t)c * plane + (size_t)y * W + x]);
        if (div2) v = __fdiv_rn(v, 255.0f);
        dst[i] = __fdiv_rn(v, 255.0f);
    }
}

template <typename S, typename D> __device__ __forceinline__ D mask_cast(S v) { return (D)v; }      // numpy astype: truncation toward zero

template <typename S, typename D>
__global__ __launch_bounds__(256) void mask_stage_kernel(const S* __restrict__ src, int n, int H, int W, unsigned long long hflip,
                                                         unsigned long long vflip, D* __restrict__ dst) {
    const long long plane = (long long)H * W, total = plane * n;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(i / plane);
        const long long p = i - (long long)j * plane;
        int y = (int)(p / W), x = (int)(p - (long long)y * W);
        if ((vflip >> j) & 1ull) y = H - 1 - y;
        if ((hflip >> j) & 1ull) x = W - 1 - x;
        dst[i] = mask_cast<S, D>(src[(size_t)j * plane + (size_t)y * W + x]);
    }
}

// DiceMulti counters (fastai metrics.py DiceMulti.accumulate; reference train.py:196): counts[0][c] += #(pred == c && targ == c),
// counts[1][c] += #(pred == c), counts[2][c] += #(clamp(targ) == c).  Integer atomics (LDS histogram per workgroup, one global add per
// class): the result does not depend on the order.
__global__ __launch_bounds__(256) void dice_counts_kernel(const long long* __restrict__ pred, const long long* __restrict__ targ, long long P,
                                                          int C, unsigned long long* __restrict__ counts) {
    __shared__ unsigned int h[3 * MAXC];
    for (int k = threadIdx.x; k < 3 * C; k += blockDim.x) h[k] = 0;
    __syncthreads();
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
        const long long p = pred[i], t = targ[i];
        const int tc = t < 0 ? 0 : (t >= C ? C - 1 : (int)t);
        if (p >= 0 && p < C) {
            atomicAdd(h + C + (int)p, 1u);
            if (p == t) atomicAdd(h + (int)p, 1u);
        }
        atomicAdd(h + 2 * C + tc, 1u);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < 3 * C; k += blockDim.x)
        if (h[k]) atomicAdd(counts + k, (unsigned long long)h[k]);
}

}  // namespace

#define ST ((hipStream_t)stream)

#define RASTER_DISPATCH(rtype, CALL)                                   \
    switch (rtype) {                                                   \
        case UNET_RASTER_U8: { using S = uint8_t; CALL; break; }       \
        case UNET_RASTER_U16: { using S = uint16_t; CALL; break; }     \
        case UNET_RASTER_I16: { using S = int16_t; CALL; break; }      \
        case UNET_RASTER_I32: { using S = int32_t; CALL; break; }      \
        case UNET_RASTER_F32: { using S = float; CALL; break; }        \
        default: unet::set_error("raster: unknown sample type %d", rtype); return UNET_E_BADARG; \
    }

extern "C" int unet_raster_nodata_zero(void* raster, int rtype, int bands, long long pixels, double nodata, void* stream) {
    UNET_CHECK_ARG(raster && bands > 0 && pixels > 0, "raster_nodata_zero: bad args");
    RASTER_DISPATCH(rtype, hipLaunchKernelGGL((nodata_zero_kernel<S>), dim3(ew_grid(pixels, 256)), dim3(256), 0, ST, (S*)raster, bands, pixels,
                                              nodata));
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_window_nonzero(const void* raster, int rtype, int bands, long long band_stride, int row_stride, const int32_t* windows,
                                   int n, int th, int tw, unsigned long long* counts, void* stream) {
    UNET_CHECK_ARG(raster && windows && counts && bands > 0 && n > 0 && th > 0 && tw > 0 && row_stride >= tw, "window_nonzero: bad args");
    UNET_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(unsigned long long) * n, ST));
    long long slices = ((long long)bands * th * tw + 16383) / 16384;        // ~64 samples per thread
    if (slices > 64) slices = 64;
    if ((long long)n * slices > 0x7fffffffLL) { unet::set_error("window_nonzero: too many windows"); return UNET_E_BADARG; }
    RASTER_DISPATCH(rtype, hipLaunchKernelGGL((window_nonzero_kernel<S>), dim3((unsigned)(n * slices)), dim3(256), 0, ST, (const S*)raster, bands,
                                              band_stride, row_stride, windows, th, tw, (int)slices, counts));
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_window_gather(const void* src, int rtype, int bands, long long src_stride, long long band_stride, int row_stride,
                                  const int32_t* windows, int n, int th, int tw, int div255_twice, void* x, int x_cs, int x_co, int dtype,
                                  void* stream) {
    UNET_CHECK_ARG(src && windows && x && bands > 0 && n > 0 && th > 0 && tw > 0 && row_stride >= tw, "window_gather: bad args");
    UNET_CHECK_ARG(x_co >= 0 && x_co + bands <= x_cs && (dtype == UNET_F32 || dtype == UNET_BF16), "window_gather: bad slice / dtype");
    const int grid = ew_grid((long long)n * th * tw, 256);
    if (dtype == UNET_F32) {
        RASTER_DISPATCH(rtype, hipLaunchKernelGGL((window_gather_kernel<S, float>), dim3(grid), dim3(256), 0, ST, (const S*)src, bands, src_stride,
                                                  band_stride, row_stride, windows, n, th, tw, div255_twice, (float*)x, x_cs, x_co));
    } else {
        RASTER_DISPATCH(rtype, hipLaunchKernelGGL((window_gather_kernel<S, unet_bf16>), dim3(grid), dim3(256), 0, ST, (const S*)src, bands,
                                                  src_stride, band_stride, row_stride, windows, n, th, tw, div255_twice, (unet_bf16*)x, x_cs,
                                                  x_co));
    }
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_mosaic_accumulate_windows(const float* z, int z_cs, int z_co, int C, int th, int tw, const int32_t* windows, int n,
                                              int origin_y, int origin_x, int mode, float* mosaic, int32_t* count, int MH, int MW, int row_lo,
                                              int row_hi, void* stream) {
    UNET_CHECK_ARG(z && windows && mosaic && count && C > 0 && C <= MAXC && th > 0 && tw > 0 && MH > 0 && MW > 0 && n > 0,
                   "mosaic_accumulate_windows: bad args");
    UNET_CHECK_ARG(z_co >= 0 && z_co + C <= z_cs && (mode == 0 || mode == 1), "mosaic_accumulate_windows: bad slice / mode");
    for (int b = 0; b < n; b += MAXWIN) {      // launches are stream ordered: later windows land on top of earlier ones
        const int nb = n - b < MAXWIN ? n - b : MAXWIN;
        hipLaunchKernelGGL(mosaic_acc_windows_kernel, dim3(ew_grid((long long)nb * th * tw, 256)), dim3(256), 0, ST,
                           z + (size_t)b * th * tw * z_cs, z_cs, z_co, C, th, tw, windows + 4 * b, nb, origin_y, origin_x, mode, mosaic, count,
                           MH, MW, row_lo, row_hi);
        UNET_CHECK_LAUNCH();
    }
    return UNET_OK;
}

extern "C" int unet_mosaic_finalize_rows(float* mosaic, const int32_t* count, int C, int MH, int MW, int row0, int nrows, uint8_t* argmax,
                                         const float* fill_host, void* stream) {
    UNET_CHECK_ARG(mosaic && count && C > 0 && MH > 0 && MW > 0 && row0 >= 0 && nrows > 0 && row0 + nrows <= MH, "mosaic_finalize_rows: bad args");
    hipLaunchKernelGGL(mosaic_fin_rows_kernel, dim3(ew_grid((long long)nrows * MW, 256)), dim3(256), 0, ST, mosaic, count, C, MH, MW, row0, nrows,
                       argmax, fill_host ? 1 : 0, fill_host ? *fill_host : 0.f);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_tiles_stage(const void* src, int rtype, int n, int bands, int H, int W, int div255_twice, unsigned long long hflip,
                                unsigned long long vflip, float* dst_nchw, void* stream) {
    UNET_CHECK_ARG(src && dst_nchw && n > 0 && n <= 64 && bands > 0 && H > 0 && W > 0, "tiles_stage: bad args (1..64 images per call)");
    const int grid = ew_grid((long long)n * bands * H * W, 256);
    RASTER_DISPATCH(rtype, hipLaunchKernelGGL((tiles_stage_kernel<S>), dim3(grid), dim3(256), 0, ST, (const S*)src, n, bands, H, W, div255_twice,
                                              hflip, vflip, dst_nchw));
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_mask_stage(const void* src, int rtype, int n, int H, int W, unsigned long long hflip, unsigned long long vflip, void* dst,
                               int dst_f32, void* stream) {
    UNET_CHECK_ARG(src && dst && n > 0 && n <= 64 && H > 0 && W > 0 && (dst_f32 == 0 || dst_f32 == 1), "mask_stage: bad args (1..64 masks per call)");
    const int grid = ew_grid((long long)n * H * W, 256);
    if (dst_f32) {
        RASTER_DISPATCH(rtype, hipLaunchKernelGGL((mask_stage_kernel<S, float>), dim3(grid), dim3(256), 0, ST, (const S*)src, n, H, W, hflip, vflip,
                                                  (float*)dst));
    } else {
        RASTER_DISPATCH(rtype, hipLaunchKernelGGL((mask_stage_kernel<S, long long>), dim3(grid), dim3(256), 0, ST, (const S*)src, n, H, W, hflip,
                                                  vflip, (long long*)dst));
    }
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_dice_counts(const int64_t* pred, const int64_t* targ, long long P, int C, unsigned long long* counts, void* stream) {
    UNET_CHECK_ARG(pred && targ && counts && P > 0 && C > 0 && C <= MAXC, "dice_counts: bad args");
    // a workgroup's LDS counters are 32 bit: at most 2^31 pixels per workgroup
    hipLaunchKernelGGL(dice_counts_kernel, dim3(ew_grid(P, 256)), dim3(256), 0, ST, (const long long*)pred, (const long long*)targ, P, C, counts);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
