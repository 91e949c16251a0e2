// HBM-bound kernels of the U-Net step for gfx950: BatchNorm statistics / apply /
// backward, pooling, pixel-shuffle + blur data movement, layout conversion,
// per-pixel weighted cross-entropy, softmax/argmax, fused Adam.
// All tensors fp32 NHWC slices (ptr, channel stride cs, channel offset co);
// every access is a 16-byte float4 with consecutive lanes on consecutive channels
// (coalesced), grid-stride loops capped at 2048 workgroups.
//
// Reference call sites these replace: fastai layers.py (BatchNorm, ResBlock.forward,
// PixelShuffle_ICNR), vision/models/unet.py (UnetBlock.forward), losses.py
// (CrossEntropyLossFlat), optimizer.py (Adam) as driven by reference train.py:128-250
// and predict.py:193-232.

#include <math.h>
#include <stdlib.h>
#include <stdarg.h>

#include "common.h"

namespace unet {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace unet

extern "C" int unet_abi_version(void) { return UNET_ABI_VERSION; }
extern "C" const char* unet_last_error(void) { return unet::g_err; }

namespace {

using unet::cdiv;
using unet::ew_grid;
using unet::roundup;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
// bf16 storage (unet_bf16 = bfloat16 bit pattern): 4 channels = one 8-byte access; arithmetic is always fp32
__device__ __forceinline__ float4 ld4(const unet_bf16* p) {
    const uint2 u = *reinterpret_cast<const uint2*>(p);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ void st4(unet_bf16* p, float4 v) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    const bf16x4 h = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};          // round to nearest even (v_cvt_pk_bf16_f32)
    *reinterpret_cast<uint2*>(p) = __builtin_bit_cast(uint2, h);
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ float ld1(const unet_bf16* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void st1(unet_bf16* p, float v) { *p = __builtin_bit_cast(unet_bf16, (__bf16)v); }
__device__ __forceinline__ float4 f4(float v) { return make_float4(v, v, v, v); }
__device__ __forceinline__ float4 operator+(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 operator-(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ float4 operator*(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
__device__ __forceinline__ float4 relu4(float4 a) { return make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f)); }
__device__ __forceinline__ float4 gate4(float4 g, float4 ref) {
    return make_float4(ref.x > 0.f ? g.x : 0.f, ref.y > 0.f ? g.y : 0.f, ref.z > 0.f ? g.z : 0.f, ref.w > 0.f ? g.w : 0.f);
}

// Grid-stride walk over (pixel, channel-quad) pairs without a 64-bit division per element: the pair is split once, then advanced
// by the constant stride (these kernels run on tensors of a few MB, where the per-element index arithmetic, not HBM, set the time).
struct QuadWalk {
    long long p, dp;
    int c4, dc, C4;
    __device__ __forceinline__ QuadWalk(int C4_) : C4(C4_) {
        const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x, stride = (long long)gridDim.x * blockDim.x;
        p = i0 / C4_; c4 = (int)(i0 - p * C4_);
        dp = stride / C4_; dc = (int)(stride - dp * C4_);
    }
    __device__ __forceinline__ void next() {
        p += dp; c4 += dc;
        if (c4 >= C4) { c4 -= C4; ++p; }
    }
};

// N consecutive elements (N * sizeof(T) a multiple of 16 bytes, 16-byte aligned) <-> floats
template <int N> __device__ __forceinline__ void ldn(const float* p, float (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; k += 4) { const float4 t = ld4(p + k); v[k] = t.x; v[k + 1] = t.y; v[k + 2] = t.z; v[k + 3] = t.w; }
}
template <int N> __device__ __forceinline__ void ldn(const unet_bf16* p, float (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; k += 8) {
        const uint4 u = *reinterpret_cast<const uint4*>(p + k);
        v[k] = __uint_as_float(u.x << 16); v[k + 1] = __uint_as_float(u.x & 0xffff0000u);
        v[k + 2] = __uint_as_float(u.y << 16); v[k + 3] = __uint_as_float(u.y & 0xffff0000u);
        v[k + 4] = __uint_as_float(u.z << 16); v[k + 5] = __uint_as_float(u.z & 0xffff0000u);
        v[k + 6] = __uint_as_float(u.w << 16); v[k + 7] = __uint_as_float(u.w & 0xffff0000u);
    }
}
template <int N> __device__ __forceinline__ void stn(float* p, const float (&v)[N]) {
#pragma unroll
    for (int k = 0; k < N; k += 4) st4(p + k, make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]));
}
template <int N> __device__ __forceinline__ void stn(unet_bf16* p, const float (&v)[N]) {
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#pragma unroll
    for (int k = 0; k < N; k += 8) {
        const bf16x8 h = {(__bf16)v[k], (__bf16)v[k + 1], (__bf16)v[k + 2], (__bf16)v[k + 3], (__bf16)v[k + 4], (__bf16)v[k + 5], (__bf16)v[k + 6], (__bf16)v[k + 7]};
        *reinterpret_cast<uint4*>(p + k) = __builtin_bit_cast(uint4, h);
    }
}


// --------------------------------------------------------------------------
// Per-channel reduction over pixels.  F(p, c4) -> two float4 values; result
// planes out0[rows][Cp], out1[rows][Cp] (one row per workgroup).
// Thread layout: tx = channel quad (TC lanes), ty = pixel sub-index (256/TC).
// --------------------------------------------------------------------------
template <typename F>
__device__ __forceinline__ void channel_reduce(F f, long long P, int C4, int TC, float* out0, float* out1, int Cp) {
    __shared__ float4 sm0[256];
    __shared__ float4 sm1[256];
    const int tx = threadIdx.x % TC, ty = threadIdx.x / TC, PY = 256 / TC;
    for (int cbase = 0; cbase < C4; cbase += TC) {
        const int c4 = cbase + tx;
        float4 s0 = f4(0.f), s1 = f4(0.f);
        if (c4 < C4) {
            // four pixels per trip: the (up to three) loads of each pixel are independent, so 12 requests are in flight per thread
            // instead of 3 (a thread walks 32-64 pixels: one HBM round trip per pixel made these kernels latency bound)
            const long long step = (long long)gridDim.x * PY;
            long long p = (long long)blockIdx.x * PY + ty;
            for (; p + 3 * step < P; p += 4 * step) {
                float4 a0, b0, a1, b1, a2, b2, a3, b3;
                f(p, c4, a0, b0);
                f(p + step, c4, a1, b1);
                f(p + 2 * step, c4, a2, b2);
                f(p + 3 * step, c4, a3, b3);
                s0 = s0 + ((a0 + a1) + (a2 + a3));
                s1 = s1 + ((b0 + b1) + (b2 + b3));
            }
            for (; p < P; p += step) {
                float4 v0, v1;
                f(p, c4, v0, v1);
                s0 = s0 + v0;
                s1 = s1 + v1;
            }
        }
        sm0[threadIdx.x] = s0;
        sm1[threadIdx.x] = s1;
        __syncthreads();
        if (ty == 0 && c4 < C4) {
            for (int j = 1; j < PY; ++j) {
                s0 = s0 + sm0[j * TC + tx];
                s1 = s1 + sm1[j * TC + tx];
            }
            st4(out0 + (size_t)blockIdx.x * Cp + 4 * c4, s0);
            if (out1 != nullptr) st4(out1 + (size_t)blockIdx.x * Cp + 4 * c4, s1);
        }
        __syncthreads();
    }
}

// bf16 8-channel (16-byte) forms of the BatchNorm apply / backward kernels; UNET_EW_OCT=0 keeps the quad forms (A/B inside one process)
static bool use_oct() {
    static const int on = [] { const char* e = getenv("UNET_EW_OCT"); return (e == nullptr || e[0] != '0') ? 1 : 0; }();
    return on != 0;
}

static int pick_tc(int C4) {
    int tc = 1;
    while (tc < C4 && tc < 64) tc <<= 1;
    return tc;
}

static int stats_rows(long long P) {
    // one partial row (= one workgroup) per 32 pixels, at most 512: the 32 x 32 and 16 x 16 stages of a batch of 16 -- and every stage of a
    // batch of 2 (BASELINE configs[0]) -- get 4 x the workgroups of the 128-pixel rule round 3 ended on, 2-4 dependent trips per thread instead
    // of 8-16 on a partly filled chip.  (Round 3 measured +0.9 % on the bf16 step and took it back because single-draw flip-noise bars tripped
    // at 2.001e-3 against 2e-3; those bars are medians over three draws now)
    long long r = (P + 31) / 32;
    if (r < 1) r = 1;
    if (r > 512) r = 512;
    return (int)r;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, int cs, int co, long long P, int C4, int TC,
                                                       float* out0, float* out1, int Cp) {
    channel_reduce(
        [&](long long p, int c4, float4& v0, float4& v1) {
            const float4 v = ld4(x + (size_t)p * cs + co + 4 * c4);
            v0 = v;
            v1 = v * v;
        },
        P, C4, TC, out0, out1, Cp);
}

__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int cs, int co, long long P, int C4, int TC,
                                                     float* out0, int Cp) {
    channel_reduce(
        [&](long long p, int c4, float4& v0, float4& v1) {
            v0 = ld4(x + (size_t)p * cs + co + 4 * c4);
            v1 = f4(0.f);
        },
        P, C4, TC, out0, nullptr, Cp);
}

// per-channel partial sums of x * y (self-attention: dL/dgamma = sum O * dout)
template <typename T>
__global__ __launch_bounds__(256) void dot_kernel(const T* __restrict__ x, int x_cs, int x_co, const T* __restrict__ y, int y_cs,
                                                  int y_co, long long P, int C4, int TC, float* out0, int Cp) {
    channel_reduce(
        [&](long long p, int c4, float4& v0, float4& v1) {
            v0 = ld4(x + (size_t)p * x_cs + x_co + 4 * c4) * ld4(y + (size_t)p * y_cs + y_co + 4 * c4);
            v1 = f4(0.f);
        },
        P, C4, TC, out0, nullptr, Cp);
}

// one workgroup: fp64 sum of n floats in a fixed order -> out[0]
__global__ __launch_bounds__(256) void sum_all_kernel(const float* __restrict__ part, long long n, float* __restrict__ out) {
    __shared__ double sh[256];
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;       // four loads in flight per thread (196 608 partials behind SelfAttention(384): 180 us as one chain)
    long long i = threadIdx.x;
    for (; i + 768 < n; i += 1024) {
        s0 += (double)part[i]; s1 += (double)part[i + 256]; s2 += (double)part[i + 512]; s3 += (double)part[i + 768];
    }
    for (; i < n; i += 256) s0 += (double)part[i];
    sh[threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if ((int)threadIdx.x < k) sh[threadIdx.x] += sh[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)sh[0];
}

// Sum rows of one or two [rows][stride] planes for 32 consecutive channels per workgroup:
// 256 threads = 32 channels (tx, coalesced 128-B row segments) x 8 row lanes (ty), fp64 accumulation,
// one LDS hop.  Result for channel c lands in thread (tx = c % 32, ty = 0).
__device__ __forceinline__ void rows_reduce2(const float* __restrict__ p0, const float* __restrict__ p1, int rows, int stride, int C,
                                             double& s, double& q, int& c_out) {
    __shared__ double sh0[256];
    __shared__ double sh1[256];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int c = blockIdx.x * 32 + tx;
    double a = 0.0, b = 0.0;
    if (c < C) {
        int r = ty;
        for (; r + 24 < rows; r += 32) {          // four independent loads per plane and trip (the chain of up to 64 dependent loads set the time)
            const float a0 = p0[(size_t)r * stride + c], a1 = p0[(size_t)(r + 8) * stride + c], a2 = p0[(size_t)(r + 16) * stride + c],
                        a3 = p0[(size_t)(r + 24) * stride + c];
            a += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
            if (p1 != nullptr) {
                const float b0 = p1[(size_t)r * stride + c], b1 = p1[(size_t)(r + 8) * stride + c], b2 = p1[(size_t)(r + 16) * stride + c],
                            b3 = p1[(size_t)(r + 24) * stride + c];
                b += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
            }
        }
        for (; r < rows; r += 8) {
            a += (double)p0[(size_t)r * stride + c];
            if (p1 != nullptr) b += (double)p1[(size_t)r * stride + c];
        }
    }
    sh0[threadIdx.x] = a;
    sh1[threadIdx.x] = b;
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int j = 1; j < 8; ++j) { a += sh0[j * 32 + tx]; b += sh1[j * 32 + tx]; }
    }
    s = a; q = b; c_out = (ty == 0 && c < C) ? c : -1;
}

__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ part, int rows, int Cp, int C, float* __restrict__ out) {
    double s, q; int c;
    rows_reduce2(part, nullptr, rows, Cp, C, s, q, c);
    if (c >= 0) out[c] = (float)s;
}

__global__ __launch_bounds__(256) void bn_finalize_kernel(const float* __restrict__ psum, const float* __restrict__ psumsq, int rows,
                                                          int rstride, double count, int C, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float* running_mean, float* running_var,
                                                          float momentum, float eps, float* scale, float* shift, float* save_mean,
                                                          float* save_invstd, long long* batches_tracked) {
    if (batches_tracked != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *batches_tracked += 1;   // BatchNorm2d.num_batches_tracked
    double s, q; int c;
    rows_reduce2(psum, psumsq, rows, rstride, C, s, q, c);
    if (c < 0) return;
    const double mean = s / count;
    double var = q / count - mean * mean;
    if (var < 0.0) var = 0.0;
    const float invstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    const float sc = g * invstd;
    scale[c] = sc;
    shift[c] = b - (float)mean * sc;
    if (save_mean) save_mean[c] = (float)mean;
    if (save_invstd) save_invstd[c] = invstd;
    if (running_mean) running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    if (running_var) {
        const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
        running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps, int C,
                                      float* scale, float* shift) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float invstd = 1.f / sqrtf(rv[c] + eps);
    const float sc = (gamma ? gamma[c] : 1.f) * invstd;
    scale[c] = sc;
    shift[c] = (beta ? beta[c] : 0.f) - rm[c] * sc;
}

template <typename T>
__global__ __launch_bounds__(256) void affine_act_kernel(const T* __restrict__ x, int x_cs, int x_co, const float* __restrict__ scale,
                                                         const float* __restrict__ shift, const T* __restrict__ x2, int x2_cs,
                                                         int x2_co, const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                         T* __restrict__ y, int y_cs, int y_co, long long P, int C4, int relu) {
    for (QuadWalk w(C4); w.p < P; w.next()) {
        const long long p = w.p;
        const int c = 4 * w.c4;
        float4 v = ld4(x + (size_t)p * x_cs + x_co + c);
        if (scale != nullptr) v = v * ld4(scale + c) + ld4(shift + c);
        if (x2 != nullptr) {
            float4 w = ld4(x2 + (size_t)p * x2_cs + x2_co + c);
            if (scale2 != nullptr) w = w * ld4(scale2 + c) + ld4(shift2 + c);
            v = v + w;
        }
        if (relu) v = relu4(v);
        st4(y + (size_t)p * y_cs + y_co + c, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, int d_cs, int d_co,
                                                            const T* __restrict__ out, int o_cs, int o_co,
                                                            const T* __restrict__ x, int x_cs, int x_co,
                                                            const float* __restrict__ mean, const float* __restrict__ invstd,
                                                            long long P, int C4, int TC, float* out0, float* out1, int Cp) {
    channel_reduce(
        [&](long long p, int c4, float4& v0, float4& v1) {
            float4 g = ld4(dout + (size_t)p * d_cs + d_co + 4 * c4);
            if (out != nullptr) g = gate4(g, ld4(out + (size_t)p * o_cs + o_co + 4 * c4));
            const float4 xh = (ld4(x + (size_t)p * x_cs + x_co + 4 * c4) - ld4(mean + 4 * c4)) * ld4(invstd + 4 * c4);
            v0 = g;
            v1 = g * xh;
        },
        P, C4, TC, out0, out1, Cp);
}

__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ part, int rows, int Cp, double count, int C,
                                                              float* dgamma, float* dbeta, float* c1, float* c2, int accumulate) {
    double s, q; int c;
    rows_reduce2(part, part + (size_t)rows * Cp, rows, Cp, C, s, q, c);
    if (c < 0) return;
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
    c1[c] = (float)(s / count);
    c2[c] = (float)(q / count);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, int d_cs, int d_co,
                                                           const T* __restrict__ out, int o_cs, int o_co,
                                                           const T* __restrict__ x, int x_cs, int x_co,
                                                           const float* __restrict__ mean, const float* __restrict__ invstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ c1,
                                                           const float* __restrict__ c2, T* __restrict__ dx, int dx_cs, int dx_co,
                                                           T* gout, int g_cs, int g_co, int g_acc, long long P, int C4) {
    for (QuadWalk w(C4); w.p < P; w.next()) {
        const long long p = w.p;
        const int c = 4 * w.c4;
        float4 g = ld4(dout + (size_t)p * d_cs + d_co + c);
        if (out != nullptr) g = gate4(g, ld4(out + (size_t)p * o_cs + o_co + c));
        const float4 is = ld4(invstd + c);
        const float4 xh = (ld4(x + (size_t)p * x_cs + x_co + c) - ld4(mean + c)) * is;
        const float4 gm = gamma ? ld4(gamma + c) : f4(1.f);
        const float4 r = gm * is * (g - ld4(c1 + c) - xh * ld4(c2 + c));
        st4(dx + (size_t)p * dx_cs + dx_co + c, r);
        if (gout != nullptr) {
            T* gp = gout + (size_t)p * g_cs + g_co + c;
            st4(gp, g_acc ? (ld4(gp) + g) : g);
        }
    }
}

// ---- bf16 storage, 8 channels (16 bytes) per access: the quad forms above move 8 bytes per lane with bf16 tensors, i.e. twice the
// load / store instructions per byte of the fp32 path they were written for (bn_bwd_reduce ran SLOWER in bf16 than in fp32 at 16 x 256^2 x 64).
// Same arithmetic per element; used when every channel count / stride / offset involved is a multiple of 8.
__global__ __launch_bounds__(256) void affine_act_oct_kernel(const unet_bf16* __restrict__ x, int x_cs, int x_co, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const unet_bf16* __restrict__ x2, int x2_cs,
                                                             int x2_co, const float* __restrict__ scale2, const float* __restrict__ shift2,
                                                             unet_bf16* __restrict__ y, int y_cs, int y_co, long long P, int C8, int relu) {
    for (QuadWalk w(C8); w.p < P; w.next()) {
        const long long p = w.p;
        const int c = 8 * w.c4;
        float v[8];
        ldn<8>(x + (size_t)p * x_cs + x_co + c, v);
        if (scale != nullptr) {
            float sc[8], sh[8];
            ldn<8>(scale + c, sc); ldn<8>(shift + c, sh);
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] * sc[k] + sh[k];
        }
        if (x2 != nullptr) {
            float u[8];
            ldn<8>(x2 + (size_t)p * x2_cs + x2_co + c, u);
            if (scale2 != nullptr) {
                float sc[8], sh[8];
                ldn<8>(scale2 + c, sc); ldn<8>(shift2 + c, sh);
#pragma unroll
                for (int k = 0; k < 8; ++k) u[k] = u[k] * sc[k] + sh[k];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = v[k] + u[k];
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        stn<8>(y + (size_t)p * y_cs + y_co + c, v);
    }
}

__global__ __launch_bounds__(256) void bn_bwd_apply_oct_kernel(const unet_bf16* __restrict__ dout, int d_cs, int d_co,
                                                               const unet_bf16* __restrict__ out, int o_cs, int o_co,
                                                               const unet_bf16* __restrict__ x, int x_cs, int x_co,
                                                               const float* __restrict__ mean, const float* __restrict__ invstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ c1,
                                                               const float* __restrict__ c2, unet_bf16* __restrict__ dx, int dx_cs, int dx_co,
                                                               unet_bf16* gout, int g_cs, int g_co, int g_acc, long long P, int C8) {
    for (QuadWalk w(C8); w.p < P; w.next()) {
        const long long p = w.p;
        const int c = 8 * w.c4;
        float g[8], xv[8], mu[8], is[8], gm[8], a1[8], a2[8], r[8];
        ldn<8>(dout + (size_t)p * d_cs + d_co + c, g);
        if (out != nullptr) {
            float o[8];
            ldn<8>(out + (size_t)p * o_cs + o_co + c, o);
#pragma unroll
            for (int k = 0; k < 8; ++k) g[k] = o[k] > 0.f ? g[k] : 0.f;
        }
        ldn<8>(x + (size_t)p * x_cs + x_co + c, xv);
        ldn<8>(mean + c, mu); ldn<8>(invstd + c, is); ldn<8>(c1 + c, a1); ldn<8>(c2 + c, a2);
        if (gamma) ldn<8>(gamma + c, gm);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float xh = (xv[k] - mu[k]) * is[k];
            r[k] = (gamma ? gm[k] : 1.f) * is[k] * (g[k] - a1[k] - xh * a2[k]);
        }
        stn<8>(dx + (size_t)p * dx_cs + dx_co + c, r);
        if (gout != nullptr) {
            unet_bf16* gp = gout + (size_t)p * g_cs + g_co + c;
            if (g_acc) {
                float old[8];
                ldn<8>(gp, old);
#pragma unroll
                for (int k = 0; k < 8; ++k) g[k] = old[k] + g[k];
            }
            stn<8>(gp, g);
        }
    }
}

// channel_reduce with 8 channels per thread (tx = channel oct): out0 / out1 rows of Cp floats as in the quad form
template <typename F>
__device__ __forceinline__ void channel_reduce8(F f, long long P, int C8, int TC, float* out0, float* out1, int Cp) {
    __shared__ float sm0[256 * 8];
    __shared__ float sm1[256 * 8];
    const int tx = threadIdx.x % TC, ty = threadIdx.x / TC, PY = 256 / TC;
    for (int cbase = 0; cbase < C8; cbase += TC) {
        const int c8 = cbase + tx;
        float s0[8], s1[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) { s0[k] = 0.f; s1[k] = 0.f; }
        if (c8 < C8) {
            const long long step = (long long)gridDim.x * PY;
            long long p = (long long)blockIdx.x * PY + ty;
            for (; p + step < P; p += 2 * step) {          // two pixels per trip: six independent 16-byte loads in flight
                float a0[8], b0[8], a1[8], b1[8];
                f(p, c8, a0, b0);
                f(p + step, c8, a1, b1);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s0[k] += a0[k] + a1[k]; s1[k] += b0[k] + b1[k]; }
            }
            for (; p < P; p += step) {
                float a0[8], b0[8];
                f(p, c8, a0, b0);
#pragma unroll
                for (int k = 0; k < 8; ++k) { s0[k] += a0[k]; s1[k] += b0[k]; }
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { sm0[k * 256 + threadIdx.x] = s0[k]; sm1[k * 256 + threadIdx.x] = s1[k]; }
        __syncthreads();
        if (ty == 0 && c8 < C8) {
            for (int j = 1; j < PY; ++j) {
#pragma unroll
                for (int k = 0; k < 8; ++k) { s0[k] += sm0[k * 256 + j * TC + tx]; s1[k] += sm1[k * 256 + j * TC + tx]; }
            }
            stn<8>(out0 + (size_t)blockIdx.x * Cp + 8 * c8, s0);
            if (out1 != nullptr) stn<8>(out1 + (size_t)blockIdx.x * Cp + 8 * c8, s1);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void bn_bwd_reduce_oct_kernel(const unet_bf16* __restrict__ dout, int d_cs, int d_co,
                                                                const unet_bf16* __restrict__ out, int o_cs, int o_co,
                                                                const unet_bf16* __restrict__ x, int x_cs, int x_co,
                                                                const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                long long P, int C8, int TC, float* out0, float* out1, int Cp) {
    channel_reduce8(
        [&](long long p, int c8, float (&v0)[8], float (&v1)[8]) {
            float xv[8], mu[8], is[8];
            ldn<8>(dout + (size_t)p * d_cs + d_co + 8 * c8, v0);
            if (out != nullptr) {
                float o[8];
                ldn<8>(out + (size_t)p * o_cs + o_co + 8 * c8, o);
#pragma unroll
                for (int k = 0; k < 8; ++k) v0[k] = o[k] > 0.f ? v0[k] : 0.f;
            }
            ldn<8>(x + (size_t)p * x_cs + x_co + 8 * c8, xv);
            ldn<8>(mean + 8 * c8, mu); ldn<8>(invstd + 8 * c8, is);
#pragma unroll
            for (int k = 0; k < 8; ++k) v1[k] = v0[k] * ((xv[k] - mu[k]) * is[k]);
        },
        P, C8, TC, out0, out1, Cp);
}

// ------------------------------------------------------------------ pooling

template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T* __restrict__ x, int x_cs, int x_co, T* __restrict__ y, int y_cs,
                                                      int y_co, uint8_t* __restrict__ idx, int N, int IH, int IW, int C4, int OH, int OW) {
    const long long total = (long long)N * OH * OW * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c = 4 * (int)(t % C4); t /= C4;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
        int mi[4] = {-1, -1, -1, -1};
        for (int r = 0; r < 3; ++r) {
            const int iy = 2 * oy - 1 + r;
            if (iy < 0 || iy >= IH) continue;
            for (int s = 0; s < 3; ++s) {
                const int ix = 2 * ox - 1 + s;
                if (ix < 0 || ix >= IW) continue;
                const float4 v = ld4(x + ((size_t)(n * IH + iy) * IW + ix) * x_cs + x_co + c);
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (mi[j] < 0 || vv[j] > m[j] || vv[j] != vv[j]) { m[j] = vv[j]; mi[j] = r * 3 + s; }
            }
        }
        const size_t po = (size_t)(n * OH + oy) * OW + ox;
        st4(y + po * y_cs + y_co + c, make_float4(m[0], m[1], m[2], m[3]));
        if (idx != nullptr) {
            uint8_t* ip = idx + po * (4 * C4) + c;
            *reinterpret_cast<uchar4*>(ip) = make_uchar4((uint8_t)mi[0], (uint8_t)mi[1], (uint8_t)mi[2], (uint8_t)mi[3]);
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, int dy_cs, int dy_co, const uint8_t* __restrict__ idx,
                                                          T* __restrict__ dx, int dx_cs, int dx_co, int N, int IH, int IW, int C4,
                                                          int OH, int OW, int accumulate) {
    const long long total = (long long)N * IH * IW * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c = 4 * (int)(t % C4); t /= C4;
        const int ix = (int)(t % IW); t /= IW;
        const int iy = (int)(t % IH);
        const int n = (int)(t / IH);
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        const int oy_lo = iy / 2, oy_hi = (iy + 1) / 2;  // windows with 2*oy-1 <= iy <= 2*oy+1
        const int ox_lo = ix / 2, ox_hi = (ix + 1) / 2;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (oy >= OH) continue;
            const int r = iy - (2 * oy - 1);
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (ox >= OW) continue;
                const int s = ix - (2 * ox - 1);
                const size_t po = (size_t)(n * OH + oy) * OW + ox;
                const uchar4 k = *reinterpret_cast<const uchar4*>(idx + po * (4 * C4) + c);
                const float4 g = ld4(dy + po * dy_cs + dy_co + c);
                const int code = r * 3 + s;
                if (k.x == code) acc[0] += g.x;
                if (k.y == code) acc[1] += g.y;
                if (k.z == code) acc[2] += g.z;
                if (k.w == code) acc[3] += g.w;
            }
        }
        T* dp = dx + ((size_t)(n * IH + iy) * IW + ix) * dx_cs + dx_co + c;
        float4 r4 = make_float4(acc[0], acc[1], acc[2], acc[3]);
        st4(dp, accumulate ? (ld4(dp) + r4) : r4);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_kernel(const T* __restrict__ x, int x_cs, int x_co, T* __restrict__ y, int y_cs,
                                                      int y_co, int N, int IH, int IW, int C4, int OH, int OW) {
    const long long total = (long long)N * OH * OW * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c = 4 * (int)(t % C4); t /= C4;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        float4 s = f4(0.f);
        int cnt = 0;
        for (int r = 0; r < 2; ++r) {
            const int iy = 2 * oy + r;
            if (iy >= IH) continue;
            for (int q = 0; q < 2; ++q) {
                const int ix = 2 * ox + q;
                if (ix >= IW) continue;
                s = s + ld4(x + ((size_t)(n * IH + iy) * IW + ix) * x_cs + x_co + c);
                ++cnt;
            }
        }
        const float inv = 1.f / (float)cnt;
        st4(y + ((size_t)(n * OH + oy) * OW + ox) * y_cs + y_co + c, make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool_bwd_kernel(const T* __restrict__ dy, int dy_cs, int dy_co, T* __restrict__ dx,
                                                          int dx_cs, int dx_co, int N, int IH, int IW, int C4, int OH, int OW,
                                                          int accumulate) {
    const long long total = (long long)N * IH * IW * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c = 4 * (int)(t % C4); t /= C4;
        const int ix = (int)(t % IW); t /= IW;
        const int iy = (int)(t % IH);
        const int n = (int)(t / IH);
        const int oy = iy / 2, ox = ix / 2;
        const int ch = (2 * oy + 1 < IH) ? 2 : 1, cw = (2 * ox + 1 < IW) ? 2 : 1;
        const float inv = 1.f / (float)(ch * cw);
        const float4 g = ld4(dy + ((size_t)(n * OH + oy) * OW + ox) * dy_cs + dy_co + c);
        T* dp = dx + ((size_t)(n * IH + iy) * IW + ix) * dx_cs + dx_co + c;
        float4 r4 = make_float4(g.x * inv, g.y * inv, g.z * inv, g.w * inv);
        st4(dp, accumulate ? (ld4(dp) + r4) : r4);
    }
}

// ------------------------------------------------- pixel shuffle (+ blur)
// yc: conv output [N,h,w,4*Cu] (post-ReLU).  P[2h+i][2w+j][c] = yc[h][w][4c+2i+j].
// blur: out[y][x] = mean P[max(y-a,0)][max(x-b,0)], a,b in {0,1}.
// One thread owns the 2x2 output block of one channel of one low-res pixel: it needs
// P at rows 2h-1..2h+1, cols 2w-1..2w+1 = yc of 4 low-res neighbours (float4 each).
template <typename T>
__global__ __launch_bounds__(256) void shuffle_blur_kernel(const T* __restrict__ yc, int yc_cs, int yc_co, T* __restrict__ X,
                                                           int X_cs, int X_co, int N, int h, int w, int Cu, int do_blur) {
    const long long total = (long long)N * h * w * Cu;
    const int H = 2 * h, W = 2 * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c = (int)(t % Cu); t /= Cu;
        const int ww = (int)(t % w); t /= w;
        const int hh = (int)(t % h);
        const int n = (int)(t / h);
        auto Y = [&](int a, int b) { return ld4(yc + ((size_t)(n * h + a) * w + b) * yc_cs + yc_co + 4 * c); };
        const float4 q11 = Y(hh, ww);  // .x=(0,0) .y=(0,1) .z=(1,0) .w=(1,1) sub-pixels
        float o00, o01, o10, o11;
        if (!do_blur) {
            o00 = q11.x; o01 = q11.y; o10 = q11.z; o11 = q11.w;
        } else {
            // P rows: 2hh-1 (clamped), 2hh, 2hh+1 ; cols likewise.  p[r][s], r,s in {0,1,2} for rows/cols -1,0,+1
            const int hm = hh > 0 ? hh - 1 : hh, wm = ww > 0 ? ww - 1 : ww;
            const float4 q01 = Y(hm, ww), q10 = Y(hh, wm), q00 = Y(hm, wm);
            float p[3][3];
            // row -1: sub-row 1 of the low-res row above (or clamp to row 0 = sub-row 0 of this one when hh == 0)
            const bool top = hh == 0, left = ww == 0;
            p[1][1] = q11.x; p[1][2] = q11.y; p[2][1] = q11.z; p[2][2] = q11.w;
            p[1][0] = left ? q11.x : q10.y;  p[2][0] = left ? q11.z : q10.w;
            p[0][1] = top ? q11.x : q01.z;   p[0][2] = top ? q11.y : q01.w;
            p[0][0] = top ? (left ? q11.x : q10.y) : (left ? q01.z : q00.w);
            o00 = 0.25f * (p[0][0] + p[0][1] + p[1][0] + p[1][1]);
            o01 = 0.25f * (p[0][1] + p[0][2] + p[1][1] + p[1][2]);
            o10 = 0.25f * (p[1][0] + p[1][1] + p[2][0] + p[2][1]);
            o11 = 0.25f * (p[1][1] + p[1][2] + p[2][1] + p[2][2]);
        }
        T* o = X + ((size_t)(n * H + 2 * hh) * W + 2 * ww) * X_cs + X_co + c;
        st1(o, o00);
        st1(o + X_cs, o01);
        st1(o + (size_t)W * X_cs, o10);
        st1(o + (size_t)W * X_cs + X_cs, o11);
    }
}

// adjoint: dyc[h][w][4c+2i+j] = (yc > 0) * dP[2h+i][2w+j][c], dP = blur^T(dX)
template <typename T>
__global__ __launch_bounds__(256) void shuffle_blur_bwd_kernel(const T* __restrict__ dX, int dX_cs, int dX_co,
                                                               const T* __restrict__ yc, int yc_cs, int yc_co,
                                                               T* __restrict__ dyc, int dyc_cs, int dyc_co, int N, int h, int w,
                                                               int Cu, int do_blur) {
    const long long total = (long long)N * h * w * Cu;
    const int H = 2 * h, W = 2 * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c = (int)(t % Cu); t /= Cu;
        const int ww = (int)(t % w); t /= w;
        const int hh = (int)(t % h);
        const int n = (int)(t / h);
        auto D = [&](int y, int x) -> float {
            return (y < H && x < W) ? ld1(dX + ((size_t)(n * H + y) * W + x) * dX_cs + dX_co + c) : 0.f;
        };
        float g[2][2];
        if (!do_blur) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) g[a][b] = D(2 * hh + a, 2 * ww + b);
        } else {
            // dP[y][x] = 1/4 sum_{i in {y, y+1}} sum_{j in {x, x+1}} wy(y,i) wx(x,j) dOut[i][j]; the clamp doubles the
            // weight of output row/col 0 on P row/col 0.
            float d[3][3];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) d[a][b] = D(2 * hh + a, 2 * ww + b);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float wy0 = (2 * hh + a == 0) ? 2.f : 1.f, wx0 = (2 * ww + b == 0) ? 2.f : 1.f;
                    g[a][b] = 0.25f * (wy0 * wx0 * d[a][b] + wy0 * d[a][b + 1] + wx0 * d[a + 1][b] + d[a + 1][b + 1]);
                }
        }
        const size_t po = ((size_t)(n * h + hh) * w + ww);
        const float4 ref = ld4(yc + po * yc_cs + yc_co + 4 * c);
        st4(dyc + po * dyc_cs + dyc_co + 4 * c, gate4(make_float4(g[0][0], g[0][1], g[1][0], g[1][1]), ref));
    }
}

// ---- vector forms: one thread owns V consecutive OUTPUT channels (V = 16 bytes of storage: 4 fp32 / 8 bf16) of one low-res pixel, i.e.
// 4 V consecutive channels of yc.  The scalar forms above store 4 (fp32) / 2 (bf16) bytes per lane and instruction; here every access
// is a 16-byte vector (the bf16 step spent 2.1 ms in the two scalar kernels at 1.9 - 3.4 TB/s).  Same arithmetic, element by element.
template <typename T> struct VecOf;
template <> struct VecOf<float> { static constexpr int V = 4; };
template <> struct VecOf<unet_bf16> { static constexpr int V = 8; };

template <typename T>
__global__ __launch_bounds__(256) void shuffle_blur_vec_kernel(const T* __restrict__ yc, int yc_cs, int yc_co, T* __restrict__ X, int X_cs,
                                                               int X_co, int N, int h, int w, int Cg, int do_blur) {
    constexpr int V = VecOf<T>::V;
    const long long total = (long long)N * h * w * Cg;
    const int H = 2 * h, W = 2 * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c0 = V * (int)(t % Cg); t /= Cg;
        const int ww = (int)(t % w); t /= w;
        const int hh = (int)(t % h);
        const int n = (int)(t / h);
        auto Y = [&](int a, int b, float (&q)[4 * V]) { ldn<4 * V>(yc + ((size_t)(n * h + a) * w + b) * yc_cs + yc_co + 4 * c0, q); };
        float q11[4 * V], o00[V], o01[V], o10[V], o11[V];
        Y(hh, ww, q11);         // [4k + s]: channel c0 + k, sub-pixel s = 2i + j
        if (!do_blur) {
#pragma unroll
            for (int k = 0; k < V; ++k) { o00[k] = q11[4 * k]; o01[k] = q11[4 * k + 1]; o10[k] = q11[4 * k + 2]; o11[k] = q11[4 * k + 3]; }
        } else {
            const int hm = hh > 0 ? hh - 1 : hh, wm = ww > 0 ? ww - 1 : ww;
            const bool top = hh == 0, left = ww == 0;
            float q01[4 * V], q10[4 * V], q00[4 * V];
            Y(hm, ww, q01); Y(hh, wm, q10); Y(hm, wm, q00);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float ax = q11[4 * k], ay = q11[4 * k + 1], az = q11[4 * k + 2], aw = q11[4 * k + 3];
                float p[3][3];
                p[1][1] = ax; p[1][2] = ay; p[2][1] = az; p[2][2] = aw;
                p[1][0] = left ? ax : q10[4 * k + 1];  p[2][0] = left ? az : q10[4 * k + 3];
                p[0][1] = top ? ax : q01[4 * k + 2];   p[0][2] = top ? ay : q01[4 * k + 3];
                p[0][0] = top ? (left ? ax : q10[4 * k + 1]) : (left ? q01[4 * k + 2] : q00[4 * k + 3]);
                o00[k] = 0.25f * (p[0][0] + p[0][1] + p[1][0] + p[1][1]);
                o01[k] = 0.25f * (p[0][1] + p[0][2] + p[1][1] + p[1][2]);
                o10[k] = 0.25f * (p[1][0] + p[1][1] + p[2][0] + p[2][1]);
                o11[k] = 0.25f * (p[1][1] + p[1][2] + p[2][1] + p[2][2]);
            }
        }
        T* o = X + ((size_t)(n * H + 2 * hh) * W + 2 * ww) * X_cs + X_co + c0;
        stn<V>(o, o00);
        stn<V>(o + X_cs, o01);
        stn<V>(o + (size_t)W * X_cs, o10);
        stn<V>(o + (size_t)W * X_cs + X_cs, o11);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void shuffle_blur_bwd_vec_kernel(const T* __restrict__ dX, int dX_cs, int dX_co, const T* __restrict__ yc,
                                                                   int yc_cs, int yc_co, T* __restrict__ dyc, int dyc_cs, int dyc_co, int N,
                                                                   int h, int w, int Cg, int do_blur) {
    constexpr int V = VecOf<T>::V;
    const long long total = (long long)N * h * w * Cg;
    const int H = 2 * h, W = 2 * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c0 = V * (int)(t % Cg); t /= Cg;
        const int ww = (int)(t % w); t /= w;
        const int hh = (int)(t % h);
        const int n = (int)(t / h);
        auto D = [&](int y, int x, float (&d)[V]) {
            if (y < H && x < W) ldn<V>(dX + ((size_t)(n * H + y) * W + x) * dX_cs + dX_co + c0, d);
            else {
#pragma unroll
                for (int k = 0; k < V; ++k) d[k] = 0.f;
            }
        };
        float g[4 * V];          // [4k + 2a + b]
        if (!do_blur) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    float d[V];
                    D(2 * hh + a, 2 * ww + b, d);
#pragma unroll
                    for (int k = 0; k < V; ++k) g[4 * k + 2 * a + b] = d[k];
                }
        } else {
            float d[3][3][V];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) D(2 * hh + a, 2 * ww + b, d[a][b]);
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float wy0 = (2 * hh + a == 0) ? 2.f : 1.f, wx0 = (2 * ww + b == 0) ? 2.f : 1.f;
#pragma unroll
                    for (int k = 0; k < V; ++k)
                        g[4 * k + 2 * a + b] = 0.25f * (wy0 * wx0 * d[a][b][k] + wy0 * d[a][b + 1][k] + wx0 * d[a + 1][b][k] + d[a + 1][b + 1][k]);
                }
        }
        const size_t po = ((size_t)(n * h + hh) * w + ww);
        float ref[4 * V];
        ldn<4 * V>(yc + po * yc_cs + yc_co + 4 * c0, ref);
#pragma unroll
        for (int k = 0; k < 4 * V; ++k) g[k] = ref[k] > 0.f ? g[k] : 0.f;
        stn<4 * V>(dyc + po * dyc_cs + dyc_co + 4 * c0, g);
    }
}

// Adjoint of PixelShuffle(2) behind a ReLU whose un-shuffled output was never stored (unet_conv_desc.pixel_shuffle): the mask comes from
// the SHUFFLED activation X itself -- yc[h][w][4c+2a+b] = X[2h+a][2w+b][c] --, read at the address dX is read at:
// dyc[h][w][4c+2a+b] = X[2h+a][2w+b][c] > 0 ? dX[2h+a][2w+b][c] : 0.  V channels per thread (16-byte accesses).
template <typename T>
__global__ __launch_bounds__(256) void shuffle_bwd_xmask_kernel(const T* __restrict__ dX, int dX_cs, int dX_co, const T* __restrict__ X, int X_cs,
                                                                int X_co, T* __restrict__ dyc, int dyc_cs, int dyc_co, int N, int h, int w, int Cg) {
    constexpr int V = VecOf<T>::V;
    const long long total = (long long)N * h * w * Cg;
    const int H = 2 * h, W = 2 * w;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c0 = V * (int)(t % Cg); t /= Cg;
        const int ww = (int)(t % w); t /= w;
        const int hh = (int)(t % h);
        const int n = (int)(t / h);
        float g[4 * V];
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const size_t px = (size_t)(n * H + 2 * hh + a) * W + 2 * ww + b;
                float d[V], m[V];
                ldn<V>(dX + px * dX_cs + dX_co + c0, d);
                ldn<V>(X + px * X_cs + X_co + c0, m);
#pragma unroll
                for (int k = 0; k < V; ++k) g[4 * k + 2 * a + b] = m[k] > 0.f ? d[k] : 0.f;
            }
        stn<4 * V>(dyc + ((size_t)(n * h + hh) * w + ww) * dyc_cs + dyc_co + 4 * c0, g);
    }
}

// ------------------------------------------------------- nearest resize
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
    int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

template <typename T>
__global__ __launch_bounds__(256) void resize_nearest_kernel(const T* __restrict__ x, int x_cs, int x_co, T* __restrict__ y,
                                                             int y_cs, int y_co, int N, int IH, int IW, int OH, int OW, int C4) {
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    const long long total = (long long)N * OH * OW * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c = 4 * (int)(t % C4); t /= C4;
        const int ox = (int)(t % OW); t /= OW;
        const int oy = (int)(t % OH);
        const int n = (int)(t / OH);
        const int iy = nearest_src(oy, sh, IH), ix = nearest_src(ox, sw, IW);
        st4(y + ((size_t)(n * OH + oy) * OW + ox) * y_cs + y_co + c, ld4(x + ((size_t)(n * IH + iy) * IW + ix) * x_cs + x_co + c));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void resize_nearest_bwd_kernel(const T* __restrict__ dy, int dy_cs, int dy_co, T* __restrict__ dx,
                                                                 int dx_cs, int dx_co, int N, int IH, int IW, int OH, int OW, int C4) {
    const float sh = (float)IH / (float)OH, sw = (float)IW / (float)OW;
    const long long total = (long long)N * IH * IW * C4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        long long t = i;
        const int c = 4 * (int)(t % C4); t /= C4;
        const int ix = (int)(t % IW); t /= IW;
        const int iy = (int)(t % IH);
        const int n = (int)(t / IH);
        // candidate destination range: conservative window around iy / scale
        int oy_lo = (int)floorf((float)iy / sh) - 2, oy_hi = (int)ceilf((float)(iy + 1) / sh) + 2;
        int ox_lo = (int)floorf((float)ix / sw) - 2, ox_hi = (int)ceilf((float)(ix + 1) / sw) + 2;
        oy_lo = oy_lo < 0 ? 0 : oy_lo; ox_lo = ox_lo < 0 ? 0 : ox_lo;
        oy_hi = oy_hi > OH - 1 ? OH - 1 : oy_hi; ox_hi = ox_hi > OW - 1 ? OW - 1 : ox_hi;
        float4 s = f4(0.f);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (nearest_src(oy, sh, IH) != iy) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (nearest_src(ox, sw, IW) != ix) continue;
                s = s + ld4(dy + ((size_t)(n * OH + oy) * OW + ox) * dy_cs + dy_co + c);
            }
        }
        st4(dx + ((size_t)(n * IH + iy) * IW + ix) * dx_cs + dx_co + c, s);
    }
}

// ------------------------------------------------------ layout conversion
template <typename T>
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, T* __restrict__ y, int y_cs, int y_co, int N,
                                                           int C, int H, int W) {
    const long long HW = (long long)H * W, total = (long long)N * HW;
    // whole channel quads at a quad-aligned offset (the 4-band tile into its own buffer and into channels 96..99 of the final concat): one
    // 16-byte (bf16: 8-byte) store per quad instead of four scalar ones -- the kernel is bound by its store instructions, not by its bytes
    const bool quads = (C & 3) == 0 && (y_cs & 3) == 0 && (y_co & 3) == 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, p = i - n * HW;
        T* o = y + (size_t)i * y_cs + y_co;
        const float* s = x + (size_t)n * C * HW + p;
        if (quads) {
            for (int c = 0; c < C; c += 4) st4(o + c, make_float4(s[(size_t)c * HW], s[(size_t)(c + 1) * HW], s[(size_t)(c + 2) * HW], s[(size_t)(c + 3) * HW]));
        } else {
            for (int c = 0; c < C; ++c) st1(o + c, s[(size_t)c * HW]);
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const float* __restrict__ x, int x_cs, int x_co, float* __restrict__ y, int N,
                                                           int C, int H, int W) {
    const long long HW = (long long)H * W, total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, p = i - n * HW;
        const float* s = x + (size_t)i * x_cs + x_co;
        for (int c = 0; c < C; ++c) y[((size_t)n * C + c) * HW + p] = s[c];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void copy_slice_kernel(const T* __restrict__ x, int x_cs, int x_co, T* __restrict__ y, int y_cs,
                                                         int y_co, long long P, int C4, int accumulate) {
    for (QuadWalk w(C4); w.p < P; w.next()) {
        const long long p = w.p;
        const int c = 4 * w.c4;
        const float4 v = ld4(x + (size_t)p * x_cs + x_co + c);
        T* o = y + (size_t)p * y_cs + y_co + c;
        st4(o, accumulate ? (ld4(o) + v) : v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void relu_mask_kernel(const T* __restrict__ g, int g_cs, int g_co, const T* __restrict__ ref,
                                                        int r_cs, int r_co, T* __restrict__ y, int y_cs, int y_co, long long P, int C4) {
    for (QuadWalk w(C4); w.p < P; w.next()) {
        const long long p = w.p;
        const int c = 4 * w.c4;
        st4(y + (size_t)p * y_cs + y_co + c, gate4(ld4(g + (size_t)p * g_cs + g_co + c), ld4(ref + (size_t)p * r_cs + r_co + c)));
    }
}

// ---------------------------------------------------------------- loss
constexpr int CE_MAXC = 64;

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ z, int z_cs, int z_co, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, long long P, int C, float* __restrict__ part) {
    float num = 0.f, den = 0.f;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long y = target[p];
        if (y < 0 || y >= C) continue;  // ignore_index semantics
        const float* zp = z + (size_t)p * z_cs + z_co;
        float m = zp[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, zp[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(zp[c] - m);
        const float nll = (m + logf(s)) - zp[y];
        const float w = weight ? weight[y] : 1.f;
        num += w * nll;
        den += w;
    }
    // wavefront (64-lane) reduction, then one LDS hop
    __shared__ float sn[4], sd[4];
    for (int o = 32; o > 0; o >>= 1) {
        num += __shfl_down(num, o);
        den += __shfl_down(den, o);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) { sn[wv] = num; sd[wv] = den; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[2 * blockIdx.x] = sn[0] + sn[1] + sn[2] + sn[3];
        part[2 * blockIdx.x + 1] = sd[0] + sd[1] + sd[2] + sd[3];
    }
}

// one wavefront: 64 lanes sum the partial rows in fp64, shuffle reduction; numden (optional) receives the numerator and the
// denominator themselves -- what the ranks of a tile-DDP step all-reduce (a rank whose tiles carry only zero-weight classes has
// den = 0: its 0/0 quotient must not reach the global loss)
__global__ __launch_bounds__(64) void ce_finalize_kernel(const float* __restrict__ part, int rows, float* loss, float* denom, float* numden) {
    double n = 0.0, d = 0.0;
    for (int r = threadIdx.x; r < rows; r += 64) { n += (double)part[2 * r]; d += (double)part[2 * r + 1]; }
    for (int o = 32; o > 0; o >>= 1) {
        n += __shfl_down(n, o);
        d += __shfl_down(d, o);
    }
    if (threadIdx.x != 0) return;
    if (loss) *loss = (float)(n / d);
    if (denom) *denom = (float)d;
    if (numden) { numden[0] = (float)n; numden[1] = (float)d; }
}

template <typename T>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ z, int z_cs, int z_co, const int64_t* __restrict__ target,
                                                     const float* __restrict__ weight, long long P, int C, const float* __restrict__ denom,
                                                     float gscale, T* __restrict__ dz, int dz_cs, int dz_co) {
    const float inv = gscale / *denom;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long y = target[p];
        T* dp = dz + (size_t)p * dz_cs + dz_co;
        if (y < 0 || y >= C) {
            for (int c = 0; c < C; ++c) st1(dp + c, 0.f);
            continue;
        }
        const float* zp = z + (size_t)p * z_cs + z_co;
        float m = zp[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, zp[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(zp[c] - m);
        const float w = (weight ? weight[y] : 1.f) * inv;
        const float is = 1.f / s;
        for (int c = 0; c < C; ++c) st1(dp + c, w * (expf(zp[c] - m) * is - (c == y ? 1.f : 0.f)));
    }
}

// FocalLossFlat(gamma, axis=1) -- the alternative classification loss the reference's configuration names (params_and_main.py:87-89:
// `FocalLossFlat(gamma=2, axis=1)`, `gamma=0.5`).  fastai 2.5.1 losses.FocalLoss.forward, per pixel:
//     ce = F.cross_entropy(inp, targ, weight, reduction="none") = w[y] * nll      (train.py:211 assigns loss_func.func.weight for every loss)
//     p_t = exp(-ce);   loss = (1 - p_t)^gamma * ce;   'mean' = sum / P over ALL pixels (a plain mean, not the weighted mean of the CE)
// d loss / d z_c = [gamma (1 - p_t)^(gamma-1) p_t ce + (1 - p_t)^gamma] * w[y] (softmax_c - [c == y]) / P.  Where ce == 0 exactly (a logit margin
// beyond fp32's exp range) the first term is 0 * inf for gamma < 1 -- torch's autograd yields NaN there; this kernel yields the limit, 0.
__device__ __forceinline__ void focal_terms(float ce, float gamma, float& val, float& dval) {
    const float pt = expf(-ce), q = 1.f - pt;
    if (q <= 0.f) { val = 0.f; dval = 0.f; return; }
    const float qg = powf(q, gamma);
    val = qg * ce;
    dval = gamma * (qg / q) * pt * ce + qg;
}

__global__ __launch_bounds__(256) void focal_fwd_kernel(const float* __restrict__ z, int z_cs, int z_co, const int64_t* __restrict__ target,
                                                        const float* __restrict__ weight, long long P, int C, float gamma, float* __restrict__ part) {
    float num = 0.f;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long y = target[p];
        if (y < 0 || y >= C) continue;  // ignore_index semantics: contributes 0 to the sum, still counts in the mean
        const float* zp = z + (size_t)p * z_cs + z_co;
        float m = zp[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, zp[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(zp[c] - m);
        const float ce = (weight ? weight[y] : 1.f) * ((m + logf(s)) - zp[y]);
        float v, dv;
        focal_terms(ce, gamma, v, dv);
        num += v;
    }
    __shared__ float sn[4];
    for (int o = 32; o > 0; o >>= 1) num += __shfl_down(num, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sn[wv] = num;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sn[0] + sn[1] + sn[2] + sn[3];
}

template <typename T>
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ z, int z_cs, int z_co, const int64_t* __restrict__ target,
                                                        const float* __restrict__ weight, long long P, int C, float gamma, float gscale,
                                                        T* __restrict__ dz, int dz_cs, int dz_co) {
    const float inv = gscale / (float)P;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long y = target[p];
        T* dp = dz + (size_t)p * dz_cs + dz_co;
        if (y < 0 || y >= C) {
            for (int c = 0; c < C; ++c) st1(dp + c, 0.f);
            continue;
        }
        const float* zp = z + (size_t)p * z_cs + z_co;
        float m = zp[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, zp[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(zp[c] - m);
        const float w = weight ? weight[y] : 1.f;
        float v, dv;
        focal_terms(w * ((m + logf(s)) - zp[y]), gamma, v, dv);
        const float g = dv * w * inv, is = 1.f / s;
        for (int c = 0; c < C; ++c) st1(dp + c, g * (expf(zp[c] - m) * is - (c == y ? 1.f : 0.f)));
    }
}

// Regression losses of the enable_regression branch (reference train.py:189-193: MSELossFlat(axis=1); utils.py:145-147:
// Smoothl1 = SmoothL1Loss(beta=0.5); fastai L1LossFlat): prediction = channel 0 of the [P,1] output slice, float targets,
// 'mean' reduction over all P pixels.  kind 0: d^2   1: |d|   2: |d| < beta ? d^2 / (2 beta) : |d| - beta / 2
__device__ __forceinline__ float regloss_val(float d, int kind, float beta) {
    const float ad = fabsf(d);
    if (kind == 0) return d * d;
    if (kind == 1) return ad;
    return ad < beta ? 0.5f * d * d / beta : ad - 0.5f * beta;
}
__device__ __forceinline__ float regloss_grad(float d, int kind, float beta) {
    const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    if (kind == 0) return 2.f * d;
    if (kind == 1) return sg;
    return fabsf(d) < beta ? d / beta : sg;
}

__global__ __launch_bounds__(256) void regloss_fwd_kernel(const float* __restrict__ z, int z_cs, int z_co, const float* __restrict__ target,
                                                          long long P, int kind, float beta, float* __restrict__ part) {
    float num = 0.f;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x)
        num += regloss_val(z[(size_t)p * z_cs + z_co] - target[p], kind, beta);
    __shared__ float sn[4];
    for (int o = 32; o > 0; o >>= 1) num += __shfl_down(num, o);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 0) sn[wv] = num;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = sn[0] + sn[1] + sn[2] + sn[3];
}

__global__ void regloss_finalize_kernel(const float* __restrict__ part, int rows, long long P, float* loss) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double n = 0.0;
    for (int r = 0; r < rows; ++r) n += (double)part[r];
    *loss = (float)(n / (double)P);
}

__global__ __launch_bounds__(256) void regloss_bwd_kernel(const float* __restrict__ z, int z_cs, int z_co, const float* __restrict__ target,
                                                          long long P, int kind, float beta, float gscale, float* __restrict__ dz, int dz_cs,
                                                          int dz_co) {
    const float inv = gscale / (float)P;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x)
        dz[(size_t)p * dz_cs + dz_co] = inv * regloss_grad(z[(size_t)p * z_cs + z_co] - target[p], kind, beta);
}

__global__ __launch_bounds__(256) void softmax_argmax_kernel(const float* __restrict__ z, int z_cs, int z_co, int N, long long HW, int C,
                                                             float* __restrict__ probs, int64_t* __restrict__ amax) {
    const long long total = (long long)N * HW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / HW, p = i - n * HW;
        const float* zp = z + (size_t)i * z_cs + z_co;
        float m = zp[0];
        for (int c = 1; c < C; ++c) m = fmaxf(m, zp[c]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(zp[c] - m);
        float best = -1.f;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            const float pr = expf(zp[c] - m) / s;
            if (probs) probs[((size_t)n * C + c) * HW + p] = pr;
            if (pr > best) { best = pr; bi = c; }
        }
        if (amax) amax[i] = bi;
    }
}

// ------------------------------------------------------- row softmax (self-attention)
// y[p][:] = softmax(x[p][:]) over C contiguous channels; one workgroup per row, wavefront + LDS reductions.
__device__ __forceinline__ float block_reduce(float v, bool is_max) {
    __shared__ float sh[4];
    for (int o = 32; o > 0; o >>= 1) {
        const float t = __shfl_xor(v, o);
        v = is_max ? fmaxf(v, t) : v + t;
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    return is_max ? fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) : (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

template <typename TO>      // logits are always fp32 (bf16 storage: the product that makes them writes fp32, unet_conv_desc.y_f32); TO = float | bf16 weights
__global__ __launch_bounds__(256) void row_softmax_kernel(const float* __restrict__ x, int x_cs, int x_co, TO* __restrict__ y, int y_cs,
                                                          int y_co, long long P, int C) {
    for (long long p = blockIdx.x; p < P; p += gridDim.x) {
        const float* xr = x + (size_t)p * x_cs + x_co;
        TO* yr = y + (size_t)p * y_cs + y_co;
        float m = -INFINITY;
        for (int c = threadIdx.x; c < C; c += 256) m = fmaxf(m, xr[c]);
        m = block_reduce(m, true);
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) s += expf(xr[c] - m);
        s = block_reduce(s, false);
        const float inv = 1.f / s;
        for (int c = threadIdx.x; c < C; c += 256) st1(yr + c, expf(xr[c] - m) * inv);
    }
}

// dx[p][c] = y[p][c] * (dy[p][c] - sum_c' y[p][c'] dy[p][c'])
template <typename TW>      // TW = storage type of the softmax weights y and of the result dx; dy (the gradient of the weights) is fp32
__global__ __launch_bounds__(256) void row_softmax_bwd_kernel(const TW* __restrict__ y, int y_cs, int y_co, const float* __restrict__ dy,
                                                              int dy_cs, int dy_co, TW* __restrict__ dx, int dx_cs, int dx_co,
                                                              long long P, int C) {
    for (long long p = blockIdx.x; p < P; p += gridDim.x) {
        const TW* yr = y + (size_t)p * y_cs + y_co;
        const float* gr = dy + (size_t)p * dy_cs + dy_co;
        TW* dr = dx + (size_t)p * dx_cs + dx_co;
        float s = 0.f;
        for (int c = threadIdx.x; c < C; c += 256) s += ld1(yr + c) * gr[c];
        s = block_reduce(s, false);
        for (int c = threadIdx.x; c < C; c += 256) st1(dr + c, ld1(yr + c) * (gr[c] - s));
    }
}

// ----------------------------------------------------------------- Adam
struct AdamArgs {
    float decay[4];     // 1 - lr*wd per group
    float step_size[4]; // -lr / debias1 per group
    float mom, one_minus_mom, sqr_mom, one_minus_sqr, debias2, eps, grad_scale;
};

// hyper-parameters read from DEVICE memory (hipGraph replay: the host refreshes the 15 floats before each replay)
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, const uint8_t* __restrict__ code, long long n,
                                                       const AdamArgs* __restrict__ ap) {
    const AdamArgs a = *ap;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int cd = code[i];
        const int grp = cd & 3;
        float pv = p[i];
        const float gv = g[i] * a.grad_scale;
        if (cd & 4) pv *= a.decay[grp];
        const float mv = m[i] * a.mom + a.one_minus_mom * gv;
        const float vv = v[i] * a.sqr_mom + a.one_minus_sqr * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float den = sqrtf(vv / a.debias2) + a.eps;
        p[i] = pv + a.step_size[grp] * (mv / den);
    }
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, const uint8_t* __restrict__ code, long long n, AdamArgs a) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const int cd = code[i];
        const int grp = cd & 3;
        float pv = p[i];
        const float gv = g[i] * a.grad_scale;
        if (cd & 4) pv *= a.decay[grp];
        const float mv = m[i] * a.mom + a.one_minus_mom * gv;
        const float vv = v[i] * a.sqr_mom + a.one_minus_sqr * gv * gv;
        m[i] = mv;
        v[i] = vv;
        const float den = sqrtf(vv / a.debias2) + a.eps;
        p[i] = pv + a.step_size[grp] * (mv / den);
    }
}

// -------------------------------------------------------------- mosaic
__global__ __launch_bounds__(256) void mosaic_acc_kernel(const float* __restrict__ probs, int C, int th, int tw, float* __restrict__ mosaic,
                                                         int32_t* __restrict__ count, int MH, int MW, int y0, int x0) {
    const long long total = (long long)th * tw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ty = (int)(i / tw), tx = (int)(i % tw);
        const int my = y0 + ty, mx = x0 + tx;
        if (my < 0 || my >= MH || mx < 0 || mx >= MW) continue;
        for (int c = 0; c < C; ++c) mosaic[((size_t)c * MH + my) * MW + mx] += probs[((size_t)c * th + ty) * tw + tx];
        count[(size_t)my * MW + mx] += 1;
    }
}

__global__ __launch_bounds__(256) void mosaic_fin_kernel(float* __restrict__ mosaic, const int32_t* __restrict__ count, int C, int MH, int MW,
                                                         uint8_t* __restrict__ amax) {
    const long long total = (long long)MH * MW;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cnt = count[i];
        float best = -INFINITY;
        int bi = 0;
        for (int c = 0; c < C; ++c) {
            float v = mosaic[(size_t)c * total + i];
            if (cnt > 0) { v = v / (float)cnt; mosaic[(size_t)c * total + i] = v; }
            if (v > best) { best = v; bi = c; }
        }
        if (amax) amax[i] = (uint8_t)bi;
    }
}

inline int c4of(int C) { return roundup(C, 4) / 4; }
inline bool pslice_ok(int cs, int co, int C) { return unet::slice_ok(cs, co, roundup(C, 4)); }

}  // namespace

#define ST ((hipStream_t)stream)

extern "C" int unet_bn_stats_rows(long long P) { return stats_rows(P); }

template <typename T>
static int bn_stats_impl(const T* x, int x_cs, int x_co, long long P, int C, float* partial, void* stream) {
    UNET_CHECK_ARG(x && partial && P > 0 && C > 0 && (C & 3) == 0, "bn_stats: bad args (C must be a multiple of 4)");
    UNET_CHECK_ARG(unet::slice_ok(x_cs, x_co, C), "bn_stats: bad slice");
    const int rows = stats_rows(P), C4 = C / 4, TC = pick_tc(C4);
    // (an 8-channel form of this ONE-tensor reduction measured slower than the quad form -- 24 vs 14 us at 16 x 256^2 x 32: half the threads
    //  per pixel row, and the quad form already keeps four loads in flight -- so the statistics pass stays as it is in both storage types)
    hipLaunchKernelGGL((bn_stats_kernel<T>), dim3(rows), dim3(256), 0, ST, x, x_cs, x_co, P, C4, TC, partial, partial + (size_t)rows * C, C);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_bn_stats(const float* x, int x_cs, int x_co, long long P, int C, float* partial, void* stream) { return bn_stats_impl<float>(x, x_cs, x_co, P, C, partial, stream); }
extern "C" int unet_bn_stats_bf16(const unet_bf16* x, int x_cs, int x_co, long long P, int C, float* partial, void* stream) { return bn_stats_impl<unet_bf16>(x, x_cs, x_co, P, C, partial, stream); }

extern "C" int unet_bn_finalize(const float* psum, const float* psumsq, int rows, long long count, int C, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, float momentum, float eps, float* scale,
                                float* shift, float* save_mean, float* save_invstd, long long* batches_tracked, void* stream) {
    UNET_CHECK_ARG(psum && psumsq && scale && shift && rows > 0 && count > 0 && C > 0, "bn_finalize: bad args");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, ST, psum, psumsq, rows, C, (double)count, C, gamma, beta,
                       running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd, batches_tracked);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var, float eps,
                                   int C, float* scale, float* shift, void* stream) {
    UNET_CHECK_ARG(running_mean && running_var && scale && shift && C > 0, "bn_eval_coeffs: bad args");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3(cdiv(C, 128)), dim3(128), 0, ST, gamma, beta, running_mean, running_var, eps, C, scale,
                       shift);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <typename T>
static int affine_act_impl(const T* x, int x_cs, int x_co, const float* scale, const float* shift, const T* x2, int x2_cs,
                               int x2_co, const float* scale2, const float* shift2, T* y, int y_cs, int y_co, long long P, int C,
                               int relu, void* stream) {
    UNET_CHECK_ARG(x && y && P > 0 && C > 0, "affine_act: bad args");
    UNET_CHECK_ARG(pslice_ok(x_cs, x_co, C) && pslice_ok(y_cs, y_co, C), "affine_act: bad slice");
    UNET_CHECK_ARG((scale == nullptr) == (shift == nullptr) && (scale2 == nullptr) == (shift2 == nullptr), "affine_act: scale/shift mismatch");
    UNET_CHECK_ARG((scale == nullptr && scale2 == nullptr) || (C & 3) == 0, "affine_act: per-channel vectors need C % 4 == 0");
    if (x2) UNET_CHECK_ARG(pslice_ok(x2_cs, x2_co, C), "affine_act: bad x2 slice");
    const int C4 = c4of(C);
    if constexpr (sizeof(T) == 2) {
        if (use_oct() && C % 8 == 0 && x_cs % 8 == 0 && x_co % 8 == 0 && y_cs % 8 == 0 && y_co % 8 == 0 && (!x2 || (x2_cs % 8 == 0 && x2_co % 8 == 0)) &&
            unet::aligned16(x) && unet::aligned16(y) && (!x2 || unet::aligned16(x2))) {
            hipLaunchKernelGGL(affine_act_oct_kernel, dim3(ew_grid(P * (C / 8), 256)), dim3(256), 0, ST, x, x_cs, x_co, scale, shift, x2, x2_cs, x2_co,
                               scale2, shift2, y, y_cs, y_co, P, C / 8, relu);
            UNET_CHECK_LAUNCH();
            return UNET_OK;
        }
    }
    hipLaunchKernelGGL((affine_act_kernel<T>), dim3(ew_grid(P * C4, 256)), dim3(256), 0, ST, x, x_cs, x_co, scale, shift, x2, x2_cs, x2_co, scale2,
                       shift2, y, y_cs, y_co, P, C4, relu);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_affine_act(const float* x, int x_cs, int x_co, const float* scale, const float* shift, const float* x2, int x2_cs,
                               int x2_co, const float* scale2, const float* shift2, float* y, int y_cs, int y_co, long long P, int C,
                               int relu, void* stream) { return affine_act_impl<float>(x, x_cs, x_co, scale, shift, x2, x2_cs, x2_co, scale2, shift2, y, y_cs, y_co, P, C, relu, stream); }
extern "C" int unet_affine_act_bf16(const unet_bf16* x, int x_cs, int x_co, const float* scale, const float* shift, const unet_bf16* x2, int x2_cs,
                               int x2_co, const float* scale2, const float* shift2, unet_bf16* y, int y_cs, int y_co, long long P, int C,
                               int relu, void* stream) { return affine_act_impl<unet_bf16>(x, x_cs, x_co, scale, shift, x2, x2_cs, x2_co, scale2, shift2, y, y_cs, y_co, P, C, relu, stream); }

template <typename T>
static int bn_bwd_reduce_impl(const T* dout, int d_cs, int d_co, const T* out, int o_cs, int o_co, const T* x, int x_cs,
                                  int x_co, const float* mean, const float* invstd, long long P, int C, float* partial, void* stream) {
    UNET_CHECK_ARG(dout && x && mean && invstd && partial && P > 0 && C > 0 && (C & 3) == 0, "bn_bwd_reduce: bad args");
    UNET_CHECK_ARG(unet::slice_ok(d_cs, d_co, C) && unet::slice_ok(x_cs, x_co, C), "bn_bwd_reduce: bad slice");
    if (out) UNET_CHECK_ARG(unet::slice_ok(o_cs, o_co, C), "bn_bwd_reduce: bad out slice");
    const int rows = stats_rows(P), C4 = C / 4, TC = pick_tc(C4);
    if constexpr (sizeof(T) == 2) {
        if (use_oct() && C % 8 == 0 && d_cs % 8 == 0 && d_co % 8 == 0 && x_cs % 8 == 0 && x_co % 8 == 0 && (!out || (o_cs % 8 == 0 && o_co % 8 == 0)) &&
            unet::aligned16(dout) && unet::aligned16(x) && (!out || unet::aligned16(out))) {
            hipLaunchKernelGGL(bn_bwd_reduce_oct_kernel, dim3(rows), dim3(256), 0, ST, dout, d_cs, d_co, out, o_cs, o_co, x, x_cs, x_co, mean, invstd,
                               P, C / 8, pick_tc(C / 8), partial, partial + (size_t)rows * C, C);
            UNET_CHECK_LAUNCH();
            return UNET_OK;
        }
    }
    hipLaunchKernelGGL((bn_bwd_reduce_kernel<T>), dim3(rows), dim3(256), 0, ST, dout, d_cs, d_co, out, o_cs, o_co, x, x_cs, x_co, mean, invstd, P,
                       C4, TC, partial, partial + (size_t)rows * C, C);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_bn_bwd_reduce(const float* dout, int d_cs, int d_co, const float* out, int o_cs, int o_co, const float* x, int x_cs,
                                  int x_co, const float* mean, const float* invstd, long long P, int C, float* partial, void* stream) { return bn_bwd_reduce_impl<float>(dout, d_cs, d_co, out, o_cs, o_co, x, x_cs, x_co, mean, invstd, P, C, partial, stream); }
extern "C" int unet_bn_bwd_reduce_bf16(const unet_bf16* dout, int d_cs, int d_co, const unet_bf16* out, int o_cs, int o_co, const unet_bf16* x, int x_cs,
                                  int x_co, const float* mean, const float* invstd, long long P, int C, float* partial, void* stream) { return bn_bwd_reduce_impl<unet_bf16>(dout, d_cs, d_co, out, o_cs, o_co, x, x_cs, x_co, mean, invstd, P, C, partial, stream); }

extern "C" int unet_bn_bwd_finalize(const float* partial, int rows, long long count, int C, float* dgamma, float* dbeta, float* c1,
                                    float* c2, void* stream) {
    UNET_CHECK_ARG(partial && c1 && c2 && rows > 0 && count > 0 && C > 0, "bn_bwd_finalize: bad args");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(cdiv(C, 32)), dim3(256), 0, ST, partial, rows, C, (double)count, C, dgamma, dbeta, c1, c2,
                       0);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <typename T>
static int bn_bwd_apply_impl(const T* dout, int d_cs, int d_co, const T* out, int o_cs, int o_co, const T* x, int x_cs,
                                 int x_co, const float* mean, const float* invstd, const float* gamma, const float* c1, const float* c2,
                                 T* dx, int dx_cs, int dx_co, T* gout, int g_cs, int g_co, int g_accumulate, long long P, int C,
                                 void* stream) {
    UNET_CHECK_ARG(dout && x && mean && invstd && c1 && c2 && dx && P > 0 && C > 0 && (C & 3) == 0, "bn_bwd_apply: bad args");
    UNET_CHECK_ARG(unet::slice_ok(d_cs, d_co, C) && unet::slice_ok(x_cs, x_co, C) && unet::slice_ok(dx_cs, dx_co, C), "bn_bwd_apply: bad slice");
    if (out) UNET_CHECK_ARG(unet::slice_ok(o_cs, o_co, C), "bn_bwd_apply: bad out slice");
    if (gout) UNET_CHECK_ARG(unet::slice_ok(g_cs, g_co, C), "bn_bwd_apply: bad gout slice");
    const int C4 = C / 4;
    if constexpr (sizeof(T) == 2) {
        if (use_oct() && C % 8 == 0 && d_cs % 8 == 0 && d_co % 8 == 0 && x_cs % 8 == 0 && x_co % 8 == 0 && dx_cs % 8 == 0 && dx_co % 8 == 0 &&
            (!out || (o_cs % 8 == 0 && o_co % 8 == 0)) && (!gout || (g_cs % 8 == 0 && g_co % 8 == 0)) && unet::aligned16(dout) && unet::aligned16(x) &&
            unet::aligned16(dx) && (!out || unet::aligned16(out)) && (!gout || unet::aligned16(gout))) {
            hipLaunchKernelGGL(bn_bwd_apply_oct_kernel, dim3(ew_grid(P * (C / 8), 256)), dim3(256), 0, ST, dout, d_cs, d_co, out, o_cs, o_co, x, x_cs,
                               x_co, mean, invstd, gamma, c1, c2, dx, dx_cs, dx_co, gout, g_cs, g_co, g_accumulate, P, C / 8);
            UNET_CHECK_LAUNCH();
            return UNET_OK;
        }
    }
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ew_grid(P * C4, 256)), dim3(256), 0, ST, dout, d_cs, d_co, out, o_cs, o_co, x, x_cs, x_co,
                       mean, invstd, gamma, c1, c2, dx, dx_cs, dx_co, gout, g_cs, g_co, g_accumulate, P, C4);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_bn_bwd_apply(const float* dout, int d_cs, int d_co, const float* out, int o_cs, int o_co, const float* x, int x_cs,
                                 int x_co, const float* mean, const float* invstd, const float* gamma, const float* c1, const float* c2,
                                 float* dx, int dx_cs, int dx_co, float* gout, int g_cs, int g_co, int g_accumulate, long long P, int C,
                                 void* stream) { return bn_bwd_apply_impl<float>(dout, d_cs, d_co, out, o_cs, o_co, x, x_cs, x_co, mean, invstd, gamma, c1, c2, dx, dx_cs, dx_co, gout, g_cs, g_co, g_accumulate, P, C, stream); }
extern "C" int unet_bn_bwd_apply_bf16(const unet_bf16* dout, int d_cs, int d_co, const unet_bf16* out, int o_cs, int o_co, const unet_bf16* x, int x_cs,
                                 int x_co, const float* mean, const float* invstd, const float* gamma, const float* c1, const float* c2,
                                 unet_bf16* dx, int dx_cs, int dx_co, unet_bf16* gout, int g_cs, int g_co, int g_accumulate, long long P, int C,
                                 void* stream) { return bn_bwd_apply_impl<unet_bf16>(dout, d_cs, d_co, out, o_cs, o_co, x, x_cs, x_co, mean, invstd, gamma, c1, c2, dx, dx_cs, dx_co, gout, g_cs, g_co, g_accumulate, P, C, stream); }

template <typename T>
static int maxpool3x3s2_impl(const T* x, int x_cs, int x_co, T* y, int y_cs, int y_co, uint8_t* idx, int N, int IH, int IW,
                                 int C, int OH, int OW, void* stream) {
    UNET_CHECK_ARG(x && y && N > 0 && C > 0 && (C & 3) == 0, "maxpool: bad args");
    UNET_CHECK_ARG(OH == (IH + 2 - 3) / 2 + 1 && OW == (IW + 2 - 3) / 2 + 1, "maxpool: bad output dims");
    UNET_CHECK_ARG(unet::slice_ok(x_cs, x_co, C) && unet::slice_ok(y_cs, y_co, C), "maxpool: bad slice");
    const int C4 = C / 4;
    hipLaunchKernelGGL((maxpool_kernel<T>), dim3(ew_grid((long long)N * OH * OW * C4, 256)), dim3(256), 0, ST, x, x_cs, x_co, y, y_cs, y_co, idx, N,
                       IH, IW, C4, OH, OW);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_maxpool3x3s2(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, uint8_t* idx, int N, int IH, int IW,
                                 int C, int OH, int OW, void* stream) { return maxpool3x3s2_impl<float>(x, x_cs, x_co, y, y_cs, y_co, idx, N, IH, IW, C, OH, OW, stream); }
extern "C" int unet_maxpool3x3s2_bf16(const unet_bf16* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, uint8_t* idx, int N, int IH, int IW,
                                 int C, int OH, int OW, void* stream) { return maxpool3x3s2_impl<unet_bf16>(x, x_cs, x_co, y, y_cs, y_co, idx, N, IH, IW, C, OH, OW, stream); }

template <typename T>
static int maxpool3x3s2_bwd_impl(const T* dy, int dy_cs, int dy_co, const uint8_t* idx, T* dx, int dx_cs, int dx_co, int N,
                                     int IH, int IW, int C, int OH, int OW, int accumulate, void* stream) {
    UNET_CHECK_ARG(dy && idx && dx && N > 0 && C > 0 && (C & 3) == 0, "maxpool_bwd: bad args");
    UNET_CHECK_ARG(unet::slice_ok(dy_cs, dy_co, C) && unet::slice_ok(dx_cs, dx_co, C), "maxpool_bwd: bad slice");
    const int C4 = C / 4;
    hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(ew_grid((long long)N * IH * IW * C4, 256)), dim3(256), 0, ST, dy, dy_cs, dy_co, idx, dx, dx_cs,
                       dx_co, N, IH, IW, C4, OH, OW, accumulate);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_maxpool3x3s2_bwd(const float* dy, int dy_cs, int dy_co, const uint8_t* idx, float* dx, int dx_cs, int dx_co, int N,
                                     int IH, int IW, int C, int OH, int OW, int accumulate, void* stream) { return maxpool3x3s2_bwd_impl<float>(dy, dy_cs, dy_co, idx, dx, dx_cs, dx_co, N, IH, IW, C, OH, OW, accumulate, stream); }
extern "C" int unet_maxpool3x3s2_bwd_bf16(const unet_bf16* dy, int dy_cs, int dy_co, const uint8_t* idx, unet_bf16* dx, int dx_cs, int dx_co, int N,
                                     int IH, int IW, int C, int OH, int OW, int accumulate, void* stream) { return maxpool3x3s2_bwd_impl<unet_bf16>(dy, dy_cs, dy_co, idx, dx, dx_cs, dx_co, N, IH, IW, C, OH, OW, accumulate, stream); }

template <typename T>
static int avgpool2_ceil_impl(const T* x, int x_cs, int x_co, T* y, int y_cs, int y_co, int N, int IH, int IW, int C, int OH,
                                  int OW, void* stream) {
    UNET_CHECK_ARG(x && y && N > 0 && C > 0 && (C & 3) == 0, "avgpool: bad args");
    UNET_CHECK_ARG(OH == (IH + 1) / 2 && OW == (IW + 1) / 2, "avgpool: bad output dims");
    UNET_CHECK_ARG(unet::slice_ok(x_cs, x_co, C) && unet::slice_ok(y_cs, y_co, C), "avgpool: bad slice");
    const int C4 = C / 4;
    hipLaunchKernelGGL((avgpool_kernel<T>), dim3(ew_grid((long long)N * OH * OW * C4, 256)), dim3(256), 0, ST, x, x_cs, x_co, y, y_cs, y_co, N, IH,
                       IW, C4, OH, OW);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_avgpool2_ceil(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, int N, int IH, int IW, int C, int OH,
                                  int OW, void* stream) { return avgpool2_ceil_impl<float>(x, x_cs, x_co, y, y_cs, y_co, N, IH, IW, C, OH, OW, stream); }
extern "C" int unet_avgpool2_ceil_bf16(const unet_bf16* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, int N, int IH, int IW, int C, int OH,
                                  int OW, void* stream) { return avgpool2_ceil_impl<unet_bf16>(x, x_cs, x_co, y, y_cs, y_co, N, IH, IW, C, OH, OW, stream); }

template <typename T>
static int avgpool2_ceil_bwd_impl(const T* dy, int dy_cs, int dy_co, T* dx, int dx_cs, int dx_co, int N, int IH, int IW, int C,
                                      int OH, int OW, int accumulate, void* stream) {
    UNET_CHECK_ARG(dy && dx && N > 0 && C > 0 && (C & 3) == 0, "avgpool_bwd: bad args");
    UNET_CHECK_ARG(OH == (IH + 1) / 2 && OW == (IW + 1) / 2, "avgpool_bwd: bad output dims");
    UNET_CHECK_ARG(unet::slice_ok(dy_cs, dy_co, C) && unet::slice_ok(dx_cs, dx_co, C), "avgpool_bwd: bad slice");
    const int C4 = C / 4;
    hipLaunchKernelGGL((avgpool_bwd_kernel<T>), dim3(ew_grid((long long)N * IH * IW * C4, 256)), dim3(256), 0, ST, dy, dy_cs, dy_co, dx, dx_cs,
                       dx_co, N, IH, IW, C4, OH, OW, accumulate);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_avgpool2_ceil_bwd(const float* dy, int dy_cs, int dy_co, float* dx, int dx_cs, int dx_co, int N, int IH, int IW, int C,
                                      int OH, int OW, int accumulate, void* stream) { return avgpool2_ceil_bwd_impl<float>(dy, dy_cs, dy_co, dx, dx_cs, dx_co, N, IH, IW, C, OH, OW, accumulate, stream); }
extern "C" int unet_avgpool2_ceil_bwd_bf16(const unet_bf16* dy, int dy_cs, int dy_co, unet_bf16* dx, int dx_cs, int dx_co, int N, int IH, int IW, int C,
                                      int OH, int OW, int accumulate, void* stream) { return avgpool2_ceil_bwd_impl<unet_bf16>(dy, dy_cs, dy_co, dx, dx_cs, dx_co, N, IH, IW, C, OH, OW, accumulate, stream); }

template <typename T>
static int shuffle_blur_impl(const T* yc, int yc_cs, int yc_co, T* X, int X_cs, int X_co, int N, int h, int w, int Cu,
                                 int do_blur, void* stream) {
    UNET_CHECK_ARG(yc && X && N > 0 && h > 0 && w > 0 && Cu > 0, "shuffle_blur: bad args");
    UNET_CHECK_ARG(unet::slice_ok(yc_cs, yc_co, 4 * Cu) && X_cs > 0 && X_co >= 0 && X_co + Cu <= X_cs, "shuffle_blur: bad slice");
    constexpr int V = VecOf<T>::V;
    // (with blur every thread re-reads three neighbours' 4 V channels: measured no faster than the scalar form, fp32 slower -- kept scalar)
    if (!do_blur && Cu % V == 0 && X_cs % V == 0 && X_co % V == 0 && yc_cs % V == 0 && yc_co % V == 0 && unet::aligned16(yc) && unet::aligned16(X)) {
        hipLaunchKernelGGL((shuffle_blur_vec_kernel<T>), dim3(ew_grid((long long)N * h * w * (Cu / V), 256)), dim3(256), 0, ST, yc, yc_cs, yc_co, X,
                           X_cs, X_co, N, h, w, Cu / V, do_blur);
    } else {
        hipLaunchKernelGGL((shuffle_blur_kernel<T>), dim3(ew_grid((long long)N * h * w * Cu, 256)), dim3(256), 0, ST, yc, yc_cs, yc_co, X, X_cs, X_co,
                           N, h, w, Cu, do_blur);
    }
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_shuffle_blur(const float* yc, int yc_cs, int yc_co, float* X, int X_cs, int X_co, int N, int h, int w, int Cu,
                                 int do_blur, void* stream) { return shuffle_blur_impl<float>(yc, yc_cs, yc_co, X, X_cs, X_co, N, h, w, Cu, do_blur, stream); }
extern "C" int unet_shuffle_blur_bf16(const unet_bf16* yc, int yc_cs, int yc_co, unet_bf16* X, int X_cs, int X_co, int N, int h, int w, int Cu,
                                 int do_blur, void* stream) { return shuffle_blur_impl<unet_bf16>(yc, yc_cs, yc_co, X, X_cs, X_co, N, h, w, Cu, do_blur, stream); }

template <typename T>
static int shuffle_bwd_xmask_impl(const T* dX, int dX_cs, int dX_co, const T* X, int X_cs, int X_co, T* dyc, int dyc_cs, int dyc_co, int N, int h, int w,
                                  int Cu, void* stream) {
    constexpr int V = VecOf<T>::V;
    UNET_CHECK_ARG(dX && X && dyc && N > 0 && h > 0 && w > 0 && Cu > 0 && Cu % V == 0, "shuffle_bwd_xmask: bad args (Cu must be a multiple of the 16-byte vector)");
    UNET_CHECK_ARG(dX_cs > 0 && dX_co >= 0 && dX_co + Cu <= dX_cs && dX_cs % V == 0 && dX_co % V == 0 && X_cs > 0 && X_co >= 0 && X_co + Cu <= X_cs &&
                   X_cs % V == 0 && X_co % V == 0 && dyc_cs % V == 0 && dyc_co % V == 0 && dyc_co + 4 * Cu <= dyc_cs, "shuffle_bwd_xmask: bad slice");
    hipLaunchKernelGGL((shuffle_bwd_xmask_kernel<T>), dim3(ew_grid((long long)N * h * w * (Cu / V), 256)), dim3(256), 0, ST, dX, dX_cs, dX_co, X, X_cs, X_co,
                       dyc, dyc_cs, dyc_co, N, h, w, Cu / V);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_shuffle_bwd_xmask(const float* dX, int dX_cs, int dX_co, const float* X, int X_cs, int X_co, float* dyc, int dyc_cs, int dyc_co, int N,
                                      int h, int w, int Cu, void* stream) { return shuffle_bwd_xmask_impl<float>(dX, dX_cs, dX_co, X, X_cs, X_co, dyc, dyc_cs, dyc_co, N, h, w, Cu, stream); }
extern "C" int unet_shuffle_bwd_xmask_bf16(const unet_bf16* dX, int dX_cs, int dX_co, const unet_bf16* X, int X_cs, int X_co, unet_bf16* dyc, int dyc_cs,
                                           int dyc_co, int N, int h, int w, int Cu, void* stream) { return shuffle_bwd_xmask_impl<unet_bf16>(dX, dX_cs, dX_co, X, X_cs, X_co, dyc, dyc_cs, dyc_co, N, h, w, Cu, stream); }

template <typename T>
static int shuffle_blur_bwd_impl(const T* dX, int dX_cs, int dX_co, const T* yc, int yc_cs, int yc_co, T* dyc, int dyc_cs,
                                     int dyc_co, int N, int h, int w, int Cu, int do_blur, void* stream) {
    UNET_CHECK_ARG(dX && yc && dyc && N > 0 && h > 0 && w > 0 && Cu > 0, "shuffle_blur_bwd: bad args");
    UNET_CHECK_ARG(unet::slice_ok(yc_cs, yc_co, 4 * Cu) && unet::slice_ok(dyc_cs, dyc_co, 4 * Cu) && dX_cs > 0 && dX_co + Cu <= dX_cs,
                   "shuffle_blur_bwd: bad slice");
    constexpr int V = VecOf<T>::V;
    if (Cu % V == 0 && dX_cs % V == 0 && dX_co % V == 0 && yc_cs % V == 0 && yc_co % V == 0 && dyc_cs % V == 0 && dyc_co % V == 0 &&
        unet::aligned16(dX) && unet::aligned16(yc) && unet::aligned16(dyc)) {
        hipLaunchKernelGGL((shuffle_blur_bwd_vec_kernel<T>), dim3(ew_grid((long long)N * h * w * (Cu / V), 256)), dim3(256), 0, ST, dX, dX_cs, dX_co,
                           yc, yc_cs, yc_co, dyc, dyc_cs, dyc_co, N, h, w, Cu / V, do_blur);
    } else {
        hipLaunchKernelGGL((shuffle_blur_bwd_kernel<T>), dim3(ew_grid((long long)N * h * w * Cu, 256)), dim3(256), 0, ST, dX, dX_cs, dX_co, yc, yc_cs,
                           yc_co, dyc, dyc_cs, dyc_co, N, h, w, Cu, do_blur);
    }
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_shuffle_blur_bwd(const float* dX, int dX_cs, int dX_co, const float* yc, int yc_cs, int yc_co, float* dyc, int dyc_cs,
                                     int dyc_co, int N, int h, int w, int Cu, int do_blur, void* stream) { return shuffle_blur_bwd_impl<float>(dX, dX_cs, dX_co, yc, yc_cs, yc_co, dyc, dyc_cs, dyc_co, N, h, w, Cu, do_blur, stream); }
extern "C" int unet_shuffle_blur_bwd_bf16(const unet_bf16* dX, int dX_cs, int dX_co, const unet_bf16* yc, int yc_cs, int yc_co, unet_bf16* dyc, int dyc_cs,
                                     int dyc_co, int N, int h, int w, int Cu, int do_blur, void* stream) { return shuffle_blur_bwd_impl<unet_bf16>(dX, dX_cs, dX_co, yc, yc_cs, yc_co, dyc, dyc_cs, dyc_co, N, h, w, Cu, do_blur, stream); }

template <typename T>
static int resize_nearest_impl(const T* x, int x_cs, int x_co, T* y, int y_cs, int y_co, int N, int IH, int IW, int OH, int OW, int C,
                               void* stream) {
    UNET_CHECK_ARG(x && y && N > 0 && C > 0, "resize_nearest: bad args");
    UNET_CHECK_ARG(pslice_ok(x_cs, x_co, C) && pslice_ok(y_cs, y_co, C), "resize_nearest: bad slice");
    const int C4 = c4of(C);
    hipLaunchKernelGGL((resize_nearest_kernel<T>), dim3(ew_grid((long long)N * OH * OW * C4, 256)), dim3(256), 0, ST, x, x_cs, x_co, y, y_cs, y_co,
                       N, IH, IW, OH, OW, C4);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_resize_nearest(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, int N, int IH, int IW, int OH, int OW,
                                   int C, void* stream) { return resize_nearest_impl<float>(x, x_cs, x_co, y, y_cs, y_co, N, IH, IW, OH, OW, C, stream); }
extern "C" int unet_resize_nearest_bf16(const unet_bf16* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, int N, int IH, int IW, int OH, int OW,
                                   int C, void* stream) { return resize_nearest_impl<unet_bf16>(x, x_cs, x_co, y, y_cs, y_co, N, IH, IW, OH, OW, C, stream); }

template <typename T>
static int resize_nearest_bwd_impl(const T* dy, int dy_cs, int dy_co, T* dx, int dx_cs, int dx_co, int N, int IH, int IW, int OH, int OW,
                                   int C, void* stream) {
    UNET_CHECK_ARG(dy && dx && N > 0 && C > 0, "resize_nearest_bwd: bad args");
    UNET_CHECK_ARG(pslice_ok(dy_cs, dy_co, C) && pslice_ok(dx_cs, dx_co, C), "resize_nearest_bwd: bad slice");
    const int C4 = c4of(C);
    hipLaunchKernelGGL((resize_nearest_bwd_kernel<T>), dim3(ew_grid((long long)N * IH * IW * C4, 256)), dim3(256), 0, ST, dy, dy_cs, dy_co, dx, dx_cs,
                       dx_co, N, IH, IW, OH, OW, C4);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_resize_nearest_bwd(const float* dy, int dy_cs, int dy_co, float* dx, int dx_cs, int dx_co, int N, int IH, int IW, int OH,
                                       int OW, int C, void* stream) { return resize_nearest_bwd_impl<float>(dy, dy_cs, dy_co, dx, dx_cs, dx_co, N, IH, IW, OH, OW, C, stream); }
extern "C" int unet_resize_nearest_bwd_bf16(const unet_bf16* dy, int dy_cs, int dy_co, unet_bf16* dx, int dx_cs, int dx_co, int N, int IH, int IW, int OH,
                                       int OW, int C, void* stream) { return resize_nearest_bwd_impl<unet_bf16>(dy, dy_cs, dy_co, dx, dx_cs, dx_co, N, IH, IW, OH, OW, C, stream); }

template <typename T>
static int nchw_to_nhwc_impl(const float* x, T* y, int y_cs, int y_co, int N, int C, int H, int W, void* stream) {
    UNET_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0 && y_co >= 0 && y_co + C <= y_cs, "nchw_to_nhwc: bad args");
    hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(ew_grid((long long)N * H * W, 256)), dim3(256), 0, ST, x, y, y_cs, y_co, N, C, H, W);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_nchw_to_nhwc(const float* x, float* y, int y_cs, int y_co, int N, int C, int H, int W, void* stream) { return nchw_to_nhwc_impl<float>(x, y, y_cs, y_co, N, C, H, W, stream); }
extern "C" int unet_nchw_to_nhwc_bf16(const float* x, unet_bf16* y, int y_cs, int y_co, int N, int C, int H, int W, void* stream) { return nchw_to_nhwc_impl<unet_bf16>(x, y, y_cs, y_co, N, C, H, W, stream); }

extern "C" int unet_nhwc_to_nchw(const float* x, int x_cs, int x_co, float* y, int N, int C, int H, int W, void* stream) {
    UNET_CHECK_ARG(x && y && N > 0 && C > 0 && H > 0 && W > 0 && x_co >= 0 && x_co + C <= x_cs, "nhwc_to_nchw: bad args");
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(ew_grid((long long)N * H * W, 256)), dim3(256), 0, ST, x, x_cs, x_co, y, N, C, H, W);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <typename T>
static int copy_slice_impl(const T* x, int x_cs, int x_co, T* y, int y_cs, int y_co, long long P, int C, int accumulate,
                               void* stream) {
    UNET_CHECK_ARG(x && y && P > 0 && C > 0, "copy_slice: bad args");
    UNET_CHECK_ARG(pslice_ok(x_cs, x_co, C) && pslice_ok(y_cs, y_co, C), "copy_slice: bad slice");
    const int C4 = c4of(C);
    hipLaunchKernelGGL((copy_slice_kernel<T>), dim3(ew_grid(P * C4, 256)), dim3(256), 0, ST, x, x_cs, x_co, y, y_cs, y_co, P, C4, accumulate);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_copy_slice(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, long long P, int C, int accumulate,
                               void* stream) { return copy_slice_impl<float>(x, x_cs, x_co, y, y_cs, y_co, P, C, accumulate, stream); }
extern "C" int unet_copy_slice_bf16(const unet_bf16* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, long long P, int C, int accumulate,
                               void* stream) { return copy_slice_impl<unet_bf16>(x, x_cs, x_co, y, y_cs, y_co, P, C, accumulate, stream); }

template <typename T>
static int relu_mask_impl(const T* g, int g_cs, int g_co, const T* ref, int r_cs, int r_co, T* y, int y_cs, int y_co, long long P, int C,
                          void* stream) {
    UNET_CHECK_ARG(g && ref && y && P > 0 && C > 0, "relu_mask: bad args");
    UNET_CHECK_ARG(pslice_ok(g_cs, g_co, C) && pslice_ok(r_cs, r_co, C) && pslice_ok(y_cs, y_co, C), "relu_mask: bad slice");
    const int C4 = c4of(C);
    hipLaunchKernelGGL((relu_mask_kernel<T>), dim3(ew_grid(P * C4, 256)), dim3(256), 0, ST, g, g_cs, g_co, ref, r_cs, r_co, y, y_cs, y_co, P, C4);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_relu_mask(const float* g, int g_cs, int g_co, const float* ref, int r_cs, int r_co, float* y, int y_cs, int y_co,
                              long long P, int C, void* stream) { return relu_mask_impl<float>(g, g_cs, g_co, ref, r_cs, r_co, y, y_cs, y_co, P, C, stream); }
extern "C" int unet_relu_mask_bf16(const unet_bf16* g, int g_cs, int g_co, const unet_bf16* ref, int r_cs, int r_co, unet_bf16* y, int y_cs, int y_co,
                              long long P, int C, void* stream) { return relu_mask_impl<unet_bf16>(g, g_cs, g_co, ref, r_cs, r_co, y, y_cs, y_co, P, C, stream); }

// fp32 -> bf16 copy of a channel slice (bf16 self-attention: fp32 weight-gradient results into the bf16 gradient of the fused QKV tensor)
__global__ __launch_bounds__(256) void cast_slice_kernel(const float* __restrict__ x, int x_cs, int x_co, unet_bf16* __restrict__ y, int y_cs,
                                                         int y_co, long long P, int C4) {
    for (QuadWalk w(C4); w.p < P; w.next())
        st4(y + (size_t)w.p * y_cs + y_co + 4 * w.c4, ld4(x + (size_t)w.p * x_cs + x_co + 4 * w.c4));
}
extern "C" int unet_cast_slice_bf16(const float* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, long long P, int C, void* stream) {
    UNET_CHECK_ARG(x && y && P > 0 && C > 0, "cast_slice: bad args");
    UNET_CHECK_ARG(pslice_ok(x_cs, x_co, C) && pslice_ok(y_cs, y_co, C), "cast_slice: bad slice");
    const int C4 = c4of(C);
    hipLaunchKernelGGL(cast_slice_kernel, dim3(ew_grid(P * C4, 256)), dim3(256), 0, ST, x, x_cs, x_co, y, y_cs, y_co, P, C4);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" size_t unet_colsum_workspace(long long P, int C) { return (size_t)stats_rows(P) * roundup(C, 4); }

extern "C" int unet_colsum(const float* x, int x_cs, int x_co, long long P, int C, float* out, float* workspace, void* stream) {
    UNET_CHECK_ARG(x && out && workspace && P > 0 && C > 0, "colsum: bad args");
    UNET_CHECK_ARG(pslice_ok(x_cs, x_co, C), "colsum: bad slice");
    const int rows = stats_rows(P), C4 = c4of(C), Cp = 4 * C4, TC = pick_tc(C4);
    hipLaunchKernelGGL(colsum_kernel, dim3(rows), dim3(256), 0, ST, x, x_cs, x_co, P, C4, TC, workspace, Cp);
    UNET_CHECK_LAUNCH();
    hipLaunchKernelGGL(rows_sum_kernel, dim3(cdiv(C, 32)), dim3(256), 0, ST, workspace, rows, Cp, C, out);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <typename T>
static int dot_impl(const T* x, int x_cs, int x_co, const T* y, int y_cs, int y_co, long long P, int C, float* out, float* workspace,
                    void* stream) {
    UNET_CHECK_ARG(x && y && out && workspace && P > 0 && C > 0, "dot: bad args");
    UNET_CHECK_ARG(pslice_ok(x_cs, x_co, C) && pslice_ok(y_cs, y_co, C), "dot: bad slice");
    const int rows = stats_rows(P), C4 = c4of(C), Cp = 4 * C4, TC = pick_tc(C4);
    hipLaunchKernelGGL((dot_kernel<T>), dim3(rows), dim3(256), 0, ST, x, x_cs, x_co, y, y_cs, y_co, P, C4, TC, workspace, Cp);
    UNET_CHECK_LAUNCH();
    hipLaunchKernelGGL(sum_all_kernel, dim3(1), dim3(256), 0, ST, workspace, (long long)rows * Cp, out);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_dot(const float* x, int x_cs, int x_co, const float* y, int y_cs, int y_co, long long P, int C, float* out,
                        float* workspace, void* stream) { return dot_impl<float>(x, x_cs, x_co, y, y_cs, y_co, P, C, out, workspace, stream); }
extern "C" int unet_dot_bf16(const unet_bf16* x, int x_cs, int x_co, const unet_bf16* y, int y_cs, int y_co, long long P, int C, float* out,
                        float* workspace, void* stream) { return dot_impl<unet_bf16>(x, x_cs, x_co, y, y_cs, y_co, P, C, out, workspace, stream); }

static int ce_rows(long long P) {
    long long r = (P + 255) / 256;
    if (r > 1024) r = 1024;
    if (r < 1) r = 1;
    return (int)r;
}

extern "C" size_t unet_ce_workspace(long long P) { return (size_t)2 * ce_rows(P); }

extern "C" int unet_ce_fwd(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, float* loss,
                           float* denom, float* workspace, void* stream) {
    UNET_CHECK_ARG(z && target && loss && denom && workspace && P > 0 && C > 0 && C <= CE_MAXC, "ce_fwd: bad args");
    UNET_CHECK_ARG(z_co >= 0 && z_co + C <= z_cs, "ce_fwd: bad slice");
    const int rows = ce_rows(P);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(rows), dim3(256), 0, ST, z, z_cs, z_co, target, weight, P, C, workspace);
    UNET_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, ST, workspace, rows, loss, denom, (float*)nullptr);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_ce_fwd_parts(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C,
                                 float* numden, float* workspace, void* stream) {
    UNET_CHECK_ARG(z && target && numden && workspace && P > 0 && C > 0 && C <= CE_MAXC, "ce_fwd_parts: bad args");
    UNET_CHECK_ARG(z_co >= 0 && z_co + C <= z_cs, "ce_fwd_parts: bad slice");
    const int rows = ce_rows(P);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(rows), dim3(256), 0, ST, z, z_cs, z_co, target, weight, P, C, workspace);
    UNET_CHECK_LAUNCH();
    hipLaunchKernelGGL(ce_finalize_kernel, dim3(1), dim3(64), 0, ST, workspace, rows, (float*)nullptr, (float*)nullptr, numden);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <typename T>
static int ce_bwd_impl(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C,
                           const float* denom, float gscale, T* dz, int dz_cs, int dz_co, void* stream) {
    UNET_CHECK_ARG(z && target && denom && dz && P > 0 && C > 0 && C <= CE_MAXC, "ce_bwd: bad args");
    UNET_CHECK_ARG(z_co >= 0 && z_co + C <= z_cs && dz_co >= 0 && dz_co + C <= dz_cs, "ce_bwd: bad slice");
    hipLaunchKernelGGL((ce_bwd_kernel<T>), dim3(ew_grid(P, 256)), dim3(256), 0, ST, z, z_cs, z_co, target, weight, P, C, denom, gscale, dz, dz_cs,
                       dz_co);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_ce_bwd(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C,
                           const float* denom, float gscale, float* dz, int dz_cs, int dz_co, void* stream) { return ce_bwd_impl<float>(z, z_cs, z_co, target, weight, P, C, denom, gscale, dz, dz_cs, dz_co, stream); }
extern "C" int unet_ce_bwd_bf16(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C,
                           const float* denom, float gscale, unet_bf16* dz, int dz_cs, int dz_co, void* stream) { return ce_bwd_impl<unet_bf16>(z, z_cs, z_co, target, weight, P, C, denom, gscale, dz, dz_cs, dz_co, stream); }

extern "C" int unet_focal_fwd(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, float gamma,
                              float* loss, float* workspace, void* stream) {
    UNET_CHECK_ARG(z && target && loss && workspace && P > 0 && C > 0 && C <= CE_MAXC && gamma >= 0.f, "focal_fwd: bad args");
    UNET_CHECK_ARG(z_co >= 0 && z_co + C <= z_cs, "focal_fwd: bad slice");
    const int rows = ce_rows(P);
    hipLaunchKernelGGL(focal_fwd_kernel, dim3(rows), dim3(256), 0, ST, z, z_cs, z_co, target, weight, P, C, gamma, workspace);
    UNET_CHECK_LAUNCH();
    hipLaunchKernelGGL(regloss_finalize_kernel, dim3(1), dim3(64), 0, ST, workspace, rows, P, loss);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <typename T>
static int focal_bwd_impl(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, float gamma,
                          float gscale, T* dz, int dz_cs, int dz_co, void* stream) {
    UNET_CHECK_ARG(z && target && dz && P > 0 && C > 0 && C <= CE_MAXC && gamma >= 0.f, "focal_bwd: bad args");
    UNET_CHECK_ARG(z_co >= 0 && z_co + C <= z_cs && dz_co >= 0 && dz_co + C <= dz_cs, "focal_bwd: bad slice");
    hipLaunchKernelGGL((focal_bwd_kernel<T>), dim3(ew_grid(P, 256)), dim3(256), 0, ST, z, z_cs, z_co, target, weight, P, C, gamma, gscale, dz, dz_cs,
                       dz_co);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_focal_bwd(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, float gamma,
                              float gscale, float* dz, int dz_cs, int dz_co, void* stream) { return focal_bwd_impl<float>(z, z_cs, z_co, target, weight, P, C, gamma, gscale, dz, dz_cs, dz_co, stream); }
extern "C" int unet_focal_bwd_bf16(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, float gamma,
                              float gscale, unet_bf16* dz, int dz_cs, int dz_co, void* stream) { return focal_bwd_impl<unet_bf16>(z, z_cs, z_co, target, weight, P, C, gamma, gscale, dz, dz_cs, dz_co, stream); }

extern "C" int unet_regloss_fwd(const float* z, int z_cs, int z_co, const float* target, long long P, int kind, float beta, float* loss,
                                float* workspace, void* stream) {
    UNET_CHECK_ARG(z && target && loss && workspace && P > 0 && kind >= 0 && kind <= 2 && (kind != 2 || beta > 0.f), "regloss_fwd: bad args");
    UNET_CHECK_ARG(z_co >= 0 && z_co < z_cs, "regloss_fwd: bad slice");
    const int rows = ce_rows(P);
    hipLaunchKernelGGL(regloss_fwd_kernel, dim3(rows), dim3(256), 0, ST, z, z_cs, z_co, target, P, kind, beta, workspace);
    UNET_CHECK_LAUNCH();
    hipLaunchKernelGGL(regloss_finalize_kernel, dim3(1), dim3(64), 0, ST, workspace, rows, P, loss);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_regloss_bwd(const float* z, int z_cs, int z_co, const float* target, long long P, int kind, float beta, float gscale,
                                float* dz, int dz_cs, int dz_co, void* stream) {
    UNET_CHECK_ARG(z && target && dz && P > 0 && kind >= 0 && kind <= 2 && (kind != 2 || beta > 0.f), "regloss_bwd: bad args");
    UNET_CHECK_ARG(z_co >= 0 && z_co < z_cs && dz_co >= 0 && dz_co < dz_cs, "regloss_bwd: bad slice");
    hipLaunchKernelGGL(regloss_bwd_kernel, dim3(ew_grid(P, 256)), dim3(256), 0, ST, z, z_cs, z_co, target, P, kind, beta, gscale, dz, dz_cs,
                       dz_co);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_softmax_argmax(const float* z, int z_cs, int z_co, int N, int H, int W, int C, float* probs_nchw, int64_t* argmax,
                                   void* stream) {
    UNET_CHECK_ARG(z && N > 0 && H > 0 && W > 0 && C > 0 && C <= CE_MAXC && z_co + C <= z_cs, "softmax_argmax: bad args");
    hipLaunchKernelGGL(softmax_argmax_kernel, dim3(ew_grid((long long)N * H * W, 256)), dim3(256), 0, ST, z, z_cs, z_co, N, (long long)H * W, C,
                       probs_nchw, argmax);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_row_softmax(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, long long P, int C, void* stream) {
    UNET_CHECK_ARG(x && y && P > 0 && C > 0 && x_co + C <= x_cs && y_co + C <= y_cs, "row_softmax: bad args");
    hipLaunchKernelGGL((row_softmax_kernel<float>), dim3((unsigned)(P < 65536 ? P : 65536)), dim3(256), 0, ST, x, x_cs, x_co, y, y_cs, y_co, P, C);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_row_softmax_bf16(const float* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, long long P, int C, void* stream) {
    UNET_CHECK_ARG(x && y && P > 0 && C > 0 && x_co + C <= x_cs && y_co + C <= y_cs, "row_softmax_bf16: bad args");
    hipLaunchKernelGGL((row_softmax_kernel<unet_bf16>), dim3((unsigned)(P < 65536 ? P : 65536)), dim3(256), 0, ST, x, x_cs, x_co, y, y_cs, y_co, P, C);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_row_softmax_bwd(const float* y, int y_cs, int y_co, const float* dy, int dy_cs, int dy_co, float* dx, int dx_cs,
                                    int dx_co, long long P, int C, void* stream) {
    UNET_CHECK_ARG(y && dy && dx && P > 0 && C > 0 && y_co + C <= y_cs && dy_co + C <= dy_cs && dx_co + C <= dx_cs, "row_softmax_bwd: bad args");
    hipLaunchKernelGGL((row_softmax_bwd_kernel<float>), dim3((unsigned)(P < 65536 ? P : 65536)), dim3(256), 0, ST, y, y_cs, y_co, dy, dy_cs, dy_co, dx,
                       dx_cs, dx_co, P, C);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
extern "C" int unet_row_softmax_bwd_bf16(const unet_bf16* y, int y_cs, int y_co, const float* dy, int dy_cs, int dy_co, unet_bf16* dx, int dx_cs,
                                         int dx_co, long long P, int C, void* stream) {
    UNET_CHECK_ARG(y && dy && dx && P > 0 && C > 0 && y_co + C <= y_cs && dy_co + C <= dy_cs && dx_co + C <= dx_cs, "row_softmax_bwd_bf16: bad args");
    hipLaunchKernelGGL((row_softmax_bwd_kernel<unet_bf16>), dim3((unsigned)(P < 65536 ? P : 65536)), dim3(256), 0, ST, y, y_cs, y_co, dy, dy_cs, dy_co,
                       dx, dx_cs, dx_co, P, C);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

static void fill_adam_args(AdamArgs& a, const float* lr, float mom, float sqr_mom, float eps, float wd, int step, float grad_scale);

extern "C" int unet_adam_hyper_floats(void) { return (int)(sizeof(AdamArgs) / sizeof(float)); }

/* host helper: the hyper-parameter block unet_adam_step_dev reads from device memory */
extern "C" int unet_adam_fill_hyper(float* hyper_host, const float* lr, float mom, float sqr_mom, float eps, float wd, int step,
                                    float grad_scale) {
    UNET_CHECK_ARG(hyper_host && lr && step >= 1, "adam_fill_hyper: bad args");
    fill_adam_args(*reinterpret_cast<AdamArgs*>(hyper_host), lr, mom, sqr_mom, eps, wd, step, grad_scale);
    return UNET_OK;
}

extern "C" int unet_adam_step_dev(float* p, const float* g, float* m, float* v, const uint8_t* code, long long n, const float* hyper_dev,
                                  void* stream) {
    UNET_CHECK_ARG(p && g && m && v && code && hyper_dev && n > 0, "adam_step_dev: bad args");
    hipLaunchKernelGGL(adam_dev_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ST, p, g, m, v, code, n,
                       reinterpret_cast<const AdamArgs*>(hyper_dev));
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

static void fill_adam_args(AdamArgs& a, const float* lr, float mom, float sqr_mom, float eps, float wd, int step, float grad_scale) {
    const double debias1 = 1.0 - pow((double)mom, (double)step);
    const double debias2 = 1.0 - pow((double)sqr_mom, (double)step);
    for (int i = 0; i < 4; ++i) {
        a.decay[i] = (float)(1.0 - (double)lr[i] * (double)wd);
        a.step_size[i] = (float)(-(double)lr[i] / debias1);
    }
    a.mom = mom; a.one_minus_mom = (float)(1.0 - (double)mom);
    a.sqr_mom = sqr_mom; a.one_minus_sqr = (float)(1.0 - (double)sqr_mom);
    a.debias2 = (float)debias2; a.eps = eps; a.grad_scale = grad_scale;
}

extern "C" int unet_adam_step(float* p, const float* g, float* m, float* v, const uint8_t* code, long long n, const float* lr, float mom,
                              float sqr_mom, float eps, float wd, int step, float grad_scale, void* stream) {
    UNET_CHECK_ARG(p && g && m && v && code && lr && n > 0 && step >= 1, "adam_step: bad args");
    AdamArgs a;
    fill_adam_args(a, lr, mom, sqr_mom, eps, wd, step, grad_scale);
    hipLaunchKernelGGL(adam_kernel, dim3(ew_grid(n, 256)), dim3(256), 0, ST, p, g, m, v, code, n, a);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}


extern "C" int unet_mosaic_accumulate(const float* probs_nchw, int C, int th, int tw, float* mosaic, int32_t* count, int MH, int MW, int y0,
                                      int x0, void* stream) {
    UNET_CHECK_ARG(probs_nchw && mosaic && count && C > 0 && th > 0 && tw > 0 && MH > 0 && MW > 0, "mosaic_accumulate: bad args");
    hipLaunchKernelGGL(mosaic_acc_kernel, dim3(ew_grid((long long)th * tw, 256)), dim3(256), 0, ST, probs_nchw, C, th, tw, mosaic, count, MH, MW,
                       y0, x0);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_mosaic_finalize(float* mosaic, const int32_t* count, int C, int MH, int MW, uint8_t* argmax, void* stream) {
    UNET_CHECK_ARG(mosaic && count && C > 0 && MH > 0 && MW > 0, "mosaic_finalize: bad args");
    hipLaunchKernelGGL(mosaic_fin_kernel, dim3(ew_grid((long long)MH * MW, 256)), dim3(256), 0, ST, mosaic, count, C, MH, MW, argmax);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
