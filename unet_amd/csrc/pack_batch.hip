// All packed filter images of a model in ONE launch.
//
// Every training step rewrites the parameters (fused Adam), so every conv's forward and input-gradient filter image is rebuilt once
// per step: 104 launches of a few microseconds each for xresnet34 (0.6 ms of device time plus as many launch gaps).  The images depend
// on nothing but the flat parameter buffer, so one kernel builds them all from a device-resident job table (addresses are static: the
// parameters live in one flat buffer, the packed images are persistent).
//
// Layouts are those of unet_pack_weights (fp32: wp[tap][chunk16][outPad][16], a reduction tail stored channel-transposed) and
// unet_pack_weights_bf16 (wp[tap][chunk32][outPad][32] + the tap-folded tail slabs, conv_common.h); see conv_igemm.hip / conv_bf16.hip.

#include "conv_common.h"

namespace {

struct Job {                 // == unet_pack_job_table entry (56 bytes)
    const float* w;
    void* wp;
    const float* scale;      // optional per-output-channel factor (eval-mode BatchNorm folded into the forward image), mode 0 only
    int Cout, Cin, T, mode, nchunks, outPad;
    unsigned block_begin, pad_;
};
static_assert(sizeof(Job) == 56, "job table entry size");

constexpr int ELEMS_PER_BLOCK = 2048;

__global__ __launch_bounds__(256) void pack_batch_kernel(const Job* __restrict__ jobs, int njobs, int bf16) {
    // the job of this workgroup: last entry with block_begin <= blockIdx.x
    int lo = 0, hi = njobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block_begin <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const Job j = jobs[lo];
    const int KC = bf16 ? 32 : 16, sh = bf16 ? 5 : 4;
    const int red = j.mode == 1 ? j.Cout : j.Cin;
    const int out = j.mode == 1 ? j.Cin : j.Cout;
    const size_t main_f32 = (size_t)j.T * j.nchunks * j.outPad * KC;
    const size_t total = bf16 ? unetconv::bf16_image_elems(red, j.outPad, j.T) : unetconv::f32_image_elems(red, out, j.T);
    const size_t base = (size_t)(blockIdx.x - j.block_begin) * ELEMS_PER_BLOCK;
#pragma unroll
    for (int k = 0; k < ELEMS_PER_BLOCK / 256; ++k) {
        const size_t i = base + threadIdx.x + k * 256;
        if (i >= total) break;
        const bool scaled = j.scale != nullptr && j.mode == 0;
        if (bf16) {
            float v = unetconv::bf16_image_value(j.w, j.Cout, j.Cin, j.T, j.mode, j.nchunks, j.outPad, i);
            const int o = (int)((i >> 5) % j.outPad);
            if (scaled && o < j.Cout) v *= j.scale[o];          // fp32 product, ONE rounding to bf16
            reinterpret_cast<unsigned short*>(j.wp)[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
            continue;
        }
        if (i >= main_f32) {          // the fp32 sliver image behind the main one
            float v = unetconv::f32_sliver_value(j.w, j.Cout, j.Cin, j.T, j.mode, j.nchunks, i - main_f32);
            const int o = (out & ~15) + (int)(((i - main_f32) >> 4) & 3);
            if (scaled && o < j.Cout) v *= j.scale[o];
            reinterpret_cast<float*>(j.wp)[i] = v;
            continue;
        }
        const int rr = (int)(i & (KC - 1));
        size_t q = i >> sh;
        const int o = (int)(q % j.outPad); q /= j.outPad;
        const int chunk = (int)(q % j.nchunks);
        const int tap = (int)(q / j.nchunks);
        const bool tail = (red & 15) != 0 && chunk == j.nchunks - 1;
        const int r = chunk * 16 + (tail ? (4 * (rr & 3) + (rr >> 2)) : rr);
        float v = 0.f;
        if (j.mode == 2) {          // forward image, columns in pixel-shuffle order (ks = 1)
            if (o < j.Cout && r < j.Cin) v = j.w[((size_t)unetconv::ps_filter_of(o, j.Cout) * j.Cin + r) * j.T + tap];
        } else if (j.mode == 0) {
            if (o < j.Cout && r < j.Cin) v = j.w[((size_t)o * j.Cin + r) * j.T + tap] * (scaled ? j.scale[o] : 1.f);
        } else {
            if (o < j.Cin && r < j.Cout) v = j.w[((size_t)r * j.Cin + o) * j.T + tap];
        }
        reinterpret_cast<float*>(j.wp)[i] = v;
    }
}

}  // namespace

extern "C" size_t unet_pack_batch_table_bytes(int njobs) { return (size_t)(njobs > 0 ? njobs : 0) * sizeof(Job); }

extern "C" int unet_pack_batch_build(const unet_pack_job* jobs, int njobs, int dtype, void* table_host, unsigned* total_blocks) {
    UNET_CHECK_ARG(jobs && table_host && total_blocks && njobs > 0, "pack_batch_build: bad args");
    UNET_CHECK_ARG(dtype == UNET_F32 || dtype == UNET_BF16, "pack_batch_build: unknown dtype %d", dtype);
    Job* t = reinterpret_cast<Job*>(table_host);
    const int KC = dtype == UNET_BF16 ? 32 : 16;
    unsigned long long blocks = 0;
    for (int i = 0; i < njobs; ++i) {
        const unet_pack_job& s = jobs[i];
        UNET_CHECK_ARG(s.w && s.wp && (s.ks == 1 || s.ks == 3) && (s.mode == 0 || s.mode == 1 || s.mode == 2) && s.Cout > 0 && s.Cin > 0,
                       "pack_batch_build: bad job %d", i);
        UNET_CHECK_ARG(s.mode != 2 || (s.ks == 1 && s.Cout % 64 == 0 && s.out_scale == nullptr), "pack_batch_build: job %d: mode 2 is a 1x1 filter with Cout %% 64 == 0", i);
        const int red = s.mode == 1 ? s.Cout : s.Cin, out = s.mode == 1 ? s.Cin : s.Cout;
        Job& j = t[i];
        j.w = s.w; j.wp = s.wp; j.scale = s.out_scale; j.Cout = s.Cout; j.Cin = s.Cin; j.T = s.ks * s.ks; j.mode = s.mode;
        j.nchunks = unet::cdiv(red, KC); j.outPad = unet::roundup(out, 128);
        j.block_begin = (unsigned)blocks; j.pad_ = 0;
        const size_t total = dtype == UNET_BF16 ? unetconv::bf16_image_elems(red, j.outPad, j.T) : unetconv::f32_image_elems(red, out, j.T);
        blocks += (total + ELEMS_PER_BLOCK - 1) / ELEMS_PER_BLOCK;
        UNET_CHECK_ARG(blocks < (1ull << 31), "pack_batch_build: too many blocks");
    }
    *total_blocks = (unsigned)blocks;
    return UNET_OK;
}

extern "C" int unet_pack_batch_run(const void* table_dev, int njobs, unsigned total_blocks, int dtype, void* stream) {
    UNET_CHECK_ARG(table_dev && njobs > 0 && total_blocks > 0, "pack_batch_run: bad args");
    UNET_CHECK_ARG(dtype == UNET_F32 || dtype == UNET_BF16, "pack_batch_run: unknown dtype %d", dtype);
    hipLaunchKernelGGL(pack_batch_kernel, dim3(total_blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const Job*>(table_dev), njobs,
                       dtype == UNET_BF16 ? 1 : 0);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}
