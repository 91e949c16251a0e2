/*
 * unet_hip.h -- C ABI of libunet_hip.so: the MI355X (gfx950) hot path of the
 * U-Net segmentation train / predict step.
 *
 * The reference (LUP-LuftbildUmweltPlanung/UNet) has no FFI of its own: its hot
 * path is two Python lines -- model construction (train.py:128-144) and
 * learn.fit_one_cycle / learn.predict (train.py:247-250, predict.py:193) -- whose
 * arithmetic is executed by fastai 2.5.1 + ATen.  Each entry point below replaces
 * the ATen operator family that graph issues; the
 * comment on each names the fastai module / reference call site it stands for.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (PyTorch allocates);
 *     the library never allocates, frees or synchronises;
 *   - all tensors are fp32 NHWC; a tensor argument is (ptr, cs, co): base
 *     pointer of the [N,H,W,cs] buffer, physical channel stride cs (multiple of
 *     4) and channel offset co (multiple of 4) -- this is how torch.cat is
 *     eliminated: producers write channel slices of one buffer;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued
 *     asynchronously on it;
 *   - return value: 0 = OK, <0 = UNET_E_*; unet_last_error() gives the text.
 */
#ifndef UNET_HIP_H
#define UNET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNET_OK 0
#define UNET_E_BADARG (-1)
#define UNET_E_UNSUPPORTED (-2)
#define UNET_E_HIP (-3)

#define UNET_ABI_VERSION 8

int unet_abi_version(void);
const char* unet_last_error(void);

/* Storage type of activations / activation gradients / packed filter images.  UNET_F32 is the parity path (the reference computes in
 * fp32: train.py:141-144 has no to_fp16()).  UNET_BF16 is the bf16-storage variant of BASELINE.json configs[1]: tensors are bfloat16
 * bit patterns (unet_bf16), channel strides / offsets are multiples of 8 (16-byte vectors), every product is accumulated in fp32
 * (v_mfma_f32_16x16x32_bf16), per-channel vectors (bias, BatchNorm coefficients), losses, master weights and optimizer state stay
 * fp32.  Entry points with a _bf16 suffix take bf16 tensors where the fp32 entry point takes float tensors. */
#define UNET_F32 0
#define UNET_BF16 1
typedef uint16_t unet_bf16;

/* ---------------------------------------------------------------- tuning --
 * Kernel-selection switches of ONE launch (A/B measurements, cross-checks between independently written kernel families in the tests).
 * A descriptor's `tuning` pointer may be NULL: the defaults.  The library keeps NO mutable process state: the switches
 * travel with the descriptor, two callers -- threads, models -- with different tunings do not see each other, every entry point is
 * re-entrant.  Always start from unet_tuning_default(): a zeroed struct is NOT the default.  Every alternative computes the same result up
 * to summation order.  Environment overrides of the DEFAULTS, read once at load: UNET_CONV_SPLITK, UNET_T256_SLIVER, UNET_CONV1X1_GEMM. */
typedef struct {
    int conv_splitk;        /* 1: the planner may split the reduction of small-grid launches (default); 0: never; n > 1: split launches of
                               fewer than n full-size tiles */
    int mfma_shape;         /* fp32 conv: 16 = v_mfma_f32_16x16x4_f32 (default), 32 = v_mfma_f32_32x32x2_f32 with an LDS filter slab */
    int f32_big_tile;       /* 1: eligible fp32 3x3 launches run on conv_bf16_t256_kernel<.., float> (default); 0: conv_igemm16_kernel */
    int bf16_big_tile;      /* bf16 conv: 1 = the 256-pixel tile for 3x3 / stride-1 launches from 64 blocks up (default), 0 = off,
                               2 = the planner order of round 3's first half, 3.. = from 64 (n - 2) blocks up */
    int t256_tiles_per_wg;  /* consecutive tiles per workgroup of conv_bf16_t256_kernel; 0 = the launcher's choice (default) */
    int t256_sliver;        /* 1: a 7-tile fp32 block whose last tile holds 1..4 channels multiplies them as a 4x4x1 sliver (default) */
    int conv1x1_gemm;       /* plain 1x1 / stride-1 convs of whole chunks on conv1x1_gemm_kernel (the flat-pixel GEMM, csrc/conv1x1.hip):
                               0 = auto (default): its LDS-staged form for bf16 storage, the implicit-GEMM kernels for fp32 (measured);
                               1 = its direct form, 2 = its staged form (both storage types), -1 = never.  pixel_shuffle descriptors always use it */
    int wgrad_mfma_shape;   /* fp32 weight gradient: 32 (default) | 16 (opt-in: loses to wave imbalance) */
    int wgrad_bf16_k4;      /* 1: bf16 3x3 / 1x1 stride-1 weight gradients on wgrad_bf16_k4_kernel (default); 0: wgrad_bf16_kernel */
    int wgrad_1x1;          /* 1: the 128x128-tiled GEMM kernel (and the small-output FMA kernel) for fp32 1x1 weight gradients (default; launches of
                               <= 64 input / output channels or < 3 GFLOP stay on the 64x64-blocked general kernel); 2: the GEMM kernel for all; 0: never */
    int wgrad_narrow;       /* 1: the narrow-output (80 < Cout <= 112) flattened-tap fp32 weight-gradient kernel (default); 2: the same without the
                               4-row v_mfma_f32_4x4x1 sliver for 97..100 output channels (seven 16-row tiles: A/B, cross-check); 3: as 1, and the
                               general fp32 kernel without the pixel sub-splits of layers with <= 32 input / output channels (A/B) */
    int plan_batch;         /* 0 (default): the conv planner sizes tiles / split reductions for the batch of the descriptor.  n > 0: it plans as
                               if the batch were n images, whatever N is -- with 1, every image of a batch runs exactly the kernels, tiles and split
                               chains it would run alone: predictions become bit-identical ACROSS batch sizes (the reference predicts tile by
                               tile, predict.py:191-193), at the price of batch-1 plans on full grids (predict.predict_raster(batch_invariant=True)) */
    int wgrad_wgs;          /* workgroups a weight-gradient launch is split into (split-K over pixel tiles).  0 (default): ~512 (two per CU); bf16
                               storage, 3x3 layers up to ~175 GFLOP: ~256 (one per CU: half the partial filter images to write and read back,
                               measured faster).  n > 0: ~n workgroups (A/B) */
    int conv_smallcin;      /* 1 (default): forward 3x3 convs of <= 8 input channels (the stem's first conv) on the direct HBM-bound kernel; 0: the
                               implicit-GEMM kernels (A/B, cross-check) */
    int conv_head1x1;       /* 1 (default): 1x1 forward convs with <= 16 produced channels (the segmentation head) on the streaming kernel that keeps
                               the filter in registers; 0: the implicit-GEMM kernels (A/B, cross-check: same bits); 2: the other pixel-tile count per trip (A/B) */
} unet_tuning;
void unet_tuning_default(unet_tuning* t);

/* ------------------------------------------------------------------ conv --
 * Implicit-GEMM convolution on fp32 MFMA, NHWC, no im2col.  Default instruction v_mfma_f32_16x16x4_f32: LDS-staged input halo
 * tile, packed filter tiles global -> VGPR (MFMA operand order); v_mfma_f32_32x32x2_f32 with an LDS filter slab is kept
 * behind unet_tuning.mfma_shape = 32.
 * Replaces: every nn.Conv2d of fastai ConvLayer (layers.py) in XResNet
 * (vision/models/xresnet.py), DynamicUnet.middle_conv / UnetBlock.conv1,conv2 /
 * PixelShuffle_ICNR 1x1 / final ResBlock / head (vision/models/unet.py), as
 * built by train.py:128-144, and their autograd input-gradients.
 */

/* flags for unet_conv2d */
#define UNET_CONV_RELU 1          /* y = max(y, 0) after bias/residual */
#define UNET_CONV_MASK 4          /* y = (maskref > 0) ? y : 0 (ReLU backward fused in dgrad) */

/* kind */
#define UNET_CONV_FWD 0           /* y = conv(x, W) stride S pad (ks-1)/2 */
#define UNET_CONV_DGRAD 1         /* y = dL/dx given x := dL/d(conv out); wp packed by unet_pack_weights(mode DGRAD) */

typedef struct {
    const float* x; int x_cs, x_co;      /* input: FWD: activations [N,IH,IW]; DGRAD: output-gradient [N,IH,IW] */
    const float* wp;                     /* packed weights (unet_pack_weights) */
    const float* bias;                   /* [Cout] or NULL */
    const float* res; int res_cs, res_co;      /* optional residual added before activation, same geometry as y */
    const float* mask; int mask_cs, mask_co;   /* UNET_CONV_MASK reference, same geometry as y */
    float* y; int y_cs, y_co;            /* output [N,OH,OW] */
    int N, IH, IW, Cin;                  /* Cin = reduction channels of this launch */
    int OH, OW, Cout;                    /* Cout = produced channels of this launch */
    int ks, stride;                      /* kernel size 1|3, stride 1|2 of the FORWARD convolution */
    int kind, flags;
    float* colsum;                       /* optional [rows][Cout] per-wave partial column sums of y (bias-grad / BN stats), or NULL */
    float* colsumsq;                     /* optional same shape, sums of y*y */
    int cout_begin, cout_count;          /* produce only channels [cout_begin, cout_begin + cout_count) of the Cout-wide problem
                                            (cout_begin % 16 == 0); 0, 0 = all.  Lets a caller issue e.g. 192 channels as 128 + 64
                                            with a channel-block width that fits each part. */
    long long wp_img_stride;             /* elements between the packed filter images of consecutive batch images; 0 = one image for all.
                                            Per-image "filters" are activations: the attention products of SelfAttention run as ONE
                                            launch over the batch (unet_pack_weights_strided once per image into one buffer). */
    int dtype;                           /* UNET_F32 (default 0) | UNET_BF16: storage type of x, res, mask, y and wp (cast the pointers);
                                            bias stays fp32.  bf16: wp from unet_pack_weights_bf16, colsum / colsumsq unsupported */
    int y_f32;                           /* dtype UNET_BF16 only: y is an fp32 buffer (the logits head feeds the fp32 loss kernels) */
    float* splitk_ws;                    /* optional scratch for split-K launches (small grid, long reduction): fp32, 16-byte aligned, */
    size_t splitk_ws_floats;             /*   >= unet_conv2d_splitk_workspace(desc) floats; NULL / too small = never split this launch */
    const unet_tuning* tuning;           /* NULL = defaults */
    int pixel_shuffle;                   /* 1: a 1x1 FWD conv of Cout = 4 nf channels that stores PixelShuffle(2)(act(conv)) directly: y is a Cout / 4 wide
                                            slice of a [N, 2 OH, 2 OW] tensor, produced channel q = ij * nf + c of pixel (h, w) lands at channel c of pixel
                                            (2 h + ij / 2, 2 w + ij % 2); wp must be the mode-2 image (columns in that order), bias stays in the filter's own
                                            order.  The PixelShuffle_ICNR without blur in front of the dense merge (reference train.py:141: blur_final applies
                                            to the LAST UnetBlock, the final upsample has none) then needs no un-shuffled copy of the conv output at all.
                                            Only descriptors unet_conv2d_variant answers 8 for are taken (UNET_E_UNSUPPORTED otherwise) */
    /* pixel_shuffle only, optional (ps_tail = NULL: none): an NHWC tensor [N, 2 OH, 2 OW] of y's storage type whose channels [ps_tail_co,
       ps_tail_co + ps_tail_c) are copied to channels [ps_tail_at, ..) of y's BUFFER (counted from the buffer, not from y_co) by the lane that
       stores the last four shuffled channels of the same pixel: the network-input half of DynamicUnet's final concat (reference train.py:141,
       last_cross: cat([up, x])) without a second pass of 8-byte writes at 208-byte stride over the buffer (158 us alone at 16 x 512^2; measured
       with bf16 storage: 490 + 158 us -> 598 us in one launch; with fp32 storage the appended stores cost more than the pass: 1021 + 155 -> 1229,
       so the model uses it for bf16 only).  Whole quads are copied (ps_tail_c rounded up to 4: pad lanes of the source are zeros);
       ps_tail_cs, ps_tail_co and ps_tail_at are multiples of 4, ps_tail_at + roundup(ps_tail_c, 4) <= y_cs */
    const void* ps_tail;
    int ps_tail_cs, ps_tail_co;
    int ps_tail_c, ps_tail_at;
} unet_conv_desc;

/* number of partial rows the colsum buffers must hold for this desc */
int unet_conv2d_colsum_rows(const unet_conv_desc* d);
int unet_conv2d(const unet_conv_desc* d, void* stream);
/* floats of split-K scratch the planner would use for this desc (0: it does not split).  A launch whose output grid cannot fill the chip
 * although its reduction is long (deep low-resolution stages, small batches) is cut into `splits` contiguous ranges of reduction chunks;
 * partial sums are added in split order by a second kernel that applies bias / residual / ReLU / mask: deterministic, and shorter
 * accumulation chains.  unet_tuning.conv_splitk = 0 switches it off for a launch (A/B). */
size_t unet_conv2d_splitk_workspace(const unet_conv_desc* d);
/* which kernel instantiation serves this desc: TW*10000 + BN*10 + (stride-2 halo variant) + 1000000 * splits -- for profilers / bench.py;
 * ...7 / ...6: the 256-pixel tile of the bf16 path (3x3 stride 1: conv_bf16_t256_kernel; 7 = 128-wide blocks, 32-pixel patches, >= 512 blocks,
 * 6 = its narrow-block / 16-pixel-patch / small-grid launches); 9: conv1x1_smallk_kernel (1x1, reduction of <= 8 channels);
 * 8: conv1x1_gemm_kernel (1x1 stride 1, whole reduction chunks, >= 256 blocks: the flat-pixel GEMM without LDS staging) */
int unet_conv2d_variant(const unet_conv_desc* d);
/* weight packing.  w is the torch-layout master parameter [Cout,Cin,ks,ks].
 * mode 0 (FWD):   wp[tap][chunk][coutPad][16]  reduction over Cin
 * mode 1 (DGRAD): wp[tap][chunk][cinPad][16]   reduction over Cout
 * mode 2 (FWD, pixel-shuffle order; ks = 1, Cout % 64 == 0): mode 0 with image column q = ij * (Cout / 4) + c holding filter 4 c + ij
 *                 (what unet_conv_desc.pixel_shuffle stores from)
 * pads are zero filled; chunk = 16 reduction channels; *Pad = roundup(.,128). */
size_t unet_pack_weights_size(int Cout, int Cin, int ks, int mode); /* floats */
int unet_pack_weights(const float* w, float* wp, int Cout, int Cin, int ks, int mode, void* stream);
/* 1x1 "weights" that are themselves activations (self-attention operands): element (out o, reduction r) = w[o*so + r*sr];
 * produces the same packed image as mode 0 with ks = 1 (size unet_pack_weights_size(O, R, 1, 0), the sliver block of a 16 n + 1..4 wide
 * output included). */
int unet_pack_weights_strided(const float* w, long long so, long long sr, float* wp, int O, int R, void* stream);
/* bf16 images from the fp32 master parameter: wp[tap][chunk][outPad][32] with chunk = 32 reduction channels (64 bytes, as in fp32);
 * a 3x3 filter whose reduction leaves a tail of 1..8 channels gets three more slabs fold[j][outPad][32] (k-slot (kq, c) = tail
 * channel c of tap 4 j + kq) that single-tap-set launches read instead of the nine tail chunks.  The size function includes them. */
size_t unet_pack_weights_size_bf16(int Cout, int Cin, int ks, int mode); /* elements */
int unet_pack_weights_bf16(const float* w, unet_bf16* wp, int Cout, int Cin, int ks, int mode, void* stream);
/* All filter images of a model in one launch (the parameters change every step, so every image is rebuilt every step).
 * unet_pack_batch_build fills a HOST table (unet_pack_batch_table_bytes(njobs) bytes) from the job list -- same layouts as
 * unet_pack_weights (dtype UNET_F32) / unet_pack_weights_bf16 (UNET_BF16) -- and returns the grid size; the caller uploads the table
 * once (addresses are static) and calls unet_pack_batch_run on it whenever the parameters have changed. */
typedef struct {
    const float* w;                      /* master parameter [Cout,Cin,ks,ks] (device) */
    void* wp;                            /* packed image (device), unet_pack_weights_size[_bf16] elements */
    int Cout, Cin, ks, mode;             /* mode 0 = forward image, 1 = input-gradient image */
    const float* out_scale;              /* optional [Cout] (device), mode 0: image of w * out_scale[cout] -- eval-mode BatchNorm folded into
                                            the filter (y = conv(x, w * scale) + shift, the conv epilogue adds shift as its bias) */
} unet_pack_job;
size_t unet_pack_batch_table_bytes(int njobs);
int unet_pack_batch_build(const unet_pack_job* jobs, int njobs, int dtype, void* table_host, unsigned* total_blocks);
int unet_pack_batch_run(const void* table_dev, int njobs, unsigned total_blocks, int dtype, void* stream);
/* weight gradient dW[Cout,Cin,ks,ks] (torch layout) = sum_pixels dy (x) x.
 * Replaces the autograd weight-gradient of the same nn.Conv2d modules.
 * workspace holds split-K partials; query its size (floats) first. */
typedef struct {
    const float* x; int x_cs, x_co;      /* forward input  [N,IH,IW,Cin] */
    const float* dy; int dy_cs, dy_co;   /* output grad    [N,OH,OW,Cout] */
    float* dw;                           /* [Cout,Cin,ks,ks] */
    float* dbias;                        /* [Cout] or NULL: also produce sum_pixels dy */
    int N, IH, IW, Cin, OH, OW, Cout, ks, stride;
    float* workspace; size_t workspace_floats;
    int accumulate;                      /* 0: dw = result, 1: dw += result */
    int dtype;                           /* UNET_F32 (default 0) | UNET_BF16: storage type of x and dy; dw, dbias and the workspace are fp32 */
    const unet_tuning* tuning;           /* NULL = defaults */
} unet_wgrad_desc;
size_t unet_conv2d_wgrad_workspace(const unet_wgrad_desc* d);
int unet_conv2d_wgrad(const unet_wgrad_desc* d, void* stream);

/* ----------------------------------------------------------- batch norm --
 * Replaces nn.BatchNorm2d(eps 1e-5, momentum 0.1) of fastai BatchNorm
 * (layers.py) in train mode (batch statistics, wavefront reductions) and eval
 * mode (running statistics), fused with residual add + ReLU of ResBlock.forward.
 */
/* per-channel partial sums of x and x*x over P pixels -> two planes partial[0:rows][C] (sums) and
 * partial[rows:2*rows][C] (sums of squares); rows = unet_bn_stats_rows(P) */
int unet_bn_stats_rows(long long P);
int unet_bn_stats(const float* x, int x_cs, int x_co, long long P, int C, float* partial, void* stream);
/* train-mode finalize: from partial sums psum[rows][C], psumsq[rows][C] (unet_bn_stats planes or the
 * colsum/colsumsq of a conv epilogue) compute mean / invstd, scale = gamma*invstd, shift = beta - mean*scale,
 * and update running stats (unbiased var, momentum); batches_tracked (BatchNorm2d.num_batches_tracked, int64, or NULL) += 1. */
int unet_bn_finalize(const float* psum, const float* psumsq, int rows, long long count, int C,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     float momentum, float eps,
                     float* scale, float* shift, float* save_mean, float* save_invstd, long long* batches_tracked, void* stream);
/* eval-mode: scale/shift from running stats */
int unet_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                        float eps, int C, float* scale, float* shift, void* stream);
/* y = act(x*scale+shift [+ x2*scale2+shift2 | + x2]) ; scale2 NULL with x2 != NULL means plain add */
int unet_affine_act(const float* x, int x_cs, int x_co, const float* scale, const float* shift,
                    const float* x2, int x2_cs, int x2_co, const float* scale2, const float* shift2,
                    float* y, int y_cs, int y_co, long long P, int C, int relu, void* stream);
/* BN backward, train mode.  g = dout * (out > 0 if out != NULL).
 * reduce: planes partial[0:rows][C] = sum g, partial[rows:2*rows][C] = sum g*xhat; rows = unet_bn_stats_rows(P) */
int unet_bn_bwd_reduce(const float* dout, int d_cs, int d_co, const float* out, int o_cs, int o_co,
                       const float* x, int x_cs, int x_co, const float* mean, const float* invstd,
                       long long P, int C, float* partial, void* stream);
/* finalize: dgamma, dbeta and coefficient vectors c1 = sum g / count, c2 = sum g xhat / count */
int unet_bn_bwd_finalize(const float* partial, int rows, long long count, int C,
                         float* dgamma, float* dbeta, float* c1, float* c2, void* stream);
/* apply: dx = gamma*invstd*(g - c1 - xhat*c2) ; optionally also writes g to gout (for the residual branch) */
int unet_bn_bwd_apply(const float* dout, int d_cs, int d_co, const float* out, int o_cs, int o_co,
                      const float* x, int x_cs, int x_co, const float* mean, const float* invstd,
                      const float* gamma, const float* c1, const float* c2,
                      float* dx, int dx_cs, int dx_co, float* gout, int g_cs, int g_co, int g_accumulate,
                      long long P, int C, void* stream);

/* -------------------------------------------------------------- pooling --
 * MaxPool2d(3,2,1) after the stem and AvgPool2d(2, ceil_mode=True) on strided
 * identity paths (fastai xresnet.py / layers.py ResBlock). */
int unet_maxpool3x3s2(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, uint8_t* idx,
                      int N, int IH, int IW, int C, int OH, int OW, void* stream);
int unet_maxpool3x3s2_bwd(const float* dy, int dy_cs, int dy_co, const uint8_t* idx, float* dx, int dx_cs, int dx_co,
                          int N, int IH, int IW, int C, int OH, int OW, int accumulate, void* stream);
int unet_avgpool2_ceil(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co,
                       int N, int IH, int IW, int C, int OH, int OW, void* stream);
int unet_avgpool2_ceil_bwd(const float* dy, int dy_cs, int dy_co, float* dx, int dx_cs, int dx_co,
                           int N, int IH, int IW, int C, int OH, int OW, int accumulate, void* stream);

/* -------------------------------------------------- decoder data movement --
 * PixelShuffle_ICNR (layers.py) tail and UnetBlock.forward (vision/models/unet.py):
 *   up = [blur](PixelShuffle(2)(yc)),  yc = relu(conv1x1(x)) of shape [N,h,w,4*Cu]
 *   PixelShuffle: P[2h+i, 2w+j, c] = yc[h, w, 4c+2i+j]
 *   blur = ReplicationPad2d((1,0,1,0)) + AvgPool2d(2, stride=1)
 * writes X[N,2h,2w][X_co : X_co+Cu]; the skip half of torch.cat is written by
 * unet_affine_act (relu(bn(skip))) into the neighbouring channel slice. */
int unet_shuffle_blur(const float* yc, int yc_cs, int yc_co, float* X, int X_cs, int X_co,
                      int N, int h, int w, int Cu, int do_blur, void* stream);
/* adjoint incl. the ReLU of the 1x1 conv: dyc = (yc > 0) * shuffle^T(blur^T(dX)) */
int unet_shuffle_blur_bwd(const float* dX, int dX_cs, int dX_co, const float* yc, int yc_cs, int yc_co,
                          float* dyc, int dyc_cs, int dyc_co, int N, int h, int w, int Cu, int do_blur, void* stream);
/* the same adjoint without blur when the forward stored the shuffled activation only (unet_conv_desc.pixel_shuffle): the ReLU mask is read
 * from X, the [N,2h,2w] forward output, at the address of dX: dyc[h,w,4c+2i+j] = X[2h+i,2w+j,c] > 0 ? dX[2h+i,2w+j,c] : 0 */
int unet_shuffle_bwd_xmask(const float* dX, int dX_cs, int dX_co, const float* X, int X_cs, int X_co,
                           float* dyc, int dyc_cs, int dyc_co, int N, int h, int w, int Cu, void* stream);
/* F.interpolate(mode='nearest') and its adjoint (UnetBlock / ResizeToOrig when sizes differ) */
int unet_resize_nearest(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co,
                        int N, int IH, int IW, int OH, int OW, int C, void* stream);
int unet_resize_nearest_bwd(const float* dy, int dy_cs, int dy_co, float* dx, int dx_cs, int dx_co,
                            int N, int IH, int IW, int OH, int OW, int C, void* stream);
/* layout conversion at the seam (torch side is NCHW): */
int unet_nchw_to_nhwc(const float* x, float* y, int y_cs, int y_co, int N, int C, int H, int W, void* stream);
int unet_nhwc_to_nchw(const float* x, int x_cs, int x_co, float* y, int N, int C, int H, int W, void* stream);
/* elementwise helpers on NHWC slices */
int unet_copy_slice(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, long long P, int C,
                    int accumulate, void* stream);
int unet_relu_mask(const float* g, int g_cs, int g_co, const float* ref, int r_cs, int r_co,
                   float* y, int y_cs, int y_co, long long P, int C, void* stream);
int unet_colsum(const float* x, int x_cs, int x_co, long long P, int C, float* out, float* workspace, void* stream);
size_t unet_colsum_workspace(long long P, int C);
/* out[0] = sum over pixels and channels of x * y (fp64 second stage); workspace = unet_colsum_workspace(P, C) floats.
 * SelfAttention.gamma gradient: sum(O * dout) (fastai layers.py SelfAttention: o = gamma * (h beta) + x). */
int unet_dot(const float* x, int x_cs, int x_co, const float* y, int y_cs, int y_co, long long P, int C, float* out,
             float* workspace, void* stream);

/* ----------------------------------------------------------------- loss --
 * CrossEntropyLossFlat(axis=1, weight=w) (fastai losses.py; train.py:195,211):
 * loss = sum w[y] * -log softmax(z)[y] / sum w[y]; activation softmax(dim=1),
 * decodes argmax(dim=1) (predict.py:193,232).
 * logits NHWC [P,C] (cs/co), targets int64 [P]. */
size_t unet_ce_workspace(long long P);   /* floats */
int unet_ce_fwd(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C,
                float* loss /*[1]*/, float* denom /*[1]*/, float* workspace, void* stream);
/* the same sums, not divided: numden[0] = sum w[y] * nll, numden[1] = sum w[y].  Tile-DDP all-reduces these two floats between the loss
 * forward and backward kernels (one cross-entropy over the global batch); a rank whose tiles carry only zero-weight classes contributes
 * 0 and 0 instead of a 0/0 quotient. */
int unet_ce_fwd_parts(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C,
                      float* numden /*[2]*/, float* workspace, void* stream);
/* dz = gscale * w[y] * (softmax(z) - onehot(y)) / denom ; gscale multiplies (loss scaling / DDP averaging) */
int unet_ce_bwd(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C,
                const float* denom, float gscale, float* dz, int dz_cs, int dz_co, void* stream);
/* FocalLossFlat(gamma, axis=1): the alternative classification loss the reference's configuration names (params_and_main.py:87-89).  fastai 2.5.1
 * losses.FocalLoss: ce = w[y] * nll per pixel (F.cross_entropy(weight, reduction="none"); train.py:211 assigns the class weights to every
 * loss), loss = mean over ALL P pixels of (1 - exp(-ce))^gamma * ce.  dz = gscale * d loss / d z.  workspace = unet_ce_workspace(P) floats.
 * Where ce == 0 exactly and gamma < 1, torch's autograd returns NaN (0 * inf); these kernels return the limit, 0. */
int unet_focal_fwd(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, float gamma,
                   float* loss /*[1]*/, float* workspace, void* stream);
int unet_focal_bwd(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, float gamma,
                   float gscale, float* dz, int dz_cs, int dz_co, void* stream);
/* Regression mode (enable_regression, reference train.py:137-138,189-193; utils.py:145-147): n_out = 1, the loss is the mean over all
 * pixels of kind 0 = (z - t)^2 (MSELossFlat), 1 = |z - t| (L1LossFlat), 2 = SmoothL1(beta) (Smoothl1: beta 0.5).  z = channel z_co of
 * the NHWC output [P,z_cs], float targets [P]; workspace = unet_ce_workspace(P) floats.
 * dz = gscale * d loss / d z. */
int unet_regloss_fwd(const float* z, int z_cs, int z_co, const float* target, long long P, int kind, float beta,
                     float* loss /*[1]*/, float* workspace, void* stream);
int unet_regloss_bwd(const float* z, int z_cs, int z_co, const float* target, long long P, int kind, float beta, float gscale,
                     float* dz, int dz_cs, int dz_co, void* stream);
/* probs NCHW [N,C,H,W] (what Learner.predict returns, predict.py:196-203) and argmax uint8/int64 mask */
int unet_softmax_argmax(const float* z, int z_cs, int z_co, int N, int H, int W, int C,
                        float* probs_nchw /*or NULL*/, int64_t* argmax /*or NULL*/, void* stream);

/* ------------------------------------------------------- self-attention --
 * fastai SelfAttention (layers.py; DynamicUnet(self_attention=True), params_and_main.py:81-83): beta = softmax(f^T g, dim=1),
 * o = gamma * h beta + x.  The matrix products run on unet_conv2d / unet_conv2d_wgrad with per-image operands packed by
 * unet_pack_weights_strided; these two kernels are the softmax over one attention row (all key positions) and its adjoint. */
int unet_row_softmax(const float* x, int x_cs, int x_co, float* y, int y_cs, int y_co, long long P, int C, void* stream);
int unet_row_softmax_bwd(const float* y, int y_cs, int y_co, const float* dy, int dy_cs, int dy_co,
                         float* dx, int dx_cs, int dx_co, long long P, int C, void* stream);

/* ------------------------------------------------------------ optimiser --
 * fastai Adam (optimizer.py: weight_decay, average_grad(dampening), average_sqr_grad,
 * step_stat, adam_step) with decoupled wd, per-group lr (train.py:78-80, 247-250).
 * code[i] = group_id (bits 0-1) | do_wd << 2.  grad_scale multiplies g first. */
int unet_adam_step(float* p, const float* g, float* m, float* v, const uint8_t* code, long long n,
                   const float* lr /*[4] host*/, float mom, float sqr_mom, float eps, float wd, int step,
                   float grad_scale, void* stream);

/* hipGraph-friendly form: the hyper-parameter block (unet_adam_hyper_floats() floats, filled on the host by
 * unet_adam_fill_hyper and copied to the device by the caller before each replay) is read from device memory, so a captured
 * step can be replayed with a new lr / momentum / step count. */
int unet_adam_hyper_floats(void);
int unet_adam_fill_hyper(float* hyper_host, const float* lr /*[4]*/, float mom, float sqr_mom, float eps, float wd, int step,
                         float grad_scale);
int unet_adam_step_dev(float* p, const float* g, float* m, float* v, const uint8_t* code, long long n,
                       const float* hyper_dev, void* stream);

/* ------------------------------------------------- overlap-merge (predict) --
 * predict.py:284-326: sum softmax probabilities of overlapping tiles into a
 * mosaic + hit counter, divide, argmax. */
int unet_mosaic_accumulate(const float* probs_nchw, int C, int th, int tw, float* mosaic /*[C,MH,MW]*/,
                           int32_t* count /*[MH,MW]*/, int MH, int MW, int y0, int x0, void* stream);
int unet_mosaic_finalize(float* mosaic, const int32_t* count, int C, int MH, int MW, uint8_t* argmax, void* stream);

/* --------------------------------------- sliding-window predict over a raster --
 * BASELINE.json configs[4]: the raster stays in HBM as the integers it was read as; windows are cut, scaled and merged on the device.
 * Replaces, with identical results, the host round trip create_tiles_unet.split_raster (create_tiles_unet.py:252-434) -> tile files
 * -> predict.save_predictions(merge=True) (predict.py:191-222, 257-334).
 * A source is band-sequential: sample (band c, row y, column x) of source s is src[s * src_stride + c * band_stride + y * row_stride + x].
 * A window table is int32 [n][4] on the device: {y0, x0, source index, unused}. */
#define UNET_RASTER_U8 0
#define UNET_RASTER_U16 1
#define UNET_RASTER_I16 2
#define UNET_RASTER_I32 3
#define UNET_RASTER_F32 4          /* cast through int32 like every tile the reference opens (data.py:24) */
/* create_tiles_unet.py:344-352: every band of a pixel becomes 0 where ANY band equals the raster's nodata value (in place) */
int unet_raster_nodata_zero(void* raster, int rtype, int bands, long long pixels, double nodata, void* stream);
/* create_tiles_unet.py:379: counts[j] = number of non-zero samples (all bands) of window j; the host drops windows with
 * counts[j] < bands * th * tw * (1 - max_empty) */
int unet_window_nonzero(const void* raster, int rtype, int bands, long long band_stride, int row_stride, const int32_t* windows, int n,
                        int th, int tw, unsigned long long* counts, void* stream);
/* x[j, y, x, x_co + c] = float(int32(sample)) / 255 (and / 255 once more when div255_twice: the reference's int16 rasters, utils.py:248-249
 * + IntToFloatTensor); x is an NHWC activation buffer [n, th, tw, x_cs] of storage type dtype (UNET_F32 | UNET_BF16).  Equals
 * open_npy + the transform + unet_nchw_to_nhwc of the same window bit for bit. */
int unet_window_gather(const void* src, int rtype, int bands, long long src_stride, long long band_stride, int row_stride,
                       const int32_t* windows, int n, int th, int tw, int div255_twice, void* x, int x_cs, int x_co, int dtype, void* stream);
/* predict.py:193-203 + 284-292 for a batch: window j's softmax probabilities (mode 0; the arithmetic of unet_softmax_argmax) or raw
 * values (mode 1, regression) of the fp32 NHWC logits z [n, th, tw, z_cs] are added to mosaic [C, MH, MW] at (y0 - origin_y,
 * x0 - origin_x) and the hit counter is incremented; only mosaic rows [row_lo, row_hi) are touched.  Contributions to one pixel are
 * added in window order (stream order across calls), so the result equals n successive unet_mosaic_accumulate calls bit for bit. */
int unet_mosaic_accumulate_windows(const float* z, int z_cs, int z_co, int C, int th, int tw, const int32_t* windows, int n, int origin_y,
                                   int origin_x, int mode, float* mosaic, int32_t* count, int MH, int MW, int row_lo, int row_hi,
                                   void* stream);
/* predict.py:306-334 on rows [row0, row0 + nrows): mosaic /= count where count > 0 (in place); argmax (uint8 [nrows, MW], may be NULL);
 * fill_host != NULL: pixels without a hit become *fill_host (regression nodata -9999, predict.py:312-315) */
int unet_mosaic_finalize_rows(float* mosaic, const int32_t* count, int C, int MH, int MW, int row0, int nrows, uint8_t* argmax,
                              const float* fill_host, void* stream);

/* ------------------------------------------------------- training feed --
 * What learn.fit_one_cycle's loader does per batch on the host in the reference (train.py:345 -> data.py:18-28 open_npy: tile -> int32 ->
 * float; utils.py:239-295 SegmentationAlbumentationsTransform: / 255 for int8 data, / 255 twice for int16 data, flips on the first
 * ceil(B * n_transform_imgs) - B images; MaskBlock: int64 masks), done on the device on the INTEGERS of the tile files: the batch crosses
 * PCIe as uint8 / uint16 samples (1 / 2 bytes instead of 4 + 8 per pixel).  src = n staged tiles [n, bands, H, W] (masks: [n, H, W]) of
 * sample type rtype (UNET_RASTER_*), n <= 64 per call; image j is mirrored along x when bit j of hflip is set and along y for vflip.
 * unet_tiles_stage: dst_nchw [n, bands, H, W] fp32 = float(int32(sample)) / 255 [/ 255], bit-equal to the host arithmetic.
 * unet_mask_stage: dst [n, H, W] int64 (dst_f32 = 0, classification) or float (dst_f32 = 1, regression targets, data.py:98-99). */
int unet_tiles_stage(const void* src, int rtype, int n, int bands, int H, int W, int div255_twice, unsigned long long hflip,
                     unsigned long long vflip, float* dst_nchw, void* stream);
int unet_mask_stage(const void* src, int rtype, int n, int H, int W, unsigned long long hflip, unsigned long long vflip, void* dst, int dst_f32,
                    void* stream);
/* DiceMulti counters of a validation batch (fastai metrics.py DiceMulti; reference train.py:196), ACCUMULATED into counts [3][C] (uint64,
 * zeroed by the caller at the start of a validation pass): [0][c] += #(pred == c and targ == c), [1][c] += #(pred == c),
 * [2][c] += #(clamp(targ, 0, C - 1) == c).  C <= 64. */
int unet_dice_counts(const int64_t* pred, const int64_t* targ, long long P, int C, unsigned long long* counts, void* stream);

/* ---------------------------------------------------- bf16-storage twins --
 * The HBM-bound kernels of the step with bf16 activation / gradient tensors (per-channel vectors, statistics, indices, losses stay
 * as in the fp32 entry point of the same name; arithmetic is fp32 per element, one rounding to bf16 at the store).  Used by
 * HipDynamicUnet(act_dtype="bf16"): BASELINE.json configs[1] "bf16" variant.  The logits head writes fp32 (unet_conv_desc.y_f32), so
 * unet_ce_fwd / unet_softmax_argmax are shared; unet_ce_bwd_bf16 writes the bf16 logit gradient. */
int unet_bn_stats_bf16(const unet_bf16* x, int x_cs, int x_co, long long P, int C, float* partial, void* stream);
int unet_affine_act_bf16(const unet_bf16* x, int x_cs, int x_co, const float* scale, const float* shift, const unet_bf16* x2, int x2_cs, int x2_co, const float* scale2, const float* shift2, unet_bf16* y, int y_cs, int y_co, long long P, int C, int relu, void* stream);
int unet_bn_bwd_reduce_bf16(const unet_bf16* dout, int d_cs, int d_co, const unet_bf16* out, int o_cs, int o_co, const unet_bf16* x, int x_cs, int x_co, const float* mean, const float* invstd, long long P, int C, float* partial, void* stream);
int unet_bn_bwd_apply_bf16(const unet_bf16* dout, int d_cs, int d_co, const unet_bf16* out, int o_cs, int o_co, const unet_bf16* x, int x_cs, int x_co, const float* mean, const float* invstd, const float* gamma, const float* c1, const float* c2, unet_bf16* dx, int dx_cs, int dx_co, unet_bf16* gout, int g_cs, int g_co, int g_accumulate, long long P, int C, void* stream);
int unet_maxpool3x3s2_bf16(const unet_bf16* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, uint8_t* idx, int N, int IH, int IW, int C, int OH, int OW, void* stream);
int unet_maxpool3x3s2_bwd_bf16(const unet_bf16* dy, int dy_cs, int dy_co, const uint8_t* idx, unet_bf16* dx, int dx_cs, int dx_co, int N, int IH, int IW, int C, int OH, int OW, int accumulate, void* stream);
int unet_avgpool2_ceil_bf16(const unet_bf16* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, int N, int IH, int IW, int C, int OH, int OW, void* stream);
int unet_avgpool2_ceil_bwd_bf16(const unet_bf16* dy, int dy_cs, int dy_co, unet_bf16* dx, int dx_cs, int dx_co, int N, int IH, int IW, int C, int OH, int OW, int accumulate, void* stream);
int unet_shuffle_blur_bf16(const unet_bf16* yc, int yc_cs, int yc_co, unet_bf16* X, int X_cs, int X_co, int N, int h, int w, int Cu, int do_blur, void* stream);
int unet_shuffle_bwd_xmask_bf16(const unet_bf16* dX, int dX_cs, int dX_co, const unet_bf16* X, int X_cs, int X_co, unet_bf16* dyc, int dyc_cs, int dyc_co, int N, int h, int w, int Cu, void* stream);
int unet_shuffle_blur_bwd_bf16(const unet_bf16* dX, int dX_cs, int dX_co, const unet_bf16* yc, int yc_cs, int yc_co, unet_bf16* dyc, int dyc_cs, int dyc_co, int N, int h, int w, int Cu, int do_blur, void* stream);
int unet_resize_nearest_bf16(const unet_bf16* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, int N, int IH, int IW, int OH, int OW, int C, void* stream);
int unet_resize_nearest_bwd_bf16(const unet_bf16* dy, int dy_cs, int dy_co, unet_bf16* dx, int dx_cs, int dx_co, int N, int IH, int IW, int OH, int OW, int C, void* stream);
int unet_nchw_to_nhwc_bf16(const float* x, unet_bf16* y, int y_cs, int y_co, int N, int C, int H, int W, void* stream);
int unet_copy_slice_bf16(const unet_bf16* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, long long P, int C, int accumulate, void* stream);
int unet_ce_bwd_bf16(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, const float* denom, float gscale, unet_bf16* dz, int dz_cs, int dz_co, void* stream);
int unet_focal_bwd_bf16(const float* z, int z_cs, int z_co, const int64_t* target, const float* weight, long long P, int C, float gamma, float gscale, unet_bf16* dz, int dz_cs, int dz_co, void* stream);
/* bf16 SelfAttention (round 3): the attention logits and the gradient of the attention weights stay fp32 (the products that make them write
 * fp32: unet_conv_desc.y_f32), the weights themselves and every other tensor are bf16 */
int unet_pack_weights_strided_bf16(const unet_bf16* w, long long so, long long sr, unet_bf16* wp, int O, int R, void* stream);
int unet_row_softmax_bf16(const float* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, long long P, int C, void* stream);
int unet_row_softmax_bwd_bf16(const unet_bf16* y, int y_cs, int y_co, const float* dy, int dy_cs, int dy_co, unet_bf16* dx, int dx_cs, int dx_co, long long P, int C, void* stream);
/* Fused SelfAttention for bf16 storage (round 5; csrc/attention.hip): beta = softmax(f^T g, dim=1), o = h beta of fastai layers.py
 * SelfAttention (params_and_main.py:81-83, train.py:141-144) without the N x N matrix in HBM.  qkv: [B][N][cq] rows with F (query conv) at
 * channel 0, G (key conv) at dp, H (value conv) at 2 dp; dp <= 64, C <= 512, both multiples of 8 (unet_sa_fused_supported; other shapes run
 * the blockwise products above).  Rows are positions h * W + w of the NHWC tensors.
 *   unet_sa_pack_bf16   image of a channel slice blocked by 64 positions with positions innermost (unet_sa_pack_elems(N, cc) elements per
 *                       image): the operand of the products that sum over positions (H for O = P H; dO, G, F in the backward pass)
 *   unet_sa_fwd_bf16    O[j] = sum_i softmax_i(G_j . F_i) H_i (bf16) and lse[b][j] = log sum_i exp(G_j . F_i) (fp32), vpack = pack of H.
 *                       lse and D are [B][Np] with Np = 64 ceil(N / 64): the rows past N are written as 1e30 (their recomputed weight is 0)
 *   unet_sa_rowdot_bf16 D[b][j] = sum_c a[b][j][c] o[b][j][c] (fp32): the softmax-adjoint term dO_j . O_j (rows past N untouched: keep them finite)
 *   unet_sa_bwd_bf16    dqkv (all three slices: dF, dG, dH) from dO, lse, D; dopack / gpack / fpack = packs of dO, G, F.  Weights are
 *                       recomputed from lse; no atomics (run-to-run identical bits) */
int unet_sa_fused_supported(int dp, int C);
size_t unet_sa_pack_elems(int N, int cc);
int unet_sa_pack_bf16(const unet_bf16* x, int x_cs, int x_co, int cc, int B, int N, unet_bf16* out, void* stream);
int unet_sa_fwd_bf16(const unet_bf16* qkv, int cq, int dp, int C, int B, int N, const unet_bf16* vpack, unet_bf16* O, int o_cs, int o_co, float* lse, void* stream);
int unet_sa_rowdot_bf16(const unet_bf16* a, int a_cs, int a_co, const unet_bf16* o, int o_cs, int o_co, int B, int N, int C, float* D, void* stream);
int unet_sa_bwd_bf16(const unet_bf16* qkv, int cq, int dp, int C, int B, int N, const unet_bf16* dO, int do_cs, int do_co, const unet_bf16* dopack,
                     const unet_bf16* gpack, const unet_bf16* fpack, const float* lse, const float* D, unet_bf16* dqkv, void* stream);
int unet_relu_mask_bf16(const unet_bf16* g, int g_cs, int g_co, const unet_bf16* ref, int r_cs, int r_co, unet_bf16* y, int y_cs, int y_co, long long P, int C, void* stream);
int unet_dot_bf16(const unet_bf16* x, int x_cs, int x_co, const unet_bf16* y, int y_cs, int y_co, long long P, int C, float* out, float* workspace, void* stream);
int unet_cast_slice_bf16(const float* x, int x_cs, int x_co, unet_bf16* y, int y_cs, int y_co, long long P, int C, void* stream);

/* ------------------------------------------------------ GeoTIFF codecs --
 * Host-side (no device code) strip / tile decoders of unet_amd/tiffio.py.  Replaces what GDAL / rasterio do when the reference opens a
 * compressed raster (create_tiles_unet.py:252-434: gdal.Open / ReadAsArray; data.py:18-28: rasterio.open().read()).  Deflate is zlib.
 * Return the number of bytes written to dst (capacity cap), -1 on a malformed stream or overflow. */
long long unet_tiff_lzw_decode(const unsigned char* src, long long n, unsigned char* dst, long long cap);      /* TIFF 6.0 LZW (compression 5) */
long long unet_tiff_packbits_decode(const unsigned char* src, long long n, unsigned char* dst, long long cap); /* PackBits (compression 32773) */

#ifdef __cplusplus
}
#endif
#endif /* UNET_HIP_H */
