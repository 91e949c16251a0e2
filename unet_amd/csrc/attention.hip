// Fused SelfAttention for bf16 storage on gfx950 (MI355X): the N x N attention matrix never reaches HBM.
//
// fastai layers.py SelfAttention (DynamicUnet(self_attention=True): /root/reference/params_and_main.py:81-83, train.py:141-144):
//     beta = softmax(f^T g, dim=1),  o = gamma * (h beta) + x,   f, g: C/8 channels, h: C channels, N = H * W positions.
// In the row notation of unet_amd/modules.py (rows = positions of the NHWC QKV buffer [B][N][2 dp + C]: F at channel 0, G at dp, H at 2 dp):
//     T[j][i] = G_j . F_i      P[j][i] = exp(T[j][i] - lse_j)  (row softmax over i)      O_j = sum_i P[j][i] H_i
//     dP[j][i] = dO_j . H_i    dT = P o (dP - D_j),  D_j = dO_j . O_j
//     dH_i = sum_j P[j][i] dO_j      dF_i = sum_j dT[j][i] G_j      dG_j = sum_i dT[j][i] F_i
// Four kernels, all v_mfma_f32_16x16x32_bf16 with fp32 accumulators, logits / weights / dP in fp32 registers:
//   sa_pack_kernel    rows-blocked transposed image of a channel slice: what a product that sums over POSITIONS reads as its B operand
//   sa_fwd_kernel     per 64 (or 128) query rows: pass 1 row maxima (logits only), pass 2 exp / row sums / P H; writes O and lse
//   sa_bwd_kv_kernel  per 64 key rows: sweeps the query rows 32 at a time, recomputes P from lse; dH and dF stay in accumulators
//   sa_bwd_q_kernel   per 64 query rows: sweeps the key rows 32 at a time; dG stays in accumulators (no atomics: results are run-to-run identical)
// Operand orientation: the logits tile is computed TRANSPOSED relative to the product that consumes it, so that its accumulator registers
// ARE the next MFMA's A fragment (k index = 8 (lane / 16) + e): rows of the tile are mapped to positions 8 (m / 4) + 4 h + m % 4 of a
// 32-position step (two tiles h = 0, 1), which puts positions 8 g .. 8 g + 7 into lane group g.  No LDS round trip for P.
// Algorithmic bytes per image and direction: the QKV tensor once + O once (forward); HBM traffic beyond that is the packed images
// (one extra copy of H, dO, F, G).  Work: 2 N^2 (dp + C) FLOP forward (+ 2 N^2 dp for the maxima pass), 2 N^2 (3 dp + 2 C) + 2 N^2 (2 dp + C) backward.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

#define ST ((hipStream_t)stream)

__device__ __forceinline__ u32x4 ldg16(const u16* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ bf16x8 as_bf(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }
__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    return (unsigned)__builtin_bit_cast(u16, (__bf16)lo) | ((unsigned)__builtin_bit_cast(u16, (__bf16)hi) << 16);
}
__device__ __forceinline__ bf16x8 pack8(const float (&p)[8]) {
    u32x4 v = {pack2(p[0], p[1]), pack2(p[2], p[3]), pack2(p[4], p[5]), pack2(p[6], p[7])};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ f32x4 mfma(bf16x8 a, bf16x8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ void st_bf(u16* p, float v) { *p = __builtin_bit_cast(u16, (__bf16)v); }

// position (inside a 32-position step) of row m of logits tile h: lane group g = m / 4 ends up with positions 8 g .. 8 g + 7
__device__ __forceinline__ int perm_pos(int m, int h) { return 8 * (m >> 2) + 4 * h + (m & 3); }

// Accumulator tiles -> whole rows.  a[ct][r] is row 4 (lane / 16) + r, channel 16 ct + lane % 16 of a 16-row block: per-lane stores would be
// 2-byte writes scattered over four rows (measured: ~250 us for the 50 MB of O at the cfg2 geometry); the wave instead writes its block into
// its own LDS region and reads it back as 16-byte pieces of whole rows (wave-local: LDS operations of one wave complete in order).
// dst = row 0 of the block, stride in elements; rows [0, nrows) and channels [0, cvalid) are stored.
template <int NT_, int PAD>
__device__ __forceinline__ void rows_out(u16* wl, const f32x4 (&a)[NT_], const float (&sc)[4], int lane, u16* dst, size_t stride, int nrows, int cvalid) {
    constexpr int PITCH = NT_ * 16 + PAD, VPR = NT_ * 2;
    const int jl = lane & 15, g = lane >> 4;
#pragma unroll
    for (int ct = 0; ct < NT_; ++ct)
#pragma unroll
        for (int r = 0; r < 4; ++r) st_bf(wl + (4 * g + r) * PITCH + ct * 16 + jl, a[ct][r] * sc[r]);
#pragma unroll
    for (int k = 0; k < (16 * VPR + 63) / 64; ++k) {
        const int v = lane + 64 * k, row = v / VPR, cv = v % VPR;
        if (row < 16 && row < nrows && cv * 8 < cvalid)
            *reinterpret_cast<u32x4*>(dst + (size_t)row * stride + cv * 8) = *reinterpret_cast<const u32x4*>(wl + row * PITCH + cv * 8);
    }
}

// Workgroups of one image run next to each other on ONE XCD (its L2 then holds the image's operands once): linear workgroup id L is
// dispatched to XCD L % 8; when the batch is a multiple of 8, XCD x takes images x, x + 8, ...
__device__ __forceinline__ void wg_image_block(int per_image, int B, int& b, int& blk) {
    const int L = blockIdx.x;
    if ((B & 7) == 0) {
        const int x = L & 7, k = L >> 3;
        b = x + 8 * (k / per_image);
        blk = k % per_image;
    } else {
        b = L / per_image;
        blk = L % per_image;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// packed image of X[b][n][co + c], c < cc:  out[b][blk][ct][ks][lane][e] = X[b][64 blk + 32 ks + 8 (lane / 16) + e][16 ct + lane % 16]
// (zeros past N and past cc): vector `lane` of (ct, ks) is the B fragment of a 16x16x32 MFMA that sums over positions.
// grid (NKB, B, ceil(NT / 24)), 256 threads, (64 x (16 ntc + 8)) bf16 of LDS
__global__ __launch_bounds__(256) void sa_pack_kernel(const u16* __restrict__ x, int cs, int co, int cc, int N, int NT, u16* __restrict__ out) {
    extern __shared__ u32x4 smem[];
    u16* s = reinterpret_cast<u16*>(smem);
    const int blk = blockIdx.x, b = blockIdx.y, ct0 = blockIdx.z * 24;
    const int ntc = min(24, NT - ct0), pitch = ntc * 16 + 8, vpr = ntc * 2;
    for (int v = threadIdx.x; v < 64 * vpr; v += 256) {
        const int row = v / vpr, c = (v % vpr) * 8, n = blk * 64 + row;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (n < N && ct0 * 16 + c < cc) val = ldg16(x + ((size_t)b * N + n) * cs + co + ct0 * 16 + c);
        *reinterpret_cast<u32x4*>(s + row * pitch + c) = val;
    }
    __syncthreads();
    u16* o = out + (((size_t)b * gridDim.x + blk) * NT + ct0) * 1024;
    for (int v = threadIdx.x; v < ntc * 128; v += 256) {
        const int lane = v & 63, ks = (v >> 6) & 1, ct = v >> 7;
        const u16* p = s + (ks * 32 + (lane >> 4) * 8) * pitch + ct * 16 + (lane & 15);
        u32x4 val;
        val.x = p[0] | ((unsigned)p[pitch] << 16);
        val.y = p[2 * pitch] | ((unsigned)p[3 * pitch] << 16);
        val.z = p[4 * pitch] | ((unsigned)p[5 * pitch] << 16);
        val.w = p[6 * pitch] | ((unsigned)p[7 * pitch] << 16);
        *reinterpret_cast<u32x4*>(o + (size_t)v * 8) = val;
    }
}

// D[b][j] = sum_c dO[b][j][c] O[b][j][c]   (one wave per row, 8 channels per lane and step); D is [B][Np], rows = B * N
__global__ __launch_bounds__(256) void sa_rowdot_kernel(const u16* __restrict__ a, int a_cs, int a_co, const u16* __restrict__ o, int o_cs, int o_co,
                                                        long long rows, int C, int N, int Np, float* __restrict__ D) {
    const int lane = threadIdx.x & 63;
    for (long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += (long long)gridDim.x * 4) {
        float s = 0.f;
        for (int c = lane * 8; c < C; c += 512) {
            const u32x4 va = ldg16(a + r * a_cs + a_co + c), vo = ldg16(o + r * o_cs + o_co + c);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                s += __uint_as_float(va[q] << 16) * __uint_as_float(vo[q] << 16);
                s += __uint_as_float(va[q] & 0xffff0000u) * __uint_as_float(vo[q] & 0xffff0000u);
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
        if (lane == 0) D[(r / N) * Np + r % N] = s;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Masking of the key rows past N (ragged last block) in the EXACT instantiations: the query / key reduction is padded to 64 lanes and
// lane dp is free (dp <= 56), so the F fragment carries -1e30 there for rows past N and every G fragment 1.0: the logit of a padded key
// comes out of the MFMA as -1e30, its weight as exp(-1e30 - m) = 0, with no compare / select per element.
#define SA_NEG_BF 0xF149u       /* bf16(-9.95e29) */
#define SA_ONE_BF 0x3F80u
#define SA_LSE_PAD 1e30f        /* lse of the query rows past N (lse is [B][64 NKB]): their recomputed weights are exp(t - 1e30) = 0 */

// Forward.  Workgroup = NW waves, wave = QT tiles of 16 query rows, all of [ct0, ct0 + ntc) value-channel tiles in accumulators.
// grid.x = B * ceil(N / (64 QT)), grid.z = ceil(NT / NTM); LDS: 2 x NTM x 2 KB (the packed H image of one 64-key block, double buffered).
// EXACT: NT == NTM (C = 16 NTM) and dp <= 56 -- no per-tile guards, so the loop body is ONE basic block and the compiler interleaves the
// LDS reads with the MFMAs; the logits / exp of block kb + 1 are issued in front of the P H products of block kb (independent work for the
// scheduler to put into the MFMA shadow).
template <int QT, int NTM, bool EXACT, int NW>
__global__ __launch_bounds__(64 * NW) void sa_fwd_kernel(const u16* __restrict__ qkv, int cq, int dp, int C, int N, int NKB, int NT, int B,
                                                     const u32x4* __restrict__ vpack, u16* __restrict__ O, int o_cs, int o_co,
                                                     float* __restrict__ lse) {
    extern __shared__ u32x4 lds[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, jl = lane & 15, g = lane >> 4;
    int b, qblk;
    wg_image_block((N + 16 * NW * QT - 1) / (16 * NW * QT), B, b, qblk);
    const int ct0 = EXACT ? 0 : blockIdx.z * NTM, ntc = EXACT ? NTM : min(NTM, NT - ct0);
    const int qw = (qblk * NW + w) * (16 * QT);
    const u16* qb = qkv + (size_t)b * N * cq;
    const u32x4* vb = vpack + (size_t)b * NKB * NT * 128;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const bool ragged = (N & 63) != 0;

    bf16x8 gq[QT][2];                                       // B fragments of T^T = F G^T: G[query jl][32 cs + 8 g + e]
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
            const int j = qw + qt * 16 + jl, c = cs * 32 + g * 8;
            u32x4 v = (j < N && c < dp) ? ldg16(qb + (size_t)j * cq + dp + c) : zero4;
            if (EXACT && c == dp) v.x = SA_ONE_BF;
            gq[qt][cs] = as_bf(v);
        }
    auto load_fa = [&](int kb, u32x4 (&fa)[4][2]) {         // A fragments: F[key perm_pos(jl, h) of step ks][32 cs + 8 g + e], tile t = 2 ks + h
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int key = kb * 64 + (t >> 1) * 32 + perm_pos(jl, t & 1);
#pragma unroll
            for (int cs = 0; cs < 2; ++cs) {
                const int c = cs * 32 + g * 8;
                u32x4 v = (key < N && c < dp) ? ldg16(qb + (size_t)key * cq + c) : zero4;
                if (EXACT && c == dp && key >= N) v.x = SA_NEG_BF;
                fa[t][cs] = v;
            }
        }
    };

    // ---- pass 1: row maxima (exact softmax, no rescaling of accumulators later)
    float mx[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) mx[qt] = -INFINITY;
    u32x4 fa[4][2], fn[4][2];
    load_fa(0, fa);
    for (int kb = 0; kb < NKB; ++kb) {
        if (kb + 1 < NKB) load_fa(kb + 1, fn);
        const bool tail = !EXACT && ragged && kb == NKB - 1;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int qt = 0; qt < QT; ++qt) {
                f32x4 s = {0.f, 0.f, 0.f, 0.f};
                s = mfma(as_bf(fa[t][0]), gq[qt][0], s);
                s = mfma(as_bf(fa[t][1]), gq[qt][1], s);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float v = s[r];
                    if (!EXACT) {
                        const int key = kb * 64 + (t >> 1) * 32 + 8 * g + 4 * (t & 1) + r;
                        if (tail && key >= N) v = -INFINITY;
                    }
                    mx[qt] = fmaxf(mx[qt], v);
                }
            }
        if (kb + 1 < NKB) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { fa[t][0] = fn[t][0]; fa[t][1] = fn[t][1]; }
        }
    }
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 16));
        mx[qt] = fmaxf(mx[qt], __shfl_xor(mx[qt], 32));
    }

    // ---- pass 2
    constexpr int NTHR = 64 * NW, NS = (NTM * 128 + NTHR - 1) / NTHR;
    u32x4 stg[NS];
    auto stage_ld = [&](int kb) {
        const u32x4* src = vb + ((size_t)kb * NT + ct0) * 128;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int idx = tid + i * NTHR;
            stg[i] = ((EXACT && NTM * 128 % NTHR == 0) || idx < ntc * 128) ? src[idx] : zero4;
        }
    };
    auto stage_st = [&](int bi) {
        u32x4* d = lds + bi * (NTM * 128);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int idx = tid + i * NTHR;
            if ((EXACT && NTM * 128 % NTHR == 0) || idx < ntc * 128) d[idx] = stg[i];
        }
    };
    f32x4 acc[QT][NTM];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt)
#pragma unroll
        for (int ct = 0; ct < NTM; ++ct) acc[qt][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
    float lsum[QT];
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) lsum[qt] = 0.f;
    // logits of block kb from the fragments in f, weights exp(t - m) as the A fragments of the P H product
    auto soft = [&](int kb, const u32x4 (&f)[4][2], bf16x8 (&pa)[QT][2]) {
        const bool tail = !EXACT && ragged && kb == NKB - 1;
#pragma unroll
        for (int qt = 0; qt < QT; ++qt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float p[8];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 s = {0.f, 0.f, 0.f, 0.f};
                    s = mfma(as_bf(f[ks * 2 + h][0]), gq[qt][0], s);
                    s = mfma(as_bf(f[ks * 2 + h][1]), gq[qt][1], s);
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float e = __expf(s[r] - mx[qt]);
                        if (!EXACT) {
                            const int key = kb * 64 + ks * 32 + 8 * g + 4 * h + r;
                            if (tail && key >= N) e = 0.f;
                        }
                        lsum[qt] += e;
                        p[4 * h + r] = e;
                    }
                }
                pa[qt][ks] = pack8(p);
            }
    };

    bf16x8 pa[QT][2], pn[QT][2];
    // P H of one key block: 2 NTM fragments of the packed H image through a ring of 8 register sets, each read 8 fragments ahead
    auto pv = [&](const u32x4* buf) {
        if constexpr (EXACT) {
            u32x4 ring[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ring[i] = buf[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 2 * NTM; ++i) {                 // fragment i = (ct, ks) = (i / 2, i % 2)
                const bf16x8 bv = as_bf(ring[i & 7]);
#pragma unroll
                for (int qt = 0; qt < QT; ++qt) acc[qt][i >> 1] = mfma(pa[qt][i & 1], bv, acc[qt][i >> 1]);
                if (i + 8 < 2 * NTM) ring[i & 7] = buf[(i + 8) * 64 + lane];
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int ct = 0; ct < NTM; ++ct)
                    if (ct < ntc) {
                        const bf16x8 bv = as_bf(buf[(ct * 2 + ks) * 64 + lane]);
#pragma unroll
                        for (int qt = 0; qt < QT; ++qt) acc[qt][ct] = mfma(pa[qt][ks], bv, acc[qt][ct]);
                    }
        }
    };
    stage_ld(0);
    load_fa(0, fa);
    stage_st(0);
    soft(0, fa, pa);
    if (NKB > 1) load_fa(1, fa);
    __syncthreads();
    for (int kb = 0; kb + 1 < NKB; ++kb) {
        stage_ld(kb + 1);
        if (kb + 2 < NKB) load_fa(kb + 2, fn);
        soft(kb + 1, fa, pn);
        pv(lds + (kb & 1) * (NTM * 128));
        stage_st((kb + 1) & 1);
#pragma unroll
        for (int qt = 0; qt < QT; ++qt) { pa[qt][0] = pn[qt][0]; pa[qt][1] = pn[qt][1]; }
        if (kb + 2 < NKB) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { fa[t][0] = fn[t][0]; fa[t][1] = fn[t][1]; }
        }
        __syncthreads();
    }
    pv(lds + ((NKB - 1) & 1) * (NTM * 128));

    // ---- epilogue: O = acc / l, lse = m + log l.  acc[qt][ct][r] belongs to query 4 g + r, channel 16 ct + jl; l of query q sits in lane q
    const int Np = NKB * 64;
    if (EXACT) __syncthreads();                              // the staging buffers become the row-store regions of the waves
#pragma unroll
    for (int qt = 0; qt < QT; ++qt) {
        float l = lsum[qt];
        l += __shfl_xor(l, 16);
        l += __shfl_xor(l, 32);
        const int jq = qw + qt * 16 + jl;
        if (g == 0 && blockIdx.z == 0 && jq < Np) lse[(size_t)b * Np + jq] = jq < N ? mx[qt] + __logf(l) : SA_LSE_PAD;
        const float linv = 1.f / l;
        float li[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) li[r] = __shfl(linv, 4 * g + r);
        if constexpr (EXACT) {
            constexpr int PAD = NW * (NTM * 16 + 8) * 32 <= 2 * NTM * 2048 ? 8 : 0;
            const int j0 = qw + qt * 16;
            rows_out<NTM, PAD>(reinterpret_cast<u16*>(lds) + w * 16 * (NTM * 16 + PAD), acc[qt], li, lane, O + ((size_t)b * N + j0) * o_cs + o_co, (size_t)o_cs,
                               N - j0, NTM * 16);
        } else {
#pragma unroll
            for (int ct = 0; ct < NTM; ++ct)
                if (ct < ntc) {
                    const int c = (ct0 + ct) * 16 + jl;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int j = qw + qt * 16 + 4 * g + r;
                        if (j < N && c < C) st_bf(O + ((size_t)b * N + j) * o_cs + o_co + c, acc[qt][ct][r] * li[r]);
                    }
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward, key side.  Workgroup = 64 key rows (4 waves x 16), sweeps the query rows 32 at a time.
// LDS image of one step (double buffered), in 16-byte vectors:
//   [0, 2 NKS 64)          dOA  [cs][h][lane]   A fragments of dP = dO H^T: dO[query perm_pos(lane % 16, h)][32 cs + 8 (lane / 16) + e]
//   [.., + NT 64)          dOT  [ct][lane]      packed dO image of the step: B fragments of dH += P^T dO
//   [.., + 256)            GA   [h][cs][lane]   A fragments of T = G F^T
//   [.., + 256)            GT   [ct][lane]      packed G image (4 tiles, zeros past dp): B fragments of dF += dT^T G
// lse and D are [B][64 NKB]; rows past N hold lse = 1e30 (weight 0) and a finite D.  EXACT: C == 16 NTM, no per-tile guards.
template <int NTM, bool EXACT>
__global__ __launch_bounds__(256) void sa_bwd_kv_kernel(const u16* __restrict__ qkv, int cq, int dp, int C, int N, int NKB, int NT, int B,
                                                        const u16* __restrict__ dO, int do_cs, int do_co, const u32x4* __restrict__ dopack,
                                                        const u32x4* __restrict__ gpack, const float* __restrict__ lse,
                                                        const float* __restrict__ D, u16* __restrict__ dqkv) {
    extern __shared__ u32x4 lds[];
    constexpr int NKSM = NTM / 2;                            // 32-channel steps of the dP reduction
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, il = lane & 15, g = lane >> 4;
    int b, kblk;
    wg_image_block(NKB, B, b, kblk);
    const int key0 = kblk * 64 + w * 16;
    const int NKS = EXACT ? NKSM : (C + 31) >> 5, NTr = EXACT ? NTM : NT, DT = (dp + 15) >> 4;
    const u16* qb = qkv + (size_t)b * N * cq;
    const u16* dob = dO + (size_t)b * N * do_cs + do_co;
    const u32x4* dpk = dopack + (size_t)b * NKB * NTr * 128;
    const u32x4* gpk = gpack + (size_t)b * NKB * DT * 128;
    const float* lseb = lse + (size_t)b * NKB * 64;
    const float* Db = D + (size_t)b * NKB * 64;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int o_dot = 2 * NKS * 64, o_ga = o_dot + NTr * 64, o_gt = o_ga + 256, nvec = o_gt + 256;
    const int NIT = NKB * 2;

    // B fragments held for the whole sweep: F and H of this wave's keys (column il of the tiles)
    bf16x8 fk[2], hk[NKSM];
    {
        const int key = key0 + il;
#pragma unroll
        for (int cs = 0; cs < 2; ++cs) {
            const int c = cs * 32 + g * 8;
            fk[cs] = as_bf((key < N && c < dp) ? ldg16(qb + (size_t)key * cq + c) : zero4);
        }
#pragma unroll
        for (int cs = 0; cs < NKSM; ++cs) {
            const int c = cs * 32 + g * 8;
            hk[cs] = as_bf(((EXACT || (cs < NKS && c < C)) && key < N) ? ldg16(qb + (size_t)key * cq + 2 * dp + c) : zero4);
        }
    }
    // staging (registers -> LDS one step ahead).  Vector (u, ln): u = w + 4 i walks a section 64 lanes at a time, ln = lane
    constexpr int NA = NKSM / 2, NTV = NTM / 4;
    u32x4 sA[NA], sT[NTV], sGA, sGT;
    f32x4 lq[2], dq[2], lqn[2], dqn[2];                      // lse / D of the step's queries 8 g + 4 h + r
    auto stage_ld = [&](int it) {
        const int q0 = it * 32;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            lqn[h] = *reinterpret_cast<const f32x4*>(lseb + q0 + 8 * g + 4 * h);
            dqn[h] = *reinterpret_cast<const f32x4*>(Db + q0 + 8 * g + 4 * h);
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {                      // dOA [cs][h][lane]
            const int u = w + 4 * i, h = u & 1, cs = u >> 1;
            const int j = q0 + perm_pos(il, h), c = cs * 32 + g * 8;
            sA[i] = ((EXACT || (cs < NKS && c < C)) && j < N) ? ldg16(dob + (size_t)j * do_cs + c) : zero4;
        }
#pragma unroll
        for (int i = 0; i < NTV; ++i) {                     // dOT [ct][lane]: packed block it / 2, step it % 2
            const int ct = w + 4 * i;
            sT[i] = (EXACT || ct < NTr) ? dpk[(((size_t)(it >> 1) * NTr + ct) * 2 + (it & 1)) * 64 + lane] : zero4;
        }
        {                                                   // GA [h][cs][lane]
            const int h = w >> 1, cs = w & 1;
            const int j = q0 + perm_pos(il, h), c = cs * 32 + g * 8;
            sGA = (j < N && c < dp) ? ldg16(qb + (size_t)j * cq + dp + c) : zero4;
        }
        sGT = w < DT ? gpk[(((size_t)(it >> 1) * DT + w) * 2 + (it & 1)) * 64 + lane] : zero4;      // GT [ct][lane]
    };
    auto stage_st = [&](int bi) {
        u32x4* d = lds + bi * nvec;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (EXACT || (w + 4 * i) < 2 * NKS) d[(w + 4 * i) * 64 + lane] = sA[i];
#pragma unroll
        for (int i = 0; i < NTV; ++i)
            if (EXACT || w + 4 * i < NTr) d[o_dot + (w + 4 * i) * 64 + lane] = sT[i];
        d[o_ga + w * 64 + lane] = sGA;
        d[o_gt + w * 64 + lane] = sGT;
    };
    f32x4 accH[NTM], accF[4];
#pragma unroll
    for (int ct = 0; ct < NTM; ++ct) accH[ct] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) accF[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_ld(0);
    stage_st(0);
    lq[0] = lqn[0]; lq[1] = lqn[1]; dq[0] = dqn[0]; dq[1] = dqn[1];
    __syncthreads();
    for (int it = 0; it < NIT; ++it) {
        if (it + 1 < NIT) stage_ld(it + 1);
        const u32x4* buf = lds + (it & 1) * nvec;
        // rows of both tiles are queries it * 32 + 8 g + 4 h + r in this lane.  The step's 56 fragments are read through a ring of 8
        // registers sets in the order the MFMAs consume them (GA h0, dOA h0, GA h1, dOA h1, dOT, GT), each read 8 MFMAs ahead of its use
        auto frag = [&](int i) -> u32x4 {
            if (i < 28) {
                const int h = i / 14, k = i % 14;
                return k < 2 ? buf[o_ga + (h * 2 + k) * 64 + lane] : buf[((k - 2) * 2 + h) * 64 + lane];
            }
            return i < 52 ? buf[o_dot + (i - 28) * 64 + lane] : buf[o_gt + (i - 52) * 64 + lane];
        };
        if constexpr (EXACT) {
            u32x4 ring[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ring[i] = frag(i);
            f32x4 t[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dpv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            bf16x8 pfrag, dtfrag;
#pragma unroll
            for (int i = 0; i < 56; ++i) {
                const bf16x8 f = as_bf(ring[i & 7]);
                if (i < 28) {
                    const int h = i / 14, k = i % 14;
                    if (k < 2) t[h] = mfma(f, fk[k], t[h]);
                    else dpv[h] = mfma(f, hk[k - 2], dpv[h]);
                } else {
                    if (i == 28) {
                        float p[8], dt[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const float pe = __expf(t[h][r] - lq[h][r]);
                                p[4 * h + r] = pe;
                                dt[4 * h + r] = pe * (dpv[h][r] - dq[h][r]);
                            }
                        pfrag = pack8(p);
                        dtfrag = pack8(dt);
                    }
                    if (i < 52) accH[i - 28] = mfma(pfrag, f, accH[i - 28]);
                    else accF[i - 52] = mfma(dtfrag, f, accF[i - 52]);
                }
                if (i + 8 < 56) ring[i & 7] = frag(i + 8);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float p[8], dt[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f}, dpv = {0.f, 0.f, 0.f, 0.f};
                t = mfma(as_bf(buf[o_ga + (h * 2 + 0) * 64 + lane]), fk[0], t);
                t = mfma(as_bf(buf[o_ga + (h * 2 + 1) * 64 + lane]), fk[1], t);
#pragma unroll
                for (int cs = 0; cs < NKSM; ++cs)
                    if (cs < NKS) dpv = mfma(as_bf(buf[(cs * 2 + h) * 64 + lane]), hk[cs], dpv);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float pe = __expf(t[r] - lq[h][r]);
                    p[4 * h + r] = pe;
                    dt[4 * h + r] = pe * (dpv[r] - dq[h][r]);
                }
            }
            const bf16x8 pfrag = pack8(p), dtfrag = pack8(dt);
#pragma unroll
            for (int ct = 0; ct < NTM; ++ct)
                if (ct < NTr) accH[ct] = mfma(pfrag, as_bf(buf[o_dot + ct * 64 + lane]), accH[ct]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) accF[ct] = mfma(dtfrag, as_bf(buf[o_gt + ct * 64 + lane]), accF[ct]);
        }
        if (it + 1 < NIT) {
            stage_st((it + 1) & 1);
            lq[0] = lqn[0]; lq[1] = lqn[1]; dq[0] = dqn[0]; dq[1] = dqn[1];
        }
        __syncthreads();
    }
    // acc[ct][r]: key key0 + 4 g + r, channel 16 ct + il
    u16* db = dqkv + (size_t)b * N * cq;
    if constexpr (EXACT) {                                   // (the loop ended on a barrier: nobody reads the step images any more)
        const float one[4] = {1.f, 1.f, 1.f, 1.f};
        u16* wl = reinterpret_cast<u16*>(lds) + w * 16 * (NTM * 16 + 8);
        rows_out<NTM, 8>(wl, accH, one, lane, db + (size_t)key0 * cq + 2 * dp, (size_t)cq, N - key0, NTM * 16);
        rows_out<4, 8>(wl, accF, one, lane, db + (size_t)key0 * cq, (size_t)cq, N - key0, dp);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int key = key0 + 4 * g + r;
            if (key >= N) continue;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int c = ct * 16 + il;
                if (c < dp) st_bf(db + (size_t)key * cq + c, accF[ct][r]);
            }
#pragma unroll
            for (int ct = 0; ct < NTM; ++ct) {
                const int c = ct * 16 + il;
                if (ct < NTr && c < C) st_bf(db + (size_t)key * cq + 2 * dp + c, accH[ct][r]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward, query side.  Workgroup = 64 query rows (4 waves x 16), sweeps the key rows 32 at a time; logits tiles transposed (rows = keys).
// LDS image of one step:  FA [h][cs 0..1][lane] | HA [cs 0..NKS)[h][lane] | FT [ct 0..3][lane] (packed F image: B fragments of dG += dT F)
// EXACT: C == 16 NTM and dp <= 56 (key rows past N masked through the bias lane, see SA_NEG_BF)
template <int NTM, bool EXACT>
__global__ __launch_bounds__(256) void sa_bwd_q_kernel(const u16* __restrict__ qkv, int cq, int dp, int C, int N, int NKB, int B,
                                                       const u16* __restrict__ dO, int do_cs, int do_co, const u32x4* __restrict__ fpack,
                                                       const float* __restrict__ lse, const float* __restrict__ D, u16* __restrict__ dqkv) {
    extern __shared__ u32x4 lds[];
    constexpr int NKSM = NTM / 2;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, jl = lane & 15, g = lane >> 4;
    int b, qblk;
    wg_image_block(NKB, B, b, qblk);
    const int q0 = qblk * 64 + w * 16;
    const int NKS = EXACT ? NKSM : (C + 31) >> 5, DT = (dp + 15) >> 4;
    const u16* qb = qkv + (size_t)b * N * cq;
    const u16* dob = dO + (size_t)b * N * do_cs + do_co;
    const u32x4* fpk = fpack + (size_t)b * NKB * DT * 128;
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const int o_ha = 256, o_ft = o_ha + 2 * NKS * 64, nvec = o_ft + 256;
    const int NIT = NKB * 2;
    const bool ragged = (N & 63) != 0;

    bf16x8 gq[2], dq[NKSM];                                  // B fragments: G and dO of query q0 + jl
    const int jq = q0 + jl;
#pragma unroll
    for (int cs = 0; cs < 2; ++cs) {
        const int c = cs * 32 + g * 8;
        u32x4 v = (jq < N && c < dp) ? ldg16(qb + (size_t)jq * cq + dp + c) : zero4;
        if (EXACT && c == dp) v.x = SA_ONE_BF;
        gq[cs] = as_bf(v);
    }
#pragma unroll
    for (int cs = 0; cs < NKSM; ++cs) {
        const int c = cs * 32 + g * 8;
        dq[cs] = as_bf(((EXACT || (cs < NKS && c < C)) && jq < N) ? ldg16(dob + (size_t)jq * do_cs + c) : zero4);
    }
    const float lse_j = lse[(size_t)b * NKB * 64 + jq];      // (rows past N: 1e30, weight 0)
    const float D_j = D[(size_t)b * NKB * 64 + jq];

    constexpr int NA = NKSM / 2;
    u32x4 sFA, sHA[NA], sFT;
    auto stage_ld = [&](int it) {
        const int k0 = it * 32;
        {                                                   // FA [h][cs][lane]
            const int h = w >> 1, cs = w & 1;
            const int key = k0 + perm_pos(jl, h), c = cs * 32 + g * 8;
            sFA = (key < N && c < dp) ? ldg16(qb + (size_t)key * cq + c) : zero4;
            if (EXACT && c == dp && key >= N) sFA.x = SA_NEG_BF;
        }
#pragma unroll
        for (int i = 0; i < NA; ++i) {                      // HA [cs][h][lane]
            const int u = w + 4 * i, h = u & 1, cs = u >> 1;
            const int key = k0 + perm_pos(jl, h), c = cs * 32 + g * 8;
            sHA[i] = ((EXACT || (cs < NKS && c < C)) && key < N) ? ldg16(qb + (size_t)key * cq + 2 * dp + c) : zero4;
        }
        sFT = w < DT ? fpk[(((size_t)(it >> 1) * DT + w) * 2 + (it & 1)) * 64 + lane] : zero4;      // FT [ct][lane]
    };
    auto stage_st = [&](int bi) {
        u32x4* d = lds + bi * nvec;
        d[w * 64 + lane] = sFA;
#pragma unroll
        for (int i = 0; i < NA; ++i)
            if (EXACT || (w + 4 * i) < 2 * NKS) d[o_ha + (w + 4 * i) * 64 + lane] = sHA[i];
        d[o_ft + w * 64 + lane] = sFT;
    };
    f32x4 accG[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) accG[ct] = f32x4{0.f, 0.f, 0.f, 0.f};

    stage_ld(0);
    stage_st(0);
    __syncthreads();
    for (int it = 0; it < NIT; ++it) {
        if (it + 1 < NIT) stage_ld(it + 1);
        const u32x4* buf = lds + (it & 1) * nvec;
        const bool tail = !EXACT && ragged && it >= NIT - 2;
        if constexpr (EXACT) {
            // 32 fragments in consumption order (FA h0, HA h0, FA h1, HA h1, FT) through a ring of 8 register sets
            auto frag = [&](int i) -> u32x4 {
                if (i < 28) {
                    const int h = i / 14, k = i % 14;
                    return k < 2 ? buf[(h * 2 + k) * 64 + lane] : buf[o_ha + ((k - 2) * 2 + h) * 64 + lane];
                }
                return buf[o_ft + (i - 28) * 64 + lane];
            };
            u32x4 ring[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) ring[i] = frag(i);
            f32x4 t[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, dpv[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            bf16x8 dtfrag;
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                const bf16x8 f = as_bf(ring[i & 7]);
                if (i < 28) {
                    const int h = i / 14, k = i % 14;
                    if (k < 2) t[h] = mfma(f, gq[k], t[h]);
                    else dpv[h] = mfma(f, dq[k - 2], dpv[h]);
                } else {
                    if (i == 28) {
                        float dt[8];
#pragma unroll
                        for (int h = 0; h < 2; ++h)
#pragma unroll
                            for (int r = 0; r < 4; ++r) dt[4 * h + r] = __expf(t[h][r] - lse_j) * (dpv[h][r] - D_j);
                        dtfrag = pack8(dt);
                    }
                    accG[i - 28] = mfma(dtfrag, f, accG[i - 28]);
                }
                if (i + 8 < 32) ring[i & 7] = frag(i + 8);
                if ((i & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            float dt[8];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 t = {0.f, 0.f, 0.f, 0.f}, dpv = {0.f, 0.f, 0.f, 0.f};
                t = mfma(as_bf(buf[(h * 2 + 0) * 64 + lane]), gq[0], t);
                t = mfma(as_bf(buf[(h * 2 + 1) * 64 + lane]), gq[1], t);
#pragma unroll
                for (int cs = 0; cs < NKSM; ++cs)
                    if (cs < NKS) dpv = mfma(as_bf(buf[o_ha + (cs * 2 + h) * 64 + lane]), dq[cs], dpv);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float de = __expf(t[r] - lse_j) * (dpv[r] - D_j);
                    const int key = it * 32 + 8 * g + 4 * h + r;
                    if (tail && key >= N) de = 0.f;
                    dt[4 * h + r] = de;
                }
            }
            const bf16x8 dtfrag = pack8(dt);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) accG[ct] = mfma(dtfrag, as_bf(buf[o_ft + ct * 64 + lane]), accG[ct]);
        }
        if (it + 1 < NIT) stage_st((it + 1) & 1);
        __syncthreads();
    }
    u16* db = dqkv + (size_t)b * N * cq;
    if constexpr (EXACT) {
        const float one[4] = {1.f, 1.f, 1.f, 1.f};
        rows_out<4, 8>(reinterpret_cast<u16*>(lds) + w * 16 * (4 * 16 + 8), accG, one, lane, db + (size_t)q0 * cq + dp, (size_t)cq, N - q0, dp);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int j = q0 + 4 * g + r;
            if (j >= N) continue;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int c = ct * 16 + jl;
                if (c < dp) st_bf(db + (size_t)j * cq + dp + c, accG[ct][r]);
            }
        }
    }
}

inline int nkb_of(int N) { return (N + 63) / 64; }
inline bool sa_shape_ok(int cq, int dp, int C, int B, int N) {
    return B > 0 && N > 0 && dp > 0 && dp <= 64 && dp % 8 == 0 && C > 0 && C <= 512 && C % 8 == 0 && cq % 8 == 0 && cq >= 2 * dp + C;
}

}  // namespace

extern "C" int unet_sa_fused_supported(int dp, int C) { return dp > 0 && dp <= 64 && dp % 8 == 0 && C > 0 && C <= 512 && C % 8 == 0; }

extern "C" size_t unet_sa_pack_elems(int N, int cc) { return (size_t)nkb_of(N) * ((cc + 15) / 16) * 1024; }

extern "C" int unet_sa_pack_bf16(const unet_bf16* x, int x_cs, int x_co, int cc, int B, int N, unet_bf16* out, void* stream) {
    UNET_CHECK_ARG(x && out && B > 0 && N > 0 && cc > 0 && cc % 8 == 0 && unet::slice_ok_v(x_cs, x_co, cc, 8) && unet::aligned16(x) && unet::aligned16(out),
                   "sa_pack_bf16: bad args");
    const int NT = (cc + 15) / 16, ntc = NT < 24 ? NT : 24;
    hipLaunchKernelGGL(sa_pack_kernel, dim3(nkb_of(N), B, unet::cdiv(NT, 24)), dim3(256), (size_t)64 * (ntc * 16 + 8) * 2, ST,
                       (const u16*)x, x_cs, x_co, cc, N, NT, (u16*)out);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_sa_rowdot_bf16(const unet_bf16* a, int a_cs, int a_co, const unet_bf16* o, int o_cs, int o_co, int B, int N, int C, float* D,
                                   void* stream) {
    UNET_CHECK_ARG(a && o && D && B > 0 && N > 0 && C > 0 && C % 8 == 0 && unet::slice_ok_v(a_cs, a_co, C, 8) && unet::slice_ok_v(o_cs, o_co, C, 8),
                   "sa_rowdot_bf16: bad args");
    const long long rows = (long long)B * N;
    hipLaunchKernelGGL(sa_rowdot_kernel, dim3(unet::ew_grid(rows * 64, 256)), dim3(256), 0, ST, (const u16*)a, a_cs, a_co, (const u16*)o, o_cs, o_co, rows,
                       C, N, nkb_of(N) * 64, D);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

template <int QT, int NTM, bool EXACT, int NW>
static int launch_sa_fwd(const u16* qkv, int cq, int dp, int C, int B, int N, const u32x4* vpack, u16* O, int o_cs, int o_co, float* lse, hipStream_t st) {
    auto kern = sa_fwd_kernel<QT, NTM, EXACT, NW>;
    static unsigned long long configured = 0;
    if (unet::first_use_on_device(&configured))
        UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int NT = (C + 15) / 16, per = unet::cdiv(N, 16 * NW * QT);
    hipLaunchKernelGGL(kern, dim3((unsigned)(B * per), 1, EXACT ? 1 : unet::cdiv(NT, NTM)), dim3(64 * NW), (size_t)2 * NTM * 2048, st, qkv, cq, dp, C, N, nkb_of(N),
                       NT, B, vpack, O, o_cs, o_co, lse);
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_sa_fwd_bf16(const unet_bf16* qkv, int cq, int dp, int C, int B, int N, const unet_bf16* vpack, unet_bf16* O, int o_cs, int o_co,
                                float* lse, void* stream) {
    UNET_CHECK_ARG(qkv && vpack && O && lse && sa_shape_ok(cq, dp, C, B, N) && unet::slice_ok_v(o_cs, o_co, C, 8) && unet::aligned16(qkv) &&
                       unet::aligned16(vpack),
                   "sa_fwd_bf16: bad args (dp <= 64, C <= 512, channel counts multiples of 8)");
#define SA_FWD(QT_, NTM_, EX_, NW_) launch_sa_fwd<QT_, NTM_, EX_, NW_>((const u16*)qkv, cq, dp, C, B, N, (const u32x4*)vpack, (u16*)O, o_cs, o_co, lse, ST)
    if (C == 384 && dp <= 56) return SA_FWD(2, 24, true, 4);     // (8 waves x 1 query tile, two waves per SIMD: measured level, 510 vs 505 us)
    if (C <= 384) return SA_FWD(2, 24, false, 4);
    return SA_FWD(1, 32, false, 4);
#undef SA_FWD
}

template <int NTM, bool EXACT>
static int launch_sa_bwd(const u16* qkv, int cq, int dp, int C, int B, int N, const u16* dO, int do_cs, int do_co, const u32x4* dopack, const u32x4* gpack,
                         const u32x4* fpack, const float* lse, const float* D, u16* dqkv, hipStream_t st) {
    auto kkv = sa_bwd_kv_kernel<NTM, EXACT>;
    static unsigned long long configured = 0;
    const bool first = unet::first_use_on_device(&configured);
    if (first) UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kkv, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int NT = (C + 15) / 16, NKS = (C + 31) / 32, NKB = nkb_of(N);
    const size_t lds_kv = (size_t)2 * (2 * NKS * 64 + NT * 64 + 512) * 16;
    const size_t lds_q = (size_t)2 * (512 + 2 * NKS * 64) * 16;
    hipLaunchKernelGGL(kkv, dim3((unsigned)(B * NKB)), dim3(256), lds_kv, st, qkv, cq, dp, C, N, NKB, NT, B, dO, do_cs, do_co, dopack, gpack, lse, D, dqkv);
    UNET_CHECK_LAUNCH();
    if (EXACT && dp <= 56) {
        auto kq = sa_bwd_q_kernel<NTM, true>;
        if (first) UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(kq, dim3((unsigned)(B * NKB)), dim3(256), lds_q, st, qkv, cq, dp, C, N, NKB, B, dO, do_cs, do_co, fpack, lse, D, dqkv);
    } else {
        auto kq = sa_bwd_q_kernel<NTM, false>;
        if (first) UNET_CHECK_HIP(hipFuncSetAttribute((const void*)kq, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        hipLaunchKernelGGL(kq, dim3((unsigned)(B * NKB)), dim3(256), lds_q, st, qkv, cq, dp, C, N, NKB, B, dO, do_cs, do_co, fpack, lse, D, dqkv);
    }
    UNET_CHECK_LAUNCH();
    return UNET_OK;
}

extern "C" int unet_sa_bwd_bf16(const unet_bf16* qkv, int cq, int dp, int C, int B, int N, const unet_bf16* dO, int do_cs, int do_co,
                                const unet_bf16* dopack, const unet_bf16* gpack, const unet_bf16* fpack, const float* lse, const float* D,
                                unet_bf16* dqkv, void* stream) {
    UNET_CHECK_ARG(qkv && dO && dopack && gpack && fpack && lse && D && dqkv && sa_shape_ok(cq, dp, C, B, N) && unet::slice_ok_v(do_cs, do_co, C, 8) &&
                       unet::aligned16(qkv) && unet::aligned16(dO) && unet::aligned16(dopack) && unet::aligned16(gpack) && unet::aligned16(fpack) &&
                       unet::aligned16(lse) && unet::aligned16(D),
                   "sa_bwd_bf16: bad args (dp <= 64, C <= 512, channel counts multiples of 8)");
#define SA_BWD(NTM_, EX_)                                                                                                                               \
    launch_sa_bwd<NTM_, EX_>((const u16*)qkv, cq, dp, C, B, N, (const u16*)dO, do_cs, do_co, (const u32x4*)dopack, (const u32x4*)gpack, (const u32x4*)fpack, \
                             lse, D, (u16*)dqkv, ST)
    if (C == 384) return SA_BWD(24, true);
    if (C < 384) return SA_BWD(24, false);
    return SA_BWD(32, false);
#undef SA_BWD
}
