// ep24 - depthwise 3x3 convolution (groups = channels) of the DWConv blocks (network_blocks.py:57-76: depthwise BaseConv + 1x1
// BaseConv; switched in by `depthwise=True` in CSPDarknet / Bottleneck / YOLOPAFPN / YOLOXHead, in no BASELINE configuration).
//
// A depthwise conv has 9 multiply-adds per output value against 2 + 2 bytes moved: pure HBM-bound elementwise work, no MFMA.
// NHWC bf16 activations, the fp32 master weights [C][kh][kw] read in place (9 C floats: no packed copy), fp32 accumulation in
// tap order, one rounding to bf16.  Every thread owns ONE 8-channel group for the whole kernel (16-byte accesses, its 72
// weights in registers) and walks pixels with a block-wide stride, as the BatchNorm-backward kernels do (RowMap below).
//   forward          z[p][c]  = sum_t x[s p + off(t)][c] w[c][t]   + the BatchNorm batch statistics of z (2^-20 fixed point, replicas)
//   input gradient   dx[q][c] (+)= sum_t dz[(q - off(t)) / s][c] w[c][t]   over the taps whose source pixel exists
//   weight gradient  dw[c][t] = sum_p x[s p + off(t)][c] dz[p][c]  as per-workgroup partial sums (slabs, folded in order by
//                    ep24_wgrad_reduce: no atomics, bitwise reproducible like every other weight gradient of the step)
#include "common.h"

namespace {

struct RowMap {        // thread -> (row slot, 8-channel group); rows strided by rows per block
    int tpr, rpb;
    __device__ RowMap(int C) { tpr = C >> 3; rpb = tpr >= 256 ? 1 : 256 / tpr; }
};

__device__ __forceinline__ void load_w72(const float* w, int c0, int C, float (&wr)[9][8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int t = 0; t < 9; ++t) wr[t][j] = c0 + j < C ? w[(long)(c0 + j) * 9 + t] : 0.f;
}

template <int S>
__global__ __launch_bounds__(256) void dwconv_fwd_kernel(const bf16* x, long ld_x, const float* w, bf16* z, long ld_z, long long* stats,
                                                         int reps, int B, int H, int W, int C, int OH, int OW) {
    __shared__ float red[256][8 + 1];
    const RowMap rm(C);
    const int tid = threadIdx.x;
    const long M = (long)B * OH * OW;
    for (int cg0 = 0; cg0 < (C >> 3); cg0 += 256) {               // only loops when C > 2048
        const int cg = cg0 + tid % rm.tpr;
        const int slot = rm.tpr >= 256 ? 0 : tid / rm.tpr;
        const bool active = slot < rm.rpb && cg < (C >> 3);
        float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (active) {
            float wr[9][8];
            load_w72(w, cg * 8, C, wr);
            const long step = (long)gridDim.x * rm.rpb;
            for (long m = (long)blockIdx.x * rm.rpb + slot; m < M; m += step) {
                const int n = (int)(m / ((long)OH * OW));
                const int rem = (int)(m - (long)n * OH * OW);
                const int oy = rem / OW, ox = rem - oy * OW;
                bf16x8 v[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) {                       // every tap requested before the first is used
                    const int iy = oy * S + t / 3 - 1, ix = ox * S + t % 3 - 1;
                    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    const long p = ((long)n * H + (ok ? iy : 0)) * W + (ok ? ix : 0);
                    v[t] = *reinterpret_cast<const bf16x8*>(x + p * ld_x + cg * 8);
                    if (!ok) v[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float a = 0.f;
#pragma unroll
                    for (int t = 0; t < 9; ++t) a = fmaf((float)v[t][j], wr[t][j], a);
                    s1[j] += a; s2[j] += a * a;
                    o[j] = (bf16)a;
                }
                *reinterpret_cast<bf16x8*>(z + m * ld_z + cg * 8) = o;
            }
        }
        if (stats) {
            const int ngrp = min(min(rm.tpr, 256), (C >> 3) - cg0);            // channel groups this pass of the block covers (<= 256 threads)
#pragma unroll
            for (int half = 0; half < 2; ++half) {
#pragma unroll
                for (int j = 0; j < 8; ++j) red[tid][j] = half ? s2[j] : s1[j];
                __syncthreads();
                for (int t = tid; t < ngrp * 8; t += 256) {
                    const int g = t >> 3, vv = t & 7;
                    float a = 0.f;
                    for (int sl = 0; sl < rm.rpb; ++sl) a += red[sl * rm.tpr + g][vv];
                    long long* dst = stats + (long)(blockIdx.x % reps) * 2 * C + (long)half * C + (cg0 + g) * 8 + vv;
                    atomicAdd((unsigned long long*)dst, (unsigned long long)to_fix(a));
                }
                __syncthreads();
            }
        }
    }
}

// dx of pixel (y, x): tap (kh, kw) of output pixel (oy, ox) read input (oy S + kh - 1, ox S + kw - 1), so it reaches (y, x) from
// oy = (y + 1 - kh) / S when that is an integer inside the output
template <int S>
__global__ __launch_bounds__(256) void dwconv_dgrad_kernel(const bf16* dz, long ld_dz, const float* w, bf16* dx, long ld_dx, int accumulate,
                                                           int B, int H, int W, int C, int OH, int OW) {
    const RowMap rm(C);
    const int tid = threadIdx.x;
    const long M = (long)B * H * W;
    for (int cg0 = 0; cg0 < (C >> 3); cg0 += 256) {
        const int cg = cg0 + tid % rm.tpr;
        const int slot = rm.tpr >= 256 ? 0 : tid / rm.tpr;
        if (!(slot < rm.rpb && cg < (C >> 3))) continue;
        float wr[9][8];
        load_w72(w, cg * 8, C, wr);
        const long step = (long)gridDim.x * rm.rpb;
        for (long m = (long)blockIdx.x * rm.rpb + slot; m < M; m += step) {
            const int n = (int)(m / ((long)H * W));
            const int rem = (int)(m - (long)n * H * W);
            const int y = rem / W, xx = rem - y * W;
            bf16x8 v[9];
            bf16x8 old = {0, 0, 0, 0, 0, 0, 0, 0};
            if (accumulate) old = *reinterpret_cast<const bf16x8*>(dx + m * ld_dx + cg * 8);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ny = y + 1 - t / 3, nx = xx + 1 - t % 3;
                const bool ok = ny >= 0 && nx >= 0 && (S == 1 || ((ny & 1) == 0 && (nx & 1) == 0)) && ny / S < OH && nx / S < OW;
                const long p = ((long)n * OH + (ok ? ny / S : 0)) * OW + (ok ? nx / S : 0);
                v[t] = *reinterpret_cast<const bf16x8*>(dz + p * ld_dz + cg * 8);
                if (!ok) v[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
            }
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float a = 0.f;
#pragma unroll
                for (int t = 0; t < 9; ++t) a = fmaf((float)v[t][j], wr[t][j], a);
                o[j] = (bf16)(accumulate ? a + (float)old[j] : a);
            }
            *reinterpret_cast<bf16x8*>(dx + m * ld_dx + cg * 8) = o;
        }
    }
}

// slab[block][c][t]: this workgroup's partial sums over its pixels (zeros where it has none)
template <int S>
__global__ __launch_bounds__(256) void dwconv_wgrad_kernel(const bf16* x, long ld_x, const bf16* dz, long ld_dz, float* slab, int B, int H,
                                                           int W, int C, int OH, int OW) {
    __shared__ float red[256][8 + 1];
    const RowMap rm(C);
    const int tid = threadIdx.x;
    const long M = (long)B * OH * OW;
    float* out = slab + (long)blockIdx.x * C * 9;
    for (int cg0 = 0; cg0 < (C >> 3); cg0 += 256) {
        const int cg = cg0 + tid % rm.tpr;
        const int slot = rm.tpr >= 256 ? 0 : tid / rm.tpr;
        const bool active = slot < rm.rpb && cg < (C >> 3);
        float a[9][8];
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) a[t][j] = 0.f;
        if (active) {
            const long step = (long)gridDim.x * rm.rpb;
            for (long m = (long)blockIdx.x * rm.rpb + slot; m < M; m += step) {
                const int n = (int)(m / ((long)OH * OW));
                const int rem = (int)(m - (long)n * OH * OW);
                const int oy = rem / OW, ox = rem - oy * OW;
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(dz + m * ld_dz + cg * 8);
                bf16x8 v[9];
#pragma unroll
                for (int t = 0; t < 9; ++t) {
                    const int iy = oy * S + t / 3 - 1, ix = ox * S + t % 3 - 1;
                    const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
                    const long p = ((long)n * H + (ok ? iy : 0)) * W + (ok ? ix : 0);
                    v[t] = *reinterpret_cast<const bf16x8*>(x + p * ld_x + cg * 8);
                    if (!ok) v[t] = (bf16x8){0, 0, 0, 0, 0, 0, 0, 0};
                }
#pragma unroll
                for (int t = 0; t < 9; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[t][j] = fmaf((float)v[t][j], (float)g[j], a[t][j]);
            }
        }
        const int ngrp = min(min(rm.tpr, 256), (C >> 3) - cg0);            // channel groups this pass of the block covers (<= 256 threads)
#pragma unroll
        for (int t = 0; t < 9; ++t) {                             // the nine taps go through the fold one after the other (9 KB of LDS)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid][j] = a[t][j];
            __syncthreads();
            for (int q = tid; q < ngrp * 8; q += 256) {
                const int g = q >> 3, vv = q & 7;
                float sum = 0.f;
                for (int sl = 0; sl < rm.rpb; ++sl) sum += red[sl * rm.tpr + g][vv];
                out[(long)((cg0 + g) * 8 + vv) * 9 + t] = sum;
            }
            __syncthreads();
        }
    }
}

int dw_check(const char* what, const void* a, const void* b, const void* c, int64_t ld1, int64_t ld2, int B, int H, int W, int C, int ksize, int stride) {
    EP24_REQUIRE(a && b && c, EP24_E_ARG, "%s: null pointer", what);
    EP24_REQUIRE(ksize == 3 && (stride == 1 || stride == 2), EP24_E_UNSUPPORTED, "%s: k=%d s=%d unsupported (the DWConv blocks are 3x3, stride 1 or 2)", what, ksize, stride);
    EP24_REQUIRE(C > 0 && C % 8 == 0 && ld1 % 8 == 0 && ld2 % 8 == 0 && (C >> 3) <= 256 * 256, EP24_E_ARG, "%s: C=%d / row strides must be multiples of 8", what, C);
    EP24_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(c)) & 15) == 0, EP24_E_ARG, "%s: 16-byte aligned rows", what);
    EP24_REQUIRE(B > 0 && H > 0 && W > 0, EP24_E_ARG, "%s: empty", what);
    return EP24_OK;
}

int dw_grid(long M, int C, int cap) {
    const int tpr = C >> 3, rpb = tpr >= 256 ? 1 : 256 / tpr;
    long b = (M + (long)rpb * 4 - 1) / ((long)rpb * 4);            // ~4 pixels per thread
    return (int)(b < 1 ? 1 : b > cap ? cap : b);
}

}  // namespace

#define S_ (hipStream_t) stream

extern "C" int ep24_dwconv_fwd_bf16(const void* x, int64_t ld_x, const float* w, void* z, int64_t ld_z, int64_t* stats, int stats_replicas,
                                    int B, int H, int W, int C, int ksize, int stride, void* stream) {
    if (int rc = dw_check("dwconv_fwd", x, w, z, ld_x, ld_z, B, H, W, C, ksize, stride)) return rc;
    EP24_REQUIRE(!stats || stats_replicas > 0, EP24_E_ARG, "dwconv_fwd: stats_replicas");
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const int grid = dw_grid((long)B * OH * OW, C, 4096);
    if (stride == 1) hipLaunchKernelGGL(dwconv_fwd_kernel<1>, dim3(grid), dim3(256), 0, S_, (const bf16*)x, ld_x, w, (bf16*)z, ld_z, (long long*)stats, stats ? stats_replicas : 1, B, H, W, C, OH, OW);
    else hipLaunchKernelGGL(dwconv_fwd_kernel<2>, dim3(grid), dim3(256), 0, S_, (const bf16*)x, ld_x, w, (bf16*)z, ld_z, (long long*)stats, stats ? stats_replicas : 1, B, H, W, C, OH, OW);
    EP24_LAUNCH_CHECK("ep24_dwconv_fwd_bf16");
    return EP24_OK;
}

extern "C" int ep24_dwconv_dgrad_bf16(const void* dz, int64_t ld_dz, const float* w, void* dx, int64_t ld_dx, int accumulate, int B, int H,
                                      int W, int C, int ksize, int stride, void* stream) {
    if (int rc = dw_check("dwconv_dgrad", dz, w, dx, ld_dz, ld_dx, B, H, W, C, ksize, stride)) return rc;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const int grid = dw_grid((long)B * H * W, C, 4096);
    if (stride == 1) hipLaunchKernelGGL(dwconv_dgrad_kernel<1>, dim3(grid), dim3(256), 0, S_, (const bf16*)dz, ld_dz, w, (bf16*)dx, ld_dx, accumulate, B, H, W, C, OH, OW);
    else hipLaunchKernelGGL(dwconv_dgrad_kernel<2>, dim3(grid), dim3(256), 0, S_, (const bf16*)dz, ld_dz, w, (bf16*)dx, ld_dx, accumulate, B, H, W, C, OH, OW);
    EP24_LAUNCH_CHECK("ep24_dwconv_dgrad_bf16");
    return EP24_OK;
}

// workgroups (= slabs of C * 9 floats) of the weight-gradient launch: fixed by the shape, so that the caller can size the slab
extern "C" int ep24_dwconv_wgrad_splits(int B, int H, int W, int C, int stride) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 8 || (stride != 1 && stride != 2)) return -1;
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    const int tpr = C >> 3, rpb = tpr >= 256 ? 1 : 256 / tpr;
    long b = ((long)B * OH * OW + (long)rpb * 16 - 1) / ((long)rpb * 16);        // >= 16 pixels per thread: the fold costs 9 passes through LDS
    return (int)(b < 1 ? 1 : b > 256 ? 256 : b);
}

extern "C" int ep24_dwconv_wgrad_slab_bf16(const void* x, int64_t ld_x, const void* dz, int64_t ld_dz, float* slab, int64_t slab_floats, int B,
                                           int H, int W, int C, int ksize, int stride, void* stream) {
    if (int rc = dw_check("dwconv_wgrad", x, dz, slab, ld_x, ld_dz, B, H, W, C, ksize, stride)) return rc;
    const int splits = ep24_dwconv_wgrad_splits(B, H, W, C, stride);
    EP24_REQUIRE(slab_floats >= (int64_t)splits * C * 9, EP24_E_ARG, "dwconv_wgrad: the slab holds %ld floats, %d splits of %d need %ld", (long)slab_floats, splits,
                 C * 9, (long)splits * C * 9);
    const int OH = (H - 1) / stride + 1, OW = (W - 1) / stride + 1;
    if (stride == 1) hipLaunchKernelGGL(dwconv_wgrad_kernel<1>, dim3(splits), dim3(256), 0, S_, (const bf16*)x, ld_x, (const bf16*)dz, ld_dz, slab, B, H, W, C, OH, OW);
    else hipLaunchKernelGGL(dwconv_wgrad_kernel<2>, dim3(splits), dim3(256), 0, S_, (const bf16*)x, ld_x, (const bf16*)dz, ld_dz, slab, B, H, W, C, OH, OW);
    EP24_LAUNCH_CHECK("ep24_dwconv_wgrad_slab_bf16");
    return EP24_OK;
}
