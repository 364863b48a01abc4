// ep24 - the element-wise / pooling pieces the swapped backbones of BASELINE config 4 need next to the conv and BN+act
// kernels (yolox_24p/models/darknet.py:179-429: ResNet stem conv 7x7 stride 2 + MaxPool2d(3, 2, 1); Bottleneck's
// "out += identity; out = relu(out)").  All HBM bound: one pass over the tensor each, 16-byte accesses.
#include "common.h"

namespace {

constexpr int MAX_BLOCKS = 16384;
int cap_grid(long work_items) {
    long b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}

// im2col of the fp32 NCHW image for a k x k stride-s conv with padding p: rows [B*OH*OW][ld] bf16, column (kh*k + kw)*C + c
// (the order of the packed weight [Cout][kh][kw][Cin]), zeros in the padding and in columns >= k*k*C.  One thread per
// (row, 8-column chunk).
__global__ __launch_bounds__(256) void im2col_kernel(const float* img, bf16* rows, long ld, int B, int C, int H, int W, int k,
                                                     int s, int p, int OH, int OW) {
    const int chunks = (int)(ld >> 3);
    const long total = (long)B * OH * OW * chunks;
    const int kkc = k * k * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ch = (int)(i % chunks);
        const long row = i / chunks;
        const int n = (int)(row / ((long)OH * OW));
        const int rem = (int)(row - (long)n * OH * OW);
        const int oy = rem / OW, ox = rem - oy * OW;
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int col = ch * 8 + j;
            float v = 0.f;
            if (col < kkc) {
                const int c = col % C, t = col / C;
                const int kh = t / k, kw = t - kh * k;
                const int y = oy * s - p + kh, x = ox * s - p + kw;
                if (y >= 0 && y < H && x >= 0 && x < W) v = img[(((long)n * C + c) * H + y) * W + x];
            }
            o[j] = (bf16)v;
        }
        *reinterpret_cast<bf16x8*>(rows + row * ld + ch * 8) = o;
    }
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(bf16* y, long ld, long M, int C) {
    const int cgs = C >> 3;
    const long total = M * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cgs;
        const int cg = (int)(i - m * cgs);
        bf16x8* p = reinterpret_cast<bf16x8*>(y + m * ld + cg * 8);
        bf16x8 v = *p;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)v[j] > 0.f ? v[j] : (bf16)0.f;
        *p = v;
    }
}

// dy *= (y > 0), in place: the gradient of out = relu(u) with respect to u, where y is the stored output
__global__ __launch_bounds__(256) void relu_bwd_kernel(bf16* dy, long ld_dy, const bf16* y, long ld_y, long M, int C) {
    const int cgs = C >> 3;
    const long total = M * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cgs;
        const int cg = (int)(i - m * cgs);
        bf16x8* p = reinterpret_cast<bf16x8*>(dy + m * ld_dy + cg * 8);
        const bf16x8 yy = *reinterpret_cast<const bf16x8*>(y + m * ld_y + cg * 8);
        bf16x8 v = *p;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)yy[j] > 0.f ? v[j] : (bf16)0.f;
        *p = v;
    }
}

// MaxPool2d(kernel 3, stride 2, padding 1) on NHWC bf16; idx keeps the winning tap (kh*3 + kw) of every output element.
// Ties go to the first tap in (kh, kw) scan order - ATen's "val > maxval" update - which matters after a ReLU (many zeros).
__global__ __launch_bounds__(256) void maxpool3s2_fwd_kernel(const bf16* x, long ld_x, bf16* y, long ld_y, uint8_t* idx, int B,
                                                             int H, int W, int C, int OH, int OW) {
    const int cgs = C >> 3;
    const long total = (long)B * OH * OW * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long op = i / cgs;
        const int n = (int)(op / ((long)OH * OW));
        const int rem = (int)(op - (long)n * OH * OW);
        const int oy = rem / OW, ox = rem - oy * OW;
        float best[8];
        uint8_t code[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best[j] = -__builtin_inff(); code[j] = 0xFF; }
        for (int kh = 0; kh < 3; ++kh) {
            const int yy = oy * 2 - 1 + kh;
            if (yy < 0 || yy >= H) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int xx = ox * 2 - 1 + kw;
                if (xx < 0 || xx >= W) continue;
                const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)(n * H + yy) * W + xx) * ld_x + cg * 8);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float f = (float)v[j];
                    if (code[j] == 0xFF || f > best[j] || f != f) { best[j] = f; code[j] = (uint8_t)(kh * 3 + kw); }
                }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)best[j];
        *reinterpret_cast<bf16x8*>(y + op * ld_y + cg * 8) = o;
        uint2 c;
        c.x = code[0] | (code[1] << 8) | (code[2] << 16) | ((unsigned)code[3] << 24);
        c.y = code[4] | (code[5] << 8) | (code[6] << 16) | ((unsigned)code[7] << 24);
        *reinterpret_cast<uint2*>(idx + op * C + cg * 8) = c;
    }
}

// gather form (no atomics): input pixel (py,px) collects dy of the up to four windows that cover it and chose it
__global__ __launch_bounds__(256) void maxpool3s2_bwd_kernel(const bf16* dy, long ld_dy, const uint8_t* idx, bf16* dx, long ld_dx,
                                                             int accumulate, int B, int H, int W, int C, int OH, int OW) {
    const int cgs = C >> 3;
    const long total = (long)B * H * W * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int n = (int)(pix / ((long)H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem - py * W;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int kh = 0; kh < 3; ++kh) {
            const int t = py + 1 - kh;                     // py = 2*oy - 1 + kh
            if (t < 0 || (t & 1)) continue;
            const int oy = t >> 1;
            if (oy >= OH) continue;
            for (int kw = 0; kw < 3; ++kw) {
                const int u = px + 1 - kw;
                if (u < 0 || (u & 1)) continue;
                const int ox = u >> 1;
                if (ox >= OW) continue;
                const long op = (long)(n * OH + oy) * OW + ox;
                const uint2 c = *reinterpret_cast<const uint2*>(idx + op * C + cg * 8);
                const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + op * ld_dy + cg * 8);
                const unsigned want = (unsigned)(kh * 3 + kw);
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if ((((j < 4 ? c.x : c.y) >> (8 * (j & 3))) & 0xFFu) == want) acc[j] += (float)g[j];
            }
        }
        bf16x8* d = reinterpret_cast<bf16x8*>(dx + pix * ld_dx + cg * 8);
        bf16x8 o;
        if (accumulate) {
            const bf16x8 old = *d;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)old[j] + acc[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)acc[j];
        }
        *d = o;
    }
}

// Per-channel batch statistics of a stored tensor (sum and sum of squares in the conv epilogue's fixed-point layout
// stats[rep][2][ld_stats]): the pre-activation BatchNorms of a dense block normalise tensors that no conv epilogue saw in
// their final form (pooled block inputs, feature maps after Dropout2d).  Added into replica 0 at channel offset c0.
__global__ __launch_bounds__(256) void colstats_kernel(const bf16* x, long ld, long long* stats, long ld_stats, long M, int C) {
    __shared__ float red[2][256][8 + 1];
    const int chunks = C >> 3;
    const int rpp = 256 / chunks;
    const int r = threadIdx.x / chunks, c = threadIdx.x - r * chunks;
    float s1[8] = {0, 0, 0, 0, 0, 0, 0, 0}, s2[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (r < rpp) {
        for (long m = (long)blockIdx.x * rpp + r; m < M; m += (long)gridDim.x * rpp) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + m * ld + c * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float f = (float)v[j]; s1[j] += f; s2[j] += f * f; }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[0][threadIdx.x][j] = s1[j]; red[1][threadIdx.x][j] = s2[j]; }
    __syncthreads();
    for (int n = threadIdx.x; n < 2 * C; n += 256) {
        const int which = n / C, ch = n - which * C;
        float s = 0.f;
        for (int rr = 0; rr < rpp; ++rr) s += red[which][rr * chunks + (ch >> 3)][ch & 7];
        atomicAdd((unsigned long long*)(stats + (long)which * ld_stats + ch), (unsigned long long)to_fix(s));
    }
}

// layer view of the block statistics: dst[rep][2][C] <- src[rep][2][ld_src] first C channels
__global__ __launch_bounds__(256) void stats_gather_kernel(const long long* src, long ld_src, long long* dst, int C, int reps) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < reps * 2 * C; i += gridDim.x * 256) {
        const int c = i % C, w = (i / C) & 1, r = i / (2 * C);
        dst[i] = src[((long)r * 2 + w) * ld_src + c];
    }
}

// nn.AvgPool2d(2, 2) on NHWC bf16 (Transition, darknet.py:551-553)
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const bf16* x, long ld_x, bf16* y, long ld_y, int B, int H, int W, int C) {
    const int cgs = C >> 3, OH = H >> 1, OW = W >> 1;
    const long total = (long)B * OH * OW * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long op = i / cgs;
        const int n = (int)(op / ((long)OH * OW));
        const int rem = (int)(op - (long)n * OH * OW);
        const int oy = rem / OW, ox = rem - oy * OW;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)(n * H + 2 * oy + (t >> 1)) * W + 2 * ox + (t & 1)) * ld_x + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)(acc[j] * 0.25f);
        *reinterpret_cast<bf16x8*>(y + op * ld_y + cg * 8) = o;
    }
}

__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const bf16* dy, long ld_dy, bf16* dx, long ld_dx, int accumulate, int B,
                                                           int H, int W, int C) {
    const int cgs = C >> 3, OH = H >> 1, OW = W >> 1;
    const long total = (long)B * H * W * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int n = (int)(pix / ((long)H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem - py * W;
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + ((long)(n * OH + (py >> 1)) * OW + (px >> 1)) * ld_dy + cg * 8);
        bf16x8* d = reinterpret_cast<bf16x8*>(dx + pix * ld_dx + cg * 8);
        bf16x8 o;
        if (accumulate) {
            const bf16x8 old = *d;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)old[j] + (float)g[j] * 0.25f);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)g[j] * 0.25f);
        }
        *d = o;
    }
}

// nn.Dropout2d: every (sample, channel) plane is kept or zeroed as a whole; keep[n*C + c] holds 0 or 1/(1-p) (the host draws
// it once per step).  In place; the same launch is the backward (the gradient takes the same factors).
__global__ __launch_bounds__(256) void chanscale_kernel(bf16* x, long ld, const float* keep, int B, long HW, int C) {
    const int cgs = C >> 3;
    const long total = (long)B * HW * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int n = (int)(pix / HW);
        bf16x8* p = reinterpret_cast<bf16x8*>(x + pix * ld + cg * 8);
        bf16x8 v = *p;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] * keep[(long)n * C + cg * 8 + j]);
        *p = v;
    }
}

// nn.MaxPool2d(kernel_size=2, stride=2) (VGG, darknet.py:481): non-overlapping windows, idx = winning tap (dy*2 + dx, first
// maximum in scan order as ATen); the backward is a gather: every input pixel belongs to exactly one window.
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const bf16* x, long ld_x, bf16* y, long ld_y, uint8_t* idx, int B, int H,
                                                           int W, int C) {
    const int cgs = C >> 3, OH = H >> 1, OW = W >> 1;
    const long total = (long)B * OH * OW * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long op = i / cgs;
        const int n = (int)(op / ((long)OH * OW));
        const int rem = (int)(op - (long)n * OH * OW);
        const int oy = rem / OW, ox = rem - oy * OW;
        float best[8];
        uint8_t code[8];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x8 v = *reinterpret_cast<const bf16x8*>(x + ((long)(n * H + 2 * oy + (t >> 1)) * W + 2 * ox + (t & 1)) * ld_x + cg * 8);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float f = (float)v[j];
                if (t == 0 || f > best[j] || f != f) { best[j] = f; code[j] = (uint8_t)t; }
            }
        }
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16)best[j];
        *reinterpret_cast<bf16x8*>(y + op * ld_y + cg * 8) = o;
        uint2 c;
        c.x = code[0] | (code[1] << 8) | (code[2] << 16) | ((unsigned)code[3] << 24);
        c.y = code[4] | (code[5] << 8) | (code[6] << 16) | ((unsigned)code[7] << 24);
        *reinterpret_cast<uint2*>(idx + op * C + cg * 8) = c;
    }
}

__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const bf16* dy, long ld_dy, const uint8_t* idx, bf16* dx, long ld_dx,
                                                           int accumulate, int B, int H, int W, int C) {
    const int cgs = C >> 3, OH = H >> 1, OW = W >> 1;
    const long total = (long)B * H * W * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int n = (int)(pix / ((long)H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem - py * W;
        const long op = (long)(n * OH + (py >> 1)) * OW + (px >> 1);
        const unsigned want = (unsigned)((py & 1) * 2 + (px & 1));
        const uint2 c = *reinterpret_cast<const uint2*>(idx + op * C + cg * 8);
        const bf16x8 g = *reinterpret_cast<const bf16x8*>(dy + op * ld_dy + cg * 8);
        bf16x8* d = reinterpret_cast<bf16x8*>(dx + pix * ld_dx + cg * 8);
        bf16x8 o;
        const bf16x8 old = accumulate ? *d : bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool hit = (((j < 4 ? c.x : c.y) >> (8 * (j & 3))) & 0xFFu) == want;
            o[j] = (bf16)((float)old[j] + (hit ? (float)g[j] : 0.f));
        }
        *d = o;
    }
}

__global__ void incr_i64_kernel(long long* p) { *p += 1; }

}  // namespace

#define S_ (hipStream_t) stream

extern "C" int ep24_maxpool2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, uint8_t* idx, int B, int H, int W, int C,
                                 void* stream) {
    EP24_REQUIRE(x && y && idx && C % 8 == 0 && ld_x % 8 == 0 && ld_y % 8 == 0 && H % 2 == 0 && W % 2 == 0, EP24_E_ARG, "maxpool2_fwd: bad arguments");
    hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(cap_grid((long)B * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, S_, (const bf16*)x, (long)ld_x,
                       (bf16*)y, (long)ld_y, idx, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_maxpool2_fwd");
    return EP24_OK;
}

extern "C" int ep24_maxpool2_bwd(const void* dy, int64_t ld_dy, const uint8_t* idx, void* dx, int64_t ld_dx, int accumulate, int B,
                                 int H, int W, int C, void* stream) {
    EP24_REQUIRE(dy && idx && dx && C % 8 == 0 && ld_dy % 8 == 0 && ld_dx % 8 == 0 && H % 2 == 0 && W % 2 == 0, EP24_E_ARG, "maxpool2_bwd: bad arguments");
    hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(cap_grid((long)B * H * W * (C / 8))), dim3(256), 0, S_, (const bf16*)dy, (long)ld_dy, idx,
                       (bf16*)dx, (long)ld_dx, accumulate, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_maxpool2_bwd");
    return EP24_OK;
}

extern "C" int ep24_incr_i64(int64_t* p, void* stream) {
    EP24_REQUIRE(p, EP24_E_ARG, "incr_i64: null pointer");
    hipLaunchKernelGGL(incr_i64_kernel, dim3(1), dim3(1), 0, S_, (long long*)p);
    EP24_LAUNCH_CHECK("ep24_incr_i64");
    return EP24_OK;
}

extern "C" int ep24_colstats(const void* x, int64_t ld, int64_t* stats, int64_t ld_stats, int64_t M, int C, void* stream) {
    EP24_REQUIRE(x && stats && C % 8 == 0 && C <= 2048 && ld % 8 == 0 && M > 0 && ld_stats >= C, EP24_E_ARG, "colstats: bad arguments");
    long blocks = (M + 15) / 16;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(colstats_kernel, dim3((unsigned)blocks), dim3(256), 0, S_, (const bf16*)x, (long)ld, (long long*)stats, (long)ld_stats,
                       (long)M, C);
    EP24_LAUNCH_CHECK("ep24_colstats");
    return EP24_OK;
}

extern "C" int ep24_stats_gather(const int64_t* src, int64_t ld_src, int64_t* dst, int C, int reps, void* stream) {
    EP24_REQUIRE(src && dst && C > 0 && reps > 0 && ld_src >= C, EP24_E_ARG, "stats_gather: bad arguments");
    hipLaunchKernelGGL(stats_gather_kernel, dim3(cap_grid((long)reps * 2 * C)), dim3(256), 0, S_, (const long long*)src, (long)ld_src,
                       (long long*)dst, C, reps);
    EP24_LAUNCH_CHECK("ep24_stats_gather");
    return EP24_OK;
}

extern "C" int ep24_avgpool2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, int B, int H, int W, int C, void* stream) {
    EP24_REQUIRE(x && y && C % 8 == 0 && ld_x % 8 == 0 && ld_y % 8 == 0 && H % 2 == 0 && W % 2 == 0, EP24_E_ARG, "avgpool2_fwd: bad arguments");
    hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3(cap_grid((long)B * (H / 2) * (W / 2) * (C / 8))), dim3(256), 0, S_, (const bf16*)x, (long)ld_x,
                       (bf16*)y, (long)ld_y, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_avgpool2_fwd");
    return EP24_OK;
}

extern "C" int ep24_avgpool2_bwd(const void* dy, int64_t ld_dy, void* dx, int64_t ld_dx, int accumulate, int B, int H, int W, int C,
                                 void* stream) {
    EP24_REQUIRE(dy && dx && C % 8 == 0 && ld_dy % 8 == 0 && ld_dx % 8 == 0 && H % 2 == 0 && W % 2 == 0, EP24_E_ARG, "avgpool2_bwd: bad arguments");
    hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3(cap_grid((long)B * H * W * (C / 8))), dim3(256), 0, S_, (const bf16*)dy, (long)ld_dy, (bf16*)dx,
                       (long)ld_dx, accumulate, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_avgpool2_bwd");
    return EP24_OK;
}

extern "C" int ep24_chanscale(void* x, int64_t ld, const float* keep, int B, int64_t HW, int C, void* stream) {
    EP24_REQUIRE(x && keep && C % 8 == 0 && ld % 8 == 0 && B > 0 && HW > 0, EP24_E_ARG, "chanscale: bad arguments");
    hipLaunchKernelGGL(chanscale_kernel, dim3(cap_grid((long)B * HW * (C / 8))), dim3(256), 0, S_, (bf16*)x, (long)ld, keep, B, (long)HW, C);
    EP24_LAUNCH_CHECK("ep24_chanscale");
    return EP24_OK;
}

extern "C" int ep24_im2col_bf16(const float* images, void* rows, int64_t ld, int B, int C, int H, int W, int k, int stride, int pad,
                                void* stream) {
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    EP24_REQUIRE(images && rows && ld % 8 == 0 && ld >= (int64_t)k * k * C && OH > 0 && OW > 0, EP24_E_ARG, "im2col_bf16: bad arguments");
    hipLaunchKernelGGL(im2col_kernel, dim3(cap_grid((long)B * OH * OW * (ld / 8))), dim3(256), 0, S_, images, (bf16*)rows, (long)ld, B, C,
                       H, W, k, stride, pad, OH, OW);
    EP24_LAUNCH_CHECK("ep24_im2col_bf16");
    return EP24_OK;
}

extern "C" int ep24_relu_fwd(void* y, int64_t ld, int64_t M, int C, void* stream) {
    EP24_REQUIRE(y && C % 8 == 0 && ld % 8 == 0 && M > 0, EP24_E_ARG, "relu_fwd: bad arguments");
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(cap_grid(M * (C / 8))), dim3(256), 0, S_, (bf16*)y, (long)ld, (long)M, C);
    EP24_LAUNCH_CHECK("ep24_relu_fwd");
    return EP24_OK;
}

extern "C" int ep24_relu_bwd(void* dy, int64_t ld_dy, const void* y, int64_t ld_y, int64_t M, int C, void* stream) {
    EP24_REQUIRE(dy && y && C % 8 == 0 && ld_dy % 8 == 0 && ld_y % 8 == 0 && M > 0, EP24_E_ARG, "relu_bwd: bad arguments");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(cap_grid(M * (C / 8))), dim3(256), 0, S_, (bf16*)dy, (long)ld_dy, (const bf16*)y, (long)ld_y,
                       (long)M, C);
    EP24_LAUNCH_CHECK("ep24_relu_bwd");
    return EP24_OK;
}

extern "C" int ep24_maxpool3s2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, uint8_t* idx, int B, int H, int W, int C,
                                   void* stream) {
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    EP24_REQUIRE(x && y && idx && C % 8 == 0 && ld_x % 8 == 0 && ld_y % 8 == 0, EP24_E_ARG, "maxpool3s2_fwd: bad arguments");
    hipLaunchKernelGGL(maxpool3s2_fwd_kernel, dim3(cap_grid((long)B * OH * OW * (C / 8))), dim3(256), 0, S_, (const bf16*)x, (long)ld_x,
                       (bf16*)y, (long)ld_y, idx, B, H, W, C, OH, OW);
    EP24_LAUNCH_CHECK("ep24_maxpool3s2_fwd");
    return EP24_OK;
}

extern "C" int ep24_maxpool3s2_bwd(const void* dy, int64_t ld_dy, const uint8_t* idx, void* dx, int64_t ld_dx, int accumulate, int B,
                                   int H, int W, int C, void* stream) {
    const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
    EP24_REQUIRE(dy && idx && dx && C % 8 == 0 && ld_dy % 8 == 0 && ld_dx % 8 == 0, EP24_E_ARG, "maxpool3s2_bwd: bad arguments");
    hipLaunchKernelGGL(maxpool3s2_bwd_kernel, dim3(cap_grid((long)B * H * W * (C / 8))), dim3(256), 0, S_, (const bf16*)dy, (long)ld_dy, idx,
                       (bf16*)dx, (long)ld_dx, accumulate, B, H, W, C, OH, OW);
    EP24_LAUNCH_CHECK("ep24_maxpool3s2_bwd");
    return EP24_OK;
}
