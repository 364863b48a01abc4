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

}  // namespace

#define S_ (hipStream_t) stream

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
