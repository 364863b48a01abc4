// ep24 - fp32 PARITY MODE of the conv graph.  The reference trains in fp32 with no AMP (yolox_24p/train_24p.py:86-104,
// models/network_blocks.py:50-51); the product path stores activations in bf16.  These kernels run the SAME launch plan
// (ep24.engine with dtype=float32: same buffers, concat slots, residual aliasing, accumulate flags, flat parameters) on
// fp32 activations so that "images in -> SimOTA indices / loss / gradients out" can be compared with the CPU oracle at
// fp32 accuracy.  Speed is irrelevant here: one thread per output element, reductions carried in double (so the
// results are the correctly rounded fp32 values up to the last bit or two, whatever order the reference summed in).
// Every entry point takes NHWC fp32 tensors with an explicit row stride (in elements), like its bf16 counterpart.
#include "common.h"

namespace {

constexpr int NTH = 256;
inline unsigned grid_for(long n) { long b = (n + NTH - 1) / NTH; return (unsigned)(b < 1 ? 1 : (b > 1048576 ? 1048576 : b)); }

__device__ __forceinline__ float act_fwd_f32(float u, int act) {
    if (act == 1) return u / (1.f + expf(-u));                // x * sigmoid(x), full-precision expf and division
    if (act == 3) return u > 0.f ? u : 0.1f * u;
    return act == 2 ? fmaxf(u, 0.f) : u;
}
__device__ __forceinline__ float act_grad_f32(float u, int act) {
    if (!act) return 1.f;
    if (act == 2) return u > 0.f ? 1.f : 0.f;
    if (act == 3) return u > 0.f ? 1.f : 0.1f;
    const float s = 1.f / (1.f + expf(-u));
    return s * (1.f + u * (1.f - s));
}

// Focus + 3x3 im2col of the stem (network_blocks.py:188-210 + the first conv): row = output pixel of the half-size map,
// column = tap * 12 + (x parity * 2 + y parity) * 3 + channel, 108 real columns, the rest zero.
__global__ void stem_pack_f32_kernel(const float* img, float* rows, long ld, int B, int IH, int IW) {
    const int FH = IH / 2, FW = IW / 2;
    const long total = (long)B * FH * FW * ld;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int col = (int)(i % ld);
        const long pix = i / ld;
        float v = 0.f;
        if (col < 108) {
            const int tap = col / 12, j = col % 12, patch = j / 3, ch = j % 3;
            const int n = (int)(pix / ((long)FH * FW));
            const int rem = (int)(pix - (long)n * FH * FW);
            const int oy = rem / FW, ox = rem % FW;
            const int fy = oy + tap / 3 - 1, fx = ox + tap % 3 - 1;
            if (fy >= 0 && fy < FH && fx >= 0 && fx < FW)
                v = img[(((long)n * 3 + ch) * IH + 2 * fy + (patch & 1)) * IW + 2 * fx + (patch >> 1)];
        }
        rows[i] = v;
    }
}

// Direct convolution, forward (transposed = 0): y[n,oh,ow,co] = bias[co] + sum_{kh,kw,ci} x[n,oh*s+kh-p,ow*s+kw-p,ci] * w[co][t][ci]
// and input gradient (transposed = 1): dx[n,h,w,ci] (+)= sum_{kh,kw,co} dy[n,(h+p-kh)/s,(w+p-kw)/s,co] * w[co][t][ci].
// w element (co, t, ci) at w[co * wco + t * wt + ci]; output pixel (n, y, x) lands on row n*dbs + dp0 + y*OW + x.
__global__ void conv_f32_kernel(const float* x, long ld_x, const float* w, long wco, long wt, float* y, long ld_y, long dbs, long dp0,
                                const float* bias, int accumulate, int B, int H, int W, int Cin, int Cout, int k, int s, int transposed) {
    const int p = (k - 1) / 2;
    const int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    const int GH = transposed ? H : OH, GW = transposed ? W : OW;          // output grid
    const int N = transposed ? Cin : Cout, K = transposed ? Cout : Cin;
    const long total = (long)B * GH * GW * N;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int c = (int)(i % N);
        const long pix = i / N;
        const int n = (int)(pix / ((long)GH * GW));
        const int rem = (int)(pix - (long)n * GH * GW);
        const int gy = rem / GW, gx = rem % GW;
        double acc = 0.0;
        for (int kh = 0; kh < k; ++kh)
            for (int kw = 0; kw < k; ++kw) {
                int sy, sx;
                if (!transposed) { sy = gy * s + kh - p; sx = gx * s + kw - p; if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue; }
                else {
                    const int ny = gy + p - kh, nx = gx + p - kw;
                    if (ny < 0 || nx < 0 || ny % s || nx % s) continue;
                    sy = ny / s; sx = nx / s;
                    if (sy >= OH || sx >= OW) continue;
                }
                const int SHh = transposed ? OH : H, SWw = transposed ? OW : W;
                const float* xs = x + (((long)n * SHh + sy) * SWw + sx) * ld_x;
                const long t = (long)(kh * k + kw) * wt;
                if (!transposed) { const float* wr = w + (long)c * wco + t; for (int q = 0; q < K; ++q) acc += (double)xs[q] * (double)wr[q]; }
                else { const float* wr = w + t + c; for (int q = 0; q < K; ++q) acc += (double)xs[q] * (double)wr[(long)q * wco]; }
            }
        float v = (float)acc + ((bias && !transposed) ? bias[c] : 0.f);
        float* d = y + ((long)n * dbs + dp0 + (long)gy * GW + gx) * ld_y + c;
        *d = accumulate ? *d + v : v;
    }
}

// Weight gradient: dw[co][t][ci] += sum over output pixels dy[pix][co] * x[pix @ t][ci]; one thread per weight element
// (ci fastest: coalesced reads of x), pixels in a fixed order: deterministic.
__global__ void conv_wgrad_f32_kernel(const float* x, long ld_x, const float* dy, long ld_dy, float* dw, long wco, long wt,
                                      int B, int H, int W, int Cin, int Cout, int k, int s) {
    const int p = (k - 1) / 2;
    const int OH = (H + 2 * p - k) / s + 1, OW = (W + 2 * p - k) / s + 1;
    const long total = (long)Cout * k * k * Cin;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int ci = (int)(i % Cin);
        const int t = (int)((i / Cin) % (k * k));
        const int co = (int)(i / ((long)Cin * k * k));
        const int kh = t / k, kw = t % k;
        double acc = 0.0;
        for (int n = 0; n < B; ++n)
            for (int oh = 0; oh < OH; ++oh) {
                const int iy = oh * s + kh - p;
                if (iy < 0 || iy >= H) continue;
                for (int ow = 0; ow < OW; ++ow) {
                    const int ix = ow * s + kw - p;
                    if (ix < 0 || ix >= W) continue;
                    acc += (double)dy[(((long)n * OH + oh) * OW + ow) * ld_dy + co] * (double)x[(((long)n * H + iy) * W + ix) * ld_x + ci];
                }
            }
        dw[(long)co * wco + (long)t * wt + ci] += (float)acc;
    }
}

// ---- BatchNorm (training mode), one workgroup per channel for the reductions
__device__ __forceinline__ double block_sum(double v, double* red) {
    red[threadIdx.x] = v;
    __syncthreads();
    for (int o = NTH / 2; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const double r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(NTH) void bn_stats_f32_kernel(const float* z, long ld, long M, int C, float* save, float* rmean, float* rvar,
                                                           long* nbt, float eps, float momentum) {
    __shared__ double red[NTH];
    const int c = blockIdx.x;
    double s = 0.0;
    for (long m = threadIdx.x; m < M; m += NTH) s += (double)z[m * ld + c];
    const double mean = block_sum(s, red) / (double)M;
    double q = 0.0;
    for (long m = threadIdx.x; m < M; m += NTH) { const double d = (double)z[m * ld + c] - mean; q += d * d; }
    const double var = block_sum(q, red) / (double)M;            // biased, two-pass
    if (threadIdx.x == 0) {
        save[c] = (float)mean;
        save[C + c] = (float)(1.0 / sqrt(var + (double)eps));
        if (rmean) {
            const double unb = M > 1 ? var * (double)M / (double)(M - 1) : var;
            rmean[c] = (1.f - momentum) * rmean[c] + momentum * (float)mean;
            rvar[c] = (1.f - momentum) * rvar[c] + momentum * (float)unb;
        }
        if (c == 0 && nbt) *nbt += 1;
    }
}

__global__ void bn_act_fwd_f32_kernel(const float* z, long ld_z, const float* save, const float* gamma, const float* beta, float* y,
                                      long ld_y, const float* res, long ld_res, long M, int C, int act) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int c = (int)(i % C);
        const long m = i / C;
        const float u = (z[m * ld_z + c] - save[c]) * save[C + c] * gamma[c] + beta[c];
        y[m * ld_y + c] = act_fwd_f32(u, act) + (res ? res[m * ld_res + c] : 0.f);
    }
}

__global__ __launch_bounds__(NTH) void bn_bwd_reduce_f32_kernel(const float* dy, long ld_dy, const float* z, long ld_z, const float* save,
                                                                const float* gamma, const float* beta, double* sums, long M, int C, int act) {
    __shared__ double red[NTH];
    const int c = blockIdx.x;
    const float mean = save[c], inv = save[C + c], g = gamma[c], b = beta[c];
    double sg = 0.0, sb = 0.0;
    for (long m = threadIdx.x; m < M; m += NTH) {
        const float zh = (z[m * ld_z + c] - mean) * inv;
        const float du = dy[m * ld_dy + c] * act_grad_f32(zh * g + b, act);
        sb += (double)du;
        sg += (double)du * (double)zh;
    }
    sg = block_sum(sg, red);
    sb = block_sum(sb, red);
    if (threadIdx.x == 0) { sums[c] = sg; sums[C + c] = sb; }
}

__global__ void bn_bwd_apply_f32_kernel(const float* dy, long ld_dy, const float* z, long ld_z, const float* save, const float* gamma,
                                        const float* beta, const double* sums, float* ggrad, float* bgrad, float* dz, long ld_dz, long M,
                                        int C, int act) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int c = (int)(i % C);
        const long m = i / C;
        const float mean = save[c], inv = save[C + c], g = gamma[c];
        const float zh = (z[m * ld_z + c] - mean) * inv;
        const float du = dy[m * ld_dy + c] * act_grad_f32(zh * g + beta[c], act);
        const double mg = sums[c] / (double)M, mb = sums[C + c] / (double)M;
        dz[m * ld_dz + c] = (float)((double)g * (double)inv * ((double)du - mb - (double)zh * mg));
        if (m == 0 && ggrad) { ggrad[c] += (float)sums[c]; bgrad[c] += (float)sums[C + c]; }
    }
}

// ---- SPP pools (5, 9, 13; stride 1; -inf padding), first maximum in row-major scan order like ATen
__global__ void spp_fwd_f32_kernel(const float* x, long ld_x, float* y5, float* y9, float* y13, long ld_y, int* idx, int B, int H, int W, int C) {
    const long total = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int n = (int)(pix / ((long)H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem % W;
        float* outs[3] = {y5, y9, y13};
        for (int k = 0; k < 3; ++k) {
            const int r = 2 + 2 * k;
            float best = -INFINITY;
            int arg = -1;
            for (int yy = py - r; yy <= py + r; ++yy) {
                if (yy < 0 || yy >= H) continue;
                for (int xx = px - r; xx <= px + r; ++xx) {
                    if (xx < 0 || xx >= W) continue;
                    const float v = x[(((long)n * H + yy) * W + xx) * ld_x + c];
                    if (v > best || arg < 0) { best = v; arg = yy * W + xx; }
                }
            }
            outs[k][pix * ld_y + c] = best;
            idx[(long)k * total + i] = arg;
        }
    }
}

__global__ void spp_bwd_f32_kernel(const float* dy5, const float* dy9, const float* dy13, long ld_dy, const int* idx, float* dx, long ld_dx,
                                   int accumulate, int B, int H, int W, int C) {
    const long total = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int n = (int)(pix / ((long)H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem % W;
        const float* dys[3] = {dy5, dy9, dy13};
        double acc = 0.0;
        for (int k = 0; k < 3; ++k) {
            const int r = 2 + 2 * k;
            for (int yy = py - r; yy <= py + r; ++yy) {
                if (yy < 0 || yy >= H) continue;
                for (int xx = px - r; xx <= px + r; ++xx) {
                    if (xx < 0 || xx >= W) continue;
                    const long q = ((long)n * H + yy) * W + xx;                 // window centred at q contains this pixel
                    if (idx[(long)k * total + q * C + c] == rem) acc += (double)dys[k][q * ld_dy + c];
                }
            }
        }
        float* d = dx + pix * ld_dx + c;
        *d = accumulate ? *d + (float)acc : (float)acc;
    }
}

__global__ void upsample2_fwd_f32_kernel(const float* x, long ld_x, float* y, long ld_y, int B, int H, int W, int C) {
    const long total = (long)B * 2 * H * 2 * W * C;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int n = (int)(pix / (4L * H * W));
        const int rem = (int)(pix - (long)n * 4 * H * W);
        const int oy = rem / (2 * W), ox = rem % (2 * W);
        y[pix * ld_y + c] = x[(((long)n * H + oy / 2) * W + ox / 2) * ld_x + c];
    }
}

__global__ void upsample2_bwd_f32_kernel(const float* dy, long ld_dy, float* dx, long ld_dx, int accumulate, int B, int H, int W, int C) {
    const long total = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int c = (int)(i % C);
        const long pix = i / C;
        const int n = (int)(pix / ((long)H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem % W;
        const float* s = dy + (((long)n * 2 * H + 2 * py) * 2 * W + 2 * px) * ld_dy + c;
        const float v = (s[0] + s[ld_dy]) + (s[2L * W * ld_dy] + s[(2L * W + 1) * ld_dy]);
        float* d = dx + pix * ld_dx + c;
        *d = accumulate ? *d + v : v;
    }
}

__global__ void rows_copy_f32_kernel(const float* src, long ld_src, float* dst, long ld_dst, int accumulate, long M, int C) {
    const long total = M * C;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int c = (int)(i % C);
        const long m = i / C;
        float* d = dst + m * ld_dst + c;
        *d = accumulate ? *d + src[m * ld_src + c] : src[m * ld_src + c];
    }
}

// gradient of the train-mode decode (yolo_head_24p.py:212-237): d_regobj [cells, 32] = (dx*s, dy*s, dr_k * r_k, dobj, 0..), d_cls [cells, ldc]
__global__ void decode_bwd_f32_kernel(const float* dout, const float* out, float* d_regobj, float* d_cls, int B, int A, int a0, int H, int W,
                                      float s, int ncols, const float* d_origin) {
    const int C = ncols - 27, ldc = (C + 7) & ~7, cols = 32 + ldc;
    const long total = (long)B * H * W * cols;
    for (long i = (long)blockIdx.x * NTH + threadIdx.x; i < total; i += (long)gridDim.x * NTH) {
        const int col = (int)(i % cols);
        const long cell = i / cols;
        const int n = (int)(cell / ((long)H * W));
        const int hw = (int)(cell - (long)n * H * W);
        const long row = ((long)n * A + a0 + hw) * ncols;
        if (col < 32) {
            float g = 0.f;
            if (col < 2) g = dout[row + col] * s;
            else if (col < 26) g = dout[row + col] * out[row + col];
            else if (col == 26) g = dout[row + 26];
            if (d_origin && col < 26) g += d_origin[((long)n * A + a0 + hw) * 26 + col];
            d_regobj[cell * 32 + col] = g;
        } else {
            const int c = col - 32;
            d_cls[cell * ldc + c] = c < C ? dout[row + 27 + c] : 0.f;
        }
    }
}

__global__ __launch_bounds__(NTH) void colsum_f32_kernel(const float* g, long ld, float* db, long M, int N) {
    __shared__ double red[NTH];
    const int c = blockIdx.x;
    double s = 0.0;
    for (long m = threadIdx.x; m < M; m += NTH) s += (double)g[m * ld + c];
    s = block_sum(s, red);
    if (threadIdx.x == 0) db[c] += (float)s;
}

}  // namespace

#define S_ (hipStream_t) stream
#define F32_LAUNCH(kern, n, ...) hipLaunchKernelGGL(kern, dim3(grid_for(n)), dim3(NTH), 0, S_, __VA_ARGS__)

extern "C" int ep24_f32_stem_pack(const float* images, float* rows, int64_t ld, int B, int H, int W, void* stream) {
    EP24_REQUIRE(images && rows && H % 2 == 0 && W % 2 == 0 && B > 0 && ld >= 108, EP24_E_ARG, "f32_stem_pack: bad arguments");
    F32_LAUNCH(stem_pack_f32_kernel, (long)B * (H / 2) * (W / 2) * ld, images, rows, ld, B, H, W);
    EP24_LAUNCH_CHECK("ep24_f32_stem_pack");
    return EP24_OK;
}

extern "C" int ep24_f32_conv(const float* x, int64_t ld_x, const float* w, int64_t w_co_stride, int64_t w_tap_stride, float* y, int64_t ld_y,
                             int64_t y_batch_rows, int64_t y_row0, const float* bias, int accumulate, int B, int H, int W, int Cin, int Cout,
                             int ksize, int stride, int transposed, void* stream) {
    EP24_REQUIRE(x && w && y && B > 0 && Cin > 0 && Cout > 0, EP24_E_ARG, "f32_conv: bad arguments");
    EP24_REQUIRE((ksize == 1 || ksize == 3 || ksize == 7) && (stride == 1 || stride == 2), EP24_E_UNSUPPORTED, "f32_conv: k=%d s=%d", ksize, stride);
    const int p = (ksize - 1) / 2;
    const int OH = (H + 2 * p - ksize) / stride + 1, OW = (W + 2 * p - ksize) / stride + 1;
    const long GH = transposed ? H : OH, GW = transposed ? W : OW;
    const long dbs = y_batch_rows > 0 ? y_batch_rows : GH * GW;
    F32_LAUNCH(conv_f32_kernel, (long)B * GH * GW * (transposed ? Cin : Cout), x, ld_x, w, w_co_stride, w_tap_stride, y, ld_y, dbs, y_row0, bias,
               accumulate, B, H, W, Cin, Cout, ksize, stride, transposed);
    EP24_LAUNCH_CHECK("ep24_f32_conv");
    return EP24_OK;
}

extern "C" int ep24_f32_conv_wgrad(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, float* dw, int64_t w_co_stride,
                                   int64_t w_tap_stride, int B, int H, int W, int Cin, int Cout, int ksize, int stride, void* stream) {
    EP24_REQUIRE(x && dy && dw, EP24_E_ARG, "f32_conv_wgrad: null pointer");
    F32_LAUNCH(conv_wgrad_f32_kernel, (long)Cout * ksize * ksize * Cin, x, ld_x, dy, ld_dy, dw, w_co_stride, w_tap_stride, B, H, W, Cin, Cout,
               ksize, stride);
    EP24_LAUNCH_CHECK("ep24_f32_conv_wgrad");
    return EP24_OK;
}

extern "C" int ep24_f32_bn_act_fwd(const float* z, int64_t ld_z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                   int64_t* num_batches, float* save, float* y, int64_t ld_y, const float* residual, int64_t ld_res, int64_t M,
                                   int C, float eps, float momentum, int act, void* stream) {
    EP24_REQUIRE(z && gamma && beta && save && y && M > 0 && C > 0, EP24_E_ARG, "f32_bn_act_fwd: bad arguments");
    hipLaunchKernelGGL(bn_stats_f32_kernel, dim3(C), dim3(NTH), 0, S_, z, ld_z, (long)M, C, save, running_mean, running_var, (long*)num_batches,
                       eps, momentum);
    F32_LAUNCH(bn_act_fwd_f32_kernel, (long)M * C, z, ld_z, save, gamma, beta, y, ld_y, residual, ld_res, (long)M, C, act);
    EP24_LAUNCH_CHECK("ep24_f32_bn_act_fwd");
    return EP24_OK;
}

extern "C" int ep24_f32_bn_act_bwd(const float* dy, int64_t ld_dy, const float* z, int64_t ld_z, const float* save, const float* gamma,
                                   const float* beta, double* sums, float* gamma_grad, float* beta_grad, float* dz, int64_t ld_dz, int64_t M,
                                   int C, int act, void* stream) {
    EP24_REQUIRE(dy && z && save && gamma && beta && sums && dz && M > 0 && C > 0, EP24_E_ARG, "f32_bn_act_bwd: bad arguments");
    hipLaunchKernelGGL(bn_bwd_reduce_f32_kernel, dim3(C), dim3(NTH), 0, S_, dy, ld_dy, z, ld_z, save, gamma, beta, sums, (long)M, C, act);
    F32_LAUNCH(bn_bwd_apply_f32_kernel, (long)M * C, dy, ld_dy, z, ld_z, save, gamma, beta, sums, gamma_grad, beta_grad, dz, ld_dz, (long)M, C, act);
    EP24_LAUNCH_CHECK("ep24_f32_bn_act_bwd");
    return EP24_OK;
}

extern "C" int ep24_f32_spp_fwd(const float* x, int64_t ld_x, float* y5, float* y9, float* y13, int64_t ld_y, int32_t* idx, int B, int H, int W,
                                int C, void* stream) {
    EP24_REQUIRE(x && y5 && y9 && y13 && idx, EP24_E_ARG, "f32_spp_fwd: null pointer");
    F32_LAUNCH(spp_fwd_f32_kernel, (long)B * H * W * C, x, ld_x, y5, y9, y13, ld_y, idx, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_f32_spp_fwd");
    return EP24_OK;
}

extern "C" int ep24_f32_spp_bwd(const float* dy5, const float* dy9, const float* dy13, int64_t ld_dy, const int32_t* idx, float* dx,
                                int64_t ld_dx, int accumulate, int B, int H, int W, int C, void* stream) {
    EP24_REQUIRE(dy5 && dy9 && dy13 && idx && dx, EP24_E_ARG, "f32_spp_bwd: null pointer");
    F32_LAUNCH(spp_bwd_f32_kernel, (long)B * H * W * C, dy5, dy9, dy13, ld_dy, idx, dx, ld_dx, accumulate, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_f32_spp_bwd");
    return EP24_OK;
}

extern "C" int ep24_f32_upsample2_fwd(const float* x, int64_t ld_x, float* y, int64_t ld_y, int B, int H, int W, int C, void* stream) {
    EP24_REQUIRE(x && y, EP24_E_ARG, "f32_upsample2_fwd: null pointer");
    F32_LAUNCH(upsample2_fwd_f32_kernel, (long)B * 4 * H * W * C, x, ld_x, y, ld_y, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_f32_upsample2_fwd");
    return EP24_OK;
}

extern "C" int ep24_f32_upsample2_bwd(const float* dy, int64_t ld_dy, float* dx, int64_t ld_dx, int accumulate, int B, int H, int W, int C,
                                      void* stream) {
    EP24_REQUIRE(dy && dx, EP24_E_ARG, "f32_upsample2_bwd: null pointer");
    F32_LAUNCH(upsample2_bwd_f32_kernel, (long)B * H * W * C, dy, ld_dy, dx, ld_dx, accumulate, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_f32_upsample2_bwd");
    return EP24_OK;
}

extern "C" int ep24_f32_rows_copy(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int accumulate, int64_t M, int C, void* stream) {
    EP24_REQUIRE(src && dst, EP24_E_ARG, "f32_rows_copy: null pointer");
    F32_LAUNCH(rows_copy_f32_kernel, (long)M * C, src, ld_src, dst, ld_dst, accumulate, (long)M, C);
    EP24_LAUNCH_CHECK("ep24_f32_rows_copy");
    return EP24_OK;
}

extern "C" int ep24_f32_head_decode_bwd(const float* dout, const float* out, float* d_regobj, float* d_cls, int B, int A, int a0, int H, int W,
                                        float stride, int ncols, const float* d_origin, void* stream) {
    EP24_REQUIRE(dout && out && d_regobj && d_cls && ncols > 27, EP24_E_ARG, "f32_head_decode_bwd: bad arguments");
    const int ldc = ((ncols - 27) + 7) & ~7;
    F32_LAUNCH(decode_bwd_f32_kernel, (long)B * H * W * (32 + ldc), dout, out, d_regobj, d_cls, B, A, a0, H, W, stride, ncols, d_origin);
    EP24_LAUNCH_CHECK("ep24_f32_head_decode_bwd");
    return EP24_OK;
}

extern "C" int ep24_f32_colsum(const float* g, int64_t ld, float* db, int64_t M, int N, void* stream) {
    EP24_REQUIRE(g && db && N > 0, EP24_E_ARG, "f32_colsum: bad arguments");
    hipLaunchKernelGGL(colsum_f32_kernel, dim3(N), dim3(NTH), 0, S_, g, ld, db, (long)M, N);
    EP24_LAUNCH_CHECK("ep24_f32_colsum");
    return EP24_OK;
}
