// ep24 - fisheye sector warp (yolox/demo_featuremap.py:244-328, Image_Distortion.sector_distort).
//
// The reference forward-scatters a [T,13200] resized image onto an annular sector with numpy fancy indexing
// (11.9 M writes, duplicates resolved by "last writer wins" in C iteration order: angle-major, radius-minor).
// MI355X form: the destination of every (angle a, radius r) pair is a pure integer function of two 1-D
// tables (cos/sin of 13200 angles, T radii) that the host mirror computes exactly as the reference does, so
//   1. sector_map   : one thread per (a,r) pair, atomicMax of the iteration index a*T+r into an int32 canvas
//                     (8 MB, L2/MALL resident) -> the winner of the scatter, bit-exact;  cached per (Theta,T)
//   2. sector_gather: one thread per output pixel reads the winner and fetches the source texel (or 114).
// Per image only step 2 runs: 4 B map + 3 B texel read + 3 B write per pixel - HBM bound, no MFMA.
#include "common.h"
#include "resize.h"

namespace {

__global__ __launch_bounds__(256) void sector_map_kernel(const double* cos_tab, const double* sin_tab, int n_ang,
                                                         const double* rho, int T, int canvas_w, int canvas_h, int* winner) {
    const long total = (long)n_ang * T;
    const double half_w = (double)canvas_w / 2.0;                 // draw_temp_w/2 is a python float
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int a = (int)(i / T), r = (int)(i - (long)a * T);
        const double rr = rho[r];
        const int px = (int)(cos_tab[a] * rr);                    // astype(int16): truncation toward zero
        const int py = (int)(sin_tab[a] * rr);
        double xf = (double)px + half_w - 1.0;                    // (new_p + w/2) - 1, then clip [0, w]
        xf = xf < 0.0 ? 0.0 : (xf > (double)canvas_w ? (double)canvas_w : xf);
        const int X = (int)xf;
        int Y = (canvas_h - py) - 1;                              // integer arithmetic, clip [0, h]
        Y = Y < 0 ? 0 : (Y > canvas_h ? canvas_h : Y);
        if (X < canvas_w && Y < canvas_h) atomicMax(winner + (long)Y * canvas_w + X, (int)i);
    }
}

__global__ __launch_bounds__(256) void sector_gather_kernel(const uint8_t* src, const int* winner, int canvas_w, int y0,
                                                            int x0, int out_h, int out_w, int T, int n_ang, uint8_t* dst,
                                                            int fill, int* src_index) {
    const long total = (long)out_h * out_w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int oy = (int)(i / out_w), ox = (int)(i - (long)oy * out_w);
        const int key = winner[(long)(y0 + oy) * canvas_w + x0 + ox];
        uint8_t c0 = (uint8_t)fill, c1 = (uint8_t)fill, c2 = (uint8_t)fill;
        int flat = -1;
        if (key >= 0) {
            const int a = key / T, r = key - a * T;
            flat = (T - 1 - r) * n_ang + (n_ang - 1 - a);         // img_resize[ptx[:, ::-1], pty[::-1, :]]
            if (dst) {                                            // index-only calls pass no source image
                const uint8_t* s = src + (long)flat * 3;
                c0 = s[0]; c1 = s[1]; c2 = s[2];
            }
        }
        if (dst) { dst[i * 3 + 0] = c0; dst[i * 3 + 1] = c1; dst[i * 3 + 2] = c2; }
        if (src_index) src_index[i] = flat;
    }
}

__global__ __launch_bounds__(256) void mask_bbox_kernel(const uint8_t* mask3, int out_h, int out_w, int* box /*xmin,ymin,xmax,ymax*/) {
    const long total = (long)out_h * out_w;
    int xmin = 1 << 30, ymin = 1 << 30, xmax = -1, ymax = -1;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        if (mask3[i * 3] != 0) {
            const int y = (int)(i / out_w), x = (int)(i - (long)y * out_w);
            xmin = min(xmin, x); ymin = min(ymin, y); xmax = max(xmax, x); ymax = max(ymax, y);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        xmin = min(xmin, __shfl_xor(xmin, o, 64)); ymin = min(ymin, __shfl_xor(ymin, o, 64));
        xmax = max(xmax, __shfl_xor(xmax, o, 64)); ymax = max(ymax, __shfl_xor(ymax, o, 64));
    }
    if ((threadIdx.x & 63) == 0 && xmax >= 0) {
        atomicMin(box + 0, xmin); atomicMin(box + 1, ymin); atomicMax(box + 2, xmax); atomicMax(box + 3, ymax);
    }
}

}  // namespace

extern "C" int ep24_sector_map(const double* cos_tab, const double* sin_tab, int n_ang, const double* rho, int T, int canvas_w,
                               int canvas_h, int32_t* winner, void* stream) {
    EP24_REQUIRE(cos_tab && sin_tab && rho && winner && n_ang > 0 && T > 0 && canvas_w > 0 && canvas_h > 0, EP24_E_ARG,
                 "sector_map: bad arguments");
    EP24_REQUIRE((long)n_ang * T < (1L << 31), EP24_E_ARG, "sector_map: pair index overflows int32");
    hipLaunchKernelGGL(sector_map_kernel, dim3(2048), dim3(256), 0, (hipStream_t)stream, cos_tab, sin_tab, n_ang, rho, T, canvas_w,
                       canvas_h, winner);
    EP24_LAUNCH_CHECK("ep24_sector_map");
    return EP24_OK;
}

extern "C" int ep24_sector_gather(const uint8_t* src, const int32_t* winner, int canvas_w, int y0, int x0, int out_h, int out_w,
                                  int T, int n_ang, uint8_t* dst, int fill, int32_t* src_index, void* stream) {
    EP24_REQUIRE(winner && (dst || src_index) && out_h > 0 && out_w > 0 && T > 0 && n_ang > 0, EP24_E_ARG,
                 "sector_gather: bad arguments");
    EP24_REQUIRE(!dst || src, EP24_E_ARG, "sector_gather: dst without src");
    long blocks = ((long)out_h * out_w + 255) / 256;
    hipLaunchKernelGGL(sector_gather_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 0, (hipStream_t)stream, src,
                       winner, canvas_w, y0, x0, out_h, out_w, T, n_ang, dst, fill, src_index);
    EP24_LAUNCH_CHECK("ep24_sector_gather");
    return EP24_OK;
}

extern "C" int ep24_mask_bbox(const uint8_t* mask3, int out_h, int out_w, int32_t* box, void* stream) {
    EP24_REQUIRE(mask3 && box && out_h > 0 && out_w > 0, EP24_E_ARG, "mask_bbox: bad arguments");
    hipLaunchKernelGGL(mask_bbox_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, mask3, out_h, out_w, box);
    EP24_LAUNCH_CHECK("ep24_mask_bbox");
    return EP24_OK;
}

// ------------------------------------------------------------------------------------------ bilinear resize
// uint8 bilinear resize with OpenCV's INTER_LINEAR fixed-point arithmetic (the reference calls
// cv2.resize(image, (13200, T)), demo_featuremap.py:285): 11-bit horizontal/vertical coefficients
// (saturate_cast<short>(w * 2048), round to nearest even), pixel-centre mapping f = (d + 0.5) * scale - 0.5 with
// border clamping, vertical pass ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.  cv2 is not installed in
// this image, so this step is checked against a numpy restatement of the same published arithmetic only.
namespace {
__global__ __launch_bounds__(256) void resize_linear_u8_kernel(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw,
                                                               double scale_x, double scale_y) {
    const long total = (long)dh * dw;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int dy = (int)(i / dw), dx = (int)(i - (long)dy * dw);
        int x0, x1, ax0, ax1, y0, y1, by0, by1;
        lin_coef(dx, scale_x, sw, x0, x1, ax0, ax1);
        lin_coef(dy, scale_y, sh, y0, y1, by0, by1);
        const uint8_t* r0 = src + (long)y0 * sw * 3;
        const uint8_t* r1 = src + (long)y1 * sw * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = r0[x0 * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;
            const int h1 = r1[x0 * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
            const int v = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
            dst[i * 3 + c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}
}  // namespace

// Gather and resize in one pass: the texel that output pixel i takes from the [T, n_ang] resized image is computed on
// the spot from the source image with the same fixed-point bilinear arithmetic, so the 35 MB intermediate (T x 13200 x 3)
// is never written or read: 114 -> ~15 us per image + mask at 1280x1280.
namespace {
__global__ __launch_bounds__(256) void sector_warp_kernel(const uint8_t* src, int sh, int sw, const int* winner, int canvas_w,
                                                          int y0, int x0, int out_h, int out_w, int T, int n_ang, uint8_t* dst,
                                                          int fill, double scale_x, double scale_y) {
    // one output pixel per thread (four per thread with packed dword stores measured 1.4x slower: the gathers serialise)
    const long total = (long)out_h * out_w;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int oy = (int)(i / out_w), ox = (int)(i - (long)oy * out_w);
        const int key = winner[(long)(y0 + oy) * canvas_w + x0 + ox];
        uint8_t c[3] = {(uint8_t)fill, (uint8_t)fill, (uint8_t)fill};
        if (key >= 0) {
            const int a = key / T, r = key - a * T;
            const int dy = T - 1 - r, dx = n_ang - 1 - a;                 // img_resize[ptx[:, ::-1], pty[::-1, :]]
            int xa, xb, axa, axb, ya, yb, bya, byb;
            lin_coef(dx, scale_x, sw, xa, xb, axa, axb);
            lin_coef(dy, scale_y, sh, ya, yb, bya, byb);
            const uint8_t* r0 = src + (long)ya * sw * 3;
            const uint8_t* r1 = src + (long)yb * sw * 3;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const int h0 = r0[xa * 3 + ch] * axa + r0[xb * 3 + ch] * axb;
                const int h1 = r1[xa * 3 + ch] * axa + r1[xb * 3 + ch] * axb;
                const int v = (((bya * (h0 >> 4)) >> 16) + ((byb * (h1 >> 4)) >> 16) + 2) >> 2;
                c[ch] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
        }
        dst[i * 3 + 0] = c[0]; dst[i * 3 + 1] = c[1]; dst[i * 3 + 2] = c[2];
    }
}
}  // namespace

extern "C" int ep24_sector_warp_u8(const uint8_t* src, int sh, int sw, const int32_t* winner, int canvas_w, int y0, int x0,
                                   int out_h, int out_w, int T, int n_ang, uint8_t* dst, int fill, void* stream) {
    EP24_REQUIRE(src && winner && dst && sh > 0 && sw > 0 && out_h > 0 && out_w > 0 && T > 0 && n_ang > 0, EP24_E_ARG,
                 "sector_warp: bad arguments");
    long blocks = ((long)out_h * out_w + 255) / 256;
    hipLaunchKernelGGL(sector_warp_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, src, sh,
                       sw, winner, canvas_w, y0, x0, out_h, out_w, T, n_ang, dst, fill, 1.0 / ((double)n_ang / sw),
                       1.0 / ((double)T / sh));
    EP24_LAUNCH_CHECK("ep24_sector_warp_u8");
    return EP24_OK;
}

extern "C" int ep24_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, void* stream) {
    EP24_REQUIRE(src && dst && sh > 0 && sw > 0 && dh > 0 && dw > 0, EP24_E_ARG, "resize_linear_u8: bad arguments");
    long blocks = ((long)dh * dw + 255) / 256;
    hipLaunchKernelGGL(resize_linear_u8_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, (hipStream_t)stream, src,
                       sh, sw, dst, dh, dw, 1.0 / ((double)dw / sw), 1.0 / ((double)dh / sh));
    EP24_LAUNCH_CHECK("ep24_resize_linear_u8");
    return EP24_OK;
}
