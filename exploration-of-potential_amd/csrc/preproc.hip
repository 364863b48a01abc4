// ep24 - input pipeline on the GPU: preproc + TrainTransform of the reference
// (yolox_24p/datasets/data_augment.py:109-174) for a whole batch in two launches.
//
// The reference resizes every image on a CPU worker (cv2.resize INTER_LINEAR to (int(w*r), int(h*r)), r = min(S/h, S/w)),
// pastes it top-left into a 114-filled canvas, transposes to CHW, converts to fp32 and ships 4.9 MB per 640x640 image over
// PCIe.  Here the raw uint8 HWC images travel (0.9 MB for 480x640) and one thread per output pixel does the rest: the
// four source texels of the fixed-point bilinear sample (resize.h), or 114 outside the resized area, written to the three
// fp32 planes of the network input (four pixels of a row per thread, 16-byte streaming stores).  HBM bound: 3 bytes read (L2 serves the neighbours) and 12 written per output pixel.
// Labels: (v * width) * r for x columns, (v * height) * r for y columns in double, rounded to fp32, zero-padded to 50 rows.
#include "common.h"
#include "resize.h"

namespace {

__global__ __launch_bounds__(256) void preproc_u8_kernel(const uint8_t* images, const long long* desc, const double* scales,
                                                         float* out, int S_h, int S_w) {
    const int n = blockIdx.y;
    const long long* d = desc + (long)n * 6;
    const uint8_t* src = images + d[0];
    const int sh = (int)d[1], sw = (int)d[2], rh = (int)d[4], rw = (int)d[5];
    const long ld = d[3];
    const double scale_x = scales[2 * n], scale_y = scales[2 * n + 1];
    const int plane = S_h * S_w;
    float* o = out + (long)n * 3 * plane;
    // four consecutive pixels of a row per thread: three 16-byte stores (S_w % 4 == 0, checked by the launcher)
    const int quads = plane >> 2;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < quads; q += gridDim.x * 256) {
        const int i = q << 2;
        const int y = i / S_w, xb = i - y * S_w;
        f32x4 v0 = {114.f, 114.f, 114.f, 114.f}, v1 = v0, v2 = v0;
        if (y < rh && xb < rw) {
            int y0, y1, by0, by1;
            lin_coef(y, scale_y, sh, y0, y1, by0, by1);
            const uint8_t* r0 = src + (long)y0 * ld;
            const uint8_t* r1 = src + (long)y1 * ld;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int x = xb + j;
                if (x < rw) {
                    int x0, x1, ax0, ax1;
                    lin_coef(x, scale_x, sw, x0, x1, ax0, ax1);
                    v0[j] = (float)lin_mix_u8(r0[x0 * 3 + 0], r0[x1 * 3 + 0], r1[x0 * 3 + 0], r1[x1 * 3 + 0], ax0, ax1, by0, by1);
                    v1[j] = (float)lin_mix_u8(r0[x0 * 3 + 1], r0[x1 * 3 + 1], r1[x0 * 3 + 1], r1[x1 * 3 + 1], ax0, ax1, by0, by1);
                    v2[j] = (float)lin_mix_u8(r0[x0 * 3 + 2], r0[x1 * 3 + 2], r1[x0 * 3 + 2], r1[x1 * 3 + 2], ax0, ax1, by0, by1);
                }
            }
        }
        __builtin_nontemporal_store(v0, reinterpret_cast<f32x4*>(o + i));
        __builtin_nontemporal_store(v1, reinterpret_cast<f32x4*>(o + plane + i));
        __builtin_nontemporal_store(v2, reinterpret_cast<f32x4*>(o + 2 * plane + i));
    }
}

__global__ __launch_bounds__(256) void preproc_labels_kernel(const double* rows, const long long* row_off, const double* whr,
                                                             float* out, int max_labels) {
    const int n = blockIdx.x;
    const long lo = row_off[n];
    const int cnt = (int)min((long long)max_labels, row_off[n + 1] - lo);
    const double w = whr[3 * n], h = whr[3 * n + 1], r = whr[3 * n + 2];
    float* o = out + (long)n * max_labels * 51;
    for (int i = threadIdx.x; i < max_labels * 51; i += 256) {
        const int j = i / 51, c = i - j * 51;
        float v = 0.f;
        if (j < cnt) {
            const double t = rows[(lo + j) * 51 + c];
            v = c == 0 ? (float)t : (float)((t * ((c & 1) ? w : h)) * r);     // columns 1,3,5.. are x, 2,4,6.. are y
        }
        o[i] = v;
    }
}

}  // namespace

extern "C" int ep24_preproc_u8(const uint8_t* images, const int64_t* desc, const double* scales, int n, float* out, int S_h,
                               int S_w, void* stream) {
    if (n == 0) return EP24_OK;
    EP24_REQUIRE(images && desc && scales && out && n > 0 && n <= 65535 && S_h > 0 && S_w > 0, EP24_E_ARG, "preproc_u8: bad arguments");
    EP24_REQUIRE(S_w % 4 == 0 && (uintptr_t)out % 16 == 0, EP24_E_ARG, "preproc_u8: the network input width must be a multiple of 4 (it is a multiple of 32)");
    int bx = (S_h * S_w / 4 + 255) / 256;
    if (bx > 4096) bx = 4096;
    hipLaunchKernelGGL(preproc_u8_kernel, dim3(bx, n), dim3(256), 0, (hipStream_t)stream, images, (const long long*)desc, scales, out,
                       S_h, S_w);
    EP24_LAUNCH_CHECK("ep24_preproc_u8");
    return EP24_OK;
}

extern "C" int ep24_preproc_labels(const double* rows, const int64_t* row_off, const double* whr, int n, float* out,
                                   int max_labels, void* stream) {
    if (n == 0) return EP24_OK;
    EP24_REQUIRE(row_off && whr && out && n > 0 && max_labels > 0, EP24_E_ARG, "preproc_labels: bad arguments");
    hipLaunchKernelGGL(preproc_labels_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, rows, (const long long*)row_off, whr, out,
                       max_labels);
    EP24_LAUNCH_CHECK("ep24_preproc_labels");
    return EP24_OK;
}
