// ep24 - OpenCV INTER_LINEAR for uint8 in its fixed-point form, shared by the sector warp and the input pipeline.
// 11-bit coefficients (saturate_cast<short>(w * 2048), round to nearest even), pixel-centre mapping
// f = (d + 0.5) * scale - 0.5 with border clamping, vertical pass (((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2) >> 2.
#pragma once
#include "common.h"

// scale is OpenCV's double 1 / (dsize / ssize) (cv::hal::resize: inv_scale = (double)dsize/ssize, scale = 1./inv_scale;
// the coefficient loop rounds (d + 0.5)*scale - 0.5 to float)
__device__ __forceinline__ void lin_coef(int d, double scale, int ssize, int& s0, int& s1, int& a0, int& a1) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
    s0 = s;
    s1 = s + 1 < ssize ? s + 1 : ssize - 1;
    a0 = (int)rintf((1.f - f) * 2048.f);
    a1 = (int)rintf(f * 2048.f);
}

__device__ __forceinline__ int lin_mix_u8(int p00, int p01, int p10, int p11, int ax0, int ax1, int by0, int by1) {
    const int h0 = p00 * ax0 + p01 * ax1;
    const int h1 = p10 * ax0 + p11 * ax1;
    const int v = (((by0 * (h0 >> 4)) >> 16) + ((by1 * (h1 >> 4)) >> 16) + 2) >> 2;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}
