// ep24 - 24-point label generation on the GPU: Polygon_24.rotation_for_24p
// (yolox_24p/datasets/2+24_labels_create.py:61-116) and the convex-hull area of its acceptance filter (:175-180).
//
// The reference paints each of the 24 rays into an image padded by the diagonal, clears the mask pixels, cuts the image
// (plus a one-pixel ring) back out and takes the marked pixel nearest to the box centre: O(diagonal^2) bytes per ray,
// "minutes per image set".  The same pixel set is a list of ceil(L/0.2) samples per ray, so here one workgroup owns one
// (object, ray): every thread walks samples s = tid, tid+256, ..., tests the mask byte under the sample and keeps the
// minimum of (distance, row-major index in the cut window) - the order np.where + np.argmin resolve ties in.  All
// arithmetic that decides a pixel or a distance is the reference's double-precision sequence (the library is built with
// -ffp-contract=off): x = s*0.2; trunc(cos*x); trunc(e + centre + L); sqrt(dx*dx + dy*dy) with the ring's one-pixel
// offset left in, as written.  cos / sin of the 24 angles come from the host (numpy's values).
#include "common.h"

namespace {

__global__ __launch_bounds__(256) void ray24_kernel(const uint8_t* masks, const long long* desc, const double* centre,
                                                    const double* rot, int* out_pts, double* out_r) {
    __shared__ double sd[4];
    __shared__ int si[4];
    const int k = blockIdx.x, obj = blockIdx.y;
    const long long* d = desc + (long)obj * 6;
    const uint8_t* mask = masks + d[0];
    const int H = (int)d[1], W = (int)d[2], L = (int)d[3], ns = (int)d[4];
    const long ld = d[5];
    const double cx = centre[2 * obj], cy = centre[2 * obj + 1];
    const double c = rot[2 * k], s = rot[2 * k + 1];
    const double inf = __longlong_as_double(0x7FF0000000000000LL);
    double best = inf;
    int best_i = 0x7FFFFFFF;
    for (int t = threadIdx.x; t < ns; t += 256) {
        const double x = (double)t * 0.2;                          // np.arange(0, L, 0.2)[t]
        const int ex = (int)(c * x), ey = (int)(s * x);            // astype(int16): truncation (|values| < L < 32768)
        const int px = (int)(((double)ex + cx) + (double)L);       // stored back into the int16 array: truncation again
        const int py = (int)(((double)ey + cy) + (double)L);
        const int ix = px - L, iy = py - L;
        const bool inside = ix >= 0 && ix < W && iy >= 0 && iy < H;
        if (inside && mask[(long)iy * ld + ix] != 0) continue;     // template[mask_x, mask_y] = 0
        const int mx = ix + 1, my = iy + 1;                        // the cut keeps one extra ring of pixels
        if (mx < 0 || mx >= W + 2 || my < 0 || my >= H + 2) continue;
        const double dx = (double)mx - cx, dy = (double)my - cy;
        const double dist = sqrt(dx * dx + dy * dy);
        const int idx = my * (W + 2) + mx;
        if (dist < best || (dist == best && idx < best_i)) { best = dist; best_i = idx; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const double ob = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(best_i, o, 64);
        if (ob < best || (ob == best && oi < best_i)) { best = ob; best_i = oi; }
    }
    if ((threadIdx.x & 63) == 0) { sd[threadIdx.x >> 6] = best; si[threadIdx.x >> 6] = best_i; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w)
            if (sd[w] < best || (sd[w] == best && si[w] < best_i)) { best = sd[w]; best_i = si[w]; }
        int* o = out_pts + ((long)obj * 24 + k) * 2;
        if (best_i == 0x7FFFFFFF) {                                // every sample masked or outside: np.argmin of an empty array
            o[0] = -1; o[1] = -1;
            out_r[(long)obj * 24 + k] = inf;
        } else {
            const int my = best_i / (W + 2), mx = best_i - my * (W + 2);
            o[0] = min(max(mx, 0), W);                             // np.clip(marker_x, 0, img_w)
            o[1] = min(max(my, 0), H);
            out_r[(long)obj * 24 + k] = best;
        }
    }
}

// one thread per object: monotone-chain hull of its 24 integer points, shoelace area (exact in int64)
__global__ __launch_bounds__(64) void hull_area24_kernel(const int* pts, int n, double* area) {
    const int obj = blockIdx.x * 64 + threadIdx.x;
    if (obj >= n) return;
    long long px[24], py[24];
    for (int i = 0; i < 24; ++i) { px[i] = pts[((long)obj * 24 + i) * 2]; py[i] = pts[((long)obj * 24 + i) * 2 + 1]; }
    for (int i = 1; i < 24; ++i) {                                 // insertion sort by (x, y)
        const long long x = px[i], y = py[i];
        int j = i - 1;
        while (j >= 0 && (px[j] > x || (px[j] == x && py[j] > y))) { px[j + 1] = px[j]; py[j + 1] = py[j]; --j; }
        px[j + 1] = x; py[j + 1] = y;
    }
    long long hx[50], hy[50];
    int m = 0;
    for (int pass = 0; pass < 2; ++pass) {
        const int base = m;
        for (int q = 0; q < 24; ++q) {
            const int i = pass ? 23 - q : q;
            if (q > 0 && px[i] == px[pass ? i + 1 : i - 1] && py[i] == py[pass ? i + 1 : i - 1]) continue;   // duplicate point
            while (m - base >= 2 &&
                   (hx[m - 1] - hx[m - 2]) * (py[i] - hy[m - 2]) - (hy[m - 1] - hy[m - 2]) * (px[i] - hx[m - 2]) <= 0)
                --m;
            hx[m] = px[i]; hy[m] = py[i]; ++m;
        }
        --m;                                                       // the last point of a chain starts the other one
    }
    long long s2 = 0;
    for (int i = 0; i < m; ++i) {
        const int j = i + 1 == m ? 0 : i + 1;
        s2 += hx[i] * hy[j] - hx[j] * hy[i];
    }
    area[obj] = (double)(s2 < 0 ? -s2 : s2) / 2.0;
}

}  // namespace

extern "C" int ep24_ray24(const uint8_t* masks, const int64_t* desc, const double* centre, const double* rot, int n,
                          int32_t* out_pts, double* out_r, void* stream) {
    if (n == 0) return EP24_OK;
    EP24_REQUIRE(masks && desc && centre && rot && out_pts && out_r && n > 0 && n <= 65535, EP24_E_ARG, "ray24: bad arguments");
    hipLaunchKernelGGL(ray24_kernel, dim3(24, n), dim3(256), 0, (hipStream_t)stream, masks, (const long long*)desc, centre, rot,
                       out_pts, out_r);
    EP24_LAUNCH_CHECK("ep24_ray24");
    return EP24_OK;
}

extern "C" int ep24_hull_area24(const int32_t* pts, int n, double* area, void* stream) {
    if (n == 0) return EP24_OK;
    EP24_REQUIRE(pts && area && n > 0, EP24_E_ARG, "hull_area24: bad arguments");
    hipLaunchKernelGGL(hull_area24_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, pts, n, area);
    EP24_LAUNCH_CHECK("ep24_hull_area24");
    return EP24_OK;
}
