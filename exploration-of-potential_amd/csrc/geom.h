// ep24 - device functions for the 24-concentric-circle geometry.  Compiled with -ffp-contract=off: the
// reference evaluates every product and sum as a separate fp32 ATen op, so no fused multiply-adds here.
//
// Follows utils.boxes.circle_inter / bboxes_iou (yolox_24p/utils/boxes.py:102-243) and
// IOUloss.circle_inter / forward (yolox_24p/models/losses.py:23-157): operation order, epsilons (1e-8 in the
// cosine denominators, 1e-6 in the IoU denominator), clip to +-0.99, float32 pi, case order
// contained -> disjoint (overrides) -> lens.
#pragma once
#include "common.h"

#define EP24_PI_F 3.14159274101257324f   // float32(np.pi)

struct RayOut {
    float giou;
};

__device__ __forceinline__ float ray_giou(float r1 /*gt*/, float r2 /*pred*/, float d) {
    const float pi = EP24_PI_F;
    const float rmin = fminf(r1, r2), rmax = fmaxf(r1, r2);
    const float rmin2 = rmin * rmin, rmax2 = rmax * rmax, d2 = d * d;
    const bool contained = fabsf(r1 - r2) >= d;
    const bool disjoint = d >= r1 + r2;
    float inter = contained ? pi * rmin2 : 0.0f;
    if (disjoint) inter = 0.0f;
    if (!(contained || disjoint)) {
        // the lens of two properly intersecting circles: the only case that needs the two acos and the sin, so a wave whose
        // pairs are all far apart (most (candidate, GT) pairs of SimOTA) skips them - the value is not used otherwise
        float c1 = (rmin2 + d2 - rmax2) / (2.0f * rmin * d + 1e-8f);
        float c2 = (rmax2 + d2 - rmin2) / (2.0f * rmax * d + 1e-8f);
        c1 = fminf(fmaxf(c1, -0.99f), 0.99f);
        c2 = fminf(fmaxf(c2, -0.99f), 0.99f);
        const float a1 = acosf(c1), a2 = acosf(c2);
        inter = a1 * rmin2 + a2 * rmax2 - rmin * d * sinf(a1);
    }
    const float area1 = pi * (r1 * r1), area2 = pi * (r2 * r2);
    const float uni = area1 + area2 - inter;
    const float iou = inter / (uni + 1e-6f);
    const float cl = contained ? rmax : (r1 + r2 + d) / 2.0f;
    const float cs = pi * (cl * cl);
    const float top = cs - uni;
    return iou - top / cs;
}

// The intersection area alone: what IOUloss.circle_inter (losses.py:23-78) and utils.boxes.circle_inter (boxes.py:102-163) return
// next to the centre distance.  Same expressions and case order as ray_giou's first half.
__device__ __forceinline__ float ray_inter(float r1 /*gt*/, float r2 /*pred*/, float d) {
    const float pi = EP24_PI_F;
    const float rmin = fminf(r1, r2), rmax = fmaxf(r1, r2);
    const float rmin2 = rmin * rmin, rmax2 = rmax * rmax, d2 = d * d;
    const bool contained = fabsf(r1 - r2) >= d;
    const bool disjoint = d >= r1 + r2;
    float inter = contained ? pi * rmin2 : 0.0f;
    if (disjoint) inter = 0.0f;
    if (!(contained || disjoint)) {
        float c1 = (rmin2 + d2 - rmax2) / (2.0f * rmin * d + 1e-8f);
        float c2 = (rmax2 + d2 - rmin2) / (2.0f * rmax * d + 1e-8f);
        c1 = fminf(fmaxf(c1, -0.99f), 0.99f);
        c2 = fminf(fmaxf(c2, -0.99f), 0.99f);
        const float a1 = acosf(c1), a2 = acosf(c2);
        inter = a1 * rmin2 + a2 * rmax2 - rmin * d * sinf(a1);
    }
    return inter;
}

// d(1 - giou)/d(r2), d(1 - giou)/d(d): analytic form of what autograd derives from the expressions above
// (clip passes gradient on the closed interval, min/max ties route to the GT side, selects by mask).
__device__ __forceinline__ void ray_loss_grad(float r1, float r2, float d, float& g_r2, float& g_d) {
    const float pi = EP24_PI_F;
    const bool pmin = r2 < r1, pmax = r2 > r1;          // torch.min/max over stack((gt, pd)): ties pick gt
    const float rmin = fminf(r1, r2), rmax = fmaxf(r1, r2);
    const float drmin = pmin ? 1.f : 0.f, drmax = pmax ? 1.f : 0.f;
    const float rmin2 = rmin * rmin, rmax2 = rmax * rmax, d2 = d * d;
    const bool contained = fabsf(r1 - r2) >= d;
    const bool disjoint = d >= r1 + r2;
    // inter and its partials
    float inter, i_r = 0.f, i_d = 0.f;
    if (disjoint) {
        inter = 0.f;
    } else if (contained) {
        inter = pi * rmin2;
        i_r = 2.f * pi * rmin * drmin;
    } else {
        const float den1 = 2.f * rmin * d + 1e-8f, den2 = 2.f * rmax * d + 1e-8f;
        const float n1 = rmin2 + d2 - rmax2, n2 = rmax2 + d2 - rmin2;
        const float c1r = n1 / den1, c2r = n2 / den2;
        const float c1 = fminf(fmaxf(c1r, -0.99f), 0.99f), c2 = fminf(fmaxf(c2r, -0.99f), 0.99f);
        const bool in1 = c1r >= -0.99f && c1r <= 0.99f, in2 = c2r >= -0.99f && c2r <= 0.99f;
        const float a1 = acosf(c1), a2 = acosf(c2);
        const float s1 = sinf(a1);
        inter = a1 * rmin2 + a2 * rmax2 - rmin * d * s1;
        // dc/dr, dc/dd
        const float dn1_r = 2.f * rmin * drmin - 2.f * rmax * drmax, dn1_d = 2.f * d;
        const float dd1_r = 2.f * d * drmin, dd1_d = 2.f * rmin;
        const float dn2_r = 2.f * rmax * drmax - 2.f * rmin * drmin, dn2_d = 2.f * d;
        const float dd2_r = 2.f * d * drmax, dd2_d = 2.f * rmax;
        const float c1_r = in1 ? (dn1_r * den1 - n1 * dd1_r) / (den1 * den1) : 0.f;
        const float c1_d = in1 ? (dn1_d * den1 - n1 * dd1_d) / (den1 * den1) : 0.f;
        const float c2_r = in2 ? (dn2_r * den2 - n2 * dd2_r) / (den2 * den2) : 0.f;
        const float c2_d = in2 ? (dn2_d * den2 - n2 * dd2_d) / (den2 * den2) : 0.f;
        const float da1 = -1.f / sqrtf(1.f - c1 * c1), da2 = -1.f / sqrtf(1.f - c2 * c2);
        const float a1_r = da1 * c1_r, a1_d = da1 * c1_d, a2_r = da2 * c2_r, a2_d = da2 * c2_d;
        const float cos1 = cosf(a1);
        i_r = a1_r * rmin2 + a1 * 2.f * rmin * drmin + a2_r * rmax2 + a2 * 2.f * rmax * drmax
              - (drmin * d * s1 + rmin * d * cos1 * a1_r);
        i_d = a1_d * rmin2 + a2_d * rmax2 - (rmin * s1 + rmin * d * cos1 * a1_d);
    }
    const float area1 = pi * (r1 * r1), area2 = pi * (r2 * r2);
    const float uni = area1 + area2 - inter;
    const float u_r = 2.f * pi * r2 - i_r, u_d = -i_d;
    const float ue = uni + 1e-6f;
    const float cl = contained ? rmax : (r1 + r2 + d) / 2.0f;
    const float cl_r = contained ? drmax : 0.5f, cl_d = contained ? 0.f : 0.5f;
    const float cs = pi * (cl * cl);
    const float cs_r = 2.f * pi * cl * cl_r, cs_d = 2.f * pi * cl * cl_d;
    // loss = 1 - inter/ue + (cs - uni)/cs
    const float iou_r = (i_r * ue - inter * u_r) / (ue * ue), iou_d = (i_d * ue - inter * u_d) / (ue * ue);
    const float t = cs - uni;
    const float q_r = ((cs_r - u_r) * cs - t * cs_r) / (cs * cs), q_d = ((cs_d - u_d) * cs - t * cs_d) / (cs * cs);
    g_r2 = -iou_r + q_r;
    g_d = -iou_d + q_d;
}
