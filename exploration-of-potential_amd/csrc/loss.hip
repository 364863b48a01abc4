// ep24 - 24-circle GIoU / objectness / class losses, dynamic task weights and the gradient w.r.t. the decoded
// head outputs (yolox_24p/models/losses.py:80-157, :283-357), plus the stand-alone forms behind
// utils.bboxes_iou (yolox_24p/utils/boxes.py:166-243) and IOUloss.forward.
//
// Sync-free: loss_terms writes per-workgroup partial sums, loss_finalize folds them in a fixed order
// (bitwise reproducible), computes num_fg / the 26 task weights / the scalar loss on the device and keeps the
// "last loss" state there; loss_grad then needs nothing from the host.
#include "geom.h"

namespace {

constexpr int G_MAX = EP24_MAX_GT;
constexpr int LCOLS = EP24_LABEL_COLS;
constexpr int NS = EP24_NUM_SUMS;          // 0..23 iou, 24 obj, 25 cls, 26 num_fg, 27 l1 (use_l1 only)
constexpr int NACC = 28;

__device__ __forceinline__ float bce_logits(float x, float y) {
    return fmaxf(x, 0.f) - x * y + log1pf(expf(-fabsf(x)));
}

// get_l1_target (losses.py:594-604) for one matched anchor: [cx/s - x_shift, cy/s - y_shift, log(|p_k|/s + 1e-8)]
// where p_k is the k-th contour POINT (absolute image coordinates, gt[:, 2::2] / gt[:, 3::2]) - as the reference has it.
__device__ __forceinline__ float l1_target(const float* lab, int c, float s, float xsh, float ysh) {
    if (c == 0) return lab[1] / s - xsh;
    if (c == 1) return lab[2] / s - ysh;
    const float px = lab[3 + 2 * (c - 2)], py = lab[4 + 2 * (c - 2)];
    return logf(sqrtf(px * px + py * py) / s + 1e-8f);
}

// The matched anchors of a workgroup (about one in a hundred), as thread indices in anchor order: ballots and a prefix over the
// four waves, so the list - and with it every sum below - is the same on every run.  Every thread of the workgroup calls it.
__device__ __forceinline__ int compact_matched(bool m, int* s_list, int* s_wc) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(m);
    if (lane == 0) s_wc[w] = __popcll(bal);
    __syncthreads();
    int base = 0;
    for (int i = 0; i < w; ++i) base += s_wc[i];
    if (m) s_list[base + __popcll(bal & ((1ull << lane) - 1ull))] = threadIdx.x;
    const int n = s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    __syncthreads();
    return n;
}

// Every anchor has an objectness term; the 24 ray terms, the class row and the L1 row exist for matched anchors only.  One thread
// per anchor left a wave waiting while one or two of its lanes walked 24 rays and 80 classes alone (75 us on the exposed path
// between forward and backward).  Now the workgroup lists its matched anchors and a WAVE takes one at a time: lane k the k-th ray,
// all lanes the classes.  The terms are the per-anchor form's, expression by expression; only the order of the sums differs.
__global__ __launch_bounds__(256) void loss_terms_kernel(const float* outputs, int ncols, const float* labels,
                                                         const int* matched_gt, const float* matched_iou, float* partials,
                                                         int A, int C, const float* origin, const float* xs,
                                                         const float* ys, const float* strides) {
    __shared__ float red[4][NS];
    __shared__ int s_list[256];
    __shared__ int s_wc[4];
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int g = -1;
    float obj = 0.f;
    if (a < A) {
        g = matched_gt[(long)b * A + a];
        obj = bce_logits(outputs[((long)b * A + a) * ncols + 26], g >= 0 ? 1.f : 0.f);
    }
    const int n = compact_matched(g >= 0, s_list, s_wc);
    float acc_r = 0.f, acc_c = 0.f, acc_l1 = 0.f, cnt = 0.f;
    for (int m = w; m < n; m += 4) {
        const int am = blockIdx.x * 256 + s_list[m];
        const float* o = outputs + ((long)b * A + am) * ncols;
        const float* lab = labels + ((long)b * G_MAX + matched_gt[(long)b * A + am]) * LCOLS;
        const float gcx = lab[1], gcy = lab[2];
        const float ddx = gcx - o[0], ddy = gcy - o[1];
        const float d = sqrtf(ddx * ddx + ddy * ddy);
        if (lane < 24) {
            const float vx = lab[3 + 2 * lane] - gcx, vy = lab[4 + 2 * lane] - gcy;
            acc_r += 1.0f - ray_giou(sqrtf(vx * vx + vy * vy), o[2 + lane], d);
        }
        const int cls = (int)lab[0];
        const float piou = matched_iou[(long)b * A + am];
        for (int c = lane; c < C; c += 64) acc_c += bce_logits(o[27 + c], c == cls ? piou : 0.f);
        cnt += 1.f;
        if (origin && lane < 26)                              // losses.py:304-307: |origin_preds - l1_target| over 26 columns
            acc_l1 += fabsf(origin[((long)b * A + am) * 26 + lane] - l1_target(lab, lane, strides[am], xs[am], ys[am]));
    }
    const float obj_w = wave_sum(obj), c_w = wave_sum(acc_c), l1_w = wave_sum(acc_l1);
    if (lane < 24) red[w][lane] = acc_r;
    if (lane == 0) { red[w][24] = obj_w; red[w][25] = c_w; red[w][26] = cnt; red[w][27] = l1_w; }
    __syncthreads();
    if (threadIdx.x < NS) {
        float v = 0.f;
        if (threadIdx.x < NACC) v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        partials[((long)b * gridDim.x + blockIdx.x) * NS + threadIdx.x] = v;
    }
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* partials, int nblocks, const int* num_gt, int B,
                                                           float* state, float* result) {
    // fixed-order two-level fold (bitwise reproducible): 8 row groups x 32 columns, then the 8 group sums in order
    __shared__ float grp[8][NS];
    __shared__ float sums[NS];
    const int t = threadIdx.x;
    {
        const int col = t & 31, g = t >> 5;
        const int per = (nblocks + 7) / 8;
        float s = 0.f;
        const int lo = g * per, hi = min(nblocks, (g + 1) * per);
        for (int i = lo; i < hi; i += 16) {                  // 16 loads in flight, added in row order
            float v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = i + k < hi ? partials[(long)(i + k) * NS + col] : 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (i + k < hi) s += v[k];
        }
        grp[g][col] = s;
    }
    __syncthreads();
    if (t < NS) {
        float s = 0.f;
        for (int g = 0; g < 8; ++g) s += grp[g][t];
        sums[t] = s;
    }
    __syncthreads();
    if (t != 0) return;
    const float nfg_raw = sums[26];
    const float nfg = fmaxf(nfg_raw, 1.f);                          // losses.py:280
    int ngts = 0;
    for (int b = 0; b < B; ++b) ngts += num_gt[b];
    float l[26], r[26], e[26];
    for (int k = 0; k < 26; ++k) l[k] = sums[k] / nfg;              // 0..23 iou, 24 obj, 25 cls
    for (int k = 0; k < 26; ++k) {
        r[k] = fminf(fmaxf(l[k] / (state[k] + 1e-8f), 0.f), 2.f);   // losses.py:316-323
        e[k] = expf(r[k] / 20.0f);
    }
    float den = 0.f;
    for (int k = 0; k < 24; ++k) den += e[k];
    den = den + e[24] + e[25];
    float loss = 0.f;
    for (int k = 0; k < 24; ++k) {
        const float w = 26.0f * e[k] / den;
        result[29 + k] = w;
        result[1 + k] = w * l[k];
        loss += w * l[k];
    }
    const float ow = 26.0f * e[24] / den, cw = 26.0f * e[25] / den;
    const float l1 = sums[27] / nfg;                                // losses.py:304-309 (0 without use_l1)
    loss = loss + ow * l[24] + cw * l[25] + l1;
    result[0] = loss;
    result[56] = l1;
    result[25] = l[24];
    result[26] = l[25];
    result[27] = nfg;
    result[28] = (float)ngts;
    result[53] = ow;
    result[54] = cw;
    result[55] = nfg_raw;
    for (int k = 0; k < 26; ++k) state[k] = l[k];                   // losses.py:343-345
}

__global__ __launch_bounds__(256) void loss_grad_kernel(const float* outputs, int ncols, const float* labels,
                                                        const int* matched_gt, const float* matched_iou, const float* result,
                                                        const float* grad_scale, float* dout, int A, int C,
                                                        const float* origin, const float* xs, const float* ys,
                                                        const float* strides, float* d_origin) {
    // 99 % of the anchors are unmatched: their rows are zero but for the objectness column.  One thread per anchor writing its own
    // 107-float row made every store instruction 64 scattered 4-byte writes (72 MB of them: 95 us on the exposed path between
    // forward and backward).  Now the block writes the rows of its 256 anchors cooperatively, consecutive lanes on consecutive
    // addresses; the few matched anchors are skipped there and written by their own threads below.  Same values as before.
    __shared__ float s_obj[256];
    __shared__ int s_g[256];
    const int b = blockIdx.y;
    const int a0 = blockIdx.x * 256;
    const int a = a0 + threadIdx.x;
    const float gs = (grad_scale ? *grad_scale : 1.0f) / result[27];
    int g = -2;                                                        // -2: no such anchor
    if (a < A) {
        g = matched_gt[(long)b * A + a];
        const float so = 1.0f / (1.0f + expf(-outputs[((long)b * A + a) * ncols + 26]));
        s_obj[threadIdx.x] = result[53] * gs * (so - (g >= 0 ? 1.f : 0.f));
    }
    s_g[threadIdx.x] = g;
    __syncthreads();
    {
        const int nrows = min(256, A - a0);
        float* base = dout + ((long)b * A + a0) * ncols;
        int row = threadIdx.x / ncols, col = threadIdx.x - row * ncols;
        const int drow = 256 / ncols, dcol = 256 - drow * ncols;
        for (int f = threadIdx.x; f < nrows * ncols; f += 256) {
            if (s_g[row] < 0) base[f] = col == 26 ? s_obj[row] : 0.f;
            row += drow; col += dcol;
            if (col >= ncols) { col -= ncols; ++row; }
        }
        if (d_origin) {
            float* ob = d_origin + ((long)b * A + a0) * 26;
            for (int f = threadIdx.x; f < nrows * 26; f += 256)
                if (s_g[f / 26] < 0) ob[f] = 0.f;
        }
    }
    // the matched anchors, one per wave at a time: lane k the k-th ray, all lanes the classes (the per-anchor loop kept one lane
    // of a wave busy for 24 rays and 80 classes); consecutive lanes write consecutive columns
    __shared__ int s_list[256];
    __shared__ int s_wc[4];
    const int n = compact_matched(g >= 0, s_list, s_wc);
    const int lane = threadIdx.x & 63;
    for (int m = threadIdx.x >> 6; m < n; m += 4) {
        const int t = s_list[m], am = a0 + t;
        const float* o = outputs + ((long)b * A + am) * ncols;
        float* d_o = dout + ((long)b * A + am) * ncols;
        const float* lab = labels + ((long)b * G_MAX + s_g[t]) * LCOLS;
        if (d_origin && lane < 26) {                                   // d |x - t| = sign(x - t), unweighted, / num_fg
            const float e = origin[((long)b * A + am) * 26 + lane] - l1_target(lab, lane, strides[am], xs[am], ys[am]);
            d_origin[((long)b * A + am) * 26 + lane] = e > 0.f ? gs : (e < 0.f ? -gs : 0.f);
        }
        const float gcx = lab[1], gcy = lab[2];
        const float ddx = gcx - o[0], ddy = gcy - o[1];
        const float d = sqrtf(ddx * ddx + ddy * ddy);
        float gdl = 0.f;
        if (lane < 24) {
            const float vx = lab[3 + 2 * lane] - gcx, vy = lab[4 + 2 * lane] - gcy;
            float g_r, g_d;
            ray_loss_grad(sqrtf(vx * vx + vy * vy), o[2 + lane], d, g_r, g_d);
            const float wk = result[29 + lane] * gs;
            d_o[2 + lane] = wk * g_r;
            gdl = wk * g_d;
        }
        const float gd = wave_sum(gdl);
        // d = sqrt((gx-cx)^2 + (gy-cy)^2): dd/dcx = -(gx-cx)/d.  d == 0 gives NaN, exactly as the reference's autograd
        if (lane == 0) {
            d_o[0] = gd * (-ddx / d);
            d_o[1] = gd * (-ddy / d);
            d_o[26] = s_obj[t];
        }
        const int cls = (int)lab[0];
        const float piou = matched_iou[(long)b * A + am];
        const float cw = result[54] * gs;
        for (int c = lane; c < C; c += 64) {
            const float sc = 1.0f / (1.0f + expf(-o[27 + c]));
            d_o[27 + c] = cw * (sc - (c == cls ? piou : 0.f));
        }
    }
}

// Round 5: loss gradient AND decode backward in one pass, for the captured step.  loss_grad_kernel writes d loss / d outputs as a dense
// fp32 [B,A,27+C] tensor (72 MB at B = 20, 99 % zeros) that head_decode_bwd then reads back to produce what the prediction convs'
// backward really consumes: per level, bf16 rows of the reg+obj gradient [cells][32] and of the class gradient [cells][ld_cls]
// (xy: dout * stride, radii: dout * decoded radius, obj / classes unchanged: yolo_head_24p.py:212-237 backward).  Here the block writes
// those rows directly - the unmatched anchors' cooperatively (zeros but for the objectness column), the matched ones by a wave each -
// with the same fp32 expressions followed by the same one rounding to bf16: bit-identical to the two-launch form
// (tests/test_gpu_loss.py::test_fused_loss_grad_decode).  Not for the L1 branch (its extra gradient is added by head_decode_bwd).
struct DecodeLevels {
    int a0[4], hw[4];             // first anchor and cells per image of each level (a0[n] = A)
    float stride[4];
    bf16* d_ro[4];
    bf16* d_cl[4];
    int n, ld_cls;
};

__global__ __launch_bounds__(256) void loss_grad_decode_kernel(const float* outputs, int ncols, const float* labels,
                                                               const int* matched_gt, const float* matched_iou, const float* result,
                                                               const DecodeLevels lv, int A, int C) {
    __shared__ float s_obj[256];
    __shared__ int s_g[256];
    __shared__ long s_cell[256];                                       // cell index inside the anchor's level
    __shared__ int s_lv[256];
    const int b = blockIdx.y;
    const int a0 = blockIdx.x * 256;
    const int a = a0 + threadIdx.x;
    const float gs = 1.0f / result[27];
    int g = -2;                                                        // -2: no such anchor
    if (a < A) {
        g = matched_gt[(long)b * A + a];
        const float so = 1.0f / (1.0f + expf(-outputs[((long)b * A + a) * ncols + 26]));
        s_obj[threadIdx.x] = result[53] * gs * (so - (g >= 0 ? 1.f : 0.f));
        int l = 0;
#pragma unroll
        for (int k = 1; k < 4; ++k)
            if (k < lv.n && a >= lv.a0[k]) l = k;
        s_lv[threadIdx.x] = l;
        s_cell[threadIdx.x] = (long)b * lv.hw[l] + (a - lv.a0[l]);
    }
    s_g[threadIdx.x] = g;
    __syncthreads();
    {
        // rows of the unmatched anchors: 4 chunks of 8 reg+obj columns, then ld_cls / 8 chunks of class columns, 16 bytes each
        const int nrows = min(256, A - a0);
        const int chunks = 4 + (lv.ld_cls >> 3);
        for (int f = threadIdx.x; f < nrows * chunks; f += 256) {
            const int row = f / chunks, ch = f - row * chunks;
            if (s_g[row] >= 0) continue;
            const int l = s_lv[row];
            bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (ch == 3) o[2] = (bf16)s_obj[row];                      // column 26
            if (ch < 4) *reinterpret_cast<bf16x8*>(lv.d_ro[l] + s_cell[row] * 32 + ch * 8) = o;
            else *reinterpret_cast<bf16x8*>(lv.d_cl[l] + s_cell[row] * lv.ld_cls + (ch - 4) * 8) = o;
        }
    }
    __shared__ int s_list[256];
    __shared__ int s_wc[4];
    const int n = compact_matched(g >= 0, s_list, s_wc);
    const int lane = threadIdx.x & 63;
    for (int m = threadIdx.x >> 6; m < n; m += 4) {
        const int t = s_list[m], am = a0 + t;
        const int l = s_lv[t];
        const float* o = outputs + ((long)b * A + am) * ncols;
        bf16* ro = lv.d_ro[l] + s_cell[t] * 32;
        bf16* cl = lv.d_cl[l] + s_cell[t] * lv.ld_cls;
        const float* lab = labels + ((long)b * G_MAX + s_g[t]) * LCOLS;
        const float gcx = lab[1], gcy = lab[2];
        const float ddx = gcx - o[0], ddy = gcy - o[1];
        const float d = sqrtf(ddx * ddx + ddy * ddy);
        float gdl = 0.f;
        if (lane < 24) {
            const float vx = lab[3 + 2 * lane] - gcx, vy = lab[4 + 2 * lane] - gcy;
            float g_r, g_d;
            ray_loss_grad(sqrtf(vx * vx + vy * vy), o[2 + lane], d, g_r, g_d);
            const float wk = result[29 + lane] * gs;
            const float dv = wk * g_r;                                 // what loss_grad_kernel stores in column 2 + lane
            ro[2 + lane] = (bf16)(dv * o[2 + lane]);                   // d exp(t) * s / dt = r
            gdl = wk * g_d;
        } else if (lane >= 27 && lane < 32) {
            ro[lane] = (bf16)0.f;                                      // padding columns of the 32-wide row
        }
        const float gd = wave_sum(gdl);
        if (lane == 0) {
            const float st = lv.stride[l];
            const float d0 = gd * (-ddx / d), d1 = gd * (-ddy / d);    // d == 0 gives NaN, exactly as the reference's autograd
            ro[0] = (bf16)(d0 * st);
            ro[1] = (bf16)(d1 * st);
            ro[26] = (bf16)s_obj[t];
        }
        const int cls = (int)lab[0];
        const float piou = matched_iou[(long)b * A + am];
        const float cw = result[54] * gs;
        for (int c = lane; c < lv.ld_cls; c += 64) {
            float v = 0.f;
            if (c < C) {
                const float sc = 1.0f / (1.0f + expf(-o[27 + c]));
                v = cw * (sc - (c == cls ? piou : 0.f));
            }
            cl[c] = (bf16)v;
        }
    }
}

// ------------------------------------------------------------------------------------------ stand-alone forms
__global__ __launch_bounds__(256) void pairwise_kernel(const float* gt50, const float* pred26, float* out, int G, int P) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)G * P) return;
    const int g = (int)(i / P), p = (int)(i - (long)g * P);
    const float* t = gt50 + (long)g * 50;
    const float* q = pred26 + (long)p * 26;
    const float ddx = t[0] - q[0], ddy = t[1] - q[1];
    const float d = sqrtf(ddx * ddx + ddy * ddy);
    float acc = 0.f;
    for (int k = 0; k < 24; ++k) {
        const float vx = t[2 + 2 * k] - t[0], vy = t[3 + 2 * k] - t[1];
        acc += 1.0f - ray_giou(sqrtf(vx * vx + vy * vy), q[2 + k], d);
    }
    out[i] = acc / 24.0f / 2.0f;
}

__global__ __launch_bounds__(256) void matched_fwd_kernel(const float* pred26, const float* target50, float* loss24, int N) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)N * 24) return;
    const int n = (int)(i / 24), k = (int)(i - (long)n * 24);
    const float* t = target50 + (long)n * 50;
    const float* q = pred26 + (long)n * 26;
    const float ddx = t[0] - q[0], ddy = t[1] - q[1];
    const float d = sqrtf(ddx * ddx + ddy * ddy);
    const float vx = t[2 + 2 * k] - t[0], vy = t[3 + 2 * k] - t[1];
    loss24[i] = 1.0f - ray_giou(sqrtf(vx * vx + vy * vy), q[2 + k], d);
}

// circle_inter of the reference as it stands: intersection areas and centre distances of (gt row, pred row) pairs over the 24 rays.
// pairwise = 0: row i of the one against row i of the other (IOUloss.circle_inter, losses.py:23-78); pairwise = 1: every gt row
// against every pred row, pair index g * P + p (utils.boxes.circle_inter, boxes.py:102-163: repeat_interleave on gt, repeat on pred).
// One thread per (pair, ray): consecutive lanes on consecutive addresses of both outputs.
__global__ __launch_bounds__(256) void circle_lens_kernel(const float* gt_cx, const float* gt_cy, const float* gt_r, const float* pd_cx,
                                                          const float* pd_cy, const float* pd_r, float* res, float* dist, long pairs,
                                                          int P, int pairwise) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs * 24) return;
    const long pr = i / 24;
    const int k = (int)(i - pr * 24);
    const long g = pairwise ? pr / P : pr, p = pairwise ? pr - g * P : pr;
    const float ddx = gt_cx[g] - pd_cx[p], ddy = gt_cy[g] - pd_cy[p];
    const float d = sqrtf(ddx * ddx + ddy * ddy);
    res[i] = ray_inter(gt_r[g * 24 + k], pd_r[p * 24 + k], d);
    dist[i] = d;
}

__global__ __launch_bounds__(256) void matched_bwd_kernel(const float* pred26, const float* target50, const float* dloss24,
                                                          float* dpred26, int N) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const float* t = target50 + (long)n * 50;
    const float* q = pred26 + (long)n * 26;
    const float ddx = t[0] - q[0], ddy = t[1] - q[1];
    const float d = sqrtf(ddx * ddx + ddy * ddy);
    float gd = 0.f;
    for (int k = 0; k < 24; ++k) {
        const float vx = t[2 + 2 * k] - t[0], vy = t[3 + 2 * k] - t[1];
        float g_r, g_d;
        ray_loss_grad(sqrtf(vx * vx + vy * vy), q[2 + k], d, g_r, g_d);
        const float w = dloss24[(long)n * 24 + k];
        dpred26[(long)n * 26 + 2 + k] = w * g_r;
        gd += w * g_d;
    }
    dpred26[(long)n * 26 + 0] = gd * (-ddx / d);
    dpred26[(long)n * 26 + 1] = gd * (-ddy / d);
}

}  // namespace

extern "C" int ep24_loss_blocks(int B, int A) { return B * ep24_cdiv(A, 256); }

extern "C" int ep24_loss_terms(const float* outputs, int ncols, const float* labels, const int32_t* matched_gt,
                               const float* matched_iou, float* partials, int B, int A, int num_classes, const float* origin,
                               const float* xs, const float* ys, const float* strides, void* stream) {
    EP24_REQUIRE(outputs && labels && matched_gt && matched_iou && partials, EP24_E_ARG, "loss_terms: null pointer");
    EP24_REQUIRE(ncols == 27 + num_classes, EP24_E_ARG, "loss_terms: ncols=%d != 27+%d", ncols, num_classes);
    EP24_REQUIRE(!origin || (xs && ys && strides), EP24_E_ARG, "loss_terms: the L1 branch needs the anchor grid");
    hipLaunchKernelGGL(loss_terms_kernel, dim3(ep24_cdiv(A, 256), B), dim3(256), 0, (hipStream_t)stream, outputs, ncols, labels,
                       matched_gt, matched_iou, partials, A, num_classes, origin, xs, ys, strides);
    EP24_LAUNCH_CHECK("ep24_loss_terms");
    return EP24_OK;
}

extern "C" int ep24_loss_finalize(const float* partials, int nblocks, const int32_t* num_gt, int B, float* state, float* result,
                                  void* stream) {
    EP24_REQUIRE(partials && num_gt && state && result && nblocks > 0, EP24_E_ARG, "loss_finalize: bad arguments");
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, nblocks, num_gt, B, state, result);
    EP24_LAUNCH_CHECK("ep24_loss_finalize");
    return EP24_OK;
}

extern "C" int ep24_loss_grad(const float* outputs, int ncols, const float* labels, const int32_t* matched_gt,
                              const float* matched_iou, const float* result, const float* grad_scale, float* dout, int B, int A,
                              int num_classes, const float* origin, const float* xs, const float* ys, const float* strides,
                              float* d_origin, void* stream) {
    EP24_REQUIRE(outputs && labels && matched_gt && matched_iou && result && dout, EP24_E_ARG, "loss_grad: null pointer");
    EP24_REQUIRE(ncols == 27 + num_classes, EP24_E_ARG, "loss_grad: ncols=%d != 27+%d", ncols, num_classes);
    EP24_REQUIRE(!d_origin || (origin && xs && ys && strides), EP24_E_ARG, "loss_grad: the L1 branch needs origin and the anchor grid");
    hipLaunchKernelGGL(loss_grad_kernel, dim3(ep24_cdiv(A, 256), B), dim3(256), 0, (hipStream_t)stream, outputs, ncols, labels,
                       matched_gt, matched_iou, result, grad_scale, dout, A, num_classes, origin, xs, ys, strides, d_origin);
    EP24_LAUNCH_CHECK("ep24_loss_grad");
    return EP24_OK;
}

extern "C" int ep24_loss_grad_decode(const float* outputs, int ncols, const float* labels, const int32_t* matched_gt, const float* matched_iou,
                                     const float* result, int B, int A, int num_classes, int n_levels, const int64_t* levels, void* stream) {
    EP24_REQUIRE(outputs && labels && matched_gt && matched_iou && result && levels, EP24_E_ARG, "loss_grad_decode: null pointer");
    EP24_REQUIRE(ncols == 27 + num_classes && n_levels >= 1 && n_levels <= 3, EP24_E_ARG, "loss_grad_decode: ncols=%d, %d levels", ncols, n_levels);
    DecodeLevels lv{};
    lv.n = n_levels;
    lv.ld_cls = (num_classes + 7) & ~7;
    int a0 = 0;
    for (int i = 0; i < n_levels; ++i) {                              // levels: HOST rows (cells per image, stride as float bits in the low word, d_ro, d_cl)
        const int64_t* r = levels + 4 * i;
        lv.a0[i] = a0; lv.hw[i] = (int)r[0];
        union { unsigned u; float f; } cv; cv.u = (unsigned)r[1]; lv.stride[i] = cv.f;
        lv.d_ro[i] = (bf16*)r[2]; lv.d_cl[i] = (bf16*)r[3];
        EP24_REQUIRE(lv.d_ro[i] && lv.d_cl[i] && ((r[2] | r[3]) & 15) == 0, EP24_E_ARG, "loss_grad_decode: level %d gradient rows must be 16-byte aligned", i);
        a0 += lv.hw[i];
    }
    lv.a0[n_levels] = a0;
    EP24_REQUIRE(a0 == A, EP24_E_ARG, "loss_grad_decode: the levels hold %d anchors, A = %d", a0, A);
    hipLaunchKernelGGL(loss_grad_decode_kernel, dim3(ep24_cdiv(A, 256), B), dim3(256), 0, (hipStream_t)stream, outputs, ncols, labels,
                       matched_gt, matched_iou, result, lv, A, num_classes);
    EP24_LAUNCH_CHECK("ep24_loss_grad_decode");
    return EP24_OK;
}

extern "C" int ep24_circle_pairwise(const float* gt50, const float* pred26, float* out, int G, int P, void* stream) {
    if (G == 0 || P == 0) return EP24_OK;       // empty GT / empty candidate set: nothing to write
    EP24_REQUIRE(gt50 && pred26 && out, EP24_E_ARG, "circle_pairwise: null pointer");
    hipLaunchKernelGGL(pairwise_kernel, dim3(ep24_cdiv((long)G * P, 256)), dim3(256), 0, (hipStream_t)stream, gt50, pred26, out, G, P);
    EP24_LAUNCH_CHECK("ep24_circle_pairwise");
    return EP24_OK;
}

extern "C" int ep24_circle_matched_fwd(const float* pred26, const float* target50, float* loss24, int N, void* stream) {
    if (N == 0) return EP24_OK;
    EP24_REQUIRE(pred26 && target50 && loss24, EP24_E_ARG, "circle_matched_fwd: null pointer");
    hipLaunchKernelGGL(matched_fwd_kernel, dim3(ep24_cdiv((long)N * 24, 256)), dim3(256), 0, (hipStream_t)stream, pred26, target50,
                       loss24, N);
    EP24_LAUNCH_CHECK("ep24_circle_matched_fwd");
    return EP24_OK;
}

extern "C" int ep24_circle_matched_bwd(const float* pred26, const float* target50, const float* dloss24, float* dpred26, int N,
                                       void* stream) {
    if (N == 0) return EP24_OK;
    EP24_REQUIRE(pred26 && target50 && dloss24 && dpred26, EP24_E_ARG, "circle_matched_bwd: null pointer");
    hipLaunchKernelGGL(matched_bwd_kernel, dim3(ep24_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, pred26, target50, dloss24,
                       dpred26, N);
    EP24_LAUNCH_CHECK("ep24_circle_matched_bwd");
    return EP24_OK;
}

extern "C" int ep24_circle_lens(const float* gt_cx, const float* gt_cy, const float* gt_r, const float* pd_cx, const float* pd_cy,
                                const float* pd_r, float* res_inter, float* dist, int G, int P, int pairwise, void* stream) {
    EP24_REQUIRE(G >= 0 && P >= 0 && (pairwise || G == P), EP24_E_ARG, "circle_lens: the matched form needs as many gt rows as pred rows (%d, %d)", G, P);
    const long pairs = pairwise ? (long)G * P : G;
    if (pairs == 0) return EP24_OK;             // the reference's placeholder path: nothing to compute
    EP24_REQUIRE(gt_cx && gt_cy && gt_r && pd_cx && pd_cy && pd_r && res_inter && dist, EP24_E_ARG, "circle_lens: null pointer");
    EP24_REQUIRE(pairs * 24 < (1L << 31) * 256, EP24_E_UNSUPPORTED, "circle_lens: %ld pairs are beyond one launch", pairs);
    hipLaunchKernelGGL(circle_lens_kernel, dim3((unsigned)ep24_cdiv(pairs * 24, 256)), dim3(256), 0, (hipStream_t)stream, gt_cx, gt_cy,
                       gt_r, pd_cx, pd_cy, pd_r, res_inter, dist, pairs, P, pairwise);
    EP24_LAUNCH_CHECK("ep24_circle_lens");
    return EP24_OK;
}
