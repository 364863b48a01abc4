// ep24 - inference path: BatchNorm with running statistics + SiLU, eval-mode head decode
// (yolo_head_24p.py:190-210, 239-256) and `postprocess` (utils/boxes.py:29-99: class max, confidence filter,
// bounding rectangle of the 24 points, per-class NMS).  Elementwise / latency kernels, no MFMA.
#include "common.h"

namespace {

// y = act(z * scale + shift) (+ residual), scale = gamma / sqrt(running_var + eps), shift = beta - running_mean * scale
__global__ __launch_bounds__(256) void bn_act_infer_kernel(const bf16* z, long ld_z, const float* gamma, const float* beta,
                                                           const float* rmean, const float* rvar, bf16* y, long ld_y,
                                                           const bf16* res, long ld_res, long M, int C, float eps, int act) {
    extern __shared__ float lds[];
    float* sc = lds;
    float* sh = lds + C;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float s_ = gamma[c] / sqrtf(rvar[c] + eps);
        sc[c] = s_;
        sh[c] = beta[c] - rmean[c] * s_;
    }
    __syncthreads();
    const int cgs = C >> 3;
    const long total = M * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / cgs;
        const int g = (int)(i - m * cgs);
        const bf16x8 v = *reinterpret_cast<const bf16x8*>(z + m * ld_z + g * 8);
        bf16x8 r = {0, 0, 0, 0, 0, 0, 0, 0};
        if (res) r = *reinterpret_cast<const bf16x8*>(res + m * ld_res + g * 8);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float u = fmaf((float)v[j], sc[g * 8 + j], sh[g * 8 + j]);
            o[j] = (bf16)(act_fwd(u, act) + (res ? (float)r[j] : 0.f));
        }
        *reinterpret_cast<bf16x8*>(y + m * ld_y + g * 8) = o;
    }
}

// BatchNorm folding for every conv unit of the network in two launches: w' = w * gamma / sqrt(running_var + eps) per output
// channel (packed bf16 [Cout][T][Cin_pad], the layout of the forward weights), bias = beta - running_mean * scale.
// desc[s][12] = {master offset, packed offset, Cout, T, Cin, Cin_pad, gamma offset, beta offset (flat), running-mean offset,
// running-var offset (statistics buffer), bias offset, 0}; prefix[s] = first element of segment s; eps[s].
__global__ __launch_bounds__(256) void fold_bn_weights_kernel(const float* flat, const float* bstat, const long* desc, const long* prefix,
                                                              const float* eps, int n_seg, bf16* wf) {
    __shared__ int s_first;
    const long total = prefix[n_seg];
    for (long base = (long)blockIdx.x * 1024; base < total; base += (long)gridDim.x * 1024) {
        if (threadIdx.x == 0) {
            int lo = 0, hi = n_seg - 1;
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (prefix[mid] <= base) lo = mid; else hi = mid - 1;
            }
            s_first = lo;
        }
        __syncthreads();
        int seg = s_first;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long i = base + threadIdx.x + 256 * k;
            if (i >= total) break;
            while (i >= prefix[seg + 1]) ++seg;
            const long* d = desc + (long)seg * 12;
            const long e = i - prefix[seg];
            const int T = (int)d[3], Cin = (int)d[4], Cin_pad = (int)d[5];
            const int co = (int)(e / ((long)T * Cin));
            const float scale = flat[d[6] + co] / sqrtf(bstat[d[9] + co] + eps[seg]);
            wf[d[1] + (e / Cin) * Cin_pad + e % Cin] = (bf16)(flat[d[0] + e] * scale);
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void fold_bn_bias_kernel(const float* flat, const float* bstat, const long* desc, const long* cprefix,
                                                           const float* eps, int n_seg, float* bias) {
    const long total = cprefix[n_seg];
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        int lo = 0, hi = n_seg - 1;
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (cprefix[mid] <= i) lo = mid; else hi = mid - 1;
        }
        const long* d = desc + (long)lo * 12;
        const int co = (int)(i - cprefix[lo]);
        const float scale = flat[d[6] + co] / sqrtf(bstat[d[9] + co] + eps[lo]);
        bias[d[10] + co] = flat[d[7] + co] - bstat[d[8] + co] * scale;
    }
}

// eval head: xy = (t + grid) * s, r = exp(t) * s, obj / cls = sigmoid(logit)
__global__ __launch_bounds__(256) void decode_eval_kernel(float* out, int B, int A, int a0, int H, int W, float s, int ncols) {
    const long total = (long)B * H * W * ncols;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % ncols);
        const long cell = i / ncols;
        const int n = (int)(cell / (H * W));
        const int hw = (int)(cell - (long)n * H * W);
        float* p = out + ((long)n * A + a0 + hw) * ncols + c;
        const float t = *p;
        float v;
        if (c == 0) v = (t + (float)(hw % W)) * s;
        else if (c == 1) v = (t + (float)(hw / W)) * s;
        else if (c < 26) v = expf(t) * s;
        else v = 1.0f / (1.0f + expf(-t));
        *p = v;
    }
}

// per anchor: best class, score = obj * class_conf (or -1 when below conf_thre), bounding rectangle of the 24 points.
// The reference multiplies the radii by theta*cos(theta) / theta*sin(theta) (boxes.py:31-33, the angle itself is a
// factor) - reproduced as written.
__global__ __launch_bounds__(256) void post_prepare_kernel(const float* pred, int ncols, int C, long N, float conf_thre,
                                                           const float* ray /*[48]: theta*cos(theta), theta*sin(theta)*/,
                                                           float* score, float* conf, int* cls, float* rect) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < N; i += (long)gridDim.x * 256) {
        const float* p = pred + i * ncols;
        float best = p[27];
        int bi = 0;
        for (int c = 1; c < C; ++c)
            if (p[27 + c] > best) { best = p[27 + c]; bi = c; }       // torch.max: first maximum
        const float sc_ = p[26] * best;
        conf[i] = best;
        cls[i] = bi;
        score[i] = sc_ >= conf_thre ? sc_ : -1.0f;
        float x0 = INFINITY, x1 = -INFINITY, y0 = INFINITY, y1 = -INFINITY;
        for (int k = 0; k < 24; ++k) {
            const float px = p[2 + k] * ray[k] + p[0];
            const float py = p[2 + k] * ray[24 + k] + p[1];
            x0 = fminf(x0, px); x1 = fmaxf(x1, px); y0 = fminf(y0, py); y1 = fmaxf(y1, py);
        }
        rect[i * 4 + 0] = x0; rect[i * 4 + 1] = y0; rect[i * 4 + 2] = x1; rect[i * 4 + 3] = y1;
    }
}

// One workgroup per image: compact the candidates, bitonic-sort them by (score desc, index asc) in global scratch,
// then greedy NMS (torchvision semantics: suppress IoU > thr, same class only unless class-agnostic).
__global__ __launch_bounds__(1024) void nms_kernel(const float* score, const int* cls, const float* rect, int A, float thr,
                                                   int agnostic, float* skey, int* sidx, unsigned char* dead, int* keep,
                                                   int* keep_count, int P /* power of two >= A */) {
    __shared__ int n_sh;
    __shared__ int alive_sh;
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* sc = score + (long)b * A;
    const int* cl = cls + (long)b * A;
    const float* rc = rect + (long)b * A * 4;
    float* key = skey + (long)b * P;
    int* idx = sidx + (long)b * P;
    unsigned char* dd = dead + (long)b * P;
    int* kp = keep + (long)b * A;
    if (tid == 0) n_sh = 0;
    __syncthreads();
    for (int a = tid; a < A; a += 1024)
        if (sc[a] >= 0.f) { const int j = atomicAdd(&n_sh, 1); key[j] = sc[a]; idx[j] = a; }
    __syncthreads();
    const int n = n_sh;
    int P2 = 1;
    while (P2 < n) P2 <<= 1;
    for (int j = n + tid; j < P2; j += 1024) { key[j] = -2.0f; idx[j] = 0x7FFFFFFF; }
    __syncthreads();
    for (int k = 2; k <= P2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < P2; i += 1024) {
                const int l = i ^ j;
                if (l > i) {
                    const bool up = (i & k) == 0;          // "up" blocks hold the better (earlier) elements first
                    const float ki = key[i], kl = key[l];
                    const int ii = idx[i], il = idx[l];
                    const bool i_first = ki > kl || (ki == kl && ii < il);
                    if (up != i_first) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
                }
            }
            __syncthreads();
        }
    for (int j = tid; j < n; j += 1024) dd[j] = 0;
    __syncthreads();
    int kept = 0;
    for (int i = 0; i < n; ++i) {
        if (tid == 0) alive_sh = dd[i] == 0;
        __syncthreads();
        const bool alive = alive_sh != 0;
        if (alive) {
            const int ai = idx[i];
            if (tid == 0) kp[kept] = ai;
            ++kept;
            const float ax0 = rc[ai * 4], ay0 = rc[ai * 4 + 1], ax1 = rc[ai * 4 + 2], ay1 = rc[ai * 4 + 3];
            const float aarea = (ax1 - ax0) * (ay1 - ay0);
            const int ac = cl[ai];
            for (int j = i + 1 + tid; j < n; j += 1024) {
                if (dd[j]) continue;
                const int bj = idx[j];
                if (!agnostic && cl[bj] != ac) continue;
                const float bx0 = rc[bj * 4], by0 = rc[bj * 4 + 1], bx1 = rc[bj * 4 + 2], by1 = rc[bj * 4 + 3];
                const float iw = fmaxf(fminf(ax1, bx1) - fmaxf(ax0, bx0), 0.f);
                const float ih = fmaxf(fminf(ay1, by1) - fmaxf(ay0, by0), 0.f);
                const float inter = iw * ih;
                const float iou = inter / (aarea + (bx1 - bx0) * (by1 - by0) - inter);
                if (iou > thr) dd[j] = 1;
            }
        }
        __syncthreads();
    }
    if (tid == 0) keep_count[b] = kept;
}

// detections [n, 29] = (pred[:, :27], class_conf, class_pred) of the kept anchors of one image, in NMS order
__global__ __launch_bounds__(256) void post_gather_kernel(const float* pred, int ncols, const float* conf, const int* cls,
                                                          const int* keep, int n, float* det) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < (long)n * 29; i += (long)gridDim.x * 256) {
        const int r = (int)(i / 29), c = (int)(i - (long)r * 29);
        const int a = keep[r];
        det[i] = c < 27 ? pred[(long)a * ncols + c] : (c == 27 ? conf[a] : (float)cls[a]);
    }
}

}  // namespace

#define S_ (hipStream_t) stream

extern "C" int ep24_bn_act_infer(const void* z, int64_t ld_z, const float* gamma, const float* beta, const float* running_mean,
                                 const float* running_var, void* y, int64_t ld_y, const void* residual, int64_t ld_res,
                                 int64_t M, int C, float eps, int act, void* stream) {
    EP24_REQUIRE(z && gamma && beta && running_mean && running_var && y, EP24_E_ARG, "bn_act_infer: null pointer");
    EP24_REQUIRE(C % 8 == 0 && ld_z % 8 == 0 && ld_y % 8 == 0 && (!residual || ld_res % 8 == 0) && M > 0, EP24_E_ARG,
                 "bn_act_infer: C=%d / strides must be multiples of 8", C);
    long blocks = (M * (C / 8) + 511) / 512;
    hipLaunchKernelGGL(bn_act_infer_kernel, dim3((unsigned)(blocks > 2048 ? 2048 : blocks)), dim3(256), 2 * C * sizeof(float), S_,
                       (const bf16*)z, ld_z, gamma, beta, running_mean, running_var, (bf16*)y, ld_y, (const bf16*)residual, ld_res, (long)M,
                       C, eps, act);
    EP24_LAUNCH_CHECK("ep24_bn_act_infer");
    return EP24_OK;
}

extern "C" int ep24_fold_bn(const float* flat, const float* bstat, const int64_t* desc, const int64_t* prefix, const int64_t* cprefix,
                            const float* eps, int n_seg, int64_t total, int64_t total_channels, void* w_folded, float* bias, void* stream) {
    EP24_REQUIRE(flat && bstat && desc && prefix && cprefix && eps && w_folded && bias && n_seg > 0 && total > 0, EP24_E_ARG,
                 "fold_bn: bad arguments");
    long blocks = (total + 1023) / 1024;
    hipLaunchKernelGGL(fold_bn_weights_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, (hipStream_t)stream, flat, bstat,
                       (const long*)desc, (const long*)prefix, eps, n_seg, (bf16*)w_folded);
    hipLaunchKernelGGL(fold_bn_bias_kernel, dim3((unsigned)((total_channels + 255) / 256)), dim3(256), 0, (hipStream_t)stream, flat, bstat,
                       (const long*)desc, (const long*)cprefix, eps, n_seg, bias);
    EP24_LAUNCH_CHECK("ep24_fold_bn");
    return EP24_OK;
}

extern "C" int ep24_head_decode_eval(float* out, int B, int A, int a0, int H, int W, float stride, int ncols, void* stream) {
    EP24_REQUIRE(out && a0 >= 0 && a0 + H * W <= A && ncols >= 27, EP24_E_ARG, "head_decode_eval: bad arguments");
    long blocks = ((long)B * H * W * ncols + 255) / 256;
    hipLaunchKernelGGL(decode_eval_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, S_, out, B, A, a0, H, W,
                       stride, ncols);
    EP24_LAUNCH_CHECK("ep24_head_decode_eval");
    return EP24_OK;
}

extern "C" int ep24_post_prepare(const float* pred, int ncols, int num_classes, int64_t n_rows, float conf_thre,
                                 const float* ray_factors, float* score, float* conf, int32_t* cls, float* rect, void* stream) {
    EP24_REQUIRE(pred && ray_factors && score && conf && cls && rect && n_rows > 0 && num_classes > 0, EP24_E_ARG,
                 "post_prepare: bad arguments");
    EP24_REQUIRE(ncols == 27 + num_classes, EP24_E_ARG, "post_prepare: ncols=%d != 27+%d", ncols, num_classes);
    long blocks = (n_rows + 255) / 256;
    hipLaunchKernelGGL(post_prepare_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, S_, pred, ncols, num_classes,
                       (long)n_rows, conf_thre, ray_factors, score, conf, cls, rect);
    EP24_LAUNCH_CHECK("ep24_post_prepare");
    return EP24_OK;
}

extern "C" int ep24_post_nms(const float* score, const int32_t* cls, const float* rect, int B, int A, float nms_thre,
                             int class_agnostic, float* sort_key, int32_t* sort_idx, uint8_t* dead, int32_t* keep,
                             int32_t* keep_count, int P, void* stream) {
    EP24_REQUIRE(score && cls && rect && sort_key && sort_idx && dead && keep && keep_count && B > 0 && A > 0, EP24_E_ARG,
                 "post_nms: bad arguments");
    EP24_REQUIRE(P >= A && (P & (P - 1)) == 0, EP24_E_ARG, "post_nms: scratch rows P=%d must be a power of two >= A=%d", P, A);
    hipLaunchKernelGGL(nms_kernel, dim3(B), dim3(1024), 0, S_, score, cls, rect, A, nms_thre, class_agnostic, sort_key, sort_idx, dead,
                       keep, keep_count, P);
    EP24_LAUNCH_CHECK("ep24_post_nms");
    return EP24_OK;
}

extern "C" int ep24_post_gather(const float* pred, int ncols, const float* conf, const int32_t* cls, const int32_t* keep, int n,
                                float* det, void* stream) {
    if (n == 0) return EP24_OK;
    EP24_REQUIRE(pred && conf && cls && keep && det && n > 0, EP24_E_ARG, "post_gather: bad arguments");
    long blocks = ((long)n * 29 + 255) / 256;
    hipLaunchKernelGGL(post_gather_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, S_, pred, ncols, conf, cls, keep,
                       n, det);
    EP24_LAUNCH_CHECK("ep24_post_gather");
    return EP24_OK;
}
