// ep24 - SimOTA label assignment for the 24-point head, batched over images with no host round trip.
//
// Reference flow per image (yolox_24p/models/losses.py:360-494): candidate masks -> boolean gathers (dynamic
// shape P) -> [G,P] pairwise/cost matrices -> per-GT python loop of topk -> .tolist()/.item() syncs.
// Here every stage is a dense kernel over [B, G<=50, A] with the candidate set carried as per-anchor 64-bit
// GT bitmasks, counts stay on the device and the stages chain on one stream:
//   candidates (a4+a5) -> cost (a6+a7) -> dynamic_k (a8, one workgroup per (image, gt)) -> resolve (a8).
// All kernels are elementwise geometry (atan2 / acos / sin / log): no MFMA, HBM/L2 resident, GT rows in LDS.
#include "geom.h"

namespace {

constexpr int G_MAX = EP24_MAX_GT;
constexpr int LCOLS = EP24_LABEL_COLS;

// rows with sum > 0 (losses.py:190); valid rows are a prefix (data_augment.py:160-174)
__device__ int count_gt(const float* lab /*[50][51]*/, float* scratch /*[64]*/) {
    const int t = threadIdx.x;
    if (t < 64) {
        float s = 0.f;
        if (t < G_MAX)
            for (int c = 0; c < LCOLS; ++c) s += lab[t * LCOLS + c];
        scratch[t] = (t < G_MAX && s > 0.f) ? 1.f : 0.f;
    }
    __syncthreads();
    int n = 0;
    for (int i = 0; i < G_MAX; ++i) n += (int)scratch[i];
    __syncthreads();
    return n;
}

#ifdef EP24_STAMPS
// diagnostic build only (make stamps; tools/step_stress.py reads it): [0..3] how often each of four evaluations of the same
// angle sum disagreed with the other three, [7] no majority, [8] threads, [9] disagreeing lanes outside 48..63
__device__ unsigned long long g_cand_dbg[16];
#endif

// ------------------------------------------------------------------------------------------ a4 + a5
__global__ __launch_bounds__(256) void candidates_kernel(const float* labels, const float* xs, const float* ys,
                                                         const float* strides, int* num_gt, unsigned long long* in_box,
                                                         unsigned long long* in_ctr, int A) {
    __shared__ float vx[G_MAX][24], vy[G_MAX][24], gcx[G_MAX], gcy[G_MAX], scratch[64];
    const int b = blockIdx.y;
    const float* lab = labels + (long)b * G_MAX * LCOLS;
    const int ng = count_gt(lab, scratch);
    for (int i = threadIdx.x; i < ng * 24; i += 256) {
        const int g = i / 24, k = i - g * 24;
        vx[g][k] = lab[g * LCOLS + 3 + 2 * k];
        vy[g][k] = lab[g * LCOLS + 4 + 2 * k];
    }
    for (int g = threadIdx.x; g < ng; g += 256) { gcx[g] = lab[g * LCOLS + 1]; gcy[g] = lab[g * LCOLS + 2]; }
    __syncthreads();
    if (blockIdx.x == 0 && threadIdx.x == 0) num_gt[b] = ng;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const float s = strides[a];
    const float xc = xs[a] * s + 0.5f * s;             // losses.py:506-516
    const float yc = ys[a] * s + 0.5f * s;
    const float rad = 2.5f * s;
    unsigned long long mb = 0ull, mc = 0ull;
#ifdef EP24_STAMPS
    float degA[16];
#endif
    for (int g = 0; g < ng; ++g) {
        // pts_in_poly (losses.py:555-592): unsigned angle of every edge seen from the anchor centre, degrees
        float deg = 0.f;
        float sx = vx[g][0] - xc, sy = vy[g][0] - yc;
        for (int k = 0; k < 24; ++k) {
            const int k1 = k == 23 ? 0 : k + 1;
            const float ex = vx[g][k1] - xc, ey = vy[g][k1] - yc;
            const float cross = sx * ey - ex * sy;
            const float dot = sx * ex + sy * ey;
            deg += atan2f(fabsf(cross), dot) * 57.2957795130823208768f;
            sx = ex; sy = ey;
        }
        if (deg >= 350.f) mb |= 1ull << g;
#ifdef EP24_STAMPS
        if (g < 16) degA[g] = deg;
#endif
        // centre square (losses.py:523-543): min of the four deltas strictly positive
        const float cl = xc - (gcx[g] - rad), cr = (gcx[g] + rad) - xc;
        const float ct = yc - (gcy[g] - rad), cb = (gcy[g] + rad) - yc;
        if (fminf(fminf(cl, ct), fminf(cr, cb)) > 0.0f) mc |= 1ull << g;
    }
    in_box[(long)b * A + a] = mb;
    in_ctr[(long)b * A + a] = mc;
#ifdef EP24_STAMPS
    // diagnostic build: the same polygon angle sum computed four ways; the pass that disagrees with the other three is counted
    auto term = [&](float sx, float sy, float ex, float ey) {
        const float cross = sx * ey - ex * sy;
        const float dot = sx * ex + sy * ey;
        return atan2f(fabsf(cross), dot) * 57.2957795130823208768f;
    };
    auto lds_nop = [&](const float* p) {             // LDS read, then a generous pause before the value may be used
        float v;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\ts_nop 7\n\ts_nop 7" : "=v"(v) : "v"((unsigned)(unsigned long long)p) : "memory");
        return v;
    };
    for (int g = 0; g < ng && g < 16; ++g) {
        float d1 = 0.f, d2 = 0.f, d3 = 0.f;
        {   // pass 1: LDS through lds_nop
            float sx = lds_nop(&vx[g][0]) - xc, sy = lds_nop(&vy[g][0]) - yc;
            for (int k = 0; k < 24; ++k) {
                const int k1 = k == 23 ? 0 : k + 1;
                const float ex = lds_nop(&vx[g][k1]) - xc, ey = lds_nop(&vy[g][k1]) - yc;
                d1 += term(sx, sy, ex, ey);
                sx = ex; sy = ey;
            }
        }
        {   // pass 2: the whole polygon in registers first, then arithmetic only
            float px[24], py[24];
#pragma unroll
            for (int k = 0; k < 24; ++k) { px[k] = lds_nop(&vx[g][k]); py[k] = lds_nop(&vy[g][k]); }
            asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
            float sx = px[0] - xc, sy = py[0] - yc;
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                const int k1 = k == 23 ? 0 : k + 1;
                const float ex = px[k1] - xc, ey = py[k1] - yc;
                d2 += term(sx, sy, ex, ey);
                sx = ex; sy = ey;
            }
        }
        {   // pass 3: global memory
            float sx = lab[g * LCOLS + 3] - xc, sy = lab[g * LCOLS + 4] - yc;
            for (int k = 0; k < 24; ++k) {
                const int k1 = k == 23 ? 0 : k + 1;
                const float ex = lab[g * LCOLS + 3 + 2 * k1] - xc, ey = lab[g * LCOLS + 4 + 2 * k1] - yc;
                d3 += term(sx, sy, ex, ey);
                sx = ex; sy = ey;
            }
        }
        const unsigned u0 = __float_as_uint(degA[g]), u1 = __float_as_uint(d1), u2 = __float_as_uint(d2), u3 = __float_as_uint(d3);
        if (u0 != u1 || u0 != u2 || u0 != u3) {
            int which = 7;                                                    // no majority
            if (u1 == u2 && u2 == u3) which = 0;
            else if (u0 == u2 && u2 == u3) which = 1;
            else if (u0 == u1 && u1 == u3) which = 2;
            else if (u0 == u1 && u1 == u2) which = 3;
            atomicAdd(&g_cand_dbg[which], 1ull);
            if ((threadIdx.x & 63) < 48) atomicAdd(&g_cand_dbg[9], 1ull);    // a lane outside the last quarter
        }
    }
    atomicAdd(&g_cand_dbg[8], 1ull);
#endif
}

// ------------------------------------------------------------------------------------------ a6 + a7
// Only the candidates of an image (anchors inside some GT box or centre square: a few per cent, in runs along x) have a
// cost row, and a row costs 24 lens evaluations per GT.  One thread per anchor left most lanes of every wave idle while one
// or two of them walked all the GTs (360 us per step at B = 20).  Here a workgroup of 256 threads compacts the candidates
// among its 64 anchors into LDS (the class-sum term, once per candidate), then all its threads share the (candidate, GT)
// pairs: four threads per anchor where the candidates are dense, an immediate exit where there are none.  The arithmetic
// of a pair is unchanged, expression by expression, so pw / cost are bit-identical to the per-anchor form (ray_giou only
// skips the lens of circle pairs that do not properly intersect, whose value it never used).
constexpr int COST_APB = 64;                                       // anchors per workgroup (one wave compacts them)
constexpr int COST_NT = 256;                                       // threads that share the workgroup's (candidate, GT) pairs

__global__ __launch_bounds__(COST_NT) void cost_kernel(const float* outputs, int ncols, const float* labels,
                                                       const int* num_gt, const unsigned long long* in_box,
                                                       const unsigned long long* in_ctr, float* pw, float* cost, int A, int C,
                                                       int a_base, int a_end) {
    __shared__ float gr[G_MAX][24], gcx[G_MAX], gcy[G_MAX];
    __shared__ float spr[24][COST_APB];                             // a candidate's 24 predicted radii, k-major
    __shared__ float s_pcx[COST_APB], s_pcy[COST_APB], s_so[COST_APB], s_s0[COST_APB];
    __shared__ unsigned long long s_both[COST_APB];
    __shared__ int s_anchor[COST_APB];
    __shared__ int gcls[G_MAX];
    __shared__ int s_nc, s_nq;
    __shared__ float terms[24][COST_NT];                           // the 24 ray terms of the batch's pairs, ray-major
    __shared__ float s_d[COST_NT];                                 // centre distance of the batch's pairs
    __shared__ unsigned short queue[COST_NT * 24];                 // (pair of the batch) * 24 + ray of the items that need the lens
    const int b = blockIdx.y;
    const int ng = num_gt[b];
    // ---- compaction by the first wave: slot of an anchor among the candidates of the workgroup (anchor order is kept)
    int slot = -1;
    unsigned long long mb = 0ull, mc = 0ull;
    const int a = a_base + blockIdx.x * COST_APB + threadIdx.x;     // the launch covers anchors [a_base, a_end) of every image
    if (threadIdx.x < COST_APB) {
        if (a < a_end) { mb = in_box[(long)b * A + a]; mc = in_ctr[(long)b * A + a]; }
        const bool cand = (mb | mc) != 0ull;                        // fg_mask
        const unsigned long long bal = __ballot(cand);
        if (cand) slot = __popcll(bal & ((1ull << threadIdx.x) - 1ull));
        if (threadIdx.x == 0) s_nc = __popcll(bal);
    }
    __syncthreads();
    const int nc = s_nc;
    if (nc == 0) return;                                            // uniform: most workgroups of an image end here
    const float* lab = labels + (long)b * G_MAX * LCOLS;
    for (int i = threadIdx.x; i < ng * 24; i += COST_NT) {
        const int g = i / 24, k = i - g * 24;
        const float dx = lab[g * LCOLS + 3 + 2 * k] - lab[g * LCOLS + 1];
        const float dy = lab[g * LCOLS + 4 + 2 * k] - lab[g * LCOLS + 2];
        gr[g][k] = sqrtf(dx * dx + dy * dy);                      // torch.norm over (x,y) (boxes.py:189-197)
    }
    for (int g = threadIdx.x; g < ng; g += COST_NT) {
        gcx[g] = lab[g * LCOLS + 1]; gcy[g] = lab[g * LCOLS + 2]; gcls[g] = (int)lab[g * LCOLS];
    }
    if (slot >= 0) {
        const float* o = outputs + ((long)b * A + a) * ncols;
        for (int k = 0; k < 24; ++k) spr[k][slot] = o[2 + k];
        s_pcx[slot] = o[0]; s_pcy[slot] = o[1];
        s_so[slot] = 1.0f / (1.0f + expf(-o[26]));
        s_both[slot] = mb & mc; s_anchor[slot] = a;
    }
    __syncthreads();
    // sum over classes of the "target 0" BCE term: -max(log1p(-p), -100), p = sqrt(sigmoid(cls)*sigmoid(obj)).  Round 5: the C terms
    // of a candidate are evaluated by four threads (the compacting wave alone used to walk all C of them while three waves
    // waited) into LDS and then added by one thread in class order - the same terms, the same order of the sum.
    {
        float* ct = &terms[0][0];                                   // [COST_APB][C] while no batch is in flight (C <= 96)
        const int sl = threadIdx.x >> 2, part = threadIdx.x & 3;
        if (sl < nc) {
            const float* o = outputs + ((long)b * A + s_anchor[sl]) * ncols;
            const float so = s_so[sl];
            for (int c = part; c < C; c += 4) {
                const float p = sqrtf((1.0f / (1.0f + expf(-o[27 + c]))) * so);
                ct[sl * C + c] = -fmaxf(log1pf(-p), -100.f);
            }
        }
        __syncthreads();
        if (threadIdx.x < nc) {
            float s0 = 0.f;
            for (int c = 0; c < C; ++c) s0 += ct[threadIdx.x * C + c];
            s_s0[threadIdx.x] = s0;
        }
        __syncthreads();
    }
    // ---- the (candidate, GT) pairs of the workgroup, candidate fastest, 256 at a time.
    // Round 5: the lens of two properly intersecting circles (two divisions, two acosf, one sinf: ~9/10 of ray_giou's instructions)
    // is needed by a few per cent of the (pair, ray) items - a candidate lies in the annulus |r1 - r2| < d < r1 + r2 of few GT rays -
    // but the lanes of a wave hold 64 neighbouring anchors (512 px at stride 8) against one GT, so some lane nearly always needed it
    // and the wave-uniform skip inside ray_giou rarely fired: 229 us of VALU time on the step's critical path
    // (profiles/r04_step_timeline.csv).  Now a pass over the batch evaluates the contained / disjoint rays in place (ray_giou's
    // lens block is skipped: no lane of the wave enters it) and queues the others in LDS; the queue is then evaluated densely, every
    // lane a lens; each pair finally adds its 24 terms in ray order.  Same expressions per term, same order of the sum: bit-identical
    // pw / cost.
    const int npair = nc * ng;
    for (int base = 0; base < npair; base += COST_NT) {
        const int pi = base + threadIdx.x;
        const bool live = pi < npair;
        int g = 0, c = 0;
        float d = 0.f;
        if (threadIdx.x == 0) s_nq = 0;
        __syncthreads();                                           // also: the previous batch's terms have been read
        if (live) {
            g = pi / nc; c = pi - g * nc;
            const float ddx = gcx[g] - s_pcx[c], ddy = gcy[g] - s_pcy[c];
            d = sqrtf(ddx * ddx + ddy * ddy);
            s_d[threadIdx.x] = d;
            for (int k = 0; k < 24; ++k) {
                const float r1 = gr[g][k], r2 = spr[k][c];
                if (fabsf(r1 - r2) >= d || d >= r1 + r2) {          // contained or disjoint: no lens (the predicates of ray_giou)
                    terms[k][threadIdx.x] = 1.0f - ray_giou(r1, r2, d);
                } else {
                    const int qi = atomicAdd(&s_nq, 1);
                    queue[qi] = (unsigned short)(threadIdx.x * 24 + k);
                }
            }
        }
        __syncthreads();
        const int nq = s_nq;
        for (int i = threadIdx.x; i < nq; i += COST_NT) {
            const int it = queue[i];
            const int t = it / 24, k = it - t * 24;
            const int p2 = base + t;
            const int g2 = p2 / nc, c2 = p2 - g2 * nc;
            terms[k][t] = 1.0f - ray_giou(gr[g2][k], spr[k][c2], s_d[t]);
        }
        __syncthreads();
        if (live) {
            const int an = s_anchor[c];
            const float* o = outputs + ((long)b * A + an) * ncols;
            float acc = 0.f;
            for (int k = 0; k < 24; ++k) acc += terms[k][threadIdx.x];
            const float v = acc / 24.0f / 2.0f;                    // boxes.py:238-241
            const float p = sqrtf((1.0f / (1.0f + expf(-o[27 + gcls[g]]))) * s_so[c]);
            const float cls_cost = s_s0[c] - (-fmaxf(log1pf(-p), -100.f)) + (-fmaxf(logf(p), -100.f));
            const bool both = (s_both[c] >> g) & 1ull;
            const float cst = cls_cost + 3.0f * (-logf(v + 1e-8f)) + 100000.0f * (both ? 0.0f : 1.0f);   // losses.py:420-424
            const long idx = ((long)b * G_MAX + g) * A + an;
            pw[idx] = v;
            cost[idx] = cst;
        }
    }
}

// ------------------------------------------------------------------------------------------ a8 part 1
// block-wide arg-best over (value, index): larger value wins when LARGEST, ties -> smaller index
template <bool LARGEST>
__device__ void block_best(float& v, int& i, float* sv, int* si) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(v, o, 64);
        const int oi = __shfl_xor(i, o, 64);
        const bool take = oi >= 0 && (i < 0 || (LARGEST ? ov > v : ov < v) || (ov == v && oi < i));
        if (take) { v = ov; i = oi; }
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sv[w] = v; si[w] = i; }
    __syncthreads();
    v = sv[0]; i = si[0];
    for (int k = 1; k < 4; ++k) {
        const bool take = si[k] >= 0 && (i < 0 || (LARGEST ? sv[k] > v : sv[k] < v) || (sv[k] == v && si[k] < i));
        if (take) { v = sv[k]; i = si[k]; }
    }
    __syncthreads();
}

// One workgroup per (image, gt).  The candidate values of the row are cached in registers (PER = ceil(A/256)
// per thread, 33 at 640x640) so the 10 + k selection rounds never go back to memory; larger anchor counts
// (PER > 40, e.g. 1280x1280) take the re-reading path.
template <int PER>
__global__ __launch_bounds__(256) void dynamic_k_kernel(const float* pw, const float* cost, const int* num_gt,
                                                        const unsigned long long* in_box, const unsigned long long* in_ctr,
                                                        unsigned long long* match, int* ks, int A) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.y, g = blockIdx.x;
    if (g >= num_gt[b]) return;                                     // uniform per block
    const float* pwr = pw + ((long)b * G_MAX + g) * A;
    const float* cr = cost + ((long)b * G_MAX + g) * A;
    const unsigned long long* mb = in_box + (long)b * A;
    const unsigned long long* mc = in_ctr + (long)b * A;
    float vpw[PER > 0 ? PER : 1], vco[PER > 0 ? PER : 1];
    if constexpr (PER > 0) {
        // Every load of the row is UNCONDITIONAL and from a clamped address, so all 4 * PER of them leave in one batch.  Written as
        // "cand ? pwr[a] : -inf" the loop compiled to a branch per element with the mask loads, a wait, and the value loads behind a
        // second branch: 2 * PER dependent memory round trips in front of the first selection round - most of the kernel's time
        // (round 5, session 3: the rounds themselves were rebuilt twice before the ISA was read).  A non-candidate's slot of pw / cost
        // may hold anything; it is read and dropped.
        unsigned long long m[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int a = threadIdx.x + 256 * j;
            const int ac = a < A ? a : A - 1;
            m[j] = mb[ac] | mc[ac];
            vpw[j] = pwr[ac];
            vco[j] = cr[ac];
        }
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int a = threadIdx.x + 256 * j;
            const bool cand = a < A && m[j] != 0ull;
            vpw[j] = cand ? vpw[j] : -INFINITY;                      // never selected
            vco[j] = cand ? vco[j] : INFINITY;
        }
    }
    // top-min(10,P) largest pairwise values, summed in descending order (losses.py:452-456)
    float prev_v = INFINITY; int prev_i = -1;
    float total = 0.f;
    for (int r = 0; r < 10; ++r) {
        float bv = -INFINITY; int bi = -1;
        if constexpr (PER > 0) {
            // the element picked in the round before leaves the thread's values (a select per element, no "after the previous pick"
            // test: 5 instead of ~20 instructions per element - a round's VALU time is A elements over the CU's four SIMDs whatever
            // the thread count); the first strictly larger element wins, so equal values go by index as before
            const int d = prev_i - (int)threadIdx.x;
            int bj = -1;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float v = vpw[j];
                v = d == 256 * j ? -INFINITY : v;
                vpw[j] = v;
                const bool t = v > bv;
                bv = t ? v : bv;
                bj = t ? j : bj;
            }
            bi = bj >= 0 ? (int)threadIdx.x + 256 * bj : -1;
        } else {
            for (int a = threadIdx.x; a < A; a += 256) {
                if ((mb[a] | mc[a]) == 0ull) continue;
                const float v = pwr[a];
                const bool after = v < prev_v || (v == prev_v && a > prev_i);
                if (after && (bi < 0 || v > bv)) { bv = v; bi = a; }     // ascending a: first hit keeps the smaller index
            }
        }
        block_best<true>(bv, bi, sv, si);
        if (bi < 0) break;                                          // fewer than 10 candidates
        total += bv;
        prev_v = bv; prev_i = bi;
    }
    int k = (int)total;                                             // .int() truncates
    if (k < 1) k = 1;
    if (threadIdx.x == 0) ks[b * G_MAX + g] = k;
    // k cheapest candidates (torch.topk(largest=False)); ties -> lower anchor index
    prev_v = -INFINITY; prev_i = -1;
    for (int r = 0; r < k; ++r) {
        float bv = INFINITY; int bi = -1;
        if constexpr (PER > 0) {
            const int d = prev_i - (int)threadIdx.x;
            int bj = -1;
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                float v = vco[j];
                v = d == 256 * j ? INFINITY : v;
                vco[j] = v;
                const bool t = v < bv;
                bv = t ? v : bv;
                bj = t ? j : bj;
            }
            bi = bj >= 0 ? (int)threadIdx.x + 256 * bj : -1;
        } else {
            for (int a = threadIdx.x; a < A; a += 256) {
                if ((mb[a] | mc[a]) == 0ull) continue;
                const float v = cr[a];
                const bool after = v > prev_v || (v == prev_v && a > prev_i);
                if (after && (bi < 0 || v < bv)) { bv = v; bi = a; }
            }
        }
        block_best<false>(bv, bi, sv, si);
        if (bi < 0) break;
        if (threadIdx.x == 0) atomicOr(match + (long)b * A + bi, 1ull << g);
        prev_v = bv; prev_i = bi;
    }
}

// ------------------------------------------------------------------------------------------ a8 part 2
__global__ __launch_bounds__(256) void resolve_kernel(const unsigned long long* match, const float* pw, const float* cost,
                                                      const int* num_gt, int* matched_gt, float* matched_iou, int A) {
    const int b = blockIdx.y;
    const int a = blockIdx.x * 256 + threadIdx.x;
    if (a >= A) return;
    const unsigned long long m = match[(long)b * A + a];
    int g = -1;
    float v = 0.f;
    if (m) {
        if (__popcll(m) == 1) {
            g = __ffsll((long long)m) - 1;
        } else {                                                    // losses.py:473-476: argmin over ALL gts
            const int ng = num_gt[b];
            float best = INFINITY;
            for (int q = 0; q < ng; ++q) {
                const float c = cost[((long)b * G_MAX + q) * A + a];
                if (c < best) { best = c; g = q; }
            }
        }
        v = pw[((long)b * G_MAX + g) * A + a];
    }
    matched_gt[(long)b * A + a] = g;
    matched_iou[(long)b * A + a] = v;
}

}  // namespace

#ifdef EP24_STAMPS
extern "C" int ep24_debug_read_cand(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_cand_dbg), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

extern "C" int ep24_assign_candidates(const float* labels, const float* xs, const float* ys, const float* strides,
                                      int32_t* num_gt, uint64_t* in_box, uint64_t* in_ctr, int B, int A, void* stream) {
    EP24_REQUIRE(labels && xs && ys && strides && num_gt && in_box && in_ctr && B > 0 && A > 0, EP24_E_ARG,
                 "assign_candidates: bad arguments");
    hipLaunchKernelGGL(candidates_kernel, dim3(ep24_cdiv(A, 256), B), dim3(256), 0, (hipStream_t)stream, labels, xs, ys, strides,
                       num_gt, (unsigned long long*)in_box, (unsigned long long*)in_ctr, A);
    EP24_LAUNCH_CHECK("ep24_assign_candidates");
    return EP24_OK;
}

// Anchors [a_lo, a_hi) of every image: a (candidate, GT) pair's pw / cost do not depend on which anchors share its workgroup, so the
// rows of a head level can be taken as soon as that level's outputs exist (ep24.train: on the forward lane that produced them) and
// the launches together write exactly what one launch over [0, A) writes.
extern "C" int ep24_assign_cost_range(const float* outputs, int ncols, const float* labels, const int32_t* num_gt,
                                      const uint64_t* in_box, const uint64_t* in_ctr, float* pw, float* cost, int B, int A,
                                      int num_classes, int a_lo, int a_hi, void* stream) {
    EP24_REQUIRE(outputs && labels && num_gt && in_box && in_ctr && pw && cost, EP24_E_ARG, "assign_cost: null pointer");
    EP24_REQUIRE(ncols == 27 + num_classes, EP24_E_ARG, "assign_cost: ncols=%d != 27+%d", ncols, num_classes);
    EP24_REQUIRE(num_classes >= 1 && COST_APB * num_classes <= 24 * COST_NT, EP24_E_UNSUPPORTED, "assign_cost: at most %d classes", 24 * COST_NT / COST_APB);
    EP24_REQUIRE(B > 0 && 0 <= a_lo && a_lo <= a_hi && a_hi <= A, EP24_E_ARG, "assign_cost: anchors [%d, %d) of %d", a_lo, a_hi, A);
    if (a_lo == a_hi) return EP24_OK;
    hipLaunchKernelGGL(cost_kernel, dim3(ep24_cdiv(a_hi - a_lo, COST_APB), B), dim3(COST_NT), 0, (hipStream_t)stream, outputs, ncols, labels, num_gt,
                       (const unsigned long long*)in_box, (const unsigned long long*)in_ctr, pw, cost, A, num_classes, a_lo, a_hi);
    EP24_LAUNCH_CHECK("ep24_assign_cost");
    return EP24_OK;
}

extern "C" int ep24_assign_cost(const float* outputs, int ncols, const float* labels, const int32_t* num_gt,
                                const uint64_t* in_box, const uint64_t* in_ctr, float* pw, float* cost, int B, int A,
                                int num_classes, void* stream) {
    return ep24_assign_cost_range(outputs, ncols, labels, num_gt, in_box, in_ctr, pw, cost, B, A, num_classes, 0, A, stream);
}

extern "C" int ep24_dynamic_k(const float* pw, const float* cost, const int32_t* num_gt, const uint64_t* in_box,
                              const uint64_t* in_ctr, uint64_t* match, int32_t* ks, int B, int A, void* stream) {
    EP24_REQUIRE(pw && cost && num_gt && in_box && in_ctr && match && ks, EP24_E_ARG, "dynamic_k: null pointer");
    if (A <= 33 * 256)
        hipLaunchKernelGGL(dynamic_k_kernel<33>, dim3(G_MAX, B), dim3(256), 0, (hipStream_t)stream, pw, cost, num_gt,
                           (const unsigned long long*)in_box, (const unsigned long long*)in_ctr, (unsigned long long*)match, ks, A);
    else
        hipLaunchKernelGGL(dynamic_k_kernel<0>, dim3(G_MAX, B), dim3(256), 0, (hipStream_t)stream, pw, cost, num_gt,
                           (const unsigned long long*)in_box, (const unsigned long long*)in_ctr, (unsigned long long*)match, ks, A);
    EP24_LAUNCH_CHECK("ep24_dynamic_k");
    return EP24_OK;
}

extern "C" int ep24_assign_resolve(const uint64_t* match, const float* pw, const float* cost, const int32_t* num_gt,
                                   int32_t* matched_gt, float* matched_iou, int B, int A, void* stream) {
    EP24_REQUIRE(match && pw && cost && num_gt && matched_gt && matched_iou, EP24_E_ARG, "assign_resolve: null pointer");
    hipLaunchKernelGGL(resolve_kernel, dim3(ep24_cdiv(A, 256), B), dim3(256), 0, (hipStream_t)stream,
                       (const unsigned long long*)match, pw, cost, num_gt, matched_gt, matched_iou, A);
    EP24_LAUNCH_CHECK("ep24_assign_resolve");
    return EP24_OK;
}
