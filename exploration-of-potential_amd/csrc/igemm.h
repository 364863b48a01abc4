// ep24 - shared pieces of the bf16 implicit-GEMM convolution kernels (conv_igemm.hip: LDS-DMA tiles and the 1x1
// streaming kernel; conv_patch.hip: the halo-patch kernel of the 3x3 stride-1 layers).
//
//   D[m][n] = sum_t sum_k  S[pix(m) + off(t)][k] * Wt[n][slot(t)][k]
//
// m runs over a pixel grid [B,GH,GW]; S is an NHWC bf16 tensor with row stride ld_src; taps t carry a spatial offset
// and a weight slot.  Forward conv, stride-1 dgrad and the four parity classes of a stride-2 dgrad are all instances
// of this one gather-GEMM.
#pragma once
#include "common.h"

namespace ep24_igemm {

struct IgemmArgs {
    const bf16* src; long ld_src; int B, SH, SW;
    int GH, GW, sy, sx;
    int T; int oy[16]; int ox[16]; int wslot[16];
    const bf16* wt; int WT; int K; int N;
    void* dst; long ld_dst; int DH, DW, dsy, dsx, dy0, dx0; long dbs, dp0;   // dst pixel = n*dbs + dp0 + (gy*dsy+dy0)*DW + gx*dsx+dx0
    int accumulate;
    const float* bias;
    long long* stats; int stats_replicas;
    // fused pass 1 of the NEXT BatchNorm backward (input gradient only): D is the complete gradient of a BN+act output
    // whose pre-activation is bn_z; the epilogue adds sum(du) / sum(du*zhat) per channel to bn_sb / bn_sg
    const bf16* bn_z; long bn_ldz; const float* bn_save; const float* bn_gamma; const float* bn_beta;
    long long* bn_sg; long long* bn_sb; int bn_act;
    // inference epilogue (BatchNorm folded into weights and bias): y = act(acc + bias) + residual
    int epi_act; const bf16* epi_res; long epi_ldres; int epi_infer;
    int toff[16];               // byte offset of tap t relative to the row's (iy0, ix0) pixel
    unsigned src_bytes, wt_bytes;   // extents for the buffer descriptors of the DMA kernels
    FastDiv d_plane, d_gw;          // row index -> (n, gy, gx)
    long M;
};

constexpr int BM = 128;
constexpr int BK = 64;
constexpr int OOB = 0x7FFFFFF0;      // a buffer offset beyond every extent: the LDS-DMA writes zeros for it

// LDS tile rows are 128 B (64 bf16); the 16-byte chunk index is XOR-ed with (row & 7): ds_read_b128 of 16 rows x
// {chunk c, c+1} by a wave is bank-conflict free, also for any common shift of the 16 rows (the halo-patch kernel).
__device__ __forceinline__ int swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Epilogue of the tiled kernels.  The workgroup has NWV waves laid out WM x WN over a (WM*MT*16) x (WN*64) tile; every
// wave holds MT x 4 accumulator tiles of 16x16 with the output channels relabelled so that a lane owns 4 consecutive
// channels of a pixel (8-byte packed bf16 stores).  Does: bias, BN batch statistics (2^-20 fixed-point int64 atomics, one
// per channel per workgroup), bf16 / fp32 stores with optional accumulate, and for
//   MODE 1: fused pass 1 of the next BatchNorm backward; the z values come from `zt` (the caller prefetched this
//           lane's [MT][4 rows] x 4 channels while the main loop was still running, so nothing is exposed here),
//   MODE 2: the inference form y = act(acc + bias) + residual.
struct ZTile4 { bf16x4 v[4][4]; };      // [m-tile][row r] -> 4 channels; MT <= 4

template <int BN, bool OUT_F32, int MT, int MODE = 0, int NWV = 4>
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& p, f32x4 (&acc)[MT][4], long m0, int n0, int tile_m, char* smem,
                                               const ZTile4* zt = nullptr) {
    constexpr int WN = BN / 64, WM = NWV / WN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int frow = lane & 15, fq = lane >> 4;
    const int c0 = n0 + wn * 64 + 4 * frow;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (c0 + j < p.N) bias4[j] = p.bias[c0 + j];
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    constexpr bool bnr = MODE == 1 && !OUT_F32;
    constexpr bool infer = MODE == 2 && !OUT_F32;
    float bsc[4] = {0.f, 0.f, 0.f, 0.f}, bsh[4] = {0.f, 0.f, 0.f, 0.f}, biv[4] = {0.f, 0.f, 0.f, 0.f}, bmi[4] = {0.f, 0.f, 0.f, 0.f};
    if (bnr) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (c0 + q < p.N) {
                const float mean = p.bn_save[c0 + q], inv = p.bn_save[p.N + c0 + q];
                bsc[q] = p.bn_gamma[c0 + q] * inv; bsh[q] = p.bn_beta[c0 + q] - mean * bsc[q]; biv[q] = inv; bmi[q] = mean * inv;
            }
    }
    const bool fast_dst = (p.dsy == 1 && p.dsx == 1 && p.dy0 == 0 && p.dx0 == 0 && p.DW == p.GW && p.dp0 == 0 &&
                           p.dbs == (long)p.GH * p.GW);
#pragma unroll
    for (int i = 0; i < MT; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = m0 + wm * (MT * 16) + i * 16 + 4 * fq + r;
            if (m >= p.M) continue;
            long dpix = m;
            if (!fast_dst) {
                int n = fdiv((int)m, p.d_plane);
                int rem = (int)m - n * (p.GH * p.GW);
                int gy = fdiv(rem, p.d_gw), gx = rem - gy * p.GW;
                dpix = (long)n * p.dbs + p.dp0 + (long)(gy * p.dsy + p.dy0) * p.DW + gx * p.dsx + p.dx0;
            }
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v[q] = acc[i][q][r] + bias4[q];
                if (!bnr) { s1[q] += v[q]; s2[q] += v[q] * v[q]; }
            }
            if constexpr (OUT_F32) {
                float* d = reinterpret_cast<float*>(p.dst) + dpix * p.ld_dst + c0;
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (c0 + q < p.N) d[q] = p.accumulate ? d[q] + v[q] : v[q];
            } else {
                bf16* d = reinterpret_cast<bf16*>(p.dst) + dpix * p.ld_dst + c0;
                if (infer) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + q < p.N) {
                            float y = act_fwd(v[q], p.epi_act);
                            if (p.epi_res) y += (float)p.epi_res[dpix * p.epi_ldres + c0 + q];
                            v[q] = y;
                        }
                }
                if (c0 + 3 < p.N) {
                    if (p.accumulate) {
                        bf16x4 o = *reinterpret_cast<const bf16x4*>(d);
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += (float)o[q];
                    }
                    bf16x4 w;
#pragma unroll
                    for (int q = 0; q < 4; ++q) w[q] = (bf16)v[q];
                    *reinterpret_cast<bf16x4*>(d) = w;
                    if (bnr) {
                        const bf16x4 zz = zt ? zt->v[i][r] : *reinterpret_cast<const bf16x4*>(p.bn_z + dpix * p.bn_ldz + c0);
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float zf = (float)zz[q];
                            const float du = (float)w[q] * act_grad(zf * bsc[q] + bsh[q], p.bn_act);
                            s1[q] += du;                             // -> sum(du)
                            s2[q] += du * (zf * biv[q] - bmi[q]);    // -> sum(du * zhat)
                        }
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + q < p.N) {
                            const bf16 w = (bf16)(p.accumulate ? (float)d[q] + v[q] : v[q]);
                            d[q] = w;
                            if (bnr) {
                                const float zf = (float)p.bn_z[dpix * p.bn_ldz + c0 + q];
                                const float du = (float)w * act_grad(zf * bsc[q] + bsh[q], p.bn_act);
                                s1[q] += du;
                                s2[q] += du * (zf * biv[q] - bmi[q]);
                            }
                        }
                }
            }
        }
    }
    if (p.stats || bnr) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);           // [NWV waves][2][64]
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = s1[q], b = s2[q];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            if (fq == 0) {
                red[(wave * 2 + 0) * 64 + 4 * frow + q] = a;
                red[(wave * 2 + 1) * 64 + 4 * frow + q] = b;
            }
        }
        __syncthreads();
        long long* st = bnr ? nullptr : p.stats + (long)(tile_m % p.stats_replicas) * 2 * p.N;
        for (int i = tid; i < 2 * BN; i += NWV * 64) {
            const int which = i / BN, c = i - which * BN;
            const int wcol = c >> 6;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < WM; ++r) v += red[((r * WN + wcol) * 2 + which) * 64 + (c & 63)];
            if (n0 + c < p.N) {
                long long* dst = bnr ? (which ? p.bn_sg : p.bn_sb) + n0 + c : st + (long)which * p.N + n0 + c;
                atomicAdd((unsigned long long*)dst, (unsigned long long)to_fix(v));
            }
        }
    }
}

// Prefetch of the z values the fused BN-reduce epilogue needs (MODE 1), into registers: issued by the caller a few K
// steps before the end of its main loop.  Plain-destination launches only (dst pixel == m).
template <int MT>
__device__ __forceinline__ void load_ztile(const IgemmArgs& p, ZTile4& zt, long m0, int n0, int wm, int wn, int lane) {
    const int frow = lane & 15, fq = lane >> 4;
    const int c0 = n0 + wn * 64 + 4 * frow;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = m0 + wm * (MT * 16) + i * 16 + 4 * fq + r;
            bf16x4 v = {0, 0, 0, 0};
            if (m < p.M && c0 + 3 < p.N) v = *reinterpret_cast<const bf16x4*>(p.bn_z + m * p.bn_ldz + c0);
            zt.v[i][r] = v;
        }
}

// conv_patch.hip: 3x3 stride-1 layers (forward and input gradient).  Returns false when the shape does not fit its
// LDS budget (the caller then uses the generic tiled kernel).
bool launch_patch(const IgemmArgs& a, hipStream_t stream);

}  // namespace ep24_igemm
