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
    // inference epilogue (BatchNorm folded into weights and bias): y = act(acc + bias) + residual
    int epi_act; const bf16* epi_res; long epi_ldres; int epi_infer;
    int toff[16];               // byte offset of tap t relative to the row's (iy0, ix0) pixel
    // the same tap tables packed four bits per tap (dy + 8, dx + 8, weight slot; prepare()): the tiled kernel takes a tap's offsets
    // out of these with scalar shifts - indexing oy[] / ox[] / toff[] / wslot[] by a run-time tap is a scalar LOAD from the kernel
    // arguments, one round trip through the scalar cache per use: 32 of them, one after the other, in its prologue and two between
    // the barrier and the DMA issue of every K step (round 5)
    unsigned long long tap_dy, tap_dx, tap_slot;
    unsigned src_bytes, wt_bytes;   // extents for the buffer descriptors of the DMA kernels
    unsigned dst_bytes;             // extent of the destination (streaming kernel: buffer stores)
    FastDiv d_plane, d_gw;          // row index -> (n, gy, gx)
    long M;
    int narrow_epi;                 // A/B option (kernel_opts bit 1 of the _ex entry points): the 8-byte-per-lane epilogue stores
    // Input gradient whose epilogue also does the BatchNorm-backward REDUCTION of the layer below (bnr_z != null): what it stores is
    // that layer's dy, so sum(du) and sum(du * zhat) (du = dy * act'(bn(z))) are taken from the tile while it leaves - the reduce
    // kernel read dy and z once more for them.  [reps][..] fixed-point sums, replica stride bnr_rep_stride, as bn_act_bwd_reduce.
    const bf16* bnr_z; long bnr_ldz; const float* bnr_mean; const float* bnr_invstd; const float* bnr_gamma; const float* bnr_beta;
    long long* bnr_dgamma; long long* bnr_dbeta; long bnr_rep_stride; int bnr_reps; int bnr_act;
    unsigned bnr_z_bytes;           // extent of bnr_z for the streaming kernel's buffer loads (its BNR forms, round 5)
    // Streaming 1x1 kernel with a TRANSFORMED A operand (igemm_stream_kernel<XF>, round 5): the rows it multiplies are not stored yet.
    // It makes them from what the BatchNorm pass in front of it would have read, stores them where that pass would have, and multiplies
    // them - one dependent launch and one read of the rows less, the same values bit for bit.
    //   forward  (XF 1, 2): src = z of the producing unit, A = y = silu(bn(z)) (+ residual rows xf_aux); xf_out receives y; block 0
    //                       does what bn_act_fwd's block 0 does (save, running statistics, num_batches)
    //   backward (XF 3):    src = dy of this unit, xf_aux = its z, A = dz of bn_act_bwd_apply; xf_out receives dz (the weight gradient
    //                       reads it); block 0 publishes the two sums into the parameter gradients
    const bf16* xf_aux; long xf_ldaux; unsigned xf_aux_bytes;
    bf16* xf_out; long xf_ldout; unsigned xf_out_bytes;
    const long long* xf_stats; int xf_reps;                   // forward: [reps][2][K] 2^-20 sums; backward: reps of the two 2^-36 sums
    const float* xf_gamma; const float* xf_beta; float xf_eps, xf_momentum;
    float* xf_rmean; float* xf_rvar; long* xf_nbt; long* xf_nbt2; float* xf_save;
    const long long* xf_dgamma; const long long* xf_dbeta; float* xf_ggrad; float* xf_bgrad;
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
// per channel per workgroup), bf16 / fp32 stores with optional accumulate, and for MODE 2 the inference form
// y = act(acc + bias) + residual.
template <int BN, bool OUT_F32, int MT, int MODE = 0, int NWV = 4, bool BNR = true>      // BNR false: no fused BatchNorm-backward sums in this instantiation; MODE 1: training without the batch statistics (an input gradient has none: round 5)
__device__ __forceinline__ void igemm_epilogue(const IgemmArgs& p, f32x4 (&acc)[MT][4], long m0, int n0, int tile_m, char* smem) {
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
    constexpr bool infer = MODE == 2 && !OUT_F32;
    constexpr bool STATS = MODE != 1;
    const bool fast_dst = (p.dsy == 1 && p.dsx == 1 && p.dy0 == 0 && p.dx0 == 0 && p.DW == p.GW && p.dp0 == 0 &&
                           p.dbs == (long)p.GH * p.GW);
    // Wide-store path (bf16 output, no fused extras, 16-byte aligned rows): the wave's 64 x 64 tile is staged through LDS
    // (free after the main loop) and leaves as 16-byte stores, 8 lanes per 128-byte row segment.  The direct form below
    // stores 8 bytes per lane, 16 instructions per lane: measured with in-kernel stamps, that store tail was 10.7 k cycles
    // per 256 x 128 tile (store-issue bound, ~7 B/clk/CU) - as long as five K steps of the main loop.
    const bool wide = (MODE == 0 || MODE == 1) && !OUT_F32 && !p.narrow_epi && (p.N & 7) == 0 && (p.ld_dst & 7) == 0 &&
                      (reinterpret_cast<unsigned long long>(p.dst) & 15) == 0;
    if (wide) {
        // fused BatchNorm-backward sums: the z rows of this wave's part of the tile are requested NOW, all of them (clamped indices, no
        // conditional load), so they arrive while the accumulators go through LDS - fetched inside the store loop they were eight
        // serial round trips per workgroup (+24 us per launch: more than the reduce kernel they replace)
        const bool bnr = BNR && (MODE == 0 || MODE == 1) && p.bnr_z != nullptr;      // uniform over the launch (the host admits it only with this store path)
        bf16x8 zr[MT * 2];
        if (bnr) {
            const int zc = n0 + wn * 64 + (lane & 7) * 8;
#pragma unroll
            for (int k = 0; k < MT * 2; ++k) {
                const long m = m0 + wm * (MT * 16) + k * 8 + (lane >> 3);
                zr[k] = *reinterpret_cast<const bf16x8*>(p.bnr_z + (m < p.M ? m : 0) * p.bnr_ldz + (zc < p.N ? zc : 0));
            }
        }
        __syncthreads();                                   // every wave is out of the main loop: LDS is free
        char* stg = smem + wave * (MT * 16 * 128);         // [MT*16 rows][128 B], 16-byte chunk index XOR (row & 7)
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + 4 * fq + r;
                const bool live = m0 + wm * (MT * 16) + row < p.M;
                bf16x4 w;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float v = acc[i][q][r] + bias4[q];
                    if constexpr (STATS) {
                        const float vs = live ? v : 0.f;         // (one select per value; `if (live) { s1 += v; s2 += v * v; }` was two, and a multiply)
                        s1[q] += vs; s2[q] = fmaf(vs, vs, s2[q]);
                    }
                    w[q] = (bf16)v;
                }
                *reinterpret_cast<bf16x4*>(stg + row * 128 + (((frow >> 1) ^ (row & 7)) << 4) + (frow & 1) * 8) = w;
            }
        }
        // the same wave reads back what it wrote (LDS operations of a wave complete in order): no barrier
        const int ch = lane & 7;
        const int cc = n0 + wn * 64 + ch * 8;
        float bsc[8], bsh[8], biv[8], bmi[8], bsg[8], bsb[8];
        if (bnr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cc + j < p.N ? cc + j : 0;
                const float mean = p.bnr_mean[c], inv = p.bnr_invstd[c];
                bsc[j] = p.bnr_gamma[c] * inv; bsh[j] = p.bnr_beta[c] - mean * bsc[j]; biv[j] = inv; bmi[j] = mean * inv;
                bsg[j] = 0.f; bsb[j] = 0.f;
            }
        }
        // Round 5: a destination whose extent fits 32-bit offsets (dst_bytes != 0: every layer of the step) leaves through buffer stores -
        // an offset per chunk instead of a 64-bit pointer, rows past M / columns past N as out-of-range offsets instead of a branch round
        // the store (profiles/r05_ring_epilogue.txt: the ring's epilogue spent 2 000 - 3 200 cycles on eight such stores)
        if (!bnr && p.dst_bytes != 0u) {
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            const auto drs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16*>(p.dst), 0, p.dst_bytes, 0x00020000);
            const int rt = lane >> 3;
            const char* lsrc = stg + rt * 128 + ((ch ^ (rt & 7)) << 4);
            const long mw = m0 + wm * (MT * 16) + rt;
#pragma unroll
            for (int k = 0; k < MT * 2; ++k) {
                const long m = mw + k * 8;
                long dpix = m;
                if (!fast_dst) {
                    const int mm = m < p.M ? (int)m : 0;
                    const int n = fdiv(mm, p.d_plane);
                    const int rem = mm - n * (p.GH * p.GW);
                    const int gy = fdiv(rem, p.d_gw), gx = rem - gy * p.GW;
                    dpix = (long)n * p.dbs + p.dp0 + (long)(gy * p.dsy + p.dy0) * p.DW + gx * p.dsx + p.dx0;
                }
                const int off = (m < p.M && cc < p.N) ? (int)((dpix * p.ld_dst + cc) * 2) : OOB;
                bf16x8 v = *reinterpret_cast<const bf16x8*>(lsrc + k * 1024);
                if (p.accumulate) {                        // (out of range: zeros)
                    const bf16x8 o = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(drs, off, 0, 0));
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)o[j]);
                }
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_, v), drs, off, 0, 0);
            }
        } else
#pragma unroll
        for (int k = 0; k < MT * 2; ++k) {
            const int row = k * 8 + (lane >> 3);
            bf16x8 v = *reinterpret_cast<const bf16x8*>(stg + row * 128 + ((ch ^ (row & 7)) << 4));
            const long m = m0 + wm * (MT * 16) + row;
            if (m >= p.M || cc >= p.N) continue;
            long dpix = m;
            if (!fast_dst) {
                int n = fdiv((int)m, p.d_plane);
                int rem = (int)m - n * (p.GH * p.GW);
                int gy = fdiv(rem, p.d_gw), gx = rem - gy * p.GW;
                dpix = (long)n * p.dbs + p.dp0 + (long)(gy * p.dsy + p.dy0) * p.DW + gx * p.dsx + p.dx0;
            }
            bf16* d = reinterpret_cast<bf16*>(p.dst) + dpix * p.ld_dst + cc;
            if (p.accumulate) {
                const bf16x8 o = *reinterpret_cast<const bf16x8*>(d);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (bf16)((float)v[j] + (float)o[j]);
            }
            *reinterpret_cast<bf16x8*>(d) = v;
            if (bnr) {                                       // the expressions of bn_act_bwd_reduce_kernel on the rounded dy it would read
                const bf16x8 vz = zr[k];                     // row m of z: the destination is plain (dpix == m)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float zz = (float)vz[j];
                    const float du = (float)v[j] * act_grad(fmaf(zz, bsc[j], bsh[j]), p.bnr_act);
                    bsb[j] += du;
                    bsg[j] = fmaf(du, fmaf(zz, biv[j], -bmi[j]), bsg[j]);
                }
            }
        }
        if (bnr) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
#pragma unroll
                for (int o = 8; o < 64; o <<= 1) { bsg[j] += __shfl_xor(bsg[j], o, 64); bsb[j] += __shfl_xor(bsb[j], o, 64); }
            }
            __syncthreads();                                   // every wave is through with its staging area
            float* red = reinterpret_cast<float*>(smem);       // [NWV waves][2][64]
            if (lane < 8) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    red[(wave * 2 + 0) * 64 + ch * 8 + j] = bsg[j];
                    red[(wave * 2 + 1) * 64 + ch * 8 + j] = bsb[j];
                }
            }
            __syncthreads();
            const long rep = (long)(tile_m % p.bnr_reps) * p.bnr_rep_stride;
            for (int i = tid; i < 2 * BN; i += NWV * 64) {
                const int which = i / BN, c = i - which * BN;
                const int wcol = c >> 6;
                float t = 0.f;
#pragma unroll
                for (int r = 0; r < WM; ++r) t += red[((r * WN + wcol) * 2 + which) * 64 + (c & 63)];
                if (n0 + c < p.N) atomicAdd((unsigned long long*)((which ? p.bnr_dbeta : p.bnr_dgamma) + rep + n0 + c), (unsigned long long)to_fix_g(t));
            }
        }
    } else
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
                s1[q] += v[q]; s2[q] += v[q] * v[q];
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
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (c0 + q < p.N) d[q] = (bf16)(p.accumulate ? (float)d[q] + v[q] : v[q]);
                }
            }
        }
    }
    if (STATS && p.stats && wide) {
        // Round 5 (the 16-byte store path; as the ring's epilogue): every wave publishes the sums of its own MT*16 x 64 piece - no fold
        // through LDS, no barriers.  After the two butterflies every lane of a quarter holds the sums of its four channels; lane
        // (frow, fq) publishes channel 4 frow + fq.  (The fixed-point conversion is per wave piece now instead of per tile: equal to
        // 2^-20 rounding; the integer sums stay order-independent.)
        float a4[4], b4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = s1[q], b = s2[q];
            a += __shfl_xor(a, 16, 64); a += __shfl_xor(a, 32, 64);
            b += __shfl_xor(b, 16, 64); b += __shfl_xor(b, 32, 64);
            a4[q] = a; b4[q] = b;
        }
        const float av = fq == 0 ? a4[0] : fq == 1 ? a4[1] : fq == 2 ? a4[2] : a4[3];
        const float bv = fq == 0 ? b4[0] : fq == 1 ? b4[1] : fq == 2 ? b4[2] : b4[3];
        const int c = n0 + wn * 64 + 4 * frow + fq;
        long long* st = p.stats + (long)(tile_m % p.stats_replicas) * 2 * p.N;
        if (c < p.N) {
            atomicAdd((unsigned long long*)(st + c), (unsigned long long)to_fix(av));
            atomicAdd((unsigned long long*)(st + (long)p.N + c), (unsigned long long)to_fix(bv));
        }
    } else if (STATS && p.stats) {
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
        long long* st = p.stats + (long)(tile_m % p.stats_replicas) * 2 * p.N;
        for (int i = tid; i < 2 * BN; i += NWV * 64) {
            const int which = i / BN, c = i - which * BN;
            const int wcol = c >> 6;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < WM; ++r) v += red[((r * WN + wcol) * 2 + which) * 64 + (c & 63)];
            if (n0 + c < p.N) atomicAdd((unsigned long long*)(st + (long)which * p.N + n0 + c), (unsigned long long)to_fix(v));
        }
    }
}

// conv_patch.hip: 3x3 stride-1 layers (forward and input gradient).  Returns false when the shape does not fit its
// LDS budget (the caller then uses the generic tiled kernel).
// dry = true only answers whether the shape is taken; *rc receives the error code of a failed launch set-up.
bool launch_patch(const IgemmArgs& a, hipStream_t stream, bool dry, int* rc);
// conv_ring.hip: the same layers as 4 consumer + 4 loader waves over an LDS ring with FULL / FREE counters (round 4); same contract.
// m16 (the default): consumers with v_mfma_f32_16x16x32_bf16, bit-identical to the tiled kernel; false: the 32 x 32 x 16 form (A/B option).
// narrow_ok (A/B option): the 256 x 64 tile wherever at least 128 of them exist.  Returns 0: not this kernel, 1: the 256 x 128 ring, 2: the narrow one.
int launch_ring(const IgemmArgs& a, hipStream_t stream, bool dry, int* rc, bool m16 = true, bool narrow_ok = false);
// conv_wreg.hip (round 5): 3x3 stride-1 layers with at most 64 channels on either side and at least 256 tiles of 256 pixels - every
// B fragment of the layer in registers, persistent workgroups, one LDS window per tap row.  Same contract as launch_patch.
bool launch_wreg(const IgemmArgs& a, hipStream_t stream, bool dry, int* rc);
// ... and the ring without a patch, for any other gather-GEMM with bf16 output that fills the chip with 256 x 128 tiles.
bool launch_ring_generic(const IgemmArgs& a, hipStream_t stream, bool dry, int* rc);

}  // namespace ep24_igemm
