// ep24 - HBM-bound glue kernels of the conv graph: BN(train)+SiLU forward/backward, stem packing, SPP pools,
// nearest upsample, head decode, bias sums, SGD.  All are streaming kernels: 16-byte (8 x bf16) accesses per
// lane, consecutive lanes on consecutive addresses, grid capped and row-strided.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int MAX_BLOCKS = 2048;

__device__ __forceinline__ void load8(const bf16* p, float (&v)[8]) {
    bf16x8 t = *reinterpret_cast<const bf16x8*>(p);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)t[i];
}
__device__ __forceinline__ void store8(bf16* p, const float (&v)[8]) {
    bf16x8 t;
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = (bf16)v[i];
    *reinterpret_cast<bf16x8*>(p) = t;
}

struct RowMap {        // thread -> (row slot, 8-channel group); rows strided by rows_per_pass
    int tpr, rpb;      // threads per row, rows per block
    __device__ RowMap(int C, int nt = 256) { tpr = C >> 3; rpb = tpr >= nt ? 1 : nt / tpr; }
};

// ---------------------------------------------------------------------------------------- BN + act forward
// Prologue: the block folds the fixed-point statistics into per-channel scale / shift in LDS (one channel per
// thread, 2*reps loads).  Body: flat grid-stride over 16-byte chunks, consecutive lanes on consecutive
// addresses whatever C is, 2 chunks in flight per lane.
template <int ACT, bool RES>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const bf16* z, long ld_z, const long long* stats, int reps,
                                                         const float* gamma, const float* beta, float* rmean,
                                                         float* rvar, long* nbt, long* nbt2, float* save, bf16* y, long ld_y,
                                                         const bf16* res_, long ld_res, long M, int C, float eps,
                                                         float momentum) {
    constexpr int act = ACT;                     // compile-time: the SiLU instantiation carries no trace of the other modes
    // ... nor of the residual: tested per element at run time, `res ? r : 0` was a scalar branch per output value (32 taken branches
    // per loop iteration in the disassembly)
    const bf16* res = RES ? res_ : nullptr;
    extern __shared__ float lds[];
    // rows of 8 constants padded to 9 floats: a thread reads the 8 of ITS channel group, and with a lane stride of 9 words the
    // 64 lanes of a wave fall on 64 different banks (an 8-word stride put lanes l and l + 8 on one bank: SQ_LDS_BANK_CONFLICT was
    // 83 % of the LDS-active cycles and those 13 % of the kernel's CU-busy cycles, profiles/r02_pmc_sq.json)
    const int CP = C + (C >> 3);
    float* sc = lds;
    float* sh = lds + CP;
    auto pad = [](int c) { return c + (c >> 3); };
    for (int c = threadIdx.x; c < C; c += 256) {
        // All replica loads of a channel are issued TOGETHER (8 per batch, index clamped instead of a conditional load): written
        // as `for (r < reps) sum += stats[...]` the compiler waited for every pair before it issued the next one - eight serial
        // L2 round trips in front of every block of every BatchNorm forward launch (most of its ~6 us fixed cost).
        long long i1 = 0, i2 = 0;
        const float gmm = gamma[c], bta = beta[c];          // requested with the statistics, not behind them
        for (int rb = 0; rb < reps; rb += 8) {
            long long a[8], b[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const int rr = rb + r < reps ? rb + r : rb;
                a[r] = stats[(long)rr * 2 * C + c];
                b[r] = stats[(long)rr * 2 * C + C + c];
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                i1 += rb + r < reps ? a[r] : 0ll;
                i2 += rb + r < reps ? b[r] : 0ll;
            }
        }
        const float mean = from_fix(i1) / (float)M;
        float var = from_fix(i2) / (float)M - mean * mean;
        var = var < 0.f ? 0.f : var;
        const float invstd = rsqrtf(var + eps);
        const float s_ = gmm * invstd;
        sc[pad(c)] = s_;
        sh[pad(c)] = bta - mean * s_;
        if (blockIdx.x == 0) {
            save[c] = mean;
            save[C + c] = invstd;
            if (rmean) {
                const float unb = M > 1 ? var * (float)M / (float)(M - 1) : var;
                rmean[c] = (1.f - momentum) * rmean[c] + momentum * mean;
                rvar[c] = (1.f - momentum) * rvar[c] + momentum * unb;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (nbt) *nbt += 1;
        if (nbt2) *nbt2 += 1;                      // a merged unit: the second module's num_batches_tracked
    }
    __syncthreads();
    const int cgs = C >> 3;
    const long total = M * cgs;
    const long stride = (long)gridDim.x * 256;
    constexpr int U = 4;
    // When the grid stride is a multiple of the channel-group count every chunk of a thread belongs to ONE group, so
    // its 16 constants live in registers.  (Reading them from LDS per chunk - 16 ds_read_b32 with a 32-byte lane
    // stride, 4-way bank conflicts - made this kernel LDS-bound at ~4.9 TB/s of input: SQ_LDS_BANK_CONFLICT was 81 %
    // of the LDS-active cycles.)
    const bool fixed_group = stride % cgs == 0;
    const int g_fixed = (int)(((long)blockIdx.x * 256 + threadIdx.x) % cgs);
    float rsc[8], rsh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { rsc[j] = sc[g_fixed * 9 + j]; rsh[j] = sh[g_fixed * 9 + j]; }
    const long m_fixed = ((long)blockIdx.x * 256 + threadIdx.x) / cgs, m_step = stride / cgs;
    int it = 0;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += U * stride, ++it) {
        long mm[U];
        int gg[U];
        bf16x8 v[U], r[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {
            const long ik = i + k * stride;
            const bool ok = ik < total;
            if (fixed_group) {                                   // row = (ik - g) / cgs without a 64-bit division
                mm[k] = ok ? m_fixed + (long)(it * U + k) * m_step : -1;
                gg[k] = g_fixed;
            } else {
                mm[k] = ok ? ik / cgs : -1;
                gg[k] = ok ? (int)(ik - mm[k] * cgs) : 0;
            }
            if (ok) {
                v[k] = *reinterpret_cast<const bf16x8*>(z + mm[k] * ld_z + gg[k] * 8);
                if constexpr (RES) r[k] = *reinterpret_cast<const bf16x8*>(res + mm[k] * ld_res + gg[k] * 8);
            }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (mm[k] < 0) break;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float a_ = fixed_group ? rsc[j] : sc[gg[k] * 9 + j];
                const float b_ = fixed_group ? rsh[j] : sh[gg[k] * 9 + j];
                const float u = fmaf((float)v[k][j], a_, b_);
                if constexpr (RES) o[j] = (bf16)(act_fwd(u, act) + (float)r[k][j]);
                else o[j] = (bf16)act_fwd(u, act);
            }
            *reinterpret_cast<bf16x8*>(y + mm[k] * ld_y + gg[k] * 8) = o;
        }
    }
}

// Backward of y = act(bn(z)) written per channel with precomputed constants:
//   u  = z*sc + sh                 (sc = gamma*invstd, sh = beta - mean*sc)
//   du = dy * act'(u)
//   zhat = z*invstd - mean*invstd
//   dz = k1*du - k2 - k3*z         (k1 = gamma*invstd, k3 = k1*invstd*mean(du*zhat), k2 = k1*(mean(du) - mean*invstd*mean(du*zhat)))
// Every thread owns one 8-channel group for the whole kernel (constants in registers) and walks rows with
// several 16-byte loads in flight.

template <int UNROLL, int NT, int ACT>
__device__ __forceinline__ void bn_bwd_reduce_body(const bf16* dy, long ld_dy, const bf16* z, long ld_z,
                                                   const float* save, const float* gamma,
                                                   const float* beta, long long* dgamma, long long* dbeta,
                                                   long M, int C, int reps, const int bid, const int nblk) {
    constexpr int act = ACT;
    // Wide blocks (NT threads) so that one batch of UNROLL rows per thread covers the tensor with few blocks: the
    // per-block cost is 2*C memory-side int64 atomics, and all of a thread's loads are in flight at once.
    // [NT][8 + 1]: the two sums go through it one after the other (round 4).  As [NT][16 + 1] it took 34.8 KB per workgroup - and beside
    // the two weight-gradient workgroups of a CU (64 KB each) that did not fit the 160 KB, so in the step this kernel could only start
    // on a CU when a weight-gradient workgroup had left it: 17 us alone, 32 us in the replayed step, the most stretched kernel of the
    // main lane.  Same values added in the same order as before: bit-identical sums.
    __shared__ float red[NT][8 + 1];
    const RowMap rm(C, NT);
    const int tid = threadIdx.x;
    for (int cg0 = 0; cg0 < (C >> 3); cg0 += NT) {               // only loops when C > 8*NT
        const int cg = cg0 + tid % rm.tpr;
        const int slot = rm.tpr >= NT ? 0 : tid / rm.tpr;
        const bool active = slot < rm.rpb && cg < (C >> 3);
        float sg[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sb[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (active) {
            const long step = (long)nblk * rm.rpb;
            const long m_first = (long)bid * rm.rpb + slot;
            bf16x8 vdy[UNROLL], vz[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {                  // first batch issued before the constants are needed
                // every load is issued (a row past M re-reads row m_first and is never used): guarded by `if (mm < M)` the
                // compiler put an s_waitcnt vmcnt(0) in front of each pair - four serial round trips, the whole kernel on a 20x20 layer
                const long mm = m_first + k * step < M ? m_first + k * step : (m_first < M ? m_first : 0);
                vdy[k] = *reinterpret_cast<const bf16x8*>(dy + mm * ld_dy + cg * 8);
                vz[k] = *reinterpret_cast<const bf16x8*>(z + mm * ld_z + cg * 8);
            }
            // ... and they STAY in front of the constants: left to itself the scheduler requested the eight constant vectors first, waited
            // for all of them (their scale / shift arithmetic), and only then issued these eight loads - a second serial round trip in
            // every block of every reduce launch (round 5, from the ISA)
            __builtin_amdgcn_sched_barrier(0);
            float sc[8], sh[8], iv[8], mi[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = cg * 8 + j;
                const float mean = save[c], inv = save[C + c];
                sc[j] = gamma[c] * inv; sh[j] = beta[c] - mean * sc[j]; iv[j] = inv; mi[j] = mean * inv;
            }
            for (long m = m_first; m < M; m += UNROLL * step) {
#pragma unroll
                for (int k = 0; k < UNROLL; ++k) {
                    if (m + k * step >= M) break;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const float zz = (float)vz[k][j];
                        const float du = (float)vdy[k][j] * act_grad(fmaf(zz, sc[j], sh[j]), act);
                        sb[j] += du;
                        sg[j] = fmaf(du, fmaf(zz, iv[j], -mi[j]), sg[j]);
                    }
                }
                if (m + UNROLL * step < M) {                      // another batch follows (uniform per thread): all of it is issued
#pragma unroll
                    for (int k = 0; k < UNROLL; ++k) {
                        const long mn = m + (UNROLL + k) * step;
                        const long mm = mn < M ? mn : m;
                        vdy[k] = *reinterpret_cast<const bf16x8*>(dy + mm * ld_dy + cg * 8);
                        vz[k] = *reinterpret_cast<const bf16x8*>(z + mm * ld_z + cg * 8);
                    }
                }
            }
        }
        const int ngrp = min(min(rm.tpr, NT), (C >> 3) - cg0);            // (<= NT: with C > 8 NT a pass covers NT groups, not tpr)
#pragma unroll
        for (int half = 0; half < 2; ++half) {                   // 0: sum of du * zhat (dgamma), 1: sum of du (dbeta)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid][j] = half ? sb[j] : sg[j];
            __syncthreads();
            // thread t sums value t%8 of channel group t/8 over the block's row slots and publishes it
            for (int t = tid; t < ngrp * 8; t += NT) {
                const int g = t >> 3, v = t & 7;
                float a = 0.f;
                for (int sl = 0; sl < rm.rpb; ++sl) a += red[sl * rm.tpr + g][v];
                // replica (block index) % reps of the sums ([reps][2][C]): with one copy every block of the launch added to the SAME 2 C
                // addresses at about the same time - 256 serial adds per address at the memory-side atomic units, ~5 us of tail
                long long* dst = (half ? dbeta : dgamma) + (long)(bid % reps) * 2 * C + (cg0 + g) * 8 + v;
                atomicAdd((unsigned long long*)dst, (unsigned long long)to_fix_g(a));
            }
            __syncthreads();
        }
    }
}

template <int UNROLL, int NT, int ACT>
__global__ __launch_bounds__(NT) void bn_act_bwd_reduce_kernel(const bf16* dy, long ld_dy, const bf16* z, long ld_z,
                                                               const float* save, const float* gamma,
                                                               const float* beta, long long* dgamma, long long* dbeta,
                                                               long M, int C, int reps) {
    bn_bwd_reduce_body<UNROLL, NT, ACT>(dy, ld_dy, z, ld_z, save, gamma, beta, dgamma, dbeta, M, C, reps, (int)blockIdx.x, (int)gridDim.x);
}

template <int UNROLL, int ACT, bool ACC, bool COH>
__device__ __forceinline__ void bn_bwd_apply_body(const bf16* dy, long ld_dy, const bf16* z, long ld_z,
                                                  const float* save, const float* gamma, const float* beta,
                                                  const long long* dgamma, const long long* dbeta, float* ggrad,
                                                  float* bgrad, bf16* dz, long ld_dz, long M, int C, int reps, const int bid, const int nblk) {
    constexpr int act = ACT;
    const RowMap rm(C);
    const int tid = threadIdx.x;
    const float invM = 1.f / (float)M;
    // fold the replicas of the two sums ([reps][2][C] fixed point) once per block: thread t takes entries t, t + 256, ... of the
    // 2 C, all replica loads of an entry issued together (index clamped, never a conditional load); exact integer sums, so the
    // result does not depend on which block added to which replica
    extern __shared__ float folded[];                           // [2][C]: sum of du * zhat, sum of du
    for (int i = tid; i < 2 * C; i += 256) {
        const long long* src = i < C ? dgamma + i : dbeta + (i - C);
        long long acc = 0;
        bool bad = false;                                       // a replica word that carries a NaN marker (common.h)
        for (int rb = 0; rb < reps; rb += 8) {
            long long a[8];
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                const long long* q = src + (long)(rb + r < reps ? rb + r : rb) * 2 * C;
                // COH (the one-launch form): the sums were added by THIS kernel's other workgroups - a device-scope load, past the
                // caches that are only made coherent at kernel boundaries
                a[r] = COH ? __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *q;
            }
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                acc += rb + r < reps ? a[r] : 0ll;
                bad = bad || fixg_bad(a[r]);
            }
        }
        folded[i] = bad ? __builtin_nanf("") : from_fix_g(acc);
    }
    __syncthreads();
    for (int cg = tid % rm.tpr; cg < (C >> 3); cg += 256) {      // only loops when C > 2048
        const int slot = rm.tpr >= 256 ? 0 : tid / rm.tpr;
        if (slot >= rm.rpb) break;
        float sc[8], sh[8], k1[8], k2[8], k3[8];
        // The 8 channels' constants come in as 16-byte vector loads, all issued before the first is used.  Channel by channel (with
        // the conditional `ggrad[c] += ...` between them, which the loads of the next channel may not pass) the prologue was eight
        // serial L2 round trips in every block: ~3 us of every launch.
        float cm[8], ci[8], cg_[8], cb[8], csg[8], csb[8];
        {
            const f32x4* pm = reinterpret_cast<const f32x4*>(save + cg * 8);
            const f32x4* pi = reinterpret_cast<const f32x4*>(save + C + cg * 8);
            const f32x4* pg = reinterpret_cast<const f32x4*>(gamma + cg * 8);
            const f32x4* pb = reinterpret_cast<const f32x4*>(beta + cg * 8);
            const f32x4 m0 = pm[0], m1 = pm[1], i0 = pi[0], i1 = pi[1], g0 = pg[0], g1 = pg[1], b0 = pb[0], b1 = pb[1];
            const f32x4 s0 = *reinterpret_cast<const f32x4*>(folded + cg * 8), s1 = *reinterpret_cast<const f32x4*>(folded + cg * 8 + 4);
            const f32x4 t0 = *reinterpret_cast<const f32x4*>(folded + C + cg * 8), t1 = *reinterpret_cast<const f32x4*>(folded + C + cg * 8 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cm[j] = m0[j]; cm[4 + j] = m1[j]; ci[j] = i0[j]; ci[4 + j] = i1[j]; cg_[j] = g0[j]; cg_[4 + j] = g1[j]; cb[j] = b0[j]; cb[4 + j] = b1[j];
                csg[j] = s0[j]; csg[4 + j] = s1[j]; csb[j] = t0[j]; csb[4 + j] = t1[j];
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float mean = cm[j], inv = ci[j], g = cg_[j];
            const float sg_ = csg[j], sb_ = csb[j];
            sc[j] = g * inv; sh[j] = cb[j] - mean * sc[j];
            k1[j] = g * inv;
            k3[j] = k1[j] * inv * (sg_ * invM);
            k2[j] = k1[j] * (sb_ * invM) - k3[j] * mean;
        }
        if (bid == 0 && slot == 0 && ggrad) {          // publish this call's sums into the parameter gradients
            // as four 16-byte read-modify-writes issued together: element by element this was 16 serial round trips in block 0,
            // which every launch then waited for
            f32x4* pgg = reinterpret_cast<f32x4*>(ggrad + cg * 8);
            f32x4* pbg = reinterpret_cast<f32x4*>(bgrad + cg * 8);
            f32x4 a0 = pgg[0], a1 = pgg[1], c0 = pbg[0], c1 = pbg[1];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0[j] += csg[j]; a1[j] += csg[4 + j];
                c0[j] += csb[j]; c1[j] += csb[4 + j];
            }
            pgg[0] = a0; pgg[1] = a1; pbg[0] = c0; pbg[1] = c1;
        }
        const long step = (long)nblk * rm.rpb;
        for (long m = (long)bid * rm.rpb + slot; m < M; m += UNROLL * step) {
            bf16x8 vdy[UNROLL], vz[UNROLL];
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const long mm = m + k * step;
                if (mm < M) {
                    vdy[k] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(dy + mm * ld_dy + cg * 8));
                    vz[k] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(z + mm * ld_z + cg * 8));
                }
            }
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) {
                const long mm = m + k * step;
                if (mm >= M) break;
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float zz = (float)vz[k][j];
                    const float du = (float)vdy[k][j] * act_grad(fmaf(zz, sc[j], sh[j]), act);
                    o[j] = (bf16)fmaf(-k3[j], zz, fmaf(k1[j], du, -k2[j]));
                }
                if (ACC) {                        // pre-activation BN over a shared input (DenseNet): dz collects every consumer
                    const bf16x8 old = *reinterpret_cast<const bf16x8*>(dz + mm * ld_dz + cg * 8);
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (bf16)((float)old[j] + (float)o[j]);
                }
                *reinterpret_cast<bf16x8*>(dz + mm * ld_dz + cg * 8) = o;
            }
        }
    }
}

template <int UNROLL, int ACT, bool ACC = false>
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const bf16* dy, long ld_dy, const bf16* z, long ld_z,
                                                               const float* save, const float* gamma, const float* beta,
                                                               const long long* dgamma, const long long* dbeta, float* ggrad,
                                                               float* bgrad, bf16* dz, long ld_dz, long M, int C, int reps) {
    bn_bwd_apply_body<UNROLL, ACT, ACC, false>(dy, ld_dy, z, ld_z, save, gamma, beta, dgamma, dbeta, ggrad, bgrad, dz, ld_dz, M, C, reps,
                                               (int)blockIdx.x, (int)gridDim.x);
}

// Round 5: both backward passes of a unit as ONE launch - reduce, a grid-wide arrive / wait on a counter in global
// memory, apply.  The grid is at most one workgroup per CU (256 threads, 9 KB + 2 C floats of LDS, <= 128 registers: it fits beside the
// weight-gradient lane's two workgroups per CU, so every workgroup is resident and the wait cannot starve), every workgroup walks the
// same rows in both passes (they come back from its XCD's L2 / the Infinity Cache), and one dependent-launch boundary per layer
// disappears.  The wait is a bounded spin: a give-up is counted (ep24_conv_ring_timeouts adds it) and carries on - wrong numbers and a
// failed test, never a hung GPU.  The sums are read back with device-scope loads (bn_bwd_apply_body<COH>).
__device__ unsigned g_bn_barrier_timeouts;

template <int ACT>
__global__ __launch_bounds__(256) void bn_act_bwd_fused_kernel(const bf16* dy, long ld_dy, const bf16* z, long ld_z,
                                                               const float* save, const float* gamma, const float* beta,
                                                               long long* dgamma, long long* dbeta, float* ggrad, float* bgrad,
                                                               bf16* dz, long ld_dz, long M, int C, int reps, unsigned* bar) {
    const int bid = (int)blockIdx.x, nblk = (int)gridDim.x;
    bn_bwd_reduce_body<4, 256, ACT>(dy, ld_dy, z, ld_z, save, gamma, beta, dgamma, dbeta, M, C, reps, bid, nblk);
    __syncthreads();                                         // every wave's atomics are acknowledged (s_waitcnt vmcnt(0) in front of the barrier)
    if (threadIdx.x == 0) {
        __threadfence();
        __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        int tries = 0;
        while (__hip_atomic_load(bar, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nblk) {
            __builtin_amdgcn_s_sleep(2);
            if (++tries > (1 << 16)) { atomicAdd(&g_bn_barrier_timeouts, 1u); break; }       // ~20 ms: orders beyond any real wait
        }
    }
    __syncthreads();
    bn_bwd_apply_body<4, ACT, false, true>(dy, ld_dy, z, ld_z, save, gamma, beta, dgamma, dbeta, ggrad, bgrad, dz, ld_dz, M, C, reps, bid, nblk);
}

int flat_grid(long M, int C, int per_thread) {       // flat 16-byte chunks, `per_thread` chunks per lane
    long blocks = (M * (C >> 3) + 256L * per_thread - 1) / (256L * per_thread);
    return (int)(blocks < 1 ? 1 : (blocks > MAX_BLOCKS ? MAX_BLOCKS : blocks));
}
int rows_grid(long M, int C, int rows_per_thread, int max_blocks, int nt = 256) {   // fixed channel group per thread, strided rows
    int tpr = C >> 3;
    int rpb = tpr >= nt ? 1 : nt / tpr;
    long blocks = (M + (long)rpb * rows_per_thread - 1) / ((long)rpb * rows_per_thread);
    return (int)(blocks < 1 ? 1 : (blocks > max_blocks ? max_blocks : blocks));
}

// ---------------------------------------------------------------------------------------- stem packing
// 32 output pixels per workgroup.  Phase 1: one (tap, pixel) item per thread, pixels fastest, so a wave reads
// 8-byte pieces (the two columns of a 2x2 Focus patch) of consecutive pixels - contiguous image rows; the 12 values
// of the item go to an LDS row image.  Phase 2: the 224-byte rows leave LDS as 16-byte chunks, consecutive lanes on
// consecutive addresses.  (The one-chunk-per-thread form did 8 scattered 4-byte reads per lane: 0.44 ms at -l.)
__global__ __launch_bounds__(256) void stem_pack_kernel(const float* img, bf16* rows, int B, int IH, int IW, int ld) {
    constexpr int TP = 32;
    __shared__ __attribute__((aligned(16))) bf16 tile[TP][120];
    const int FH = IH >> 1, FW = IW >> 1;
    const long npix = (long)B * FH * FW;
    const int chunks = ld >> 3;
    for (long p0 = (long)blockIdx.x * TP; p0 < npix; p0 += (long)gridDim.x * TP) {
        for (int it = threadIdx.x; it < 9 * TP; it += 256) {
            const int tap = it / TP, lp = it - tap * TP;
            const long pix = p0 + lp;
            float v[12] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (pix < npix) {
                const int n = (int)(pix / ((long)FH * FW));
                const int rem = (int)(pix - (long)n * FH * FW);
                const int oy = rem / FW, ox = rem - oy * FW;
                const int fy = oy + tap / 3 - 1, fx = ox + tap % 3 - 1;
                if (fy >= 0 && fy < FH && fx >= 0 && fx < FW) {
#pragma unroll
                    for (int ch = 0; ch < 3; ++ch)
#pragma unroll
                        for (int yp = 0; yp < 2; ++yp) {
                            const float2 t = *reinterpret_cast<const float2*>(img + (((long)n * 3 + ch) * IH + 2 * fy + yp) * IW + 2 * fx);
                            v[(0 * 2 + yp) * 3 + ch] = t.x;          // patch = (x parity) * 2 + (y parity): TL, BL, TR, BR
                            v[(1 * 2 + yp) * 3 + ch] = t.y;
                        }
                }
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) tile[lp][tap * 12 + j] = (bf16)v[j];
        }
        if (threadIdx.x < TP) {
#pragma unroll
            for (int j = 108; j < 120; ++j) tile[threadIdx.x][j] = (bf16)0.f;
        }
        __syncthreads();
        for (int it = threadIdx.x; it < TP * chunks; it += 256) {
            const int lp = it / chunks, c = it - lp * chunks;
            const long pix = p0 + lp;
            if (pix >= npix) continue;
            bf16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
            if (c * 8 < 120) o = *reinterpret_cast<const bf16x8*>(&tile[lp][c * 8]);
            *reinterpret_cast<bf16x8*>(rows + pix * ld + c * 8) = o;
        }
        __syncthreads();
    }
}

// Focus as a tensor (network_blocks.py Focus.forward: cat(top-left, bottom-left, top-right, bottom-right) of the 2x2 pixel
// patches): fp32 NCHW image -> bf16 [B][IH/2][IW/2][16], channel (x parity * 2 + y parity) * 3 + c, channels 12..15 zero.  One
// pixel per thread: a wave reads 512 contiguous bytes per (channel, row) and writes 2 KB contiguous.
__global__ __launch_bounds__(256) void focus_pack_kernel(const float* img, bf16* out, int B, int IH, int IW) {
    const int FH = IH >> 1, FW = IW >> 1;
    const long npix = (long)B * FH * FW;
    for (long pix = (long)blockIdx.x * 256 + threadIdx.x; pix < npix; pix += (long)gridDim.x * 256) {
        const int n = (int)(pix / ((long)FH * FW));
        const int rem = (int)(pix - (long)n * FH * FW);
        const int fy = rem / FW, fx = rem - fy * FW;
        float v[12];
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)
#pragma unroll
            for (int yp = 0; yp < 2; ++yp) {
                const float2 t = *reinterpret_cast<const float2*>(img + (((long)n * 3 + ch) * IH + 2 * fy + yp) * IW + 2 * fx);
                v[(0 * 2 + yp) * 3 + ch] = t.x;
                v[(1 * 2 + yp) * 3 + ch] = t.y;
            }
        bf16x8 lo, hi = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 8; ++j) lo[j] = (bf16)v[j];
#pragma unroll
        for (int j = 8; j < 12; ++j) hi[j - 8] = (bf16)v[j];
        *reinterpret_cast<bf16x8*>(out + pix * 16) = lo;
        *reinterpret_cast<bf16x8*>(out + pix * 16 + 8) = hi;
    }
}

// ---------------------------------------------------------------------------------------- SPP pools
__global__ __launch_bounds__(256) void spp_fwd_kernel(const bf16* x, long ld_x, bf16* y5, bf16* y9, bf16* y13, long ld_y,
                                                      uint8_t* idx, int B, int H, int W, int C) {
    const int cgs = C >> 3;
    const long total = (long)B * H * W * cgs;
    const long plane = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int n = (int)(pix / (H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem - py * W;
        float m5[8], m9[8], m13[8];
        uint8_t i5[8], i9[8], i13[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m5[j] = m9[j] = m13[j] = -INFINITY; i5[j] = i9[j] = i13[j] = 0x88; }
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = py + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = px + dx;
                if (xx < 0 || xx >= W) continue;
                float v[8];
                load8(x + ((long)(n * H + yy) * W + xx) * ld_x + cg * 8, v);
                const uint8_t code = (uint8_t)(((dy + 8) << 4) | (dx + 8));
                const bool in9 = dy >= -4 && dy <= 4 && dx >= -4 && dx <= 4;
                const bool in5 = dy >= -2 && dy <= 2 && dx >= -2 && dx <= 2;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (v[j] > m13[j] || v[j] != v[j]) { m13[j] = v[j]; i13[j] = code; }
                    if (in9 && (v[j] > m9[j] || v[j] != v[j])) { m9[j] = v[j]; i9[j] = code; }
                    if (in5 && (v[j] > m5[j] || v[j] != v[j])) { m5[j] = v[j]; i5[j] = code; }
                }
            }
        }
        store8(y5 + pix * ld_y + cg * 8, m5);
        store8(y9 + pix * ld_y + cg * 8, m9);
        store8(y13 + pix * ld_y + cg * 8, m13);
        uint8_t* ip = idx + pix * C + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ip[j] = i5[j]; ip[plane + j] = i9[j]; ip[2 * plane + j] = i13[j]; }
    }
}

// Separable form (same results, ~8x less work): a horizontal pass leaves, per pixel and window size, the row maximum
// and the FIRST column offset that attains it; the vertical pass then takes the first row offset whose row maximum is
// the window maximum.  Row-major first-maximum order = smallest dy, then smallest dx, which is exactly that pair.
__global__ __launch_bounds__(256) void spp_row_kernel(const bf16* x, long ld_x, bf16* h, uint8_t* hx, int B, int H, int W, int C) {
    const int cgs = C >> 3;
    const long total = (long)B * H * W * cgs;
    const long plane = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int px = (int)(pix % W);
        float m5[8], m9[8], m13[8];
        uint8_t i5[8], i9[8], i13[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m5[j] = m9[j] = m13[j] = -INFINITY; i5[j] = i9[j] = i13[j] = 8; }
#pragma unroll
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = px + dx;
            if (xx < 0 || xx >= W) continue;
            float v[8];
            load8(x + (pix + dx) * ld_x + cg * 8, v);
            const uint8_t code = (uint8_t)(dx + 8);
            const bool in9 = dx >= -4 && dx <= 4, in5 = dx >= -2 && dx <= 2;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (v[j] > m13[j] || v[j] != v[j]) { m13[j] = v[j]; i13[j] = code; }
                if (in9 && (v[j] > m9[j] || v[j] != v[j])) { m9[j] = v[j]; i9[j] = code; }
                if (in5 && (v[j] > m5[j] || v[j] != v[j])) { m5[j] = v[j]; i5[j] = code; }
            }
        }
        bf16* hp = h + pix * C + cg * 8;
        store8(hp, m5); store8(hp + plane, m9); store8(hp + 2 * plane, m13);
        uint8_t* ip = hx + pix * C + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ip[j] = i5[j]; ip[plane + j] = i9[j]; ip[2 * plane + j] = i13[j]; }
    }
}

__global__ __launch_bounds__(256) void spp_col_kernel(const bf16* h, const uint8_t* hx, bf16* y5, bf16* y9, bf16* y13, long ld_y,
                                                      uint8_t* idx, int B, int H, int W, int C) {
    const int cgs = C >> 3;
    const long total = (long)B * H * W * cgs;
    const long plane = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int py = (int)((pix / W) % H);
        float m[3][8];
        uint8_t id[3][8];
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
            for (int j = 0; j < 8; ++j) { m[w][j] = -INFINITY; id[w][j] = 0x88; }
#pragma unroll
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = py + dy;
            if (yy < 0 || yy >= H) continue;
            const long q = (pix + (long)dy * W) * C + cg * 8;
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const int r = 2 + 2 * w;                       // window radius 2, 4, 6
                if (dy < -r || dy > r) continue;
                float v[8];
                load8(h + w * plane + q, v);
                const uint2 cx = *reinterpret_cast<const uint2*>(hx + w * plane + q);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned c = ((j < 4 ? cx.x : cx.y) >> (8 * (j & 3))) & 0xFF;
                    if (v[j] > m[w][j] || v[j] != v[j]) { m[w][j] = v[j]; id[w][j] = (uint8_t)(((dy + 8) << 4) | c); }
                }
            }
        }
        store8(y5 + pix * ld_y + cg * 8, m[0]);
        store8(y9 + pix * ld_y + cg * 8, m[1]);
        store8(y13 + pix * ld_y + cg * 8, m[2]);
        uint8_t* ip = idx + pix * C + cg * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) { ip[j] = id[0][j]; ip[plane + j] = id[1][j]; ip[2 * plane + j] = id[2][j]; }
    }
}

__global__ __launch_bounds__(256) void spp_bwd_kernel(const bf16* dy5, const bf16* dy9, const bf16* dy13, long ld_dy,
                                                      const uint8_t* idx, bf16* dx, long ld_dx, int accumulate, int B,
                                                      int H, int W, int C) {
    const int cgs = C >> 3;
    const long total = (long)B * H * W * cgs;
    const long plane = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int n = (int)(pix / (H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem - py * W;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        // output pixel o = this - (dy,dx) selected this pixel iff its stored offset equals (dy,dx)
        for (int dy = -6; dy <= 6; ++dy) {
            const int oy = py - dy;
            if (oy < 0 || oy >= H) continue;
            for (int dxo = -6; dxo <= 6; ++dxo) {
                const int ox = px - dxo;
                if (ox < 0 || ox >= W) continue;
                const long op = (long)(n * H + oy) * W + ox;
                const uint8_t code = (uint8_t)(((dy + 8) << 4) | (dxo + 8));
                const uint8_t* ip = idx + op * C + cg * 8;
                // the 8 winner codes of a window arrive as one 8-byte load; a matching byte XORs to zero
                const unsigned rep = code * 0x01010101u;
                auto gather = [&](const bf16* g, const uint8_t* codes) {
                    uint2 c = *reinterpret_cast<const uint2*>(codes);
                    c.x ^= rep; c.y ^= rep;
                    float v[8];
                    load8(g + op * ld_dy + cg * 8, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        if ((((j < 4 ? c.x : c.y) >> (8 * (j & 3))) & 0xFFu) == 0u) acc[j] += v[j];
                };
                gather(dy13, ip + 2 * plane);
                if (dy >= -4 && dy <= 4 && dxo >= -4 && dxo <= 4) gather(dy9, ip + plane);
                if (dy >= -2 && dy <= 2 && dxo >= -2 && dxo <= 2) gather(dy5, ip);
            }
        }
        bf16* d = dx + pix * ld_dx + cg * 8;
        if (accumulate) {
            float o[8];
            load8(d, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += o[j];
        }
        store8(d, acc);
    }
}

// The routing done separably in LDS.  A k x k max pool is a row pass after a column pass, and the stored 2-D winner code of an
// output (y, x) is (dy, dx) = (row offset chosen in column x, column offset chosen in row y + dy).  One workgroup per (image,
// 8-channel group) stages the three incoming gradients and codes of the whole H x W map (72 bytes per pixel), then per pool:
//   vertical    g_h[r][x] = sum over y in the window with dy(y, x) = r - y of dy_pool[y][x]   (fp32, in LDS; the dx of those
//               outputs is the same for all of them - it is the row pass's choice at (r, x) - and is kept next to the sum)
//   horizontal  dx[r][c] += sum over x in the window with that choice = c - x of g_h[r][x]
// 2 x (5 + 9 + 13) window positions per pixel instead of 25 + 81 + 169: the gather above spent 229 us per step at -l (compute
// bound on byte compares, not on memory).  Sums are associated differently (per column first), so results agree with the gather
// to fp32 rounding of a handful of terms, not bit for bit; the order is fixed, so they are reproducible.
__global__ __launch_bounds__(256) void spp_bwd_lds_kernel(const bf16* dy5, const bf16* dy9, const bf16* dy13, long ld_dy,
                                                          const uint8_t* idx, bf16* dx, long ld_dx, int accumulate, int B,
                                                          int H, int W, int C) {
    extern __shared__ __attribute__((aligned(16))) char spp_lds[];
    const int HW = H * W;
    bf16x8* gv = reinterpret_cast<bf16x8*>(spp_lds);                          // [3][HW] gradients of the 5 / 9 / 13 pools
    uint2* cv = reinterpret_cast<uint2*>(spp_lds + (size_t)3 * HW * 16);      // [3][HW] winner codes, one byte per channel
    float* gh = reinterpret_cast<float*>(spp_lds + (size_t)3 * HW * 24);      // [HW][8] column sums of the pool in work
    uint2* ds = reinterpret_cast<uint2*>(spp_lds + (size_t)3 * HW * 24 + (size_t)HW * 32);   // [HW] dx + 8 per channel, 0xFF = none
    const int cg = blockIdx.x, n = blockIdx.y;
    const long plane = (long)B * HW * C;
    const long base = (long)n * HW;
    for (int q = threadIdx.x; q < HW; q += 256) {
        const long op = base + q;
        gv[q] = *reinterpret_cast<const bf16x8*>(dy5 + op * ld_dy + cg * 8);
        gv[HW + q] = *reinterpret_cast<const bf16x8*>(dy9 + op * ld_dy + cg * 8);
        gv[2 * HW + q] = *reinterpret_cast<const bf16x8*>(dy13 + op * ld_dy + cg * 8);
        const uint8_t* ip = idx + op * C + cg * 8;
        cv[q] = *reinterpret_cast<const uint2*>(ip);
        cv[HW + q] = *reinterpret_cast<const uint2*>(ip + plane);
        cv[2 * HW + q] = *reinterpret_cast<const uint2*>(ip + 2 * plane);
    }
    __syncthreads();
    constexpr int NPASS = 4;                                                   // pixels per thread: H * W <= 1024
    float acc[NPASS][8];
#pragma unroll
    for (int t = 0; t < NPASS; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    for (int pool = 2; pool >= 0; --pool) {                                    // 13, 9, 5: the gather's order per window position
        const int R = 2 * pool + 2;
        // ---- vertical
        for (int q = threadIdx.x; q < HW; q += 256) {
            const int r = q / W, x = q - r * W;
            float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            unsigned char sel[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) sel[j] = 0xFF;
            for (int dy = -R; dy <= R; ++dy) {
                const int y = r - dy;
                if (y < 0 || y >= H) continue;
                const int o = y * W + x;
                const uint2 c = cv[pool * HW + o];
                const bf16x8 t = gv[pool * HW + o];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned byte = ((j < 4 ? c.x : c.y) >> (8 * (j & 3))) & 0xFFu;
                    if ((int)(byte >> 4) == dy + 8) { s[j] += (float)t[j]; sel[j] = (unsigned char)(byte & 15u); }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) gh[q * 8 + j] = s[j];
            uint2 d;
            d.x = sel[0] | (sel[1] << 8) | (sel[2] << 16) | ((unsigned)sel[3] << 24);
            d.y = sel[4] | (sel[5] << 8) | (sel[6] << 16) | ((unsigned)sel[7] << 24);
            ds[q] = d;
        }
        __syncthreads();
        // ---- horizontal
#pragma unroll
        for (int t = 0; t < NPASS; ++t) {
            const int q = threadIdx.x + 256 * t;
            if (q < HW) {
                const int r = q / W, c = q - r * W;
                for (int dxo = -R; dxo <= R; ++dxo) {
                    const int x = c - dxo;
                    if (x < 0 || x >= W) continue;
                    const int o = r * W + x;
                    const uint2 d = ds[o];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned byte = ((j < 4 ? d.x : d.y) >> (8 * (j & 3))) & 0xFFu;
                        if ((int)byte == dxo + 8) acc[t][j] += gh[o * 8 + j];
                    }
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < NPASS; ++t) {
        const int q = threadIdx.x + 256 * t;
        if (q < HW) {
            bf16* d = dx + (base + q) * ld_dx + cg * 8;
            float out[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) out[j] = acc[t][j];
            if (accumulate) {
                float o[8];
                load8(d, o);
#pragma unroll
                for (int j = 0; j < 8; ++j) out[j] += o[j];
            }
            store8(d, out);
        }
    }
}

// ---------------------------------------------------------------------------------------- upsample / copy
__global__ __launch_bounds__(256) void upsample2_fwd_kernel(const bf16* x, long ld_x, bf16* y, long ld_y, int B, int H,
                                                            int W, int C) {
    const int cgs = C >> 3;
    const long total = (long)B * 2 * H * 2 * W * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long opix = i / cgs;
        const int n = (int)(opix / (4 * H * W));
        const int rem = (int)(opix - (long)n * 4 * H * W);
        const int oy = rem / (2 * W), ox = rem - oy * 2 * W;
        const long ip = (long)(n * H + (oy >> 1)) * W + (ox >> 1);
        *reinterpret_cast<bf16x8*>(y + opix * ld_y + cg * 8) = *reinterpret_cast<const bf16x8*>(x + ip * ld_x + cg * 8);
    }
}

__global__ __launch_bounds__(256) void upsample2_bwd_kernel(const bf16* dy, long ld_dy, bf16* dx, long ld_dx,
                                                            int accumulate, int B, int H, int W, int C) {
    const int cgs = C >> 3;
    const long total = (long)B * H * W * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long pix = i / cgs;
        const int n = (int)(pix / (H * W));
        const int rem = (int)(pix - (long)n * H * W);
        const int py = rem / W, px = rem - py * W;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                float v[8];
                load8(dy + ((long)(n * 2 * H + 2 * py + a) * 2 * W + 2 * px + b) * ld_dy + cg * 8, v);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[j];
            }
        bf16* d = dx + pix * ld_dx + cg * 8;
        if (accumulate) {
            float o[8];
            load8(d, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += o[j];
        }
        store8(d, acc);
    }
}

__global__ __launch_bounds__(256) void rows_copy_kernel(const bf16* src, long ld_src, bf16* dst, long ld_dst,
                                                        int accumulate, long M, int C) {
    const int cgs = C >> 3;
    const long total = M * cgs;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cg = (int)(i % cgs);
        const long m = i / cgs;
        float v[8];
        load8(src + m * ld_src + cg * 8, v);
        bf16* d = dst + m * ld_dst + cg * 8;
        if (accumulate) {
            float o[8];
            load8(d, o);
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] += o[j];
        }
        store8(d, v);
    }
}

// fp32 <-> bf16 casts of a flat gradient bucket (the data-parallel reducer's bf16 wire format): n multiple of 4
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* src, bf16* dst, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const float4 v = reinterpret_cast<const float4*>(src)[i];
        bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
        reinterpret_cast<bf16x4*>(dst)[i] = o;
    }
}
__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16* src, float* dst, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const bf16x4 v = reinterpret_cast<const bf16x4*>(src)[i];
        reinterpret_cast<float4*>(dst)[i] = make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
    }
}

// ---------------------------------------------------------------------------------------- head decode
__global__ __launch_bounds__(256) void decode_fwd_kernel(float* out, int B, int A, int a0, int H, int W, float s,
                                                         int ncols, float* origin) {
    const long total = (long)B * H * W * 26;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % 26);
        const long cell = i / 26;
        const int n = (int)(cell / (H * W));
        const int hw = (int)(cell - (long)n * H * W);
        float* p = out + ((long)n * A + a0 + hw) * ncols + c;
        const float t = *p;
        if (origin) origin[((long)n * A + a0 + hw) * 26 + c] = t;
        float v;
        if (c == 0) v = (t + (float)(hw % W)) * s;
        else if (c == 1) v = (t + (float)(hw / W)) * s;
        else v = expf(t) * s;
        *p = v;
    }
}

__global__ __launch_bounds__(256) void decode_bwd_kernel(const float* dout, const float* out, bf16* d_regobj, bf16* d_cls,
                                                         int B, int A, int a0, int H, int W, float s, int ncols,
                                                         const float* d_origin) {
    // one thread per (cell, 8-column chunk): 4 chunks of the 32-wide reg+obj gradient, then ceil(C/8) class chunks
    const int C = ncols - 27;
    const int ld_cls = (C + 7) & ~7;
    const int chunks = 4 + (ld_cls >> 3);
    const long total = (long)B * H * W * chunks;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int chunk = (int)(i % chunks);
        const long cell = i / chunks;
        const int n = (int)(cell / (H * W));
        const int hw = (int)(cell - (long)n * H * W);
        const long row = ((long)n * A + a0 + hw) * ncols;
        bf16x8 o;
        if (chunk < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = chunk * 8 + j;
                float g = 0.f;
                if (c < 2) g = dout[row + c] * s;
                else if (c < 26) g = dout[row + c] * out[row + c];      // d exp(t)*s / dt = r
                else if (c == 26) g = dout[row + 26];
                if (d_origin && c < 26) g += d_origin[((long)n * A + a0 + hw) * 26 + c];
                o[j] = (bf16)g;
            }
            *reinterpret_cast<bf16x8*>(d_regobj + cell * 32 + chunk * 8) = o;
        } else {
            const int k = chunk - 4;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = k * 8 + j;
                o[j] = (bf16)(c < C ? dout[row + 27 + c] : 0.f);
            }
            *reinterpret_cast<bf16x8*>(d_cls + cell * ld_cls + k * 8) = o;
        }
    }
}

__global__ __launch_bounds__(256) void colsum_kernel(const bf16* g, long ld, float* db, long M, int N) {
    // block = 256 threads = 4 row lanes x 64 columns (N <= 128 handled in two column passes)
    __shared__ float red[256];
    for (int c0 = 0; c0 < N; c0 += 64) {
        const int c = c0 + (threadIdx.x & 63);
        float acc = 0.f;
        if (c < N)
            for (long m = (long)blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += (long)gridDim.x * 4) acc += (float)g[m * ld + c];
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < 64 && c < N) atomicAdd(db + c, red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] + red[threadIdx.x + 192]);
        __syncthreads();
    }
}

// the same partial sums without atomics: block b stores its partial at slab[b * N + c]; ep24_wgrad_reduce folds the
// gridDim.x "splits" in order (bitwise reproducible bias gradients)
__global__ __launch_bounds__(256) void colsum_slab_kernel(const bf16* g, long ld, float* slab, long M, int N) {
    // thread = (row lane, 8-column chunk): 16-byte loads, 4 rows in flight, then a fixed-order fold over the row lanes
    __shared__ float red[256][8 + 1];
    const int chunks = (N + 7) >> 3;                       // ld >= 8 * chunks (rows are padded to a multiple of 8)
    const int rpp = 256 / chunks;
    const int r = threadIdx.x / chunks, c = threadIdx.x - r * chunks;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (r < rpp) {
        const long step = (long)gridDim.x * rpp;
        for (long m = (long)blockIdx.x * rpp + r; m < M; m += 4 * step) {
            bf16x8 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (m + k * step < M) v[k] = *reinterpret_cast<const bf16x8*>(g + (m + k * step) * ld + c * 8);
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (m + k * step < M) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[j] += (float)v[k][j];
                }
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = acc[j];
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += 256) {
        float s = 0.f;
        for (int rr = 0; rr < rpp; ++rr) s += red[rr * chunks + (n >> 3)][n & 7];
        slab[(long)blockIdx.x * N + n] = s;
    }
}

// ---------------------------------------------------------------------------------------- weights / SGD
__global__ __launch_bounds__(256) void pack_weights_kernel(const float* w, long ld_w, bf16* wf, bf16* wd, int Cout, int T,
                                                           int Cin, int Cin_pad, int Cout_pad) {
    // only real elements are written: padding columns are zeroed once at allocation and never touched
    const long total = (long)Cout * T * Cin;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int ci = (int)(i % Cin);
        const long r = i / Cin;
        const int t = (int)(r % T);
        const int co = (int)(r / T);
        const bf16 v = (bf16)w[(long)co * ld_w + (long)t * Cin + ci];
        if (wf) wf[((long)co * T + t) * Cin_pad + ci] = v;
        if (wd) wd[((long)ci * T + t) * Cout_pad + co] = v;
    }
}

// All conv segments in ONE launch: desc[s] = {master offset, fwd offset, dgrad offset (-1: none), Cout, T, Cin,
// Cin_pad, Cout_pad}; prefix[s] = first flat element index of segment s in the packed enumeration.
__global__ __launch_bounds__(256) void pack_batched_kernel(const float* flat, const long* desc, const long* prefix, int n_seg,
                                                          bf16* wf, bf16* wd, const int* chunk_seg) {
    __shared__ int s_first;
    const long total = prefix[n_seg];
    for (long base = (long)blockIdx.x * 4096; base < total; base += (long)gridDim.x * 4096) {
        // chunk_seg[c] = segment of element 4096 c (built once by the host): one load.  Without it thread 0 bisects the prefix table
        // - eight dependent loads while 255 threads wait, three quarters of this kernel's time.
        int seg;
        if (chunk_seg) {
            seg = chunk_seg[base >> 12];
        } else {
            if (threadIdx.x == 0) {
                int lo = 0, hi = n_seg - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (prefix[mid] <= base) lo = mid; else hi = mid - 1;
                }
                s_first = lo;
            }
            __syncthreads();
            seg = s_first;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const long i = base + 4 * (threadIdx.x + 256 * k);       // four consecutive elements per thread
            if (i >= total) break;
            while (i >= prefix[seg + 1]) ++seg;
            const long* d = desc + (long)seg * 8;
            const int e = (int)(i - prefix[seg]);                    // a segment has < 2^31 elements
            const int Cin = (int)d[5], Cin_pad = (int)d[6];
            const float* src = flat + d[0] + e;
            bf16* dst = wf + d[1] + e;
            if (Cin == Cin_pad && i + 3 < prefix[seg + 1] && (((uintptr_t)src & 15) | ((uintptr_t)dst & 7)) == 0) {
                const float4 v = *reinterpret_cast<const float4*>(src);   // the forward copy is a pure conversion
                bf16x4 o = {(bf16)v.x, (bf16)v.y, (bf16)v.z, (bf16)v.w};
                *reinterpret_cast<bf16x4*>(dst) = o;
            } else {
                for (int j = 0; j < 4 && i + j < total; ++j) {
                    long ii = i + j;
                    int sg2 = seg;
                    while (ii >= prefix[sg2 + 1]) ++sg2;
                    const long* d2 = desc + (long)sg2 * 8;
                    const int e2 = (int)(ii - prefix[sg2]);
                    const int c2 = (int)d2[5], cp2 = (int)d2[6];
                    const long o2 = c2 == cp2 ? e2 : (long)(e2 / c2) * cp2 + e2 % c2;
                    wf[d2[1] + o2] = (bf16)flat[d2[0] + e2];
                }
            }
        }
        if (!chunk_seg) __syncthreads();
    }
}

// dgrad copy [Cin][T][Cout_pad] = transpose of the master [Cout][T][Cin] per tap: 64x64 tiles through LDS so that both
// the fp32 reads (along ci) and the bf16 writes (along co) are coalesced.  tprefix[s] = first tile of segment s.
__global__ __launch_bounds__(256) void pack_transpose_kernel(const float* flat, const long* desc, const long* tprefix, int n_seg,
                                                            bf16* wd, const int* tile_seg) {
    __shared__ float tile[64][65];
    __shared__ int s_seg;
    const long total = tprefix[n_seg];
    for (long tl = blockIdx.x; tl < total; tl += gridDim.x) {
        int sg;
        if (tile_seg) {
            sg = tile_seg[tl];                                  // segment of tile tl, built once by the host
        } else {
            if (threadIdx.x == 0) {
                int lo = 0, hi = n_seg - 1;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (tprefix[mid] <= tl) lo = mid; else hi = mid - 1;
                }
                s_seg = lo;
            }
            __syncthreads();
            sg = s_seg;
        }
        const long* d = desc + (long)sg * 8;
        const int Cout = (int)d[3], T = (int)d[4], Cin = (int)d[5];
        const int tci = (Cin + 63) >> 6, tco = (Cout + 63) >> 6;
        long r = tl - tprefix[sg];
        const int ci0 = (int)(r % tci) * 64; r /= tci;
        const int co0 = (int)(r % tco) * 64;
        const int t = (int)(r / tco);
        const int c = threadIdx.x & 63, r4 = threadIdx.x >> 6;
        const long src0 = d[0];
        // 16-byte reads and writes where the layer allows (every conv of the YOLOX path but the 12- / 27-column stems): the scalar form
        // moved 4 bytes in and 2 bytes out per lane and instruction - 1.1 TB/s, 0.29 ms per step for the 54 M weights of YOLOX-l
        if ((Cin & 3) == 0 && (src0 & 3) == 0) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int idx = threadIdx.x + 256 * k;
                const int row = idx >> 4, c4 = (idx & 15) * 4;
                const int co = co0 + row, ci = ci0 + c4;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (co < Cout && ci < Cin) v = *reinterpret_cast<const f32x4*>(flat + src0 + ((long)co * T + t) * Cin + ci);
                tile[row][c4] = v[0]; tile[row][c4 + 1] = v[1]; tile[row][c4 + 2] = v[2]; tile[row][c4 + 3] = v[3];
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                const int co = co0 + r4 + 4 * k, ci = ci0 + c;
                tile[r4 + 4 * k][c] = (co < Cout && ci < Cin) ? flat[src0 + ((long)co * T + t) * Cin + ci] : 0.f;
            }
        }
        __syncthreads();
        if (d[2] >= 0) {
            const int cop = (int)d[7];
            if ((cop & 7) == 0 && (d[2] & 7) == 0) {              // rows of Cout_pad: the padding columns hold zeros already, zeros are written again
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int idx = threadIdx.x + 256 * k;
                    const int cl = idx >> 3, j = idx & 7;
                    const int ci = ci0 + cl, co = co0 + 8 * j;
                    if (ci < Cin && co < cop) {
                        bf16x8 o;
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = (bf16)tile[8 * j + e][cl];
                        *reinterpret_cast<bf16x8*>(wd + d[2] + ((long)ci * T + t) * cop + co) = o;
                    }
                }
            } else {
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int ci = ci0 + r4 + 4 * k, co = co0 + c;
                    if (ci < Cin && co < Cout) wd[d[2] + ((long)ci * T + t) * cop + co] = (bf16)tile[c][r4 + 4 * k];
                }
            }
        }
        __syncthreads();
    }
}

// ModelEMA.update on one element (utils/ema.py:57-60): v *= d; v += (1 - d) * p - two rounded products and a rounded sum
__device__ __forceinline__ float ema_step(float e, float pnew, float d, float omd) {
    return __fadd_rn(__fmul_rn(e, d), __fmul_rn(omd, pnew));
}

// Round 5: the update keeps the packed bf16 FORWARD copy of the conv weights current itself.  The flat buffer holds a conv weight
// physically as [Cout][kh][kw][Cin] - for Cin % 8 == 0 exactly the layout of its packed copy - and segments start at multiples of
// 64 elements, so a 4-element group of the update lies in one segment: `wf_delta[(base + i) >> 6]` is that segment's
// (offset in wf) - (offset in the flat buffer), or INT_MIN for a group that has no such copy (BatchNorm vectors, biases, a conv with
// a padded Cin row).  The step used to start with pack_batched_kernel re-reading all 217 MB of masters on the main lane (98 us on
// the critical path of YOLOX-l, profiles/r04_step_timeline.csv); now the freshly written values leave as 8-byte bf16 stores from
// the pass that has them in registers.  base = index of p[0] in the whole flat buffer (the range form passes p + first).
constexpr int WF_NONE = -2147483647 - 1;
__global__ __launch_bounds__(256) void sgd_kernel(float* p, const float* g, float* buf, long n, float lr, float mom,
                                                  float gscale, const int* first_flag, const float* hp, float* ema,
                                                  long base = 0, const int* wf_delta = nullptr, bf16* wf = nullptr) {
    const int first = *first_flag;
    float d = 0.f, omd = 0.f;
    if (hp) { lr = hp[0]; mom = hp[1]; gscale = hp[2]; d = hp[3]; omd = hp[4]; }
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 3 < n) {
            f32x4 gv = *reinterpret_cast<const f32x4*>(g + i);
            f32x4 pv = *reinterpret_cast<const f32x4*>(p + i);
            f32x4 bv = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : *reinterpret_cast<const f32x4*>(buf + i);
            const int wd = wf_delta ? wf_delta[(base + i) >> 6] : WF_NONE;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float gg = gv[j] * gscale;
                bv[j] = first ? gg : mom * bv[j] + gg;
                pv[j] -= lr * (gg + mom * bv[j]);
            }
            *reinterpret_cast<f32x4*>(buf + i) = bv;
            *reinterpret_cast<f32x4*>(p + i) = pv;
            if (wd != WF_NONE) {
                bf16x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = (bf16)pv[j];
                *reinterpret_cast<bf16x4*>(wf + (base + i + wd)) = o;
            }
            if (ema) {
                f32x4 ev = *reinterpret_cast<const f32x4*>(ema + i);
#pragma unroll
                for (int j = 0; j < 4; ++j) ev[j] = ema_step(ev[j], pv[j], d, omd);
                *reinterpret_cast<f32x4*>(ema + i) = ev;
            }
        } else {
            for (long k = i; k < n; ++k) {
                const float gg = g[k] * gscale;
                const float b = first ? gg : mom * buf[k] + gg;
                buf[k] = b;
                p[k] -= lr * (gg + mom * b);
                if (ema) ema[k] = ema_step(ema[k], p[k], d, omd);
                const int wd = wf_delta ? wf_delta[(base + k) >> 6] : WF_NONE;
                if (wd != WF_NONE) wf[base + k + wd] = (bf16)p[k];
            }
        }
    }
}

__global__ __launch_bounds__(256) void ema_kernel(float* ema, const float* src, long n, float d, float omd, const float* hp) {
    if (hp) { d = hp[3]; omd = hp[4]; }
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) ema[i] = ema_step(ema[i], src[i], d, omd);
}

__global__ void set_hparams_kernel(float* hp, float lr, float mom, float gscale, float d, float omd) {
    hp[0] = lr; hp[1] = mom; hp[2] = gscale; hp[3] = d; hp[4] = omd;
}
__global__ void clear_flag_kernel(int* f) { *f = 0; }

int cap_grid(long work_items) {
    long b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > MAX_BLOCKS ? MAX_BLOCKS : b));
}

}  // namespace

#define S_ (hipStream_t) stream

extern "C" int ep24_bn_act_fwd(const void* z, int64_t ld_z, const int64_t* stats, int reps, const float* gamma,
                               const float* beta, float* running_mean, float* running_var, int64_t* num_batches,
                               int64_t* num_batches2, float* save, void* y, int64_t ld_y, const void* residual, int64_t ld_res,
                               int64_t M, int C, float eps, float momentum, int act, void* stream) {
    EP24_REQUIRE(z && stats && gamma && beta && save && y, EP24_E_ARG, "bn_act_fwd: null pointer");
    EP24_REQUIRE(C % 8 == 0 && ld_z % 8 == 0 && ld_y % 8 == 0 && (!residual || ld_res % 8 == 0), EP24_E_ARG,
                 "bn_act_fwd: C=%d / strides must be multiples of 8", C);
    EP24_REQUIRE(M > 0 && reps > 0, EP24_E_ARG, "bn_act_fwd: empty");
    // chunks per lane and launch.  Every block folds the statistics of ALL channels in its prologue (2 * reps * C 8-byte loads out of
    // L2), so thin blocks multiply that traffic: at 2 chunks per lane a 20x20x1024 layer ran 2 000 blocks that read 262 MB of
    // statistics for a 16 MB tensor.  4 = one batch of the body's unrolled loop.
    // (swept again in round 3 with the batched prologue, tools/bn_probe.py: 8 wins where C >= 1024 or the tensor is large and narrow)
    const int fw_per = (C >= 1024 || (C <= 128 && M * C >= (12L << 20))) ? 8 : 4;
    auto kfn = residual ? (act == 1 ? bn_act_fwd_kernel<1, true> : act == 2 ? bn_act_fwd_kernel<2, true> : act == 3 ? bn_act_fwd_kernel<3, true> : bn_act_fwd_kernel<0, true>)
                        : (act == 1 ? bn_act_fwd_kernel<1, false> : act == 2 ? bn_act_fwd_kernel<2, false> : act == 3 ? bn_act_fwd_kernel<3, false> : bn_act_fwd_kernel<0, false>);
    hipLaunchKernelGGL(kfn, dim3(flat_grid(M, C, fw_per)), dim3(256), 2 * (C + C / 8) * sizeof(float), S_, (const bf16*)z, ld_z, (const long long*)stats, reps, gamma,
                       beta, running_mean, running_var, (long*)num_batches, (long*)num_batches2, save, (bf16*)y, ld_y, (const bf16*)residual,
                       ld_res, M, C, eps, momentum);
    EP24_LAUNCH_CHECK("ep24_bn_act_fwd");
    return EP24_OK;
}

extern "C" int ep24_bn_act_bwd_reduce(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save,
                                      const float* gamma, const float* beta, int64_t* dgamma, int64_t* dbeta, int64_t M, int C,
                                      int act, int reps, void* stream) {
    EP24_REQUIRE(dy && z && save && gamma && beta && dgamma && dbeta && reps > 0, EP24_E_ARG, "bn_act_bwd_reduce: null pointer / reps");
    EP24_REQUIRE(C % 8 == 0 && ld_dy % 8 == 0 && ld_z % 8 == 0, EP24_E_ARG, "bn_act_bwd_reduce: alignment");
    // 256-thread blocks, two per CU (round 4).  Alone, one 512-thread block per CU is as fast or 0.5 us faster (tools/bn_probe.py,
    // tools/bn_ab.py) - but in the step this kernel runs beside the weight-gradient lane, whose two workgroups per CU hold 368 of a
    // SIMD's 512 registers: a 512-thread block (two waves per SIMD, 240 registers) does not fit beside them and waited for one of them to
    // leave, a 256-thread block (one wave per SIMD, 120) does.  21.73 against 22.22 ms per step, twice, one box.
    int red_cap = 512;
    int red_rows = 4;
    auto kfn = act == 1 ? bn_act_bwd_reduce_kernel<4, 256, 1> : act == 2 ? bn_act_bwd_reduce_kernel<4, 256, 2> : act == 3 ? bn_act_bwd_reduce_kernel<4, 256, 3> : bn_act_bwd_reduce_kernel<4, 256, 0>;
    hipLaunchKernelGGL(kfn, dim3(rows_grid(M, C, red_rows, red_cap, 256)), dim3(256), 0, S_, (const bf16*)dy, ld_dy, (const bf16*)z, ld_z,
                       save, gamma, beta, (long long*)dgamma, (long long*)dbeta, M, C, reps);
    EP24_LAUNCH_CHECK("ep24_bn_act_bwd_reduce");
    return EP24_OK;
}

extern "C" int ep24_bn_act_bwd_apply(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save,
                                     const float* gamma, const float* beta, const int64_t* dgamma, const int64_t* dbeta,
                                     float* gamma_grad, float* beta_grad, void* dz, int64_t ld_dz, int64_t M, int C, int act,
                                     int reps, void* stream) {
    EP24_REQUIRE(dy && z && save && gamma && beta && dgamma && dbeta && dz && reps > 0, EP24_E_ARG, "bn_act_bwd_apply: null pointer / reps");
    EP24_REQUIRE(C % 8 == 0 && ld_dy % 8 == 0 && ld_z % 8 == 0 && ld_dz % 8 == 0, EP24_E_ARG, "bn_act_bwd_apply: alignment");
    EP24_REQUIRE((((uintptr_t)save | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, EP24_E_ARG,
                 "bn_act_bwd_apply: save / gamma / beta must be 16-byte aligned (they are read as vectors)");
    EP24_REQUIRE((((uintptr_t)gamma_grad | (uintptr_t)beta_grad) & 15) == 0, EP24_E_ARG, "bn_act_bwd_apply: gamma_grad / beta_grad must be 16-byte aligned");
    // rows per lane: 8 on the small tensors (more, thinner blocks: -1.5 ... -2.7 us on the 20x20 / 40x40 layers), 16 on the large
    const int ap_rows = M * C <= (16L << 20) ? 8 : 16, ap_cap = 2048;
    auto kfn = act == 1 ? bn_act_bwd_apply_kernel<4, 1> : act == 2 ? bn_act_bwd_apply_kernel<4, 2> : act == 3 ? bn_act_bwd_apply_kernel<4, 3> : bn_act_bwd_apply_kernel<4, 0>;
    hipLaunchKernelGGL(kfn, dim3(rows_grid(M, C, ap_rows, ap_cap)), dim3(256), 2 * (size_t)C * sizeof(float), S_, (const bf16*)dy, ld_dy,
                       (const bf16*)z, ld_z, save, gamma, beta, (const long long*)dgamma, (const long long*)dbeta, gamma_grad, beta_grad,
                       (bf16*)dz, ld_dz, M, C, reps);
    EP24_LAUNCH_CHECK("ep24_bn_act_bwd_apply");
    return EP24_OK;
}

extern "C" int ep24_bn_act_bwd_fused(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save, const float* gamma,
                                     const float* beta, int64_t* dgamma, int64_t* dbeta, float* gamma_grad, float* beta_grad, void* dz,
                                     int64_t ld_dz, int64_t M, int C, int act, int reps, int32_t* barrier, void* stream) {
    EP24_REQUIRE(dy && z && save && gamma && beta && dgamma && dbeta && dz && barrier && reps > 0, EP24_E_ARG, "bn_act_bwd_fused: null pointer / reps");
    EP24_REQUIRE(C % 8 == 0 && C <= 2048 && ld_dy % 8 == 0 && ld_z % 8 == 0 && ld_dz % 8 == 0, EP24_E_ARG, "bn_act_bwd_fused: alignment / C <= 2048");
    EP24_REQUIRE((((uintptr_t)save | (uintptr_t)gamma | (uintptr_t)beta | (uintptr_t)gamma_grad | (uintptr_t)beta_grad) & 15) == 0, EP24_E_ARG,
                 "bn_act_bwd_fused: save / gamma / beta and their gradients must be 16-byte aligned (they are read as vectors)");
    // at most one workgroup per CU: all of them resident at once is what makes the grid-wide wait safe
    int grid = rows_grid(M, C, 4, 256, 256);
    auto kfn = act == 1 ? bn_act_bwd_fused_kernel<1> : act == 2 ? bn_act_bwd_fused_kernel<2> : act == 3 ? bn_act_bwd_fused_kernel<3> : bn_act_bwd_fused_kernel<0>;
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(256), 2 * (size_t)C * sizeof(float), S_, (const bf16*)dy, ld_dy, (const bf16*)z, ld_z, save, gamma, beta,
                       (long long*)dgamma, (long long*)dbeta, gamma_grad, beta_grad, (bf16*)dz, ld_dz, M, C, reps, (unsigned*)barrier);
    EP24_LAUNCH_CHECK("ep24_bn_act_bwd_fused");
    return EP24_OK;
}

namespace ep24_igemm {
int bn_barrier_timeouts() {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bn_barrier_timeouts), sizeof(v)) != hipSuccess) return -1;
    return (int)v;
}
}

extern "C" int ep24_bn_act_bwd_apply_acc(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save,
                                         const float* gamma, const float* beta, const int64_t* dgamma, const int64_t* dbeta,
                                         float* gamma_grad, float* beta_grad, void* dz, int64_t ld_dz, int64_t M, int C, int act,
                                         int reps, void* stream) {
    EP24_REQUIRE(dy && z && save && gamma && beta && dgamma && dbeta && dz && reps > 0, EP24_E_ARG, "bn_act_bwd_apply_acc: null pointer / reps");
    EP24_REQUIRE(C % 8 == 0 && ld_dy % 8 == 0 && ld_z % 8 == 0 && ld_dz % 8 == 0, EP24_E_ARG, "bn_act_bwd_apply_acc: alignment");
    EP24_REQUIRE((((uintptr_t)save | (uintptr_t)gamma | (uintptr_t)beta) & 15) == 0, EP24_E_ARG,
                 "bn_act_bwd_apply_acc: save / gamma / beta must be 16-byte aligned (they are read as vectors)");
    EP24_REQUIRE((((uintptr_t)gamma_grad | (uintptr_t)beta_grad) & 15) == 0, EP24_E_ARG, "bn_act_bwd_apply_acc: gamma_grad / beta_grad must be 16-byte aligned");
    auto kfn = act == 1 ? bn_act_bwd_apply_kernel<4, 1, true> : act == 2 ? bn_act_bwd_apply_kernel<4, 2, true> : bn_act_bwd_apply_kernel<4, 0, true>;
    hipLaunchKernelGGL(kfn, dim3(rows_grid(M, C, 16, 2048)), dim3(256), 2 * (size_t)C * sizeof(float), S_, (const bf16*)dy, ld_dy, (const bf16*)z, ld_z, save, gamma, beta,
                       (const long long*)dgamma, (const long long*)dbeta, gamma_grad, beta_grad, (bf16*)dz, ld_dz, M, C, reps);
    EP24_LAUNCH_CHECK("ep24_bn_act_bwd_apply_acc");
    return EP24_OK;
}

extern "C" int ep24_stem_pack(const float* images, void* rows, int64_t ld, int B, int H, int W, void* stream) {
    EP24_REQUIRE(images && rows && H % 2 == 0 && W % 2 == 0 && B > 0 && ld >= 108 && ld % 8 == 0, EP24_E_ARG, "stem_pack: bad arguments");
    long sp_blocks = ((long)B * (H / 2) * (W / 2) + 31) / 32;
    hipLaunchKernelGGL(stem_pack_kernel, dim3((unsigned)(sp_blocks > 16384 ? 16384 : sp_blocks)), dim3(256), 0, S_, images,
                       (bf16*)rows, B, H, W, (int)ld);
    EP24_LAUNCH_CHECK("ep24_stem_pack");
    return EP24_OK;
}

extern "C" int ep24_focus_pack(const float* images, void* f16, int B, int H, int W, void* stream) {
    EP24_REQUIRE(images && f16 && H % 2 == 0 && W % 2 == 0 && B > 0 && (reinterpret_cast<unsigned long long>(images) & 7) == 0, EP24_E_ARG,
                 "focus_pack: bad arguments");
    const long blocks = ((long)B * (H / 2) * (W / 2) + 255) / 256;
    hipLaunchKernelGGL(focus_pack_kernel, dim3((unsigned)(blocks > 65536 ? 65536 : blocks)), dim3(256), 0, S_, images, (bf16*)f16, B, H, W);
    EP24_LAUNCH_CHECK("ep24_focus_pack");
    return EP24_OK;
}

extern "C" int ep24_spp_fwd(const void* x, int64_t ld_x, void* y5, void* y9, void* y13, int64_t ld_y, uint8_t* idx, int B,
                            int H, int W, int C, void* scratch, void* stream) {
    EP24_REQUIRE(x && y5 && y9 && y13 && idx && C % 8 == 0 && ld_x % 8 == 0 && ld_y % 8 == 0, EP24_E_ARG, "spp_fwd: bad arguments");
    if (scratch) {
        const long n = (long)B * H * W * C;
        bf16* h = (bf16*)scratch;
        uint8_t* hx = (uint8_t*)scratch + 6 * n;
        hipLaunchKernelGGL(spp_row_kernel, dim3(cap_grid(n / 8)), dim3(256), 0, S_, (const bf16*)x, ld_x, h, hx, B, H, W, C);
        hipLaunchKernelGGL(spp_col_kernel, dim3(cap_grid(n / 8)), dim3(256), 0, S_, (const bf16*)h, (const uint8_t*)hx, (bf16*)y5,
                           (bf16*)y9, (bf16*)y13, ld_y, idx, B, H, W, C);
        EP24_LAUNCH_CHECK("ep24_spp_fwd");
        return EP24_OK;
    }
    hipLaunchKernelGGL(spp_fwd_kernel, dim3(cap_grid((long)B * H * W * (C / 8))), dim3(256), 0, S_, (const bf16*)x, ld_x,
                       (bf16*)y5, (bf16*)y9, (bf16*)y13, ld_y, idx, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_spp_fwd");
    return EP24_OK;
}

extern "C" int ep24_spp_bwd(const void* dy5, const void* dy9, const void* dy13, int64_t ld_dy, const uint8_t* idx, void* dx,
                            int64_t ld_dx, int accumulate, int B, int H, int W, int C, void* stream) {
    EP24_REQUIRE(dy5 && dy9 && dy13 && idx && dx && C % 8 == 0 && ld_dy % 8 == 0 && ld_dx % 8 == 0, EP24_E_ARG, "spp_bwd: bad arguments");
    const size_t lds = (size_t)H * W * 112;                     // the whole map of one (image, channel group) in LDS
    if (lds <= 64 * 1024 && H * W <= 1024)
        hipLaunchKernelGGL(spp_bwd_lds_kernel, dim3(C / 8, B), dim3(256), lds, S_, (const bf16*)dy5, (const bf16*)dy9, (const bf16*)dy13,
                           ld_dy, idx, (bf16*)dx, ld_dx, accumulate, B, H, W, C);
    else
        hipLaunchKernelGGL(spp_bwd_kernel, dim3(cap_grid((long)B * H * W * (C / 8))), dim3(256), 0, S_, (const bf16*)dy5,
                           (const bf16*)dy9, (const bf16*)dy13, ld_dy, idx, (bf16*)dx, ld_dx, accumulate, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_spp_bwd");
    return EP24_OK;
}

extern "C" int ep24_upsample2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, int B, int H, int W, int C, void* stream) {
    EP24_REQUIRE(x && y && C % 8 == 0 && ld_x % 8 == 0 && ld_y % 8 == 0, EP24_E_ARG, "upsample2_fwd: bad arguments");
    hipLaunchKernelGGL(upsample2_fwd_kernel, dim3(cap_grid((long)B * 4 * H * W * (C / 8))), dim3(256), 0, S_, (const bf16*)x,
                       ld_x, (bf16*)y, ld_y, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_upsample2_fwd");
    return EP24_OK;
}

extern "C" int ep24_upsample2_bwd(const void* dy, int64_t ld_dy, void* dx, int64_t ld_dx, int accumulate, int B, int H, int W,
                                  int C, void* stream) {
    EP24_REQUIRE(dy && dx && C % 8 == 0 && ld_dy % 8 == 0 && ld_dx % 8 == 0, EP24_E_ARG, "upsample2_bwd: bad arguments");
    hipLaunchKernelGGL(upsample2_bwd_kernel, dim3(cap_grid((long)B * H * W * (C / 8))), dim3(256), 0, S_, (const bf16*)dy, ld_dy,
                       (bf16*)dx, ld_dx, accumulate, B, H, W, C);
    EP24_LAUNCH_CHECK("ep24_upsample2_bwd");
    return EP24_OK;
}

extern "C" int ep24_rows_copy(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int accumulate, int64_t M, int C,
                              void* stream) {
    EP24_REQUIRE(src && dst && C % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0, EP24_E_ARG, "rows_copy: bad arguments");
    hipLaunchKernelGGL(rows_copy_kernel, dim3(cap_grid(M * (C / 8))), dim3(256), 0, S_, (const bf16*)src, ld_src, (bf16*)dst,
                       ld_dst, accumulate, M, C);
    EP24_LAUNCH_CHECK("ep24_rows_copy");
    return EP24_OK;
}

extern "C" int ep24_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
    EP24_REQUIRE(src && dst && n % 4 == 0 && n >= 0, EP24_E_ARG, "cast_f32_bf16: n must be a multiple of 4");
    if (n) hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(cap_grid(n / 4)), dim3(256), 0, S_, src, (bf16*)dst, (long)(n / 4));
    EP24_LAUNCH_CHECK("ep24_cast_f32_bf16");
    return EP24_OK;
}

extern "C" int ep24_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream) {
    EP24_REQUIRE(src && dst && n % 4 == 0 && n >= 0, EP24_E_ARG, "cast_bf16_f32: n must be a multiple of 4");
    if (n) hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(cap_grid(n / 4)), dim3(256), 0, S_, (const bf16*)src, dst, (long)(n / 4));
    EP24_LAUNCH_CHECK("ep24_cast_bf16_f32");
    return EP24_OK;
}

extern "C" int ep24_head_decode_fwd(float* out, int B, int A, int a0, int H, int W, float stride, int ncols, float* origin,
                                    void* stream) {
    EP24_REQUIRE(out && a0 >= 0 && a0 + H * W <= A && ncols >= 27, EP24_E_ARG, "head_decode_fwd: bad arguments");
    hipLaunchKernelGGL(decode_fwd_kernel, dim3(cap_grid((long)B * H * W * 26)), dim3(256), 0, S_, out, B, A, a0, H, W, stride, ncols, origin);
    EP24_LAUNCH_CHECK("ep24_head_decode_fwd");
    return EP24_OK;
}

extern "C" int ep24_head_decode_bwd(const float* dout, const float* out, void* d_regobj, void* d_cls, int B, int A, int a0,
                                    int H, int W, float stride, int ncols, const float* d_origin, void* stream) {
    EP24_REQUIRE(dout && out && d_regobj && d_cls && a0 >= 0 && a0 + H * W <= A && ncols > 27, EP24_E_ARG,
                 "head_decode_bwd: bad arguments");
    hipLaunchKernelGGL(decode_bwd_kernel, dim3(cap_grid((long)B * H * W * (4 + (ncols - 27 + 7) / 8))), dim3(256), 0, S_, dout, out, (bf16*)d_regobj,
                       (bf16*)d_cls, B, A, a0, H, W, stride, ncols, d_origin);
    EP24_LAUNCH_CHECK("ep24_head_decode_bwd");
    return EP24_OK;
}

extern "C" int ep24_colsum_splits(int64_t M) {
    long blocks = (M + 3) / 4;                         // few partial rows: the ordered fold reads them one after another
    return (int)(blocks > 64 ? 64 : (blocks < 1 ? 1 : blocks));
}

extern "C" int ep24_colsum_slab(const void* g, int64_t ld, float* slab, int64_t M, int N, void* stream) {
    EP24_REQUIRE(g && slab && N > 0 && N <= 2048 && M > 0 && ld % 8 == 0 && ld >= ((N + 7) / 8) * 8, EP24_E_ARG,
                 "colsum_slab: bad arguments (rows must be padded to a multiple of 8 columns)");
    hipLaunchKernelGGL(colsum_slab_kernel, dim3((unsigned)ep24_colsum_splits(M)), dim3(256), 0, S_, (const bf16*)g, ld, slab, M, N);
    EP24_LAUNCH_CHECK("ep24_colsum_slab");
    return EP24_OK;
}

extern "C" int ep24_colsum(const void* g, int64_t ld, float* db, int64_t M, int N, void* stream) {
    EP24_REQUIRE(g && db && N > 0, EP24_E_ARG, "colsum: bad arguments");
    long blocks = (M + 3) / 4;
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)blocks), dim3(256), 0, S_, (const bf16*)g, ld, db, M, N);
    EP24_LAUNCH_CHECK("ep24_colsum");
    return EP24_OK;
}

extern "C" int ep24_pack_weights(const float* w, int64_t ld_w, void* w_fwd, void* w_dgrad, int Cout, int T, int Cin, int Cin_pad,
                                 int Cout_pad, void* stream) {
    EP24_REQUIRE(w && (w_fwd || w_dgrad) && Cin_pad >= Cin && Cout_pad >= Cout, EP24_E_ARG, "pack_weights: bad arguments");
    long n = (long)Cout * T * Cin;
    hipLaunchKernelGGL(pack_weights_kernel, dim3(cap_grid(n)), dim3(256), 0, S_, w, ld_w, (bf16*)w_fwd, (bf16*)w_dgrad, Cout, T, Cin,
                       Cin_pad, Cout_pad);
    EP24_LAUNCH_CHECK("ep24_pack_weights");
    return EP24_OK;
}

extern "C" int ep24_sgd_nesterov(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float grad_scale,
                                 int32_t* first_flag, void* stream) {
    EP24_REQUIRE(p && g && buf && first_flag && n > 0, EP24_E_ARG, "sgd_nesterov: bad arguments");
    EP24_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf) % 16 == 0, EP24_E_ARG, "sgd_nesterov: 16-byte alignment");
    hipLaunchKernelGGL(sgd_kernel, dim3(cap_grid((n + 3) / 4)), dim3(256), 0, S_, p, g, buf, n, lr, momentum, grad_scale, first_flag,
                       (const float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(clear_flag_kernel, dim3(1), dim3(1), 0, S_, first_flag);
    EP24_LAUNCH_CHECK("ep24_sgd_nesterov");
    return EP24_OK;
}

extern "C" int ep24_sgd_nesterov_hp(float* p, const float* g, float* buf, int64_t n, const float* hp, int32_t* first_flag,
                                    float* ema, void* stream) {
    EP24_REQUIRE(p && g && buf && hp && first_flag && n > 0, EP24_E_ARG, "sgd_nesterov_hp: bad arguments");
    EP24_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf | (uintptr_t)ema) % 16 == 0, EP24_E_ARG, "sgd_nesterov_hp: 16-byte alignment");
    hipLaunchKernelGGL(sgd_kernel, dim3(cap_grid((n + 3) / 4)), dim3(256), 0, S_, p, g, buf, n, 0.f, 0.f, 0.f, first_flag, hp, ema);
    hipLaunchKernelGGL(clear_flag_kernel, dim3(1), dim3(1), 0, S_, first_flag);
    EP24_LAUNCH_CHECK("ep24_sgd_nesterov_hp");
    return EP24_OK;
}

// The same update on elements [first, first + n) of the flat buffers: the update of the parameters whose gradients are complete
// can run while the tail of backward still produces the others (ep24.train).  `last` != 0 on the call that finishes the step: only
// that one clears the first-step flag.
extern "C" int ep24_sgd_nesterov_hp_range(float* p, const float* g, float* buf, int64_t first, int64_t n, const float* hp,
                                          int32_t* first_flag, float* ema, int last, void* stream) {
    EP24_REQUIRE(p && g && buf && hp && first_flag && n > 0 && first >= 0 && first % 4 == 0, EP24_E_ARG, "sgd_nesterov_hp_range: bad arguments");
    EP24_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf | (uintptr_t)ema) % 16 == 0, EP24_E_ARG, "sgd_nesterov_hp_range: 16-byte alignment");
    hipLaunchKernelGGL(sgd_kernel, dim3(cap_grid((n + 3) / 4)), dim3(256), 0, S_, p + first, g + first, buf + first, n, 0.f, 0.f, 0.f, first_flag, hp,
                       ema ? ema + first : nullptr);
    if (last) hipLaunchKernelGGL(clear_flag_kernel, dim3(1), dim3(1), 0, S_, first_flag);
    EP24_LAUNCH_CHECK("ep24_sgd_nesterov_hp_range");
    return EP24_OK;
}

// ... and with the packed forward copy of the conv weights written by the same pass (see sgd_kernel): wf_delta has one int32 per 64
// elements of the WHOLE flat buffer, wf is the packed buffer's base.
extern "C" int ep24_sgd_nesterov_hp_range_pack(float* p, const float* g, float* buf, int64_t first, int64_t n, const float* hp,
                                               int32_t* first_flag, float* ema, int last, const int32_t* wf_delta, void* wf, void* stream) {
    EP24_REQUIRE(p && g && buf && hp && first_flag && wf_delta && wf && n > 0 && first >= 0 && first % 4 == 0, EP24_E_ARG,
                 "sgd_nesterov_hp_range_pack: bad arguments");
    EP24_REQUIRE(((uintptr_t)p | (uintptr_t)g | (uintptr_t)buf | (uintptr_t)ema) % 16 == 0 && (uintptr_t)wf % 16 == 0, EP24_E_ARG,
                 "sgd_nesterov_hp_range_pack: 16-byte alignment");
    hipLaunchKernelGGL(sgd_kernel, dim3(cap_grid((n + 3) / 4)), dim3(256), 0, S_, p + first, g + first, buf + first, n, 0.f, 0.f, 0.f, first_flag, hp,
                       ema ? ema + first : nullptr, (long)first, (const int*)wf_delta, (bf16*)wf);
    if (last) hipLaunchKernelGGL(clear_flag_kernel, dim3(1), dim3(1), 0, S_, first_flag);
    EP24_LAUNCH_CHECK("ep24_sgd_nesterov_hp_range_pack");
    return EP24_OK;
}

extern "C" int ep24_ema_update(float* ema, const float* src, int64_t n, float decay, float one_minus_decay, const float* hp,
                               void* stream) {
    EP24_REQUIRE(ema && src && n > 0, EP24_E_ARG, "ema_update: bad arguments");
    hipLaunchKernelGGL(ema_kernel, dim3(cap_grid(n)), dim3(256), 0, S_, ema, src, n, decay, one_minus_decay, hp);
    EP24_LAUNCH_CHECK("ep24_ema_update");
    return EP24_OK;
}

extern "C" int ep24_set_hparams(float* hp, float lr, float momentum, float grad_scale, float ema_decay, float one_minus_decay,
                                void* stream) {
    EP24_REQUIRE(hp, EP24_E_ARG, "set_hparams: null pointer");
    hipLaunchKernelGGL(set_hparams_kernel, dim3(1), dim3(1), 0, S_, hp, lr, momentum, grad_scale, ema_decay, one_minus_decay);
    EP24_LAUNCH_CHECK("ep24_set_hparams");
    return EP24_OK;
}

// A kernel, not hipMemsetAsync: a captured memset NODE refilled its buffer with a stale 16-byte pattern (pointer-like
// garbage) from the second replay of the graph on (ROCm 7.2), which silently corrupted the BatchNorm statistics.
namespace {
__global__ __launch_bounds__(256) void zero_fill_kernel(uint4* p, long n16, unsigned char* tail, int ntail) {
    const uint4 z = {0u, 0u, 0u, 0u};
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n16; i += (long)gridDim.x * 256) p[i] = z;
    if (blockIdx.x == 0 && (int)threadIdx.x < ntail) tail[threadIdx.x] = 0;
}
__global__ __launch_bounds__(256) void zero_fill_bytes_kernel(unsigned char* p, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) p[i] = 0;
}
}  // namespace

extern "C" int ep24_memset_zero(void* p, int64_t bytes, void* stream) {
    EP24_REQUIRE(p && bytes >= 0, EP24_E_ARG, "memset_zero: bad arguments");
    if (bytes == 0) return EP24_OK;
    if (((uintptr_t)p & 15) == 0) {
        const long n16 = bytes / 16;
        long blocks = (n16 + 1023) / 1024;                 // 4 stores per thread
        blocks = blocks < 1 ? 1 : (blocks > 8192 ? 8192 : blocks);
        hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, S_, (uint4*)p, n16,
                           (unsigned char*)p + n16 * 16, (int)(bytes - n16 * 16));
    } else {
        long blocks = (bytes + 255) / 256;
        hipLaunchKernelGGL(zero_fill_bytes_kernel, dim3((unsigned)(blocks > 8192 ? 8192 : blocks)), dim3(256), 0, S_,
                           (unsigned char*)p, (long)bytes);
    }
    EP24_LAUNCH_CHECK("ep24_memset_zero");
    return EP24_OK;
}

extern "C" int ep24_pack_weights_batched(const float* flat, const int64_t* desc, const int64_t* prefix, const int64_t* tile_prefix,
                                         int n_seg, void* w_fwd, void* w_dgrad, int64_t total, int64_t total_tiles,
                                         const int32_t* chunk_seg, const int32_t* tile_seg, int which, void* stream) {
    EP24_REQUIRE(flat && desc && prefix && tile_prefix && w_fwd && w_dgrad && n_seg > 0 && total > 0, EP24_E_ARG,
                 "pack_weights_batched: bad arguments");
    long blocks = (total + 4095) / 4096;
    // which: 0 both copies, 1 the forward copy only, 2 the input-gradient (transposed) copy only - backward is the first to need
    // the second, so a captured step packs it on another stream beside the start of the forward pass
    if (which != 2)
        hipLaunchKernelGGL(pack_batched_kernel, dim3((unsigned)(blocks > 4096 ? 4096 : blocks)), dim3(256), 0, S_, flat, (const long*)desc,
                           (const long*)prefix, n_seg, (bf16*)w_fwd, (bf16*)w_dgrad, (const int*)chunk_seg);
    if (which != 1)
        hipLaunchKernelGGL(pack_transpose_kernel, dim3((unsigned)(total_tiles > 8192 ? 8192 : total_tiles)), dim3(256), 0, S_, flat,
                           (const long*)desc, (const long*)tile_prefix, n_seg, (bf16*)w_dgrad, (const int*)tile_seg);
    EP24_LAUNCH_CHECK("ep24_pack_weights_batched");
    return EP24_OK;
}
