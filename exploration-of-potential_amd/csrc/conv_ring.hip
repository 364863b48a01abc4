// ep24 - loader / consumer form of the halo-patch kernel for the 3x3 stride-1 convolutions (forward and input gradient).
//
// conv_patch.hip runs 8 waves in lockstep on a 256 x 128 tile: every wave issues its share of the LDS-DMA, every wave multiplies a
// 64 x 64 piece, one s_barrier per tap step.  Its stamps (tools/conv_stamps.py) say where a 1 800-cycle step goes against 1 024
// cycles of MFMA issue for the two waves of a SIMD: ~500 cycles at the barrier waiting for the SIMD partner (two MFMA-bound waves
// share one matrix pipe and arrive one after the other), ~180 for DMA data, the rest issue slots.  Here the roles are split
// (the FULL / FREE ring of MI355X_MICROARCH.md, row ring-gemm):
//
//   waves 0..3  CONSUMERS, one per SIMD, a 128 x 64 piece each (8 x 4 accumulator tiles of 16 x 16: 128 registers): 24 fragment
//               reads per 64 MFMAs instead of 32, nobody to share the matrix pipe with, no workgroup barrier in the loop
//   waves 4..7  LOADERS, one per SIMD beside a consumer: every LDS-DMA and its address arithmetic.  Per tap step: the weight tile
//               of the step (16 KB, 4 instructions per loader) and from tap 2 on two pieces of the NEXT channel chunk's patch
//
// Hand-off through two arrays of monotone counters in LDS, one word per wave, no atomics and no barrier:
//   full[l]  = steps whose weight tile (and every DMA loader l issued before it, the patch pieces among them) has LANDED: written
//              by loader l behind a counted s_waitcnt vmcnt.  A consumer starts step j once min(full) > j.
//   free[c]  = steps whose fragment reads consumer c has ISSUED (LDS executes a wave's operations in order, so a later DMA cannot
//              overtake them).  A loader refills ring stage j % 3 for step j once min(free) >= j - 2; the patch buffer of chunk
//              kc + 1 is refilled from tap 2 of chunk kc on, when that condition already implies that chunk kc - 1 is read.
// Every wait is a bounded spin (s_sleep between polls); a spin that gives up sets a word that the host-side tests read back and
// carries on, so that a protocol error shows as wrong numbers and a counter, never as a hung GPU.
//
// Arithmetic: the same products in the same order as conv_patch.hip and the tiled kernel (chunks outer, taps inner, two k halves of
// 32, v_mfma_f32_16x16x32_bf16, fp32 accumulate) - results are bit-identical, tests/test_gpu_conv.py asserts it on every hot shape.
#include <atomic>
#include <type_traits>
#include "igemm.h"

using namespace ep24_igemm;

namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

constexpr int NCW = 4, NLW = 4;                  // consumer / loader waves
constexpr int NS = 3;                            // weight ring stages: 9 taps = 3 x 3, so the stage of a step is its tap column
constexpr int RBN = 128, RBM = 256;              // tile
constexpr int B_BYTES = RBN * 128;               // one weight tile: [128 channels][64 k] bf16
[[maybe_unused]] constexpr int B_INSTR = RBN / 8 / NLW;           // weight-tile DMA instructions per loader and step (4)
constexpr int MT = 8, NT = 4;                    // accumulator tiles per consumer: 128 rows x 64 channels
constexpr int ZB = 128;                          // LDS bytes 0..127 stay zero
constexpr int LA = 8;                            // groups of 4 MFMAs between the request of an A fragment and its use (a divisor of 8)
constexpr int SPIN_LIMIT = 1 << 14;              // polls of >= 64 cycles each: ~2 ms, three orders beyond any real wait (and short enough
                                                 // that a broken protocol fails a test run in seconds instead of stalling it)

// spins that gave up, over every launch of the process (tests read it through ep24_conv_ring_timeouts and require 0)
__device__ unsigned g_ring_timeouts;

#ifdef EP24_STAMPS
__device__ unsigned long long g_ring_stamps[64 * 8];
__device__ unsigned long long g_ring_estamps[64 * 8];    // the epilogue of wave 0 (a consumer) and wave 4 (a loader) of the first 32 workgroups, piece by piece
#define STAMP() __builtin_amdgcn_s_memtime()
#define ESTAMP(k) do { if (blockIdx.x < 32 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == NCW)) est[k] = STAMP(); } while (0)
#else
#define ESTAMP(k) do { } while (0)
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }          // compiler barrier: no memory access moves across

// The counters are addressed as LDS (address space 3) objects.  Through generic pointers (the first form) every access was a FLAT
// instruction: it counts on vmcnt AND lgkmcnt, so the compiler drained both before using the value - a consumer's look at the FULL
// counters stalled on every fragment read in flight, and a loader's poll of the FREE counters waited for every DMA it had in flight
// (running further ahead made the loaders slower, which is how it was found: stamps of the narrow tile, round 4).
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) v4u lds_v4u;
__device__ __forceinline__ unsigned ld_flag(const lds_u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void st_flag(lds_u32* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ v4u ld_flags4(const lds_u32* p) { return *reinterpret_cast<const volatile lds_v4u*>(p); }

// One MFMA with its accumulator tile in the ACCUMULATOR half of the register file, destination = source.  Left to the builtin the
// compiler kept all 128 accumulator registers in the VGPR class, gave most MFMAs a destination different from their C operand (tiles
// migrate, 4 .. 16 registers of slack) and, at 256 registers, spilled a loop-invariant address term to scratch: a memory round trip
// with s_waitcnt vmcnt(0) in EVERY step (1 860 cycles per step against 1 517 without the reload).  With the tiles pinned to AGPRs
// the vector half holds fragments and addresses only (~90 registers).  Operands come straight from ds_read (the compiler waits for
// them before the statement); no VALU result feeds an MFMA here, and the accumulators are first read in the epilogue behind a barrier.
__device__ __forceinline__ void mfma_acc(f32x4& c, const bf16x8& a, const bf16x8& b) {
#if (RING_VAR & 8)
    asm volatile("" : "+a"(c) : "v"(a), "v"(b));             // diagnostic: operands stay live, no MFMA
#else
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#endif
}

// The MFMAs above are plain inline asm: the compiler's hazard recogniser does not see an XDL write, so nothing is inserted between
// the last v_mfma of the main loop and the first v_accvgpr_read of the epilogue.  The ISA asks for 11 wait states behind an 8-pass
// MFMA (16 x 16 x 32) and 19 behind a 16-pass one (32 x 32 x 16) before a VALU instruction reads its destination; the epilogues open
// with an s_barrier that takes far longer in practice, but "in practice" is not the contract.  20 wait states, once per
// kernel, in front of every epilogue call.
__device__ __forceinline__ void mfma_drain() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }

typedef __attribute__((ext_vector_type(16))) float f32x16_;
// ... and the 32 x 32 x 16 form: D[32 x 32] in 16 registers per lane (row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5), column = lane & 31)
__device__ __forceinline__ void mfma32_acc(f32x16_& c, const bf16x8& a, const bf16x8& b) {
#if (RING_VAR & 8)
    asm volatile("" : "+a"(c) : "v"(a), "v"(b));
#else
    asm("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
#endif
}

// min over the four counters; the words sit in one 16-byte line, every lane reads the same addresses (broadcast)
__device__ __forceinline__ unsigned min4(const lds_u32* f) {
    const unsigned a = ld_flag(f), b = ld_flag(f + 1), c = ld_flag(f + 2), d = ld_flag(f + 3);
    return __builtin_amdgcn_readfirstlane(min(min(a, b), min(c, d)));
}

// Diagnostic builds only (make stamps EXTRA=-DRING_VAR=n, tools/ring_stamps.py): what a step costs with one thing changed.
// bit 0: consumers at s_setprio 2; bit 1: loaders sleep 512 cycles between polls instead of 64; bit 2: no border masks (wrong
// results, timing only); bit 3: no MFMAs (the loop's fragment traffic and bookkeeping alone).  The product is RING_VAR 0.
#ifndef RING_VAR
#define RING_VAR 0
#endif

// bounded wait until min(f[0..3]) >= need
__device__ __forceinline__ void spin_until(const lds_u32* f, unsigned need, lds_u32* err) {
    int tries = 0;
    while (min4(f) < need) {
        if constexpr ((RING_VAR & 2) != 0) { if (threadIdx.x >= NCW * 64) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(1); }
        else __builtin_amdgcn_s_sleep(1);
        if (++tries > SPIN_LIMIT) {
            if ((threadIdx.x & 63) == 0) { atomicAdd(&g_ring_timeouts, 1u); *err = 1u; }
            break;
        }
    }
    cbar();
}

template <int TBN>
__device__ __forceinline__ void ring_store_tile(const IgemmArgs& p, long m0, int n0, char* smem);
template <int HALF, bool ALL>
__device__ __forceinline__ void ring_store_half(const IgemmArgs& p, long m0, int n0, char* smem);

// Epilogue with all eight waves (16-byte store path only).  The stamps of the first form - the four consumers alone, each staging and
// storing its 128 x 64 piece - read 10.5 k cycles per tile, a quarter of a 36-step kernel: the tail is store-ISSUE bound (MI355X_MICROARCH
// .md, attention epilogue), so the loaders take half of the stores.  Consumers round their accumulators to bf16 into the four staging
// areas ([128 rows][128 B], 16-byte chunk index XOR (row & 7), as igemm_epilogue) and keep the BatchNorm statistics of what they
// rounded FROM (fp32, as the tiled kernels); after a barrier every wave stores 8 chunks per lane, 8 lanes per 128-byte row segment.
// Same values, same statistics and the same order of the fixed-point sums as igemm_epilogue: bit-identical outputs.
// TBN = 64 (the narrow tile, 256 x 64): the four consumers hold 64 x 64 pieces stacked over the rows, CMT = 4.
template <int TBN, int CMT, int EMODE = 0>                  // EMODE: 0 training (statistics), 1 inference (bias, act, residual), 2 training without statistics (input gradients)
__device__ __forceinline__ void ring_epilogue(const IgemmArgs& p, f32x4 (&acc)[CMT][NT], const bool consumer, long m0, int n0, int tile_m, char* smem) {
    constexpr bool INFER = EMODE == 1, STATS = EMODE == 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
#ifdef EP24_STAMPS
    unsigned long long est[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    ESTAMP(0);
    __syncthreads();                                         // every wave is out of the main loop: LDS is free
    ESTAMP(1);
    if (consumer) {
        const int wm = TBN == 128 ? wave >> 1 : wave;
        const int wn = TBN == 128 ? wave & 1 : 0;
        char* stg = smem + wave * (CMT * 16 * 128);
        // eval-mode unit: y = act(acc + bias) + residual on the way into the staging area, the expressions and their
        // order as in igemm_epilogue's inference form (a lane's four values of a row are four consecutive channels c0 .. c0 + 3)
        constexpr bool infer = INFER;                          // its own instantiation: the training kernel carries no trace of it
        const int c0 = n0 + wn * 64 + 4 * frow;
        float bias4[4] = {0.f, 0.f, 0.f, 0.f};
        if (infer && p.bias) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (c0 + q < p.N) bias4[q] = p.bias[c0 + q];
        }
        // Round 5: the stamps of this epilogue (tools/ring_estamps.py, profiles/r05_ring_epilogue.txt) put 4 850 of its 10 500 cycles
        // HERE - ~1 050 instructions of one wave per SIMD, 256 of them the selects of `if (row < M) { s1 += v; s2 += v * v; }` (two per
        // value) and 128 the multiplies.  Now the accumulators of rows past M are cleared up front - only the launch's last row of tiles
        // has any (uniform branch) - so that they add zeros, and the sum of squares is one fused multiply-add per value.
        if (STATS && m0 + RBM > p.M) {
#pragma unroll
            for (int i = 0; i < CMT; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const bool dead = m0 + wm * (CMT * 16) + i * 16 + 4 * fq + r >= p.M;
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[i][q][r] = dead ? 0.f : acc[i][q][r];
                }
        }
        // a lane's four staging addresses (r = 0 .. 3) do not depend on the row block: row & 7 = 4 (fq & 1) + r, a block is 2 KB further
        // (an immediate) - written as `stg + row * 128 + swizzle(row)` inside the loop they were 190 of the loop's 600 instructions
        int sadr[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sadr[r] = (4 * fq + r) * 128 + (((frow >> 1) ^ (4 * (fq & 1) + r)) << 4) + (frow & 1) * 8;
        auto stage = [&](auto lo_c, auto hi_c) {
#pragma unroll
        for (int i = decltype(lo_c)::value; i < decltype(hi_c)::value; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i * 16 + 4 * fq + r;
                const long m = m0 + wm * (CMT * 16) + row;
                [[maybe_unused]] const bool live = m < p.M;
                bf16x4 w;
                bf16x4 res4 = {0, 0, 0, 0};
                if (infer && p.epi_res && live && c0 + 3 < p.N) res4 = *reinterpret_cast<const bf16x4*>(p.epi_res + m * p.epi_ldres + c0);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float v = acc[i][q][r];
                    if (infer) {
                        v = act_fwd(v + bias4[q], p.epi_act);
                        if (p.epi_res) v += (float)res4[q];
                    }
                    // (an input gradient has no statistics - EMODE 2: the sums are 256 of the loop's ~500 instructions; the eval-mode form stores none either)
                    if constexpr (STATS) { s1[q] += v; s2[q] = fmaf(v, v, s2[q]); }
                    w[q] = (bf16)v;
                }
                *reinterpret_cast<bf16x4*>(stg + sadr[r] + i * 2048) = w;
            }
        }
        };
        if constexpr (TBN == 128 && !INFER) {
            // two halves: while the consumers stage the second 64 rows of their pieces, the loaders - idle during the staging until
            // now: 3 000 cycles - store the first
            stage(std::integral_constant<int, 0>{}, std::integral_constant<int, CMT / 2>{});
            ESTAMP(2);
            __syncthreads();
            ESTAMP(3);
            stage(std::integral_constant<int, CMT / 2>{}, std::integral_constant<int, CMT>{});
        } else {
            stage(std::integral_constant<int, 0>{}, std::integral_constant<int, CMT>{});
        }
    }
    if constexpr (TBN == 128 && !INFER) {
        if (!consumer) {
            ESTAMP(2);
            __syncthreads();
            ESTAMP(3);
            ring_store_half<0, false>(p, m0, n0, smem);
        }
        ESTAMP(4);
        __syncthreads();
        ring_store_half<1, true>(p, m0, n0, smem);
    } else {
        ESTAMP(2);
        __syncthreads();
        ESTAMP(3);
        ring_store_tile<TBN>(p, m0, n0, smem);
        ESTAMP(4);
    }
    if (STATS && p.stats && consumer) {
        // Round 5: every consumer publishes the sums of ITS 128 x 64 piece (64 x 64: the narrow tile) - no barrier, no LDS.  The fold of
        // the two row halves of a tile through LDS sat behind two workgroup barriers, the first of which waited for the loaders' half of
        // the stores (1 300 + 1 600 cycles of the epilogue's 10 500).  After the two butterflies every lane of a quarter holds the sums of
        // its four channels; lane (frow, fq) publishes channel 4 frow + fq.  (The fixed-point conversion is per 128 rows now, as in the
        // tiled kernels, instead of per 256: equal to 2^-20 rounding.)
        const int wn = TBN == 128 ? wave & 1 : 0;
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
    }
    ESTAMP(6);
#ifdef EP24_STAMPS
    if (blockIdx.x < 32 && (threadIdx.x & 63) == 0 && ((threadIdx.x >> 6) == 0 || (threadIdx.x >> 6) == NCW)) {
        unsigned long long* o = g_ring_estamps + (blockIdx.x * 2 + ((threadIdx.x >> 6) != 0)) * 8;
        for (int k = 0; k < 7; ++k) o[k] = est[k];
        o[7] = STAMP();
    }
#endif
}

// The staged 256 x 128 bf16 tile (four areas of [128 rows][128 B], 16-byte chunk index XOR (row & 7)) leaves as 16-byte stores, eight
// chunks per lane of all eight waves, 8 lanes per 128-byte row segment.  Called behind the barrier that follows the staging writes.
template <int TBN>
__device__ __forceinline__ void ring_store_tile(const IgemmArgs& p, long m0, int n0, char* smem) {
    const int tid = threadIdx.x;
    const bool fast_dst = (p.dsy == 1 && p.dsx == 1 && p.dy0 == 0 && p.dx0 == 0 && p.DW == p.GW && p.dp0 == 0 && p.dbs == (long)p.GH * p.GW);
    constexpr int CMT = TBN == 128 ? 8 : 4;
    constexpr int CPT = NCW * CMT * 16 * 8 / ((NCW + NLW) * 64);     // 16-byte chunks per thread: 8 (4 for the narrow tile)
    if constexpr (TBN == 128) {
        // Round 5: the destination of this kernel is plain and below 2 GiB (launch_ring): it leaves through buffer stores with 32-bit
        // offsets.  Chunk k of thread t is LDS byte k * 8192 + (t >> 3) * 128 + (((t & 7) ^ ((t >> 3) & 7)) << 4) and row
        // (k >> 2) * 128 + (k & 1) * 64 + (t >> 3), channel ((k >> 1) & 1) * 64 + (t & 7) * 8 of the tile: one address per thread, the rest
        // immediates and one add per chunk.  (The pointer form below - still the narrow tile's - makes a 64-bit address per chunk: ~700
        // instructions for eight stores; the stamps read 2 000 cycles for a consumer and 3 200 for a loader, the consumers then
        // waiting for the loaders at the next barrier.)
        typedef int v4i_ __attribute__((ext_vector_type(4)));
        const auto drs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16*>(p.dst), 0, p.dst_bytes, 0x00020000);
        const int rt = tid >> 3, ch = tid & 7;
        const char* lsrc = smem + rt * 128 + ((ch ^ (rt & 7)) << 4);
        const int ld2 = (int)p.ld_dst * 2;
        const long mt = m0 + rt;
        const int cc0 = n0 + ch * 8;
        const int off_t = (int)((mt * p.ld_dst + cc0) * 2);
        bf16x8 v_[CPT];
        int off[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) v_[k] = *reinterpret_cast<const bf16x8*>(lsrc + k * 8192);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int dm = (k >> 2) * 128 + (k & 1) * 64, dc = ((k >> 1) & 1) * 64;
            off[k] = (mt + dm < p.M && cc0 + dc < p.N) ? off_t + dm * ld2 + dc * 2 : OOB;
        }
        if (p.accumulate) {                                  // every old value is requested before the first is added (out of range: zeros)
            bf16x8 o_[CPT];
#pragma unroll
            for (int k = 0; k < CPT; ++k) o_[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(drs, off[k], 0, 0));
#pragma unroll
            for (int k = 0; k < CPT; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) v_[k][e] = (bf16)((float)v_[k][e] + (float)o_[k][e]);
        }
#pragma unroll
        for (int k = 0; k < CPT; ++k) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_, v_[k]), drs, off[k], 0, 0);
        return;
    }
    bf16* dptr[CPT];
    bf16x8 val[CPT], old[CPT];
    bool ok[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = k * ((NCW + NLW) * 64) + tid;
        const int area = TBN == 128 ? c >> 10 : c >> 9, row = (c >> 3) & (CMT * 16 - 1), ch = c & 7;
        val[k] = *reinterpret_cast<const bf16x8*>(smem + area * (CMT * 16 * 128) + row * 128 + ((ch ^ (row & 7)) << 4));
        const long m = m0 + (TBN == 128 ? area >> 1 : area) * (CMT * 16) + row;
        const int cc = n0 + (TBN == 128 ? (area & 1) * 64 : 0) + ch * 8;
        ok[k] = m < p.M && cc < p.N;
        long dpix = ok[k] ? m : 0;
        if (!fast_dst) {
            const int mm = (int)dpix;
            const int n = fdiv(mm, p.d_plane);
            const int rem = mm - n * (p.GH * p.GW);
            const int gy = fdiv(rem, p.d_gw), gx = rem - gy * p.GW;
            dpix = (long)n * p.dbs + p.dp0 + (long)(gy * p.dsy + p.dy0) * p.DW + gx * p.dsx + p.dx0;
        }
        dptr[k] = reinterpret_cast<bf16*>(p.dst) + dpix * p.ld_dst + (ok[k] ? cc : 0);
    }
    if (p.accumulate) {                                      // every old value is requested before the first is added
#pragma unroll
        for (int k = 0; k < CPT; ++k) old[k] = *reinterpret_cast<const bf16x8*>(dptr[k]);
#pragma unroll
        for (int k = 0; k < CPT; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) val[k][e] = (bf16)((float)val[k][e] + (float)old[k][e]);
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k)
        if (ok[k]) *reinterpret_cast<bf16x8*>(dptr[k]) = val[k];
}

// Half of the staged 256 x 128 tile - rows HALF * 64 .. + 63 of each of the four 128-row areas - as 16-byte buffer stores: by all eight
// waves (ALL: 4 chunks per thread) or by the four loader waves alone (8 chunks per thread).  Chunk c = k * T + t of the half is area
// c >> 9, row (c >> 3) & 63 of the half, 16-byte chunk c & 7; a thread's row & 7 and chunk do not depend on k.
template <int HALF, bool ALL>
__device__ __forceinline__ void ring_store_half(const IgemmArgs& p, long m0, int n0, char* smem) {
    typedef int v4i_ __attribute__((ext_vector_type(4)));
    constexpr int T = ALL ? (NCW + NLW) * 64 : NLW * 64, NK = 4 * 64 * 8 / T;
    const int t = ALL ? (int)threadIdx.x : (int)threadIdx.x - NCW * 64;
    const auto drs = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16*>(p.dst), 0, p.dst_bytes, 0x00020000);
    const int rt = t >> 3, ch = t & 7;
    const char* lsrc = smem + HALF * 8192 + rt * 128 + ((ch ^ (rt & 7)) << 4);
    const int ld2 = (int)p.ld_dst * 2;
    const long mt = m0 + HALF * 64 + rt;
    const int cc0 = n0 + ch * 8;
    const int off_t = (int)((mt * p.ld_dst + cc0) * 2);
    bf16x8 v_[NK];
    int off[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) {
        const int area = ALL ? k : k >> 1, rr = ALL ? 0 : (k & 1) * 32;          // (compile-time after unrolling)
        v_[k] = *reinterpret_cast<const bf16x8*>(lsrc + area * 16384 + rr * 128);
        const int dm = (area >> 1) * 128 + rr, dc = (area & 1) * 64;
        off[k] = (mt + dm < p.M && cc0 + dc < p.N) ? off_t + dm * ld2 + dc * 2 : OOB;
    }
    if (p.accumulate) {                                      // every old value is requested before the first is added (out of range: zeros)
        bf16x8 o_[NK];
#pragma unroll
        for (int k = 0; k < NK; ++k) o_[k] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(drs, off[k], 0, 0));
#pragma unroll
        for (int k = 0; k < NK; ++k)
#pragma unroll
            for (int e = 0; e < 8; ++e) v_[k][e] = (bf16)((float)v_[k][e] + (float)o_[k][e]);
    }
#pragma unroll
    for (int k = 0; k < NK; ++k) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_, v_[k]), drs, off[k], 0, 0);
}

// Epilogue of the 32 x 32 x 16 consumers.  A consumer holds D[channel][pixel] tiles: lane = pixel (lane & 31), register quad qd of a
// tile = four consecutive channels 8 qd + 4 (lane >> 5) .. + 3 - one 8-byte packed store into the pixel's staging row, 32 per lane
// as in the 16 x 16 form.  BatchNorm statistics: a lane's 32 channels (x 2 sums) summed over its four pixel tiles, then over the 32
// lanes that share them through LDS (fixed order: bitwise reproducible).
__device__ __forceinline__ void ring_epilogue32(const IgemmArgs& p, f32x16_ (&acc)[2][4], const bool consumer, long m0, int n0, int tile_m, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pr = lane & 31, h = lane >> 5;
    float ps1[2][16], ps2[2][16];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ps1[ct][r] = 0.f; ps2[ct][r] = 0.f; }
    __syncthreads();                                         // every wave is out of the main loop: LDS is free
    if (consumer) {
        const int wm = wave >> 1;
        char* stg = smem + wave * (MT * 16 * 128);
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const int row = pt * 32 + pr;
            const bool live = m0 + wm * (MT * 16) + row < p.M;
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    bf16x4 w;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float v = acc[ct][pt][qd * 4 + e];
                        if (live) { ps1[ct][qd * 4 + e] += v; ps2[ct][qd * 4 + e] += v * v; }
                        w[e] = (bf16)v;
                    }
                    *reinterpret_cast<bf16x4*>(stg + row * 128 + (((ct * 4 + qd) ^ (row & 7)) << 4) + h * 8) = w;
                }
            }
        }
    }
    __syncthreads();
    ring_store_tile<128>(p, m0, n0, smem);
    if (p.stats) {
        __syncthreads();                                     // the staging areas have been read
        constexpr int LP = 33;                               // 32 lanes + 1: lanes of one channel on consecutive banks
        float* red = reinterpret_cast<float*>(smem);         // [NCW waves][2 sums][64 channels][LP]
        if (consumer) {
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ch = ct * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                    red[((wave * 2 + 0) * 64 + ch) * LP + pr] = ps1[ct][r];
                    red[((wave * 2 + 1) * 64 + ch) * LP + pr] = ps2[ct][r];
                }
        }
        __syncthreads();
        long long* st = p.stats + (long)(tile_m % p.stats_replicas) * 2 * p.N;
        if (tid < 2 * RBN) {
            const int which = tid / RBN, c = tid - which * RBN;
            const int wcol = c >> 6;
            float v = 0.f;
            for (int r = 0; r < 2; ++r) {
                const float* src = red + (((r * 2 + wcol) * 2 + which) * 64 + (c & 63)) * LP;
#pragma unroll
                for (int l = 0; l < 32; ++l) v += src[l];
            }
            if (n0 + c < p.N) atomicAdd((unsigned long long*)(st + (long)which * p.N + n0 + c), (unsigned long long)to_fix(v));
        }
    }
}

// M32: the consumers multiply with v_mfma_f32_32x32x16_bf16 (weights as the A operand, pixels as B: D[channel][pixel]) instead of
// v_mfma_f32_16x16x32_bf16.  Same FLOPs per matrix-pipe cycle, HALF the MFMA instructions: a step is 32 MFMAs that leave 24 free
// issue cycles each (768 per step) where the 64 of the 16 x 16 form leave 8 each (512) - and the stamps of the 16 x 16 consumer
// said its other ~85 instructions per step need ~740 cycles of issue by themselves (profiles/r04_ring_variants_16x16.txt).
// What changes with it: LDS rows keep 128 B, but the 16-byte chunk of k-step s (16 deep) and lane half h is 2 s + h, and the
// swizzle key is (row >> 1) & 7 - conflict free for ds_read_b128 of 32 consecutive rows at ANY row shift (brute-forced over the
// instruction's lane groups), which row & 7 is not for this fragment shape; the weight rows sit in natural channel order (a
// register quad of D is four consecutive channels already); the sum over a 64-channel chunk runs in four k-steps of 16 instead of
// two of 32, so results differ from the tiled kernel in the last bit of the fp32 sums (and agree run to run, bit for bit).
//
// TBN = 64: the NARROW tile, 256 pixels x 64 channels, for the layers that 256 x 128 tiles do not spread over the chip (the 20 x 20
// level at B = 20: 128 tiles) or whose N is 64.  Consumers are stacked over the rows (64 x 64 each, 4 x 4 accumulator tiles): a step is
// 8 groups of 4 MFMAs and 16 fragment reads, LDS read bandwidth and MFMA issue in balance (512 cycles each), so the loaders run TWO
// steps ahead here (the next step's DMAs are issued before the wait for this step's: one step per round trip would bound the loop).
template <int PPS, bool M32, int TBN = RBN, int INFER = 0>        // INFER: the epilogue's EMODE (0 training, 1 inference, 2 training without statistics)
__global__ __launch_bounds__((NCW + NLW) * 64) void conv_ring_kernel(const IgemmArgs p, const int NP, const int halo, const int npb) {
    static_assert(TBN == 128 || (TBN == 64 && !M32), "tile widths: 128, or 64 with the 16 x 16 consumers");
    constexpr int CMT = TBN == 128 ? MT : 4;                 // accumulator row tiles per consumer
    constexpr int TB_BYTES = TBN * 128;                      // one weight tile: [TBN channels][64 k] bf16
    constexpr int TB_INSTR = TBN / 8 / NLW;                  // weight-tile DMA instructions per loader and step
    constexpr int TNS = TBN == 128 ? NS : 9;                 // weight ring stages; the narrow tile: one per tap (8 KB each)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tile_id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const int tiles_n = (p.N + TBN - 1) / TBN;
    const int tile_m = tile_id / tiles_n;
    const long m0 = (long)tile_m * RBM;
    const int n0 = (tile_id - tile_m * tiles_n) * TBN;
    const int KC = (p.K + BK - 1) / BK;
    // LDS: [128 zero bytes][patch buffers: npb x NP pieces of 8 rows x 128 B][weight ring: 3 x 16 KB][counters]
    const int PBYTES = NP * 1024;
    const int bb = npb * PBYTES;                             // offset of the weight ring behind ZB
    unsigned* const flags = reinterpret_cast<unsigned*>(smem + ZB + bb + TNS * TB_BYTES);
    lds_u32* const f_full = (lds_u32*)(lptr_t)flags;         // [NLW]
    lds_u32* const f_free = f_full + 4;                      // [NCW]
    lds_u32* const f_err = f_full + 8;
    // taps in row-major order; forward reads pixel (y + ty - 1, x + tx - 1), the input gradient (y + 1 - ty, x + 1 - tx)
    const int sgn = p.oy[0] < 0 ? 1 : -1;

    // the fragment reads address LDS by offset: smem must start at LDS offset 0 (no static LDS object in this kernel)
    if (reinterpret_cast<unsigned long>((lptr_t)smem) != 0ul) {
        if (tid == 0) atomicAdd(&g_ring_timeouts, 1u << 16);
        return;
    }
    if (tid < 12) flags[tid] = 0u;
    if (tid >= 64 && tid < 64 + ZB / 4) reinterpret_cast<unsigned*>(smem)[tid - 64] = 0u;     // the zero row masked fragment rows read
    __syncthreads();                                         // the only workgroup barrier before the epilogue; no DMA is in flight yet

    // the 16-byte store path of the epilogue (bf16 output, rows aligned): uniform over the launch
    const bool wide = !p.narrow_epi && (p.N & 7) == 0 && (p.ld_dst & 7) == 0 && (reinterpret_cast<unsigned long long>(p.dst) & 15) == 0;

    f32x4 acc[MT][NT];                                       // consumers only; zeros in the loaders (one epilogue call site for both);
                                                             // the narrow tile uses rows 0..3, the rest folds away
    f32x16_ acc32[2][4];                                     // M32: [channel tile][pixel tile]; whichever form is unused folds away
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int pt = 0; pt < 4; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[ct][pt][r] = 0.f;
#ifdef EP24_STAMPS
    unsigned long long st_t0 = 0, st_r0 = 0, st_t1 = 0, st_t2 = 0, st_spin = 0;
    unsigned st_nspin = 0;
#endif

    auto loader_main = [&]() {
        // =================================================================== LOADER
        const int lw = wave - NCW;
        // source chunk of the lane's LDS slot: slot ^ key(row).  16 x 16 form: key = row & 7 = (lane >> 3) & 7 for every 8-row piece.
        // M32: key = (row >> 1) & 7 = (lane >> 4) | 4 (piece & 1): the odd pieces' chunk is the even pieces' with bit 2 flipped.
        const int lchunk = M32 ? ((lane & 7) ^ ((lane >> 4) & 3)) : ((lane & 7) ^ ((lane >> 3) & 7));
        const int lchunk_odd = M32 ? (lchunk ^ 4) : lchunk;
        const auto src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.src), 0, p.src_bytes, 0x00020000);
        const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.wt), 0, p.wt_bytes, 0x00020000);
        const int kmax = (p.K - lchunk * 8 + BK - 1) / BK;   // chunks kc < kmax hold real channels for this lane
        const int kmax_odd = (p.K - lchunk_odd * 8 + BK - 1) / BK;
        const int ktail = (p.K & (BK - 1)) ? KC - 1 : KC;    // chunks >= ktail need the per-lane check
        const int NPW = (NP + NLW - 1) / NLW;                // patch pieces per loader and chunk (the last may be a duplicate)
        const unsigned msrc = (unsigned)((long)p.B * p.SH * p.SW);
        const int prow0 = (int)(m0 - halo) + (lane >> 3);
        const int ld2 = (int)p.ld_src * 2;
        const unsigned pv0 = (unsigned)prow0 * (unsigned)ld2 + lchunk * 16;
        const unsigned pv0_odd = (unsigned)prow0 * (unsigned)ld2 + lchunk_odd * 16;
        // piece i of this loader = piece g = min(i * NLW + lw, NP - 1) of the patch (a clamped index re-loads the last piece:
        // identical bytes to the same place, so that every step issues a fixed number of DMAs and the waits are immediates)
        auto issue_patch = [&](int pbuf_off, int kc, int i) {
            int g = (i < NPW ? i : NPW - 1) * NLW + lw;
            g = g < NP ? g : NP - 1;                           // scalar
            const unsigned ps = (unsigned)(prow0 + 8 * g);     // a row before the tensor wraps to a huge value and fails the test
            bool ok = ps < msrc;
            const bool odd = M32 && (g & 1);                   // scalar
            if (kc >= ktail) ok = ok && kc < (odd ? kmax_odd : kmax);
            const int vo = ok ? (int)((odd ? pv0_odd : pv0) + (unsigned)(8 * g * ld2 + kc * (BK * 2))) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lptr_t)(smem + ZB + pbuf_off + g * 1024), 16, vo, 0, 0, 0);
        };
        int wvoff[TB_INSTR];
#pragma unroll
        for (int i = 0; i < TB_INSTR; ++i) {
            const int q = (lw * TB_INSTR + i) * 8 + (lane >> 3);
            // 16 x 16 form: channel relabelling of the shared epilogue; M32: natural order (piece lw * 4 + i is odd when i is)
            const int r = M32 ? q : (q & ~63) + ((q & 15) << 2) + ((q >> 4) & 3);
            const int lc = (i & 1) ? lchunk_odd : lchunk;
            wvoff[i] = n0 + r < p.N ? (int)((((long)(n0 + r) * p.WT * p.K) + lc * 8) * 2) : OOB;
        }
        auto issue_b = [&](int stage, int t, int kc) {
            const int b_s = (t * p.K + kc * BK) * 2;
#pragma unroll
            for (int i = 0; i < TB_INSTR; ++i) {
                int vo = wvoff[i] == OOB ? OOB : wvoff[i] + b_s;
                if (kc >= ktail) vo = kc < ((i & 1) ? kmax_odd : kmax) ? vo : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wt_rsrc, (lptr_t)(smem + ZB + bb + stage * TB_BYTES + (lw * TB_INSTR + i) * 1024), 16, vo, 0, 0, 0);
            }
        };
        // the whole patch of chunk 0 first: the wait behind the first weight tile covers it
        for (int i = 0; i < NPW; ++i) issue_patch(0, 0, i);
        unsigned j = 0;
        if constexpr (TBN == 64) {
            // DEPTH + 1 steps in flight: a consumer step is ~600 - 900 cycles here and a DMA round trip 1 - 3 k (the 20 x 20 layers'
            // weights do not fit one XCD's L2), so the loaders run DEPTH steps ahead of the step they wait for; the ring has a stage
            // per tap (stage = tap) and the wait is counted exactly (loads return in order): the DMAs of the steps behind step jj stay
            // outstanding.  Patch pieces of the next chunk ride on taps >= S0 = DEPTH, so that running ahead never waits for the
            // consumers to leave the buffer they go to (chunk kc - 1's: read once min(free) >= 9 kc).
            constexpr int DEPTH = 5, S0 = 5;
            const int nsteps = 9 * KC;
            int t1 = 0, kc1 = 0, nissued = 0;                  // (tap, chunk) of the next step to issue
            auto issue_step = [&]() {
                if (nissued >= TNS) spin_until(f_free, (unsigned)(nissued - (TNS - 1)), f_err);
                issue_b(t1, t1, kc1);
                if (kc1 + 1 < KC && t1 >= S0) {
                    if (kc1 >= 1) spin_until(f_free, (unsigned)(9 * kc1), f_err);
                    const int pn = ((kc1 + 1) & 1) * PBYTES;
#pragma unroll
                    for (int i = 0; i < PPS; ++i) issue_patch(pn, kc1 + 1, (t1 - S0) * PPS + i);
                }
                if (++t1 == 9) { t1 = 0; ++kc1; }
                ++nissued;
            };
            auto wait_n = [&](int n) {                         // s_waitcnt takes an immediate
                switch (n) {
#define EP24_WCASE(N) case N: wait_vmcnt_c<N>(); break;
                    EP24_WCASE(0) EP24_WCASE(1) EP24_WCASE(2) EP24_WCASE(3) EP24_WCASE(4) EP24_WCASE(5) EP24_WCASE(6) EP24_WCASE(7)
                    EP24_WCASE(8) EP24_WCASE(9) EP24_WCASE(10) EP24_WCASE(11) EP24_WCASE(12) EP24_WCASE(13) EP24_WCASE(14) EP24_WCASE(15)
                    EP24_WCASE(16) EP24_WCASE(17) EP24_WCASE(18) EP24_WCASE(19) EP24_WCASE(20) EP24_WCASE(21) EP24_WCASE(22) EP24_WCASE(23)
                    EP24_WCASE(24) EP24_WCASE(25)
#undef EP24_WCASE
                    default: wait_vmcnt_c<0>(); break;
                }
            };
            static_assert(DEPTH * (TB_INSTR + PPS) <= 25 && DEPTH <= TNS - 2, "counted wait range / ring depth");
            for (int i = 0; i <= DEPTH && i < nsteps; ++i) issue_step();
            int tj = 1, kj = 0;                                // (tap, chunk) of step jj + 1
#pragma unroll 1
            for (int jj = 0; jj < nsteps; ++jj) {
                int n = 0, ts = tj, ks = kj;
                for (int q = jj + 1; q < nissued; ++q) {
                    n += TB_INSTR + ((ks + 1 < KC && ts >= S0) ? PPS : 0);
                    if (++ts == 9) { ts = 0; ++ks; }
                }
                wait_n(n);
                if (lane == 0) st_flag(f_full + lw, (unsigned)(jj + 1));
                cbar();
                if (nissued < nsteps) issue_step();
                if (++tj == 9) { tj = 0; ++kj; }
            }
            return;
        }
        for (int kc = 0; kc < KC; ++kc) {
            const int pnext = ((kc + 1) & 1) * PBYTES;
            const bool more = kc + 1 < KC;
#pragma unroll 1
            for (int ty = 0; ty < 3; ++ty) {
                auto step = [&](auto txc) {
                    constexpr int tx = decltype(txc)::value;
                    const int t = ty * 3 + tx;
                    // stage tx was last read in step j - 3: every consumer must have issued the reads of steps < j - 2
                    if (j >= (unsigned)NS) spin_until(f_free, j - (NS - 1), f_err);
                    issue_b(tx, t, kc);
                    const bool slice = more && t >= 2;         // taps 2..8: the next chunk's patch, PPS pieces per step
                    if (slice) {
#pragma unroll
                        for (int i = 0; i < PPS; ++i) issue_patch(pnext, kc + 1, (t - 2) * PPS + i);
                        wait_vmcnt_c<PPS>();                   // this step's weight tile and everything older has landed
                    } else {
                        wait_vmcnt_c<0>();
                    }
                    ++j;
                    if (lane == 0) st_flag(f_full + lw, j);
                    cbar();
                };
                step(std::integral_constant<int, 0>{});
                step(std::integral_constant<int, 1>{});
                step(std::integral_constant<int, 2>{});
            }
        }
        // nothing in flight: every step ended behind its wait (the last: vmcnt(0))
    };

    auto consumer_main = [&]() {
    // ======================================================================= CONSUMER
    if constexpr ((RING_VAR & 1) != 0) __builtin_amdgcn_s_setprio(2);
    const int cw = wave, wm = cw >> 1, wn = cw & 1;
    const int frow = lane & 15, fq = lane >> 4;

    // per-lane tap masks of the eight fragment rows: bit t = tap t reads a pixel inside the image (three row cases x three column
    // cases composed from two small tables).  Packed nine bits per row, three rows per register.
    unsigned vmp[3] = {0u, 0u, 0u};
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long m = m0 + wm * (MT * 16) + i * 16 + frow;
        unsigned mk = 0;
        if (m < p.M) {
            const int mm = (int)m;
            const int n = fdiv(mm, p.d_plane);
            const int rem = mm - n * (p.GH * p.GW);
            const int y = fdiv(rem, p.d_gw), x = rem - y * p.GW;
            unsigned rowm = 0x038u, colm = 0x092u;             // the middle tap row (taps 3..5) / middle column (taps 1, 4, 7)
            if (y - sgn >= 0 && y - sgn < p.SH) rowm |= 0x007u;    // tap row 0 reads image row y - sgn
            if (y + sgn >= 0 && y + sgn < p.SH) rowm |= 0x1C0u;    // tap row 2
            if (x - sgn >= 0 && x - sgn < p.SW) colm |= 0x049u;    // tap column 0
            if (x + sgn >= 0 && x + sgn < p.SW) colm |= 0x124u;    // tap column 2
            mk = rowm & colm;
        }
        vmp[i / 3] |= mk << (9 * (i % 3));
    }

    // LDS read addresses.  Fragment row i of the wave sits at patch row arow0 + 16 i + shift(tap); 16 i and the k half leave the
    // swizzle key (row & 7) alone, so the address of row i is the tap's address + 2048 i and the second k half is that with bit 6
    // flipped.  A row that the tap takes from across an image border (or a row >= M) is not masked in registers as conv_patch.hip
    // does (4 v_and per fragment behind a scalar branch): its ADDRESS is and-ed with 0 / -1, and LDS bytes 0..127 are zeros -
    // two instructions per row, no branch, the step is one basic block.
    const int arow0 = wm * (MT * 16) + frow + halo;
    const int bo0 = ZB + bb + (wn * 64 + frow) * 128 + ((fq ^ (frow & 7)) << 4);
    auto tap_addr = [&](int pbuf_off, int sh) {
        const int q0 = arow0 + sh;
        return ZB + pbuf_off + q0 * 128 + ((fq ^ (q0 & 7)) << 4);
    };
    // masked address of fragment row i (k half 0) for the tap whose address is ta and whose mask bit is t
    auto row_addr = [&](int ta, int i, int t) {
        const int m = __builtin_amdgcn_sbfe((int)vmp[i / 3], (unsigned)(9 * (i % 3) + t), 1u);      // 0 or -1
        if constexpr ((RING_VAR & 4) != 0) return ta + i * 2048;
        return (ta + i * 2048) & m;
    };
    // fragment reads take the LDS byte offset as the address: smem is the kernel's only LDS object (offset 0, checked at the top),
    // and written as smem + off the compiler materialised "0 + off" in a VALU instruction per read
    typedef const __attribute__((address_space(3))) bf16x8* lds_frag_p;
    auto lds_frag = [&](int off) { return *reinterpret_cast<lds_frag_p>((unsigned long)(unsigned)off); };
    auto pin = [](int& v) { asm volatile("" : "+v"(v)); };   // keeps an address where it is computed (the compiler sank them behind the branch)
    auto mma_row = [&](int i, const bf16x8& fa, const bf16x8 (&fb)[NT]) {
#pragma unroll
        for (int q = 0; q < NT; ++q) mfma_acc(acc[i][q], fa, fb[q]);
    };

#ifdef EP24_STAMPS
    st_t0 = STAMP(); st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    // A step is 16 groups of 4 MFMAs (k half h = g / 8, fragment row i = g % 8: 64 matrix-pipe cycles each).  The A fragment of group
    // g is requested LA groups earlier, the weight fragments of a half during the half before: LA + 1 A fragments and both halves'
    // weight fragments are live (52 registers at LA = 4 beside 128 accumulators; a half-step double buffer held 96 and spilled, and
    // so do LA = 6 and 8 - 40 / 80 registers in the loop).  Groups 0..LA-1 of a step are requested during the previous step (pa[],
    // fb0[]), behind the check that the step has landed: that check sits in the middle of a step, the loaders' counters are
    // requested four groups before it.
    spin_until(f_full, 1u, f_err);                           // step 0: the patch of chunk 0 and weight tile 0
    bf16x8 pa[LA], fb0[NT], fb1[NT];
    int ta = tap_addr(0, sgn * (-p.SW - 1));                 // tap 0 of chunk 0
    int ra[MT];                                              // masked row addresses of the current tap (k half 0)
#pragma unroll
    for (int i = 0; i < MT; ++i) ra[i] = row_addr(ta, i, 0);
#pragma unroll
    for (int g = 0; g < LA; ++g) pa[g] = lds_frag(ra[g % MT] ^ (g >= MT ? 64 : 0));
#pragma unroll
    for (int q = 0; q < NT; ++q) fb0[q] = lds_frag(bo0 + q * 2048);
#ifdef EP24_STAMPS
    st_t1 = STAMP();
#endif
    unsigned j = 0;
    for (int kc = 0; kc < KC; ++kc) {
        const int pcur = (npb == 2) ? (kc & 1) * PBYTES : 0;
        const int pnext = ((kc + 1) & 1) * PBYTES;
        const bool more = kc + 1 < KC;
#pragma unroll 1
        for (int ty = 0; ty < 3; ++ty) {
            const int shrow = sgn * (ty - 1) * p.SW;
            auto tap_step = [&](auto txc) {
                constexpr int tx = decltype(txc)::value;
                const int t = ty * 3 + tx;
                const int bst = bo0 + tx * B_BYTES;            // this step's weight tile (ring stage = tap column)
                const int bsn = bo0 + ((tx + 1) % 3) * B_BYTES;   // the next step's
                const bool last = tx == 2 && ty == 2 && !more;
                const int tn = (tx < 2 || ty < 2) ? t + 1 : 0;
                const int tan = (tx < 2) ? tap_addr(pcur, shrow + sgn * tx)
                                         : (ty < 2 ? tap_addr(pcur, shrow + sgn * (p.SW - 1))
                                                   : tap_addr(pnext, sgn * (-p.SW - 1)));
                bf16x8 F[16];                                  // this step's A fragments; F[0..LA-1] arrive as pa[]
                v4u fl = {0u, 0u, 0u, 0u};
                // group 8 looks at the loaders' counters (requested at group 4): from there on the next step's fragments may be
                // requested - its weight fragments at groups 8..11 (fb0 is dead behind group 7), its A fragments from group 16 - LA on
                // (requests for a step that does not exist - behind the last one - read LDS bytes nobody uses: unconditional, so that
                // the groups stay in one basic block; only the check itself looks at `last`)
                constexpr int G_CHECK = 8, G_FLAGS = 4;
                static_assert(LA >= 1 && LA <= 8, "LA");
                auto group = [&](auto gc) {
                    constexpr int g = decltype(gc)::value;
                    constexpr int gr = g + LA;                 // the group whose A fragment is requested now
                    if constexpr (g == G_CHECK) {
                        if (!last) {
                            // the next step's weight tile (and at tap 8 the next chunk's patch) must have landed before its first reads
                            const unsigned have = __builtin_amdgcn_readfirstlane(min(min(fl.x, fl.y), min(fl.z, fl.w)));
                            // (this is step j, 0-based: the next one has landed once the loaders' counters have reached j + 2)
                            if (have < j + 2) {
#ifdef EP24_STAMPS
                                const unsigned long long s0 = STAMP();
                                spin_until(f_full, j + 2, f_err);
                                st_spin += STAMP() - s0; ++st_nspin;
#else
                                spin_until(f_full, j + 2, f_err);
#endif
                            }
                            cbar();
                        }
                    }
                    if constexpr (gr < 16) {
                        F[gr] = lds_frag(ra[gr % MT] ^ (gr >= MT ? 64 : 0));
                    } else {
                        pa[gr - 16] = lds_frag(ra[(gr - 16) % MT] ^ (gr - 16 >= MT ? 64 : 0));
                    }
                    if constexpr (g < 4) fb1[g] = lds_frag((bst ^ 64) + g * 2048);
                    if constexpr (g >= G_CHECK && g < G_CHECK + 4) {
                        fb0[g - G_CHECK] = lds_frag(bsn + (g - G_CHECK) * 2048);
                    }
                    if constexpr (g == G_FLAGS) {
                        fl = ld_flags4(f_full);
                    }
                    if constexpr (gr == 15) {
                        // every fragment read of this step is issued (LDS executes a wave's operations in order): its ring stage
                        // may be refilled
                        cbar();
                        if (lane == 0) st_flag(f_free + cw, j + 1);
                        cbar();
                    }
                    if constexpr (g < LA) mma_row(g % MT, pa[g], g < MT ? fb0 : fb1);
                    else mma_row(g % MT, F[g], g < MT ? fb0 : fb1);
                    // the next tap's masked row addresses, in place: row i was last used by group 8 + i - LA (its k half 1 request)
                    // and is next needed by group 16 - LA + i (the next step's k half 0 request)
                    if constexpr (g >= 9 - LA && g < 9 - LA + MT) {
                        constexpr int i = g - (9 - LA);
                        ra[i] = row_addr(tan, i, tn);
                        pin(ra[i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
                group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
                group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
                group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
                group(std::integral_constant<int, 8>{}); group(std::integral_constant<int, 9>{});
                group(std::integral_constant<int, 10>{}); group(std::integral_constant<int, 11>{});
                group(std::integral_constant<int, 12>{}); group(std::integral_constant<int, 13>{});
                group(std::integral_constant<int, 14>{}); group(std::integral_constant<int, 15>{});
                ++j;
            };
            tap_step(std::integral_constant<int, 0>{});
            tap_step(std::integral_constant<int, 1>{});
            tap_step(std::integral_constant<int, 2>{});
        }
    }
#ifdef EP24_STAMPS
    st_t2 = STAMP();
#endif
    };

    // ======================================================================= CONSUMER of the narrow tile (TBN = 64)
    // Consumer cw multiplies rows [64 cw, 64 cw + 64) with all 64 channels.  A step is 8 groups of 4 MFMAs (k half h = g / 4, fragment
    // row i = g % 4); the A fragment of a group is requested 4 groups earlier, this step's second-half weight fragments in groups 0 - 1,
    // the next step's first-half ones in groups 4 - 5 behind the FULL check of group 4 (flags requested in group 2).
    auto consumer_main_narrow = [&]() {
        const int cw = wave;
        const int frow = lane & 15, fq = lane >> 4;
        unsigned vmp[2] = {0u, 0u};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long m = m0 + cw * 64 + i * 16 + frow;
            unsigned mk = 0;
            if (m < p.M) {
                const int mm = (int)m;
                const int n = fdiv(mm, p.d_plane);
                const int rem = mm - n * (p.GH * p.GW);
                const int y = fdiv(rem, p.d_gw), x = rem - y * p.GW;
                unsigned rowm = 0x038u, colm = 0x092u;
                if (y - sgn >= 0 && y - sgn < p.SH) rowm |= 0x007u;
                if (y + sgn >= 0 && y + sgn < p.SH) rowm |= 0x1C0u;
                if (x - sgn >= 0 && x - sgn < p.SW) colm |= 0x049u;
                if (x + sgn >= 0 && x + sgn < p.SW) colm |= 0x124u;
                mk = rowm & colm;
            }
            vmp[i >> 1] |= mk << (9 * (i & 1));
        }
        const int arow0 = cw * 64 + frow + halo;
        const int bo0 = ZB + bb + frow * 128 + ((fq ^ (frow & 7)) << 4);
        auto tap_addr = [&](int pbuf_off, int sh) {
            const int q0 = arow0 + sh;
            return ZB + pbuf_off + q0 * 128 + ((fq ^ (q0 & 7)) << 4);
        };
        auto row_addr = [&](int ta, int i, int t) {
            const int m = __builtin_amdgcn_sbfe((int)vmp[i >> 1], (unsigned)(9 * (i & 1) + t), 1u);      // 0 or -1
            return (ta + i * 2048) & m;
        };
        typedef const __attribute__((address_space(3))) bf16x8* lds_frag_p;
        auto lds_frag = [&](int off) { return *reinterpret_cast<lds_frag_p>((unsigned long)(unsigned)off); };
        auto pin = [](int& v) { asm volatile("" : "+v"(v)); };
        auto mma_row = [&](int i, const bf16x8& fa, const bf16x8 (&fb)[NT]) {
#pragma unroll
            for (int q = 0; q < NT; ++q) mfma_acc(acc[i][q], fa, fb[q]);
        };
#ifdef EP24_STAMPS
        st_t0 = STAMP(); st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        spin_until(f_full, 1u, f_err);
        bf16x8 pa[4], fb0[NT], fb1[NT];
        int ra[4];
        {
            const int ta = tap_addr(0, sgn * (-p.SW - 1));
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = row_addr(ta, i, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) pa[g] = lds_frag(ra[g]);
#pragma unroll
        for (int q = 0; q < NT; ++q) fb0[q] = lds_frag(bo0 + q * 2048);
#ifdef EP24_STAMPS
        st_t1 = STAMP();
#endif
        unsigned j = 0;
        for (int kc = 0; kc < KC; ++kc) {
            const int pcur = (npb == 2) ? (kc & 1) * PBYTES : 0;
            const int pnext = ((kc + 1) & 1) * PBYTES;
            const bool more = kc + 1 < KC;
#pragma unroll 1
            for (int ty = 0; ty < 3; ++ty) {
                const int shrow = sgn * (ty - 1) * p.SW;
                auto tap_step = [&](auto txc) {
                    constexpr int tx = decltype(txc)::value;
                    const int t = ty * 3 + tx;
                    const bool last = tx == 2 && ty == 2 && !more;
                    const int tn = (tx < 2 || ty < 2) ? t + 1 : 0;
                    const int bst = bo0 + t * TB_BYTES;        // ring stage = tap
                    const int bsn = bo0 + tn * TB_BYTES;
                    const int tan = (tx < 2) ? tap_addr(pcur, shrow + sgn * tx)
                                             : (ty < 2 ? tap_addr(pcur, shrow + sgn * (p.SW - 1))
                                                       : tap_addr(pnext, sgn * (-p.SW - 1)));
                    bf16x8 F[4];                               // this step's k half 1 fragments (k half 0 arrives as pa[])
                    v4u fl = {0u, 0u, 0u, 0u};
                    auto group = [&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        if constexpr (g == 4) {
                            if (!last) {
                                const unsigned have = __builtin_amdgcn_readfirstlane(min(min(fl.x, fl.y), min(fl.z, fl.w)));
                                if (have < j + 2) {            // this is step j (0-based): the next one must have landed
#ifdef EP24_STAMPS
                                    const unsigned long long s0 = STAMP();
                                    spin_until(f_full, j + 2, f_err);
                                    st_spin += STAMP() - s0; ++st_nspin;
#else
                                    spin_until(f_full, j + 2, f_err);
#endif
                                }
                                cbar();
                            }
                        }
                        if constexpr (g < 4) F[g] = lds_frag(ra[g] ^ 64);
                        else pa[g - 4] = lds_frag(ra[g - 4]);
                        if constexpr (g < 2) {
                            fb1[2 * g] = lds_frag((bst ^ 64) + (2 * g) * 2048);
                            fb1[2 * g + 1] = lds_frag((bst ^ 64) + (2 * g + 1) * 2048);
                        }
                        if constexpr (g == 4 || g == 5) {
                            fb0[2 * (g - 4)] = lds_frag(bsn + (2 * (g - 4)) * 2048);
                            fb0[2 * (g - 4) + 1] = lds_frag(bsn + (2 * (g - 4) + 1) * 2048);
                        }
                        if constexpr (g == 2) fl = ld_flags4(f_full);
                        if constexpr (g == 3) {
                            // every fragment read of this step is issued: its ring stage may be refilled
                            cbar();
                            if (lane == 0) st_flag(f_free + cw, j + 1);
                            cbar();
                        }
                        if constexpr (g < 4) mma_row(g, pa[g], fb0);
                        else mma_row(g - 4, F[g - 4], fb1);
                        // the next tap's masked row addresses, in place: row i was last used by group i (its k half 1 request) and is next
                        // needed by group 4 + i (the next step's k half 0 request)
                        if constexpr (g >= 1 && g < 5) {
                            constexpr int i = g - 1;
                            ra[i] = row_addr(tan, i, tn);
                            pin(ra[i]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
                    group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
                    group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
                    group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
                    ++j;
                };
                tap_step(std::integral_constant<int, 0>{});
                tap_step(std::integral_constant<int, 1>{});
                tap_step(std::integral_constant<int, 2>{});
            }
        }
#ifdef EP24_STAMPS
        st_t2 = STAMP();
#endif
    };

    // ======================================================================= CONSUMER, 32 x 32 x 16 form
    auto consumer_main32 = [&]() {
        const int cw = wave, wm = cw >> 1, wn = cw & 1;
        const int pr = lane & 31, h = lane >> 5;
        // per-lane tap masks of the four pixel tiles (9 bits each): two registers
        unsigned vmp[2] = {0u, 0u};
#pragma unroll
        for (int pt = 0; pt < 4; ++pt) {
            const long m = m0 + wm * (MT * 16) + pt * 32 + pr;
            unsigned mk = 0;
            if (m < p.M) {
                const int mm = (int)m;
                const int n = fdiv(mm, p.d_plane);
                const int rem = mm - n * (p.GH * p.GW);
                const int y = fdiv(rem, p.d_gw), x = rem - y * p.GW;
                unsigned rowm = 0x038u, colm = 0x092u;
                if (y - sgn >= 0 && y - sgn < p.SH) rowm |= 0x007u;
                if (y + sgn >= 0 && y + sgn < p.SH) rowm |= 0x1C0u;
                if (x - sgn >= 0 && x - sgn < p.SW) colm |= 0x049u;
                if (x + sgn >= 0 && x + sgn < p.SW) colm |= 0x124u;
                mk = rowm & colm;
            }
            vmp[pt >> 1] |= mk << (9 * (pt & 1));
        }
        // Fragment of (pixel tile pt, k-step s): patch row q = arow0 + 32 pt + shift(tap), chunk 2 s + h, key (q >> 1) & 7 - 32 pt
        // leaves the key alone and s only flips bits 5..6 of the address: address = ((tap address + 4096 pt) & mask) ^ (s << 5).
        // Masked rows read the zero bytes 0..127 (0 ^ 32 s stays inside them).
        const int arow0 = wm * (MT * 16) + pr + halo;
        auto tap_addr = [&](int pbuf_off, int sh) {
            const int q0 = arow0 + sh;
            return ZB + pbuf_off + q0 * 128 + ((h ^ ((q0 >> 1) & 7)) << 4);
        };
        auto row_addr = [&](int ta, int pt, int t) {
            const int m = __builtin_amdgcn_sbfe((int)vmp[pt >> 1], (unsigned)(9 * (pt & 1) + t), 1u);
            if constexpr ((RING_VAR & 4) != 0) return ta + pt * 4096;
            return (ta + pt * 4096) & m;
        };
        // weight fragment (channel tile ct, k-step s): tile row wn * 64 + 32 ct + pr; stage as an offset constant
        int wb[2][4];
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int n = wn * 64 + ct * 32 + pr;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) wb[ct][ks] = ZB + bb + n * 128 + (((2 * ks + h) ^ ((n >> 1) & 7)) << 4);
        }
        typedef const __attribute__((address_space(3))) bf16x8* lds_frag_p;
        auto lds_frag = [&](int off) { return *reinterpret_cast<lds_frag_p>((unsigned long)(unsigned)off); };
        auto pin = [](int& v) { asm volatile("" : "+v"(v)); };
#ifdef EP24_STAMPS
        st_t0 = STAMP(); st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
        // A step is 16 groups of 2 MFMAs (k-step s = g / 4, pixel tile pt = g % 4, both channel tiles: 64 matrix-pipe cycles); the
        // pixel fragment of group g is requested LA groups earlier, the two weight fragments of a k-step during the k-step before.
        spin_until(f_full, 1u, f_err);
        bf16x8 pa[LA], wp[2];
        int ra[4];
        {
            const int ta = tap_addr(0, sgn * (-p.SW - 1));
#pragma unroll
            for (int pt = 0; pt < 4; ++pt) ra[pt] = row_addr(ta, pt, 0);
        }
#pragma unroll
        for (int g = 0; g < LA; ++g) pa[g] = lds_frag(ra[g & 3] ^ ((g >> 2) << 5));
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) wp[ct] = lds_frag(wb[ct][0]);
#ifdef EP24_STAMPS
        st_t1 = STAMP();
#endif
        unsigned j = 0;
        for (int kc = 0; kc < KC; ++kc) {
            const int pcur = (npb == 2) ? (kc & 1) * PBYTES : 0;
            const int pnext = ((kc + 1) & 1) * PBYTES;
            const bool more = kc + 1 < KC;
#pragma unroll 1
            for (int ty = 0; ty < 3; ++ty) {
                const int shrow = sgn * (ty - 1) * p.SW;
                auto tap_step = [&](auto txc) {
                    constexpr int tx = decltype(txc)::value;
                    const int t = ty * 3 + tx;
                    constexpr int so = tx * B_BYTES, son = ((tx + 1) % 3) * B_BYTES;     // ring stage of this step / the next
                    const bool last = tx == 2 && ty == 2 && !more;
                    const int tn = (tx < 2 || ty < 2) ? t + 1 : 0;
                    const int tan = (tx < 2) ? tap_addr(pcur, shrow + sgn * tx)
                                             : (ty < 2 ? tap_addr(pcur, shrow + sgn * (p.SW - 1))
                                                       : tap_addr(pnext, sgn * (-p.SW - 1)));
                    bf16x8 F[16], W[4][2];
                    v4u fl = {0u, 0u, 0u, 0u};
                    constexpr int G_CHECK = 8, G_FLAGS = 4;
                    auto group = [&](auto gc) {
                        constexpr int g = decltype(gc)::value;
                        constexpr int gr = g + LA, ks = g >> 2, pt = g & 3;
                        if constexpr (g == G_CHECK) {
                            if (!last) {
                                const unsigned have = __builtin_amdgcn_readfirstlane(min(min(fl.x, fl.y), min(fl.z, fl.w)));
                                if (have < j + 2) {           // this is step j (0-based): the next one must have landed
#ifdef EP24_STAMPS
                                    const unsigned long long s0 = STAMP();
                                    spin_until(f_full, j + 2, f_err);
                                    st_spin += STAMP() - s0; ++st_nspin;
#else
                                    spin_until(f_full, j + 2, f_err);
#endif
                                }
                                cbar();
                            }
                        }
                        // pixel fragment of group g + LA (this step's, or the next step's behind the check)
                        if constexpr (gr < 16) F[gr] = lds_frag(ra[gr & 3] ^ ((gr >> 2) << 5));
                        else pa[gr - 16] = lds_frag(ra[(gr - 16) & 3] ^ (((gr - 16) >> 2) << 5));
                        // weight fragments of the next k-step: two of the four groups of a k-step request one each
                        if constexpr (pt < 2) {
                            if constexpr (ks < 3) W[ks + 1][pt] = lds_frag(wb[pt][ks + 1] + so);
                            else wp[pt] = lds_frag(wb[pt][0] + son);
                        }
                        if constexpr (g == G_FLAGS) fl = ld_flags4(f_full);
                        if constexpr (g == 9) {
                            // every fragment read of this step is issued (pixel fragments by group 7, the last weight fragments by
                            // group 9): its ring stage may be refilled
                            cbar();
                            if (lane == 0) st_flag(f_free + cw, j + 1);
                            cbar();
                        }
                        auto mma2 = [&](const bf16x8& fx) {
                            if constexpr (ks == 0) { mfma32_acc(acc32[0][pt], wp[0], fx); mfma32_acc(acc32[1][pt], wp[1], fx); }
                            else { mfma32_acc(acc32[0][pt], W[ks][0], fx); mfma32_acc(acc32[1][pt], W[ks][1], fx); }
                        };
                        if constexpr (g < LA) mma2(pa[g]); else mma2(F[g]);
                        // the next tap's masked row addresses, in place: tile pt was last used by group 12 + pt - LA (its k-step 3
                        // request) and is next needed by group 16 - LA + pt
                        if constexpr (g >= 13 - LA && g < 13 - LA + 4) {
                            constexpr int i = g - (13 - LA);
                            ra[i] = row_addr(tan, i, tn);
                            pin(ra[i]);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    };
                    group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
                    group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
                    group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
                    group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
                    group(std::integral_constant<int, 8>{}); group(std::integral_constant<int, 9>{});
                    group(std::integral_constant<int, 10>{}); group(std::integral_constant<int, 11>{});
                    group(std::integral_constant<int, 12>{}); group(std::integral_constant<int, 13>{});
                    group(std::integral_constant<int, 14>{}); group(std::integral_constant<int, 15>{});
                    ++j;
                };
                tap_step(std::integral_constant<int, 0>{});
                tap_step(std::integral_constant<int, 1>{});
                tap_step(std::integral_constant<int, 2>{});
            }
        }
#ifdef EP24_STAMPS
        st_t2 = STAMP();
#endif
    };

    const bool consumer = wave < NCW;
    if constexpr (M32) {
        if (consumer) consumer_main32(); else loader_main();
        mfma_drain();
        ring_epilogue32(p, acc32, consumer, m0, n0, tile_m, smem);      // the host admits M32 only with the 16-byte store path
    } else {
        if constexpr (TBN == 64) { if (consumer) consumer_main_narrow(); else loader_main(); }
        else { if (consumer) consumer_main(); else loader_main(); }
        mfma_drain();
        // LDS is nobody's any more: the last DMA landed before the last full count.  16-byte store path: all eight waves (the loaders
        // carry half of the stores); otherwise the loaders are done (a terminated wave no longer counts at s_barrier) and the four
        // consumers run the shared epilogue (a 2 x 2 grid of 128 x 64 pieces; the narrow tile: four 64 x 64 pieces over the rows).
        f32x4 (&acc_t)[CMT][NT] = reinterpret_cast<f32x4 (&)[CMT][NT]>(acc);
        if (wide) ring_epilogue<TBN, CMT, INFER>(p, acc_t, consumer, m0, n0, tile_m, smem);
        else if (consumer) igemm_epilogue<TBN, false, CMT, 0, NCW, false>(p, acc_t, m0, n0, tile_m, smem);
        else return;
    }
    const int cw = wave; (void)cw;
#ifdef EP24_STAMPS
    if (blockIdx.x < 32 && lane == 0 && (cw == 0 || cw == NCW - 1)) {
        unsigned long long* o = g_ring_stamps + (blockIdx.x * 2 + (cw != 0)) * 8;
        o[0] = st_t1 - st_t0; o[1] = st_t2 - st_t1; o[2] = STAMP() - st_t2; o[3] = st_spin; o[4] = st_nspin;
        o[5] = __builtin_amdgcn_s_memrealtime() - st_r0; o[6] = STAMP() - st_t0; o[7] = 9 * KC;
    }
#endif
}

#ifdef EP24_AB_VARIANTS      // measured and lost (profiles/r04_ring_generic_ab.txt): only in the A/B library (make variants)
// ---------------------------------------------------------------------------------------------------------------------------
// The same ring WITHOUT a patch: any gather-GEMM of igemm.h (1x1 layers with K > 128, stride-2 3x3 layers, ...).  A step is
// (tap t, 64-channel chunk kc), chunk outer / tap inner as the tiled kernel walks them (bit-identical sums); the loaders fetch a
// [256 pixels][64 k] A tile per step the way the tiled kernel does (per-lane row offsets and tap masks hoisted out of the loop,
// padding / tails as out-of-range offsets that the DMA zero-fills) next to the weight tile: 48 KB per stage, three stages.  What
// it buys over igemm_dma_kernel: a 4 .. 16-step layer is ONE DMA round trip deep instead of one per step (three stages are
// requested before the first is needed, nobody waits at a barrier), 24 fragment reads per 64 MFMAs instead of 32, and the
// consumers have nothing to compute but MFMAs: every fragment address is a base plus an immediate.
constexpr int GA_BYTES = RBM * 128, GSTAGE = GA_BYTES + B_BYTES;       // 32 KB + 16 KB
constexpr int GA_INSTR = RBM / 8 / NLW;                                  // A-tile DMA instructions per loader and step (8)

__global__ __launch_bounds__((NCW + NLW) * 64) void conv_ring_generic_kernel(const IgemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tile_id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const int tiles_n = (p.N + RBN - 1) / RBN;
    const int tile_m = tile_id / tiles_n;
    const long m0 = (long)tile_m * RBM;
    const int n0 = (tile_id - tile_m * tiles_n) * RBN;
    const int KC = (p.K + BK - 1) / BK;
    const unsigned S = (unsigned)(p.T * KC);                 // steps
    unsigned* const flags = reinterpret_cast<unsigned*>(smem + ZB + NS * GSTAGE);
    lds_u32* const f_full = (lds_u32*)(lptr_t)flags;
    lds_u32* const f_free = f_full + 4;
    lds_u32* const f_err = f_full + 8;
    if (reinterpret_cast<unsigned long>((lptr_t)smem) != 0ul) {
        if (tid == 0) atomicAdd(&g_ring_timeouts, 1u << 16);
        return;
    }
    if (tid < 12) flags[tid] = 0u;
    __syncthreads();
    const bool wide = !p.narrow_epi && (p.N & 7) == 0 && (p.ld_dst & 7) == 0 && (reinterpret_cast<unsigned long long>(p.dst) & 15) == 0;

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto loader_main = [&]() {
        const int lw = wave - NCW;
        const int lchunk = (lane & 7) ^ ((lane >> 3) & 7);
        const auto src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.src), 0, p.src_bytes, 0x00020000);
        const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.wt), 0, p.wt_bytes, 0x00020000);
        const int kmax = (p.K - lchunk * 8 + BK - 1) / BK;
        const int ktail = (p.K & (BK - 1)) ? KC - 1 : KC;
        int rowoff[GA_INSTR];
        unsigned vmask[GA_INSTR];
#pragma unroll
        for (int i = 0; i < GA_INSTR; ++i) {
            const long m = m0 + (lw * GA_INSTR + i) * 8 + (lane >> 3);
            const bool rv = m < p.M;
            const int mm = rv ? (int)m : 0;
            const int n = fdiv(mm, p.d_plane);
            const int rem = mm - n * (p.GH * p.GW);
            const int gy = fdiv(rem, p.d_gw), gx = rem - gy * p.GW;
            const int iy0 = gy * p.sy, ix0 = gx * p.sx;
            rowoff[i] = (int)((((long)n * p.SH * p.SW + (long)iy0 * p.SW + ix0) * p.ld_src + lchunk * 8) * 2);
            unsigned mk = 0;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int iy = iy0 + p.oy[t], ix = ix0 + p.ox[t];
                if (t < p.T && rv && iy >= 0 && iy < p.SH && ix >= 0 && ix < p.SW) mk |= 1u << t;
            }
            vmask[i] = mk;
        }
        int wvoff[B_INSTR];
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int q = (lw * B_INSTR + i) * 8 + (lane >> 3);
            const int r = (q & ~63) + ((q & 15) << 2) + ((q >> 4) & 3);
            wvoff[i] = n0 + r < p.N ? (int)((((long)(n0 + r) * p.WT * p.K) + lchunk * 8) * 2) : OOB;
        }
        int is_t = 0, is_kc = 0, st = 0;
        for (unsigned j = 0; j < S; ++j) {
            // stage st was last read in step j - 3
            if (j >= (unsigned)NS) spin_until(f_free, j - (NS - 1), f_err);
            const int t = is_t, kc = is_kc;
            if (++is_t == p.T) { is_t = 0; ++is_kc; }
            const bool tail = kc >= ktail;
            const int ktm = tail ? -(int)(kc < kmax) : -1;
            const int a_s = p.toff[t] + kc * (BK * 2);
            const unsigned b_s = (unsigned)((p.wslot[t] * p.K + kc * BK) * 2);
            char* stage = smem + ZB + st * GSTAGE;
#pragma unroll
            for (int i = 0; i < GA_INSTR; ++i) {
                const int m = -(int)((vmask[i] >> t) & 1u) & ktm;
                const int vo = ((rowoff[i] + a_s) & m) | (OOB & ~m);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lptr_t)(stage + (lw * GA_INSTR + i) * 1024), 16, vo, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < B_INSTR; ++i) {
                const int v = (int)((unsigned)wvoff[i] + b_s);
                const int vo = (v & ktm) | (OOB & ~ktm);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(wt_rsrc, (lptr_t)(stage + GA_BYTES + (lw * B_INSTR + i) * 1024), 16, vo, 0, 0, 0);
            }
            st = st == NS - 1 ? 0 : st + 1;
            // two steps in flight: the wait behind the issue of step j retires step j - 1 (everything but the youngest 12)
            if (j > 0) {
                wait_vmcnt_c<GA_INSTR + B_INSTR>();
                if (lane == 0) st_flag(f_full + lw, j);
                cbar();
            }
        }
        wait_vmcnt_c<0>();
        if (lane == 0) st_flag(f_full + lw, S);
        cbar();
    };

    auto consumer_main = [&]() {
        const int cw = wave, wm = cw >> 1, wn = cw & 1;
        const int frow = lane & 15, fq = lane >> 4;
        const int ao0 = ZB + (wm * (MT * 16) + frow) * 128 + ((fq ^ (frow & 7)) << 4);
        const int bo0 = ZB + GA_BYTES + (wn * 64 + frow) * 128 + ((fq ^ (frow & 7)) << 4);
        typedef const __attribute__((address_space(3))) bf16x8* lds_frag_p;
        auto lds_frag = [&](int off) { return *reinterpret_cast<lds_frag_p>((unsigned long)(unsigned)off); };
        auto mma_row = [&](int i, const bf16x8& fa, const bf16x8 (&fb)[NT]) {
#pragma unroll
            for (int q = 0; q < NT; ++q) mfma_acc(acc[i][q], fa, fb[q]);
        };
        spin_until(f_full, 1u, f_err);
        bf16x8 pa[LA], fb0[NT], fb1[NT];
#pragma unroll
        for (int g = 0; g < LA; ++g) pa[g] = lds_frag((g >= MT ? ao0 ^ 64 : ao0) + (g % MT) * 2048);
#pragma unroll
        for (int q = 0; q < NT; ++q) fb0[q] = lds_frag(bo0 + q * 2048);
        unsigned j = 0;
        int st = 0;
#pragma unroll 1
        for (unsigned jj = 0; jj < S; ++jj) {
            const int stn = st == NS - 1 ? 0 : st + 1;
            const bool last = jj + 1 == S;
            const int a0 = ao0 + st * GSTAGE, a1 = a0 ^ 64, an = ao0 + stn * GSTAGE;
            const int bst = bo0 + st * GSTAGE, bsn = bo0 + stn * GSTAGE;
            bf16x8 F[16];
            v4u fl = {0u, 0u, 0u, 0u};
            constexpr int G_CHECK = 8, G_FLAGS = 4;
            auto group = [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int gr = g + LA;
                if constexpr (g == G_CHECK) {
                    if (!last) {
                        const unsigned have = __builtin_amdgcn_readfirstlane(min(min(fl.x, fl.y), min(fl.z, fl.w)));
                        if (have < j + 2) spin_until(f_full, j + 2, f_err);     // this is step j (0-based): the next one must have landed
                        cbar();
                    }
                }
                if constexpr (gr < 16) {
                    F[gr] = lds_frag((gr >= MT ? a1 : a0) + (gr % MT) * 2048);
                } else {
                    pa[gr - 16] = lds_frag((gr - 16 >= MT ? an ^ 64 : an) + ((gr - 16) % MT) * 2048);
                }
                if constexpr (g < 4) fb1[g] = lds_frag((bst ^ 64) + g * 2048);
                if constexpr (g >= G_CHECK && g < G_CHECK + 4) {
                    fb0[g - G_CHECK] = lds_frag(bsn + (g - G_CHECK) * 2048);
                }
                if constexpr (g == G_FLAGS) {
                    fl = ld_flags4(f_full);
                }
                if constexpr (gr == 15) {
                    cbar();
                    if (lane == 0) st_flag(f_free + cw, j + 1);
                    cbar();
                }
                if constexpr (g < LA) mma_row(g % MT, pa[g], g < MT ? fb0 : fb1);
                else mma_row(g % MT, F[g], g < MT ? fb0 : fb1);
                __builtin_amdgcn_sched_barrier(0);
            };
            group(std::integral_constant<int, 0>{}); group(std::integral_constant<int, 1>{});
            group(std::integral_constant<int, 2>{}); group(std::integral_constant<int, 3>{});
            group(std::integral_constant<int, 4>{}); group(std::integral_constant<int, 5>{});
            group(std::integral_constant<int, 6>{}); group(std::integral_constant<int, 7>{});
            group(std::integral_constant<int, 8>{}); group(std::integral_constant<int, 9>{});
            group(std::integral_constant<int, 10>{}); group(std::integral_constant<int, 11>{});
            group(std::integral_constant<int, 12>{}); group(std::integral_constant<int, 13>{});
            group(std::integral_constant<int, 14>{}); group(std::integral_constant<int, 15>{});
            ++j;
            st = stn;
        }
    };

    const bool consumer = wave < NCW;
    if (consumer) consumer_main(); else loader_main();
    mfma_drain();
    if (wide) ring_epilogue<RBN, MT>(p, acc, consumer, m0, n0, tile_m, smem);
    else if (consumer) igemm_epilogue<RBN, false, MT, 0, NCW, false>(p, acc, m0, n0, tile_m, smem);
}
#endif   // EP24_AB_VARIANTS

template <int PPS, bool M32, int TBN = RBN, int INFER = 0>        // INFER: the epilogue's EMODE (0 training, 1 inference, 2 training without statistics)
int launch_ring_pps(const IgemmArgs& a, int NP, int halo, int npb, size_t lds, hipStream_t stream) {
    const unsigned tiles = (unsigned)ep24_cdiv(a.M, RBM) * (unsigned)ep24_cdiv(a.N, TBN);
    static std::atomic<unsigned long long> done{0};          // per-device attribute, set once (conv_patch.hip has the reasons)
    int dev = 0;
    EP24_REQUIRE(hipGetDevice(&dev) == hipSuccess, EP24_E_LAUNCH, "conv_ring: hipGetDevice failed");
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv_ring_kernel<PPS, M32, TBN, INFER>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        EP24_REQUIRE(e == hipSuccess, EP24_E_LAUNCH, "conv_ring: hipFuncSetAttribute(MaxDynamicSharedMemorySize, 160 KB) failed on device %d: %s", dev,
                     hipGetErrorString(e));
        done.fetch_or(bit, std::memory_order_release);
    }
    IgemmArgs b = a;
    const long dst_b = ((a.M - 1) * a.ld_dst + a.N) * 2;     // a destination the 32-bit offsets of the epilogue's buffer stores cover (else 0: its pointer form)
    EP24_REQUIRE(dst_b < 0x7FFF0000L, EP24_E_UNSUPPORTED, "conv_ring: a destination of %ld bytes is beyond the 32-bit offsets of its stores (launch_ring sends such a layer to the tiled kernel)", dst_b);
    b.dst_bytes = (unsigned)dst_b;
    hipLaunchKernelGGL((conv_ring_kernel<PPS, M32, TBN, INFER>), dim3(tiles), dim3((NCW + NLW) * 64), lds, stream, b, NP, halo, npb);
    return EP24_OK;
}

}  // namespace

namespace ep24_igemm { int wgrad_ring_timeouts(); int bn_barrier_timeouts(); }   // conv_wgrad.hip: the weight-gradient ring's counter; elementwise.hip: the
                                                                                  // fused BatchNorm backward's grid-wide wait

extern "C" int ep24_conv_ring_timeouts(void) {
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_ring_timeouts), sizeof(v)) != hipSuccess) return -1;
    const int w = ep24_igemm::wgrad_ring_timeouts(), bb = ep24_igemm::bn_barrier_timeouts();
    return (w < 0 || bb < 0) ? -1 : (int)v + w + bb;
}

#ifdef EP24_STAMPS
extern "C" int ep24_debug_read_ring_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ring_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
extern "C" int ep24_debug_read_ring_estamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_ring_estamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

namespace ep24_igemm {

// Same shapes as the halo-patch kernel takes (3x3 stride-1, N > 64, >= 200 tiles of 256 x 128, row-major nine-tap table), plus
// room for the counters behind the weight ring.  The patch is stored in whole 8-row pieces (NP of them), not rounded up to a
// multiple of the loader count: at W = 80 that is what leaves the 64 bytes (2 x 53 KB + 48 KB + counters < 160 KB).
// `narrow_ok` (an A/B option, off by default): the NARROW tile (256 x 64) for every layer with at least 128 of those - also the ones
// 256 x 128 tiles do not spread over the chip (the 20 x 20 level) or whose N is 64, which otherwise stay with the tiled kernel.
// Measured slower than what it replaces on every layer.  Returns 0: not this kernel, 1: the 256 x 128 ring, 2: the narrow one.
int launch_ring(const IgemmArgs& a, hipStream_t stream, bool dry, int* rc, bool m16, bool narrow_ok) {
    const int sgn = a.oy[0] < 0 ? 1 : -1;
    for (int t = 0; t < 9; ++t)
        if (a.oy[t] != sgn * (t / 3 - 1) || a.ox[t] != sgn * (t % 3 - 1) || a.wslot[t] != t) return 0;
    if (a.K % 8 != 0) return 0;
    if (a.bnr_z) return 0;                                   // the fused BatchNorm-backward sums (an A/B option) stay with the 8-wave kernel
    if (((a.M - 1) * a.ld_dst + a.N) * 2 >= 0x7FFF0000L) return 0;      // its epilogue stores with 32-bit offsets
#ifdef EP24_AB_VARIANTS
    const bool narrow = narrow_ok;
#else
    const bool narrow = false;                               // the narrow tile and the 32 x 32 x 16 consumers left the product library (round 5)
    (void)narrow_ok;
    m16 = true;
#endif
    if (a.epi_infer && (narrow || !m16)) return 0;            // the eval-mode epilogue lives in the 256 x 128 tile's 16 x 16 form
    if (!narrow && (a.N <= 64 || (long)ep24_cdiv(a.M, RBM) * ep24_cdiv(a.N, RBN) < 200)) return 0;
    if (narrow && (long)ep24_cdiv(a.M, RBM) * ep24_cdiv(a.N, 64) < 128) return 0;
    const int tb_bytes = (narrow ? 64 : RBN) * 128;
    const int halo = a.SW + 1;
    const int npb = a.K > BK ? 2 : 1;
    const int np = (RBM + 2 * halo + 7) / 8;
    const int ns = narrow ? 9 : NS;                            // the narrow ring: a stage per tap
    const size_t lds = ZB + (size_t)npb * np * 1024 + ns * (size_t)tb_bytes + 64;
    const int npw = (np + NLW - 1) / NLW;
    // patch pieces of the next chunk ride on taps 2..8 (the narrow ring: 5..8), at most three per loader and step
    // (a single-chunk layer has no next chunk: its whole patch is requested up front, <= 21 pieces per loader beside 12 weight DMAs)
    if (lds > 160 * 1024 || npw > ((narrow && npb == 2) ? 4 : 7) * 3) return 0;
    // the epilogue stages the tile (4 x 16 KB, the narrow one 4 x 8 KB) through the patch area
    if ((size_t)npb * np * 1024 + ns * (size_t)tb_bytes < (size_t)NCW * (narrow ? 4 : MT) * 16 * 128) return 0;
    if (!dry) {
        const int pps = narrow ? (npb == 2 ? (npw + 3) / 4 : 1) : (npw + 6) / 7;
        // the 32 x 32 x 16 consumers need the 16-byte store path (bf16 rows aligned to 16 bytes) and room for the statistics fold
        const bool m32 = !narrow && !m16 && !a.narrow_epi && (a.N & 7) == 0 && (a.ld_dst & 7) == 0 && (reinterpret_cast<unsigned long long>(a.dst) & 15) == 0 &&
                         lds >= (size_t)NCW * 2 * 64 * 33 * 4;
#ifdef EP24_AB_VARIANTS
        if (narrow)
            *rc = pps <= 1 ? launch_ring_pps<1, false, 64>(a, np, halo, npb, lds, stream)
                : pps == 2 ? launch_ring_pps<2, false, 64>(a, np, halo, npb, lds, stream)
                           : launch_ring_pps<3, false, 64>(a, np, halo, npb, lds, stream);
        else if (m32)
            *rc = pps <= 1 ? launch_ring_pps<1, true>(a, np, halo, npb, lds, stream)
                : pps == 2 ? launch_ring_pps<2, true>(a, np, halo, npb, lds, stream)
                           : launch_ring_pps<3, true>(a, np, halo, npb, lds, stream);
        else
#else
        (void)m32;
#endif
        if (a.epi_infer)
            *rc = pps <= 1 ? launch_ring_pps<1, false, RBN, true>(a, np, halo, npb, lds, stream)
                : pps == 2 ? launch_ring_pps<2, false, RBN, true>(a, np, halo, npb, lds, stream)
                           : launch_ring_pps<3, false, RBN, true>(a, np, halo, npb, lds, stream);
        else
            if (a.stats)
                *rc = pps <= 1 ? launch_ring_pps<1, false>(a, np, halo, npb, lds, stream)
                    : pps == 2 ? launch_ring_pps<2, false>(a, np, halo, npb, lds, stream)
                               : launch_ring_pps<3, false>(a, np, halo, npb, lds, stream);
            else                                                 // an input gradient: the form without the BatchNorm sums
                *rc = pps <= 1 ? launch_ring_pps<1, false, RBN, 2>(a, np, halo, npb, lds, stream)
                    : pps == 2 ? launch_ring_pps<2, false, RBN, 2>(a, np, halo, npb, lds, stream)
                               : launch_ring_pps<3, false, RBN, 2>(a, np, halo, npb, lds, stream);
    }
    return narrow ? 2 : 1;
}

// Generic form: any gather-GEMM with bf16 output, no bias, N > 64 and enough 256 x 128 tiles to fill the chip.
bool launch_ring_generic(const IgemmArgs& a, hipStream_t stream, bool dry, int* rc) {
#ifndef EP24_AB_VARIANTS
    (void)a; (void)stream; (void)dry; (void)rc;
    return false;
#else
    if (a.K % 8 != 0 || a.N <= 64 || a.T > 16 || a.bnr_z || a.bias || a.epi_infer) return false;
    if ((long)ep24_cdiv(a.M, RBM) * ep24_cdiv(a.N, RBN) < 200) return false;
    const size_t lds = ZB + (size_t)NS * GSTAGE + 64;
    if (!dry) {
        static std::atomic<unsigned long long> done{0};
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) { ep24_set_error("conv_ring: hipGetDevice failed"); *rc = EP24_E_LAUNCH; return true; }
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            const hipError_t e = hipFuncSetAttribute((const void*)conv_ring_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) { ep24_set_error("conv_ring: hipFuncSetAttribute failed on device %d: %s", dev, hipGetErrorString(e)); *rc = EP24_E_LAUNCH; return true; }
            done.fetch_or(bit, std::memory_order_release);
        }
        const unsigned tiles = (unsigned)ep24_cdiv(a.M, RBM) * (unsigned)ep24_cdiv(a.N, RBN);
        hipLaunchKernelGGL(conv_ring_generic_kernel, dim3(tiles), dim3((NCW + NLW) * 64), lds, stream, a);
        *rc = EP24_OK;
    }
    return true;
#endif
}

}  // namespace ep24_igemm
