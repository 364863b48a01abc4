// ep24 - halo-patch kernel for the 3x3 stride-1 convolutions (forward and input gradient): 63 % of the network's FLOPs.
//
// Why a second kernel.  The generic tiled kernel (conv_igemm.hip) fetches a fresh [128 pixels][64 channels] A tile for
// every one of the nine taps: 32 KB of LDS fill per 2.1 MFLOP, and the fill path of a CU (L2 -> LDS-DMA -> LDS write,
// ~70 GB/s per CU measured for gathers out of L2) is what bounds it at ~35 % of the MFMA peak - removing all global
// traffic from that kernel bought only 19 %, the fill instructions and their LDS writes stay.  For a same-size 3x3 conv
// the nine A tiles are shifted views of ONE run of input pixels, so here a workgroup loads that run once per 64-channel
// chunk - the PATCH: rows [m0 - (W+1), m0 + BM + (W+1)) of the flattened [B*H*W, C] activation - and the nine taps read
// their MFMA A fragments from it at row offsets dy*W + dx.  Rows that a tap takes from across an image border (left /
// right edge, top / bottom, the next image of the batch) are zeroed in registers with a per-lane 9-bit tap mask.
// LDS fill per (9 taps x 64 channels) drops from 9 x (A + B) tiles to 1 patch + 9 B tiles: for a 256 x 128 tile at
// W = 40, 288 KB -> 187 KB for twice the FLOPs (3.1x less per FLOP), and the number of DMA instructions per wave and
// barrier per MFMA halves with the 256-row tile (8 waves).
//
// Pipeline: the patch of chunk kc+1 streams in (a few DMA pieces per tap step) while the nine taps of chunk kc are
// multiplied; the weight tile of the next tap is one step ahead in a two-stage ring.  One barrier per tap step.
// A fragments come from the resident patch, so the reads of step j+1 do not wait for any DMA.
// Tile mapping, LDS row swizzle (on the DMA source side), channel relabelling and the epilogue are the tiled kernel's.
#include <atomic>
#include <type_traits>
#include "igemm.h"

using namespace ep24_igemm;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

// Diagnostic build only (make stamps -> libep24_stamps.so, tools/conv_stamps.py): wave 0 of the first workgroups records
// s_memtime at a few points of the kernel.  No stamp exists in the product library.
#ifdef EP24_STAMPS
__device__ unsigned long long g_stamps[64 * 8];
#define STAMP() __builtin_amdgcn_s_memtime()
#endif

template <int N>
__device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int BMP, int BN, int NB, int PPS>
__global__ __launch_bounds__((BMP / 64) * (BN / 64) * 64) void conv_patch_kernel(const IgemmArgs p, const int PR, const int halo,
                                                                                const int npb) {
    constexpr int WN = BN / 64, WM = BMP / 64, NW = WM * WN;
    constexpr int MT = 4, NT = 4;
    constexpr int B_BYTES = BN * 128;
    constexpr int B_INSTR = BN / 8 / NW;                     // weight-tile DMA instructions per wave and step
    static_assert(B_INSTR >= 1, "tile too narrow for its wave count");
    static_assert(NB == 3, "nine taps per chunk: the stage of a step is its tap column, a compile-time constant");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef EP24_STAMPS
    const unsigned long long st_t0 = STAMP(), st_r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long st_wait = 0, st_work = 0;
#endif
    const int wm = wave / WN, wn = wave % WN;
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    const int tile_id = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
    const int tiles_n = (p.N + BN - 1) / BN;
    const int tile_m = tile_id / tiles_n;
    const long m0 = (long)tile_m * BMP;
    const int n0 = (tile_id - tile_m * tiles_n) * BN;
    const int KC = (p.K + BK - 1) / BK;
    const int PBYTES = PR * 128;
    const int bb = npb * PBYTES;                             // LDS offset of the weight ring (smem is the only LDS object: offset 0)

    const int lchunk = (lane & 7) ^ ((lane >> 3) & 7);
    const auto src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.src), 0, p.src_bytes, 0x00020000);
    const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.wt), 0, p.wt_bytes, 0x00020000);
    const int kmax = (p.K - lchunk * 8 + BK - 1) / BK;       // chunks kc < kmax hold real channels for this lane
    const int ktail = (p.K & (BK - 1)) ? KC - 1 : KC;        // chunks >= ktail need the per-lane check (K not a multiple of 64)

    // ---- patch pieces: piece g = i*NW + wave covers patch rows 8g .. 8g+7 (1 KiB, one DMA instruction); the lane's row
    // in piece g is pixel m0 - halo + 8g + (lane >> 3) of the flattened source.  All offsets are 32-bit (the launcher
    // refuses tensors of 2 GiB and more); a row before the tensor wraps to a huge unsigned value and fails the range test.
    const int NPW = PR / (8 * NW);                           // pieces per wave and chunk; NPW <= 7 * PPS: all issued by tap step 6
    const unsigned msrc = (unsigned)((long)p.B * p.SH * p.SW);
    const int prow0 = (int)(m0 - halo) + (lane >> 3);
    const int ld2 = (int)p.ld_src * 2;
    const unsigned pv0 = (unsigned)prow0 * (unsigned)ld2 + lchunk * 16;   // byte offset of the lane's row in piece 0, chunk 0 (wraps like prow0)
    auto issue_patch = [&](int pbuf_off, int kc, int i) {
        if (i >= NPW) i = NPW - 1;                             // filler: the same piece again (identical bytes), so that every
                                                               // step issues a fixed number of DMAs and the waits are immediates
        const int g = i * NW + wave;                           // scalar
        const unsigned ps = (unsigned)(prow0 + 8 * g);
        bool ok = ps < msrc;
        if (kc >= ktail) ok = ok && kc < kmax;
        const int vo = ok ? (int)(pv0 + (unsigned)(8 * g * ld2 + kc * (BK * 2))) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lptr_t)(smem + pbuf_off + g * 1024), 16, vo, 0, 0, 0);
    };
    // ---- weight tile rows (relabelled inside each 64 span, as in the tiled kernel); rows beyond N keep an offset that
    // stays out of range whatever is added to it
    int wvoff[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int q = (wave * B_INSTR + i) * 8 + (lane >> 3);
        const int r = (q & ~63) + ((q & 15) << 2) + ((q >> 4) & 3);
        wvoff[i] = n0 + r < p.N ? (int)((((long)(n0 + r) * p.WT * p.K) + lchunk * 8) * 2) : OOB;
    }
    // tile (tap t, chunk kc) into ring stage `stage`; nothing is issued beyond the last chunk
    auto issue_b = [&](int stage, int t, int kc) {
        if (kc >= KC) return;
        const int b_s = (t * p.K + kc * BK) * 2;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            int vo = wvoff[i] == OOB ? OOB : wvoff[i] + b_s;
            if (kc >= ktail) vo = kc < kmax ? vo : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wt_rsrc, (lptr_t)(smem + bb + stage * B_BYTES + (wave * B_INSTR + i) * 1024), 16, vo, 0, 0, 0);
        }
    };

    const int frow = lane & 15, fq = lane >> 4;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: whole patch of chunk 0, then the weight tiles of steps 0 and 1
    for (int i = 0; i < NPW; ++i) issue_patch(0, 0, i);
    issue_b(0, 0, 0);
    issue_b(1, 1, 0);

    // taps in row-major order; forward reads pixel (y + ty - 1, x + tx - 1), the input gradient (y + 1 - ty, x + 1 - tx):
    // p.oy[0] tells which (the host builds both tables with weight slot t = tap t)
    const int sgn = p.oy[0] < 0 ? 1 : -1;

    // ---- per-lane tap masks of the four fragment rows (computed while the first DMAs are in flight), and per fragment
    // the taps for which ANY lane of the wave is masked: the others skip the masking instructions altogether (a 16-pixel
    // run touches an image border for one tap in five at W = 80)
    unsigned vm[MT];
    unsigned need[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const long m = m0 + wm * 64 + i * 16 + frow;
        unsigned mk = 0;
        if (m < p.M) {
            const int mm = (int)m;
            const int n = fdiv(mm, p.d_plane);
            const int rem = mm - n * (p.GH * p.GW);
            const int y = fdiv(rem, p.d_gw), x = rem - y * p.GW;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = y + sgn * (t / 3 - 1), ix = x + sgn * (t % 3 - 1);
                if (iy >= 0 && iy < p.SH && ix >= 0 && ix < p.SW) mk |= 1u << t;
            }
        }
        vm[i] = mk;
        unsigned nd = 0;
#pragma unroll
        for (int t = 0; t < 9; ++t)
            if (__ballot(!((mk >> t) & 1u)) != 0ull) nd |= 1u << t;
        need[i] = __builtin_amdgcn_readfirstlane(nd);
    }

    // ---- LDS read addresses.  A: fragment row i of the wave sits at patch row arow0 + 16 i + shift(tap); 16 i and the
    // k half leave the swizzle key (row & 7) alone, so one address per tap serves the four rows through the instruction's
    // offset field and the second k half is that address with bit 6 flipped.  B: constant per lane; stage, fragment and
    // k half are offset-field constants (the stage of a step is its tap column).
    const int arow0 = wm * 64 + frow + halo;
    const int bo0 = bb + (wn * 64 + frow) * 128 + ((fq ^ (frow & 7)) << 4);
    auto a_off = [&](int pbuf_off, int sh) {                  // ks = 0 address of a tap whose row shift is sh
        const int q0 = arow0 + sh;
        return pbuf_off + q0 * 128 + ((fq ^ (q0 & 7)) << 4);
    };
    auto read_a = [&](bf16x8 (&fa)[MT], int off) {
#pragma unroll
        for (int i = 0; i < MT; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(smem + off + i * 2048);
    };
    auto read_b = [&](bf16x8 (&fb)[NT], int off, int stage) {
#pragma unroll
        for (int q = 0; q < NT; ++q) fb[q] = *reinterpret_cast<const bf16x8*>(smem + off + stage * B_BYTES + q * 2048);
    };
    // rows a tap takes from across an image border are zeroed, right before the MFMAs that use the fragments
    auto mask_half = [&](bf16x8 (&fa)[MT], int t) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if ((need[i] >> t) & 1u) {
                asm volatile("" ::: "memory");                 // keeps this a scalar branch (no select over the whole wave)
                v4i v = __builtin_bit_cast(v4i, fa[i]);
                v &= -(int)((vm[i] >> t) & 1u);
                fa[i] = __builtin_bit_cast(bf16x8, v);
            }
        }
    };
    auto mma = [&](const bf16x8 (&fa)[MT], const bf16x8 (&fb)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NT; ++q)
                acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[q], acc[i][q], 0, 0, 0);
    };

    // Software pipeline, half a step deep: when step j starts, the fragments of its first k half are already in registers
    // (read during step j-1, behind its MFMAs), so the MFMA pipe starts right after the barrier instead of waiting for
    // an LDS round trip; the second half is read under the first half's MFMAs, the next step's first half under the
    // second's.  For that the weight tile of step j+1 must be visible during step j: the wait at the start of step j
    // covers it (it was issued FIRST in step j-1; only that step's patch pieces may still be in flight).
    // Issue order per step: [weight tile of step j+2][PPS patch pieces in tap steps 0..6 of a chunk that has a successor].
    // Weight tiles are retired with a COUNTED vmcnt and a raw s_barrier: a __syncthreads() would drain every DMA in flight.
    // The tap loop is 3 rows x 3 unrolled columns: tap column, ring stage and every LDS offset of a step are constants,
    // which took the loop from 253 to ~130 instructions per 32 MFMAs (it was issue-bound: 104 vector + 98 scalar
    // instructions per step next to the MFMAs of two waves per SIMD).
    wait_vmcnt_c<0>();
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    int offa = a_off(0, sgn * (-p.SW - 1));                   // tap 0 of chunk 0
    read_a(fa0, offa);
    read_b(fb0, bo0, 0);
#ifdef EP24_STAMPS
    const unsigned long long st_t1 = STAMP();
#endif
    for (int kc = 0; kc < KC; ++kc) {
        const int pcur = (npb == 2) ? (kc & 1) * PBYTES : 0;
        const int pnext = ((kc + 1) & 1) * PBYTES;
        const bool more = kc + 1 < KC;                       // another chunk follows: its patch pieces are issued in these steps
#pragma unroll 1
        for (int ty = 0; ty < 3; ++ty) {
            const int shrow = sgn * (ty - 1) * p.SW;
            auto tap_step = [&](auto txc) {
                constexpr int tx = decltype(txc)::value;
                const int t = ty * 3 + tx;
#ifdef EP24_STAMPS
                const unsigned long long st_a = STAMP();
#endif
                // taps 1..7 of a chunk with a successor: the pieces issued in the previous step may stay in flight
                const bool counted = more && (tx == 1 || (tx == 0 ? ty != 0 : ty != 2));
                if (counted) wait_vmcnt_c<PPS>(); else wait_vmcnt_c<0>();
#ifdef EP24_STAMPS
                const unsigned long long st_m = STAMP();
                st_work += st_m - st_a;                        // diagnostic build: "work" column = the counted DMA wait alone
#endif
                __builtin_amdgcn_s_barrier();  // weight tile of step j+1 (and at tap 8 the next chunk's patch) landed for every
                                               // wave; nobody still reads the stage / patch buffer refilled below
#ifdef EP24_STAMPS
                const unsigned long long st_b = STAMP();
                st_wait += st_b - st_m;                        // "wait" column = the barrier alone
#endif
                // weight tile of step j+2: tap t+2 of this chunk, or tap t-7 of the next
                if (tx == 0 || ty < 2) issue_b((tx + 2) % 3, t + 2, kc);
                else issue_b((tx + 2) % 3, t - 7, kc + 1);
                if (more && (ty < 2 || tx == 0)) {             // tap steps 0..6
#pragma unroll
                    for (int i = 0; i < PPS; ++i) issue_patch(pnext, kc + 1, t * PPS + i);
                }
                read_a(fa1, offa ^ 64);
                read_b(fb1, bo0 ^ 64, tx);
                mask_half(fa0, t);
                mma(fa0, fb0);
                if (tx < 2 || ty < 2 || more) {                // not the last step: first k half of the next tap
                    offa = (tx < 2) ? a_off(pcur, shrow + sgn * tx)                              // same row, next column
                                    : (ty < 2 ? a_off(pcur, shrow + sgn * (p.SW - 1))            // first column of the next row
                                              : a_off(pnext, sgn * (-p.SW - 1)));                // tap 0 of the next chunk
                    read_a(fa0, offa);
                    read_b(fb0, bo0, (tx + 1) % 3);
                }
                mask_half(fa1, t);
                mma(fa1, fb1);
            };
            tap_step(std::integral_constant<int, 0>{});
            tap_step(std::integral_constant<int, 1>{});
            tap_step(std::integral_constant<int, 2>{});
        }
    }
#ifdef EP24_STAMPS
    const unsigned long long st_t2 = STAMP();
    const int n_steps = 9 * KC;
#endif
    igemm_epilogue<BN, false, MT, 0, NW>(p, acc, m0, n0, tile_m, smem);
#ifdef EP24_STAMPS
    if (blockIdx.x < 32 && lane == 0 && (wave == 0 || wave == NW - 1)) {     // the oldest and the youngest wave of a workgroup
        unsigned long long* o = g_stamps + (blockIdx.x * 2 + (wave != 0)) * 8;
        o[0] = st_t1 - st_t0; o[1] = st_t2 - st_t1; o[2] = STAMP() - st_t2; o[3] = st_wait; o[4] = st_work;
        o[5] = __builtin_amdgcn_s_memrealtime() - st_r0; o[6] = STAMP() - st_t0; o[7] = n_steps;
    }
#endif
}

template <int BMP, int BN, int NB, int PPS>
int launch_pps(const IgemmArgs& a, int PR, int halo, int npb, size_t lds, hipStream_t stream) {
    constexpr int NT_ = (BMP / 64) * (BN / 64) * 64;
    const unsigned tiles = (unsigned)ep24_cdiv(a.M, BMP) * (unsigned)ep24_cdiv(a.N, BN);
    // more than 64 KB of dynamic LDS needs the attribute on the function, once per DEVICE (a process that launches on a second
    // device would otherwise fail there with a generic launch error); the set is cheap and idempotent, a mutex-free bitmask of
    // the devices already done keeps it off the launch path
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    EP24_REQUIRE(hipGetDevice(&dev) == hipSuccess, EP24_E_LAUNCH, "conv_patch: hipGetDevice failed");
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void*)conv_patch_kernel<BMP, BN, NB, PPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        EP24_REQUIRE(e == hipSuccess, EP24_E_LAUNCH, "conv_patch: hipFuncSetAttribute(MaxDynamicSharedMemorySize, 160 KB) failed on device %d: %s", dev,
                     hipGetErrorString(e));
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((conv_patch_kernel<BMP, BN, NB, PPS>), dim3(tiles), dim3(NT_), lds, stream, a, PR, halo, npb);
    return EP24_OK;
}

template <int BMP, int BN, int NB>
int launch_cfg(const IgemmArgs& a, int PR, int halo, int npb, size_t lds, hipStream_t stream) {
    const int npw = PR / (8 * (BMP / 64) * (BN / 64));
    if (npw <= 7) return launch_pps<BMP, BN, NB, 1>(a, PR, halo, npb, lds, stream);
    if (npw <= 14) return launch_pps<BMP, BN, NB, 2>(a, PR, halo, npb, lds, stream);
    return launch_pps<BMP, BN, NB, 3>(a, PR, halo, npb, lds, stream);
}

}  // namespace

#ifdef EP24_STAMPS
extern "C" int ep24_debug_read_stamps(unsigned long long* host, int n) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * (size_t)n) == hipSuccess ? 0 : -1;
}
#endif

namespace ep24_igemm {

// One tile shape: 256 x 128 (8 waves).  Measured against the generic tiled kernel in one process (tools/conv_ab.py, MI355X, B = 20):
// -6 % time on 40x40x256->256, -4 % on 80x80x128->128, -2.5 % on 80x80x256->256, -4 % on 256->512; with 128-row tiles
// (the 20x20 level, one wave per SIMD) and with 64-wide N (160x160x64) it LOSES 13 - 30 %, so those shapes stay with the
// tiled kernel, as does anything whose patch does not fit the LDS next to the three-stage weight ring.
bool launch_patch(const IgemmArgs& a, hipStream_t stream, bool dry, int* rc) {
    // The kernel does not read the tap tables: it derives every shift from sgn = (oy[0] < 0 ? +1 : -1) and assumes the row-major
    // layout oy[t] = sgn * (t / 3 - 1), ox[t] = sgn * (t % 3 - 1), weight slot t (forward: sgn +1, input gradient: -1).  Any other
    // nine-tap table goes to the tiled kernel, which does read them.
    const int sgn = a.oy[0] < 0 ? 1 : -1;
    for (int t = 0; t < 9; ++t)
        if (a.oy[t] != sgn * (t / 3 - 1) || a.ox[t] != sgn * (t % 3 - 1) || a.wslot[t] != t) return false;
    if (a.K % 8 != 0 || a.N <= 64) return false;
    if ((long)ep24_cdiv(a.M, 256) * ep24_cdiv(a.N, 128) < 200) return false;      // would leave CUs idle
    const int halo = a.SW + 1;
    const int npb = a.K > BK ? 2 : 1;
    const int nw = 8;
    const int pr = (256 + 2 * halo + 8 * nw - 1) / (8 * nw) * (8 * nw);
    const size_t lds = (size_t)npb * pr * 128 + 3 * (size_t)128 * 128;
    if (lds > 160 * 1024 || pr / (8 * nw) > 21) return false;
    if (!dry) *rc = launch_cfg<256, 128, 3>(a, pr, halo, npb, lds, stream);
    return true;
}

}  // namespace ep24_igemm
