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
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    char* const bbase = smem + npb * PBYTES;

    const int lchunk = (lane & 7) ^ ((lane >> 3) & 7);
    const auto src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.src), 0, p.src_bytes, 0x00020000);
    const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.wt), 0, p.wt_bytes, 0x00020000);
    const int kmax = (p.K - lchunk * 8 + BK - 1) / BK;       // chunks kc < kmax hold real channels for this lane

    // ---- patch pieces: piece g = i*NW + wave covers patch rows 8g .. 8g+7 (1 KiB, one DMA instruction); the lane's row
    // in piece g is pixel m0 - halo + 8g + (lane >> 3) of the flattened source
    const int NPW = PR / (8 * NW);                           // pieces per wave and chunk; NPW <= 7 * PPS: all issued by tap step 6
    static_assert(NB == 3, "the half-step pipeline needs the three-stage weight ring");
    const long msrc = (long)p.B * p.SH * p.SW;
    const long prow0 = m0 - halo + (lane >> 3);
    const long ld2 = p.ld_src * 2;
    auto issue_patch = [&](char* pbuf, int kc, int i) {
        if (i >= NPW) i = NPW - 1;                             // filler: the same piece again (identical bytes), so that every
                                                               // step issues a fixed number of DMAs and the waits are immediates
        const int g = i * NW + wave;
        const long ps = prow0 + 8 * g;
        const int vo = (ps >= 0 && ps < msrc && kc < kmax) ? (int)(ps * ld2) + lchunk * 16 + kc * (BK * 2) : OOB;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lptr_t)(pbuf + g * 1024), 16, vo, 0, 0, 0);
    };
    // ---- weight tile rows (relabelled inside each 64 span, as in the tiled kernel)
    int wvoff[B_INSTR];
#pragma unroll
    for (int i = 0; i < B_INSTR; ++i) {
        const int q = (wave * B_INSTR + i) * 8 + (lane >> 3);
        const int r = (q & ~63) + ((q & 15) << 2) + ((q >> 4) & 3);
        wvoff[i] = n0 + r < p.N ? (int)((((long)(n0 + r) * p.WT * p.K) + lchunk * 8) * 2) : OOB;
    }
    auto issue_b = [&](int step_, int wslot, int kc) {
        const int b_s = (wslot * p.K + kc * BK) * 2;
        char* st = bbase + (step_ % NB) * B_BYTES;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int vo = (kc < kmax && wvoff[i] != OOB) ? wvoff[i] + b_s : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wt_rsrc, (lptr_t)(st + (wave * B_INSTR + i) * 1024), 16, vo, 0, 0, 0);
        }
    };

    const int frow = lane & 15, fq = lane >> 4;
    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // ---- prologue: whole patch of chunk 0 (the weight tiles of the first steps follow below)
    for (int i = 0; i < NPW; ++i) issue_patch(smem, 0, i);

    // taps in row-major order; forward reads pixel (y + ty - 1, x + tx - 1), the input gradient (y + 1 - ty, x + 1 - tx):
    // p.oy[0] tells which (the host builds both tables with weight slot t = tap t)
    const int sgn = p.oy[0] < 0 ? 1 : -1;
    // Weight tiles run D = NB - 1 steps ahead in a ring of NB stages and are retired with a COUNTED vmcnt: per step a
    // wave issues [weight tile of step j + D][patch pieces of step j], so when step j starts everything except the
    // D - 1 younger weight tiles and the previous step's patch pieces must have landed.  Raw s_barrier: a
    // __syncthreads() would drain every DMA in flight.
    int fetch_t = 0, fetch_kc = 0;                            // (tap, chunk) of the next weight tile to issue
    auto issue_next_b = [&]() {
        issue_b(fetch_t + 9 * fetch_kc, fetch_t, fetch_kc);    // beyond the last chunk: zero fill into a stage nobody reads again
        if (++fetch_t == 9) { fetch_t = 0; ++fetch_kc; }
    };
    issue_next_b();                                            // steps 0 and 1 (the patch of chunk 0 was issued before them)
    issue_next_b();

    // ---- per-lane fragment rows and their tap masks (computed while the first DMAs are in flight)
    int qb[MT];
    unsigned vm[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int r = wm * 64 + i * 16 + frow;
        qb[i] = r + halo;
        const long m = m0 + r;
        unsigned mk = 0;
        if (m < p.M) {
            const int mm = (int)m;
            const int n = fdiv(mm, p.d_plane);
            const int rem = mm - n * (p.GH * p.GW);
            const int y = fdiv(rem, p.d_gw), x = rem - y * p.GW;
            const int sg = p.oy[0] < 0 ? 1 : -1;
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int iy = y + sg * (t / 3 - 1), ix = x + sg * (t % 3 - 1);
                if (iy >= 0 && iy < p.SH && ix >= 0 && ix < p.SW) mk |= 1u << t;
            }
        }
        vm[i] = mk;
    }


    // fragments of one 32-deep k half: A rows from the patch at this tap's row shift (masked), B rows from a weight stage
    auto read_half = [&](bf16x8 (&fa)[MT], bf16x8 (&fb)[NT], const char* pb, const char* lb, int t, int sh, int ks) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int q = qb[i] + sh;
            fa[i] = *reinterpret_cast<const bf16x8*>(pb + q * 128 + (((ks * 4 + fq) ^ (q & 7)) << 4));
        }
#pragma unroll
        for (int q = 0; q < NT; ++q)
            fb[q] = *reinterpret_cast<const bf16x8*>(lb + swz(wn * 64 + q * 16 + frow, ks * 4 + fq));
    };
    // rows a tap takes from across an image border are zeroed; applied right before the MFMAs that use the fragments (the
    // reads were issued half a step earlier, so this does not wait for the LDS)
    auto mask_half = [&](bf16x8 (&fa)[MT], int t) {
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            v4i v = __builtin_bit_cast(v4i, fa[i]);
            v &= -(int)((vm[i] >> t) & 1u);
            fa[i] = __builtin_bit_cast(bf16x8, v);
        }
    };
    auto mma = [&](const bf16x8 (&fa)[MT], const bf16x8 (&fb)[NT]) {
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int q = 0; q < NT; ++q)
                acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[q], acc[i][q], 0, 0, 0);
    };
    auto shift_of = [&](int t) { return sgn * ((t / 3 - 1) * p.SW + (t % 3 - 1)); };

    // Software pipeline, half a step deep: when step j starts, the fragments of its first k half are already in registers
    // (read during step j-1, behind its MFMAs), so the MFMA pipe starts right after the barrier instead of waiting for
    // an LDS round trip; the second half is read under the first half's MFMAs, the next step's first half under the
    // second's.  For that the weight tile of step j+1 must be visible during step j: the wait at the start of step j
    // covers it (it was issued FIRST in step j-1; only that step's patch pieces may still be in flight).
    // Issue order per step: [weight tile of step j+2][PPS patch pieces in tap steps 0..6 of a chunk that has a successor].
    wait_vmcnt_c<0>();
    __builtin_amdgcn_s_barrier();
    bf16x8 fa0[MT], fb0[NT], fa1[MT], fb1[NT];
    read_half(fa0, fb0, smem, bbase, 0, shift_of(0), 0);
    int step = 0;
    const int n_steps = 9 * KC;
#ifdef EP24_STAMPS
    const unsigned long long st_t1 = STAMP();
#endif
    for (int kc = 0; kc < KC; ++kc) {
        const char* pb = smem + ((npb == 2) ? (kc & 1) : 0) * PBYTES;
        char* const pnext = smem + ((kc + 1) & 1) * PBYTES;
        const bool more = kc + 1 < KC;                       // another chunk follows: its patch pieces are issued in these steps
#pragma unroll 1
        for (int t = 0; t < 9; ++t, ++step) {
#ifdef EP24_STAMPS
            const unsigned long long st_a = STAMP();
#endif
            if (more && t >= 1 && t <= 7) wait_vmcnt_c<PPS>(); else wait_vmcnt_c<0>();
            __builtin_amdgcn_s_barrier();      // weight tile of step j+1 (and at tap 8 the next chunk's patch) landed for every
                                               // wave; nobody still reads the stage / patch buffer refilled below
#ifdef EP24_STAMPS
            const unsigned long long st_b = STAMP();
            st_wait += st_b - st_a;
#endif
            issue_next_b();
            if (more && t <= 6) {
#pragma unroll
                for (int i = 0; i < PPS; ++i) issue_patch(pnext, kc + 1, t * PPS + i);
            }
            const char* lb = bbase + (step % NB) * B_BYTES;
            read_half(fa1, fb1, pb, lb, t, shift_of(t), 1);
            mask_half(fa0, t);
            mma(fa0, fb0);
            if (step + 1 < n_steps) {
                const int tn = t == 8 ? 0 : t + 1;
                read_half(fa0, fb0, t == 8 ? pnext : pb, bbase + ((step + 1) % NB) * B_BYTES, tn, shift_of(tn), 0);
            }
            mask_half(fa1, t);
            mma(fa1, fb1);
#ifdef EP24_STAMPS
            st_work += STAMP() - st_b;
#endif
        }
    }
#ifdef EP24_STAMPS
    const unsigned long long st_t2 = STAMP();
#endif
    igemm_epilogue<BN, false, MT, 0, NW>(p, acc, m0, n0, tile_m, smem);
#ifdef EP24_STAMPS
    if (blockIdx.x < 64 && tid == 0) {
        unsigned long long* o = g_stamps + blockIdx.x * 8;
        o[0] = st_t1 - st_t0; o[1] = st_t2 - st_t1; o[2] = STAMP() - st_t2; o[3] = st_wait; o[4] = st_work;
        o[5] = __builtin_amdgcn_s_memrealtime() - st_r0; o[6] = STAMP() - st_t0; o[7] = n_steps;
    }
#endif
}

template <int BMP, int BN, int NB, int PPS>
void launch_pps(const IgemmArgs& a, int PR, int halo, int npb, size_t lds, hipStream_t stream) {
    constexpr int NT_ = (BMP / 64) * (BN / 64) * 64;
    const unsigned tiles = (unsigned)ep24_cdiv(a.M, BMP) * (unsigned)ep24_cdiv(a.N, BN);
    static bool attr = false;
    if (!attr) { (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BMP, BN, NB, PPS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr = true; }
    hipLaunchKernelGGL((conv_patch_kernel<BMP, BN, NB, PPS>), dim3(tiles), dim3(NT_), lds, stream, a, PR, halo, npb);
}

template <int BMP, int BN, int NB>
void launch_cfg(const IgemmArgs& a, int PR, int halo, int npb, size_t lds, hipStream_t stream) {
    const int npw = PR / (8 * (BMP / 64) * (BN / 64));
    if (npw <= 7) launch_pps<BMP, BN, NB, 1>(a, PR, halo, npb, lds, stream);
    else if (npw <= 14) launch_pps<BMP, BN, NB, 2>(a, PR, halo, npb, lds, stream);
    else launch_pps<BMP, BN, NB, 3>(a, PR, halo, npb, lds, stream);
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
bool launch_patch(const IgemmArgs& a, hipStream_t stream) {
    for (int t = 0; t < 9; ++t)
        if (a.oy[t] < -1 || a.oy[t] > 1 || a.ox[t] < -1 || a.ox[t] > 1 || a.wslot[t] != t) return false;
    if (a.K % 8 != 0 || a.N <= 64) return false;
    if ((long)ep24_cdiv(a.M, 256) * ep24_cdiv(a.N, 128) < 200) return false;      // would leave CUs idle
    const int halo = a.SW + 1;
    const int npb = a.K > BK ? 2 : 1;
    const int nw = 8;
    const int pr = (256 + 2 * halo + 8 * nw - 1) / (8 * nw) * (8 * nw);
    const size_t lds = (size_t)npb * pr * 128 + 3 * (size_t)128 * 128;
    if (lds > 160 * 1024 || pr / (8 * nw) > 21) return false;
    launch_cfg<256, 128, 3>(a, pr, halo, npb, lds, stream);
    return true;
}

}  // namespace ep24_igemm
