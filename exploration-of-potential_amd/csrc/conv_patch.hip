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
#include "igemm.h"

using namespace ep24_igemm;

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));

template <int BMP, int BN, int MODE>
__global__ __launch_bounds__((BMP / 64) * (BN / 64) * 64) void conv_patch_kernel(const IgemmArgs p, const int PR, const int halo,
                                                                                const int npb) {
    constexpr int WN = BN / 64, WM = BMP / 64, NW = WM * WN;
    constexpr int MT = 4, NT = 4;
    constexpr int B_BYTES = BN * 128;
    constexpr int B_INSTR = BN / 8 / NW;                     // weight-tile DMA instructions per wave and step
    static_assert(B_INSTR >= 1, "tile too narrow for its wave count");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const int NPW = PR / (8 * NW);                           // pieces per wave and chunk
    const int PPS = (NPW + 7) >> 3;                          // pieces per tap step: all issued by step 7
    const long msrc = (long)p.B * p.SH * p.SW;
    const long prow0 = m0 - halo + (lane >> 3);
    const long ld2 = p.ld_src * 2;
    auto issue_patch = [&](char* pbuf, int kc, int i) {
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
    auto issue_b = [&](int stage, int wslot, int kc) {
        const int b_s = (wslot * p.K + kc * BK) * 2;
        char* st = bbase + stage * B_BYTES;
#pragma unroll
        for (int i = 0; i < B_INSTR; ++i) {
            const int vo = (kc < kmax && wvoff[i] != OOB) ? wvoff[i] + b_s : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wt_rsrc, (lptr_t)(st + (wave * B_INSTR + i) * 1024), 16, vo, 0, 0, 0);
        }
    };

    // ---- per-lane fragment rows and their tap masks
    const int frow = lane & 15, fq = lane >> 4;
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

    f32x4 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int q = 0; q < NT; ++q) acc[i][q] = (f32x4){0.f, 0.f, 0.f, 0.f};

    ZTile4 zt;
    constexpr bool zpre = MODE == 1;

    // ---- prologue: whole patch of chunk 0, weight tile of step 0
    for (int i = 0; i < NPW; ++i) issue_patch(smem, 0, i);
    issue_b(0, 0, 0);

    // taps in row-major order; forward reads pixel (y + ty - 1, x + tx - 1), the input gradient (y + 1 - ty, x + 1 - tx):
    // p.oy[0] tells which (the host builds both tables with weight slot t = tap t)
    const int sgn = p.oy[0] < 0 ? 1 : -1;
    int step = 0;
    for (int kc = 0; kc < KC; ++kc) {
        const char* pb = smem + ((npb == 2) ? (kc & 1) : 0) * PBYTES;
        char* const pnext = smem + ((kc + 1) & 1) * PBYTES;
        const bool more = kc + 1 < KC;
        int ty = 0, tx = 0;
#pragma unroll 1
        for (int t = 0; t < 9; ++t, ++step) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();                   // everything issued one step ago has landed for every wave; the stage the next
                                               // weight tile goes to and the other patch buffer are no longer being read
            if (t < 8) issue_b((step + 1) & 1, t + 1, kc);
            else if (more) issue_b((step + 1) & 1, 0, kc + 1);
            if (more) {
                const int hi = (t + 1) * PPS < NPW ? (t + 1) * PPS : NPW;
                for (int i = t * PPS; i < hi; ++i) issue_patch(pnext, kc + 1, i);
            }
            if (zpre && !more && t == 7) load_ztile<MT>(p, zt, m0, n0, wm, wn, lane);
            const char* lb = bbase + (step & 1) * B_BYTES;
            const int sh = sgn * ((ty - 1) * p.SW + (tx - 1));
            if (++tx == 3) { tx = 0; ++ty; }
            int keep[MT], arow[MT], asw[MT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                const int q = qb[i] + sh;
                keep[i] = -(int)((vm[i] >> t) & 1u);
                arow[i] = q * 128;
                asw[i] = q & 7;
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                bf16x8 fa[MT], fb[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    v4i v = *reinterpret_cast<const v4i*>(pb + arow[i] + (((ks * 4 + fq) ^ asw[i]) << 4));
                    v &= keep[i];
                    fa[i] = __builtin_bit_cast(bf16x8, v);
                }
#pragma unroll
                for (int q = 0; q < NT; ++q)
                    fb[q] = *reinterpret_cast<const bf16x8*>(lb + swz(wn * 64 + q * 16 + frow, ks * 4 + fq));
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc[i][q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[q], acc[i][q], 0, 0, 0);
            }
        }
    }
    igemm_epilogue<BN, false, MT, MODE, NW>(p, acc, m0, n0, tile_m, smem, zpre ? &zt : nullptr);
}

template <int BMP, int BN>
void launch_cfg(const IgemmArgs& a, int PR, int halo, int npb, size_t lds, hipStream_t stream) {
    constexpr int NT_ = (BMP / 64) * (BN / 64) * 64;
    const unsigned tiles = (unsigned)ep24_cdiv(a.M, BMP) * (unsigned)ep24_cdiv(a.N, BN);
    if (a.bn_z) {
        static bool attr1 = false;
        if (!attr1) { (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BMP, BN, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr1 = true; }
        hipLaunchKernelGGL((conv_patch_kernel<BMP, BN, 1>), dim3(tiles), dim3(NT_), lds, stream, a, PR, halo, npb);
    } else {
        static bool attr0 = false;
        if (!attr0) { (void)hipFuncSetAttribute((const void*)conv_patch_kernel<BMP, BN, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr0 = true; }
        hipLaunchKernelGGL((conv_patch_kernel<BMP, BN, 0>), dim3(tiles), dim3(NT_), lds, stream, a, PR, halo, npb);
    }
}

}  // namespace

namespace ep24_igemm {

// Tile choice: the largest tile that still gives the chip about one workgroup per CU and whose patch fits the LDS.
bool launch_patch(const IgemmArgs& a, hipStream_t stream) {
    for (int t = 0; t < 9; ++t)
        if (a.oy[t] < -1 || a.oy[t] > 1 || a.ox[t] < -1 || a.ox[t] > 1) return false;
    if (a.K % 8 != 0 || a.N < 16) return false;
    const int halo = a.SW + 1;
    const int KC = (a.K + BK - 1) / BK;
    const int npb = KC > 1 ? 2 : 1;
    const bool wide_n = a.N > 64;
    const int bn = wide_n ? 128 : 64;
    const long tn = ep24_cdiv(a.N, bn);
    const int order[2] = {256, 128};
    for (int k = 0; k < 2; ++k) {
        const int bmp = order[k];
        if (bmp == 256 && (long)ep24_cdiv(a.M, 256) * tn < 200) continue;     // would leave CUs idle: smaller tiles
        const int nw = (bmp / 64) * (bn / 64);
        const int pr = (bmp + 2 * halo + 8 * nw - 1) / (8 * nw) * (8 * nw);
        const size_t lds = (size_t)npb * pr * 128 + 2 * (size_t)bn * 128;
        if (lds > 160 * 1024) continue;
        if (bmp == 256 && wide_n) launch_cfg<256, 128>(a, pr, halo, npb, lds, stream);
        else if (bmp == 256) launch_cfg<256, 64>(a, pr, halo, npb, lds, stream);
        else if (wide_n) launch_cfg<128, 128>(a, pr, halo, npb, lds, stream);
        else launch_cfg<128, 64>(a, pr, halo, npb, lds, stream);
        return true;
    }
    return false;
}

}  // namespace ep24_igemm
