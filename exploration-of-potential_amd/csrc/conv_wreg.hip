// ep24 - 3x3 stride-1 convolutions with at most 64 input and 64 output channels (forward and input gradient): the WEIGHTS LIVE IN
// REGISTERS.  (YOLOX-l: the three Bottleneck 3x3 layers of dark2, 160 x 160 x 64 at 640 x 640 - 512 000 pixels per launch at B = 20.)
//
// Why its own kernel (round 5).  The tiled kernel runs these layers in 128 x 64 tiles, four waves of 32 x 64: six fragment reads per
// eight MFMAs, 48 KB of LDS reads + 24 KB of LDS-DMA writes per 64-MFMA step of a workgroup - 576 cycles of the LDS pipe against 256
// of the matrix pipe (two workgroups per CU: 1 152 against 512).  It is LDS-bound by a factor of 2.25 (SQ: MFMA busy 0.23; 78 us
// alone, 132 us beside the weight-gradient lane), and the ring's 256 x 64 tile has the same read ratio (profiles/r04_ring_narrow_ab.txt).
// The whole weight tensor of such a layer is 9 x 64 x 64 bf16 = 72 KB.  Here:
//
//   * one workgroup of 8 waves per CU (two per SIMD: each hides the other's address arithmetic, waits and stores - the first form of
//     this kernel, four waves of 64 x 64 with 288 weight registers each, spent 3/4 of its time on single-issue latency), a wave =
//     64 pixels x 32 output channels of a 256-pixel tile; it keeps the weight fragments of ITS 32 channels (9 taps x 2 k halves x 2
//     blocks of 16 = 144 registers) for the whole launch: no weight traffic in the loop, the only LDS reads are the activation
//     fragments, 4 per 8 MFMAs
//   * PERSISTENT over tiles (an XCD takes a contiguous run of tiles: neighbours share their halo rows in its L2); the weights come in
//     once per CU (one coalesced copy into LDS, fragments read from there), the activation stream never drains: the windows of the
//     next tile are in flight while this one multiplies, the stores of the previous one leave meanwhile
//   * the activations of one tap ROW (dy) come as one window per tile: the 256 pixels in PADDED coordinates (row length W + 2, one
//     zero column left and right), so that the three taps dx = -1, 0, 1 read the same window at slot + dx - a third of the tiled
//     kernel's L2 -> LDS traffic, and no masks at the fragments: the padding is in the window (out-of-range DMA offsets write zeros).
//     Slot s of a window holds padded position x0 + s counted from the tile's first pixel; a lane's fragment slots are per-tile
//     registers.  Three window buffers = the three dy of a tile; two windows in flight behind counted waits.
//   * the weights are the MFMA's FIRST operand (D[channel][pixel]): with the fragment rows relabelled (row j of block qq = channel
//     8 (j >> 2) + 4 qq + (j & 3)) a lane ends up with 8 consecutive channels of a pixel - 16-byte stores straight from the
//     accumulators, no staging through LDS.  BatchNorm statistics accumulate in registers over all tiles of the workgroup and leave
//     as one set of fixed-point atomics per workgroup.
//
// Arithmetic: the products of a pixel in the order of the tiled kernel (taps 0 .. 8, two k halves of 32 each, v_mfma_f32_16x16x32_bf16,
// fp32 accumulate): the OUTPUTS are bit-identical to the tiled kernel's (tests/test_gpu_conv.py asserts it).  The statistics are sums
// of the same fp32 values in another order (a lane adds its pixels of every tile before the fixed-point conversion; the tiled kernel
// converts per 128-row tile): equal to fp32 rounding, reproducible from run to run (the tile -> workgroup map is static).
#include <atomic>
#include <type_traits>
#include "igemm.h"

using namespace ep24_igemm;

namespace {

constexpr int TM = 256;                          // pixels per tile
constexpr int NW = 8;                            // waves: (pixel block of 64) x (channel half of 32)
constexpr int NSLOT = 320;                       // window slots: 256 + 2 per image row the tile touches (W >= 32: at most 9) + 2 = 276, rounded up to 8 x NW
constexpr int NI = NSLOT / 8 / NW;               // LDS-DMA instructions per wave and window (5)
constexpr int WBYTES = NSLOT * 128;              // one window: [slot][64 channels] bf16, 16-byte chunk index XOR (slot & 7)
constexpr int NB = 3;                            // window buffers = the three dy of a tile
constexpr int RED_OFF = NB * WBYTES;             // [NW waves][2][32] floats
constexpr int WL_OFF = RED_OFF + NW * 2 * 32 * 4;   // the weight fragments of the taps that do not fit the registers: [tap - NRT][h][channel block][lane] x 16 B
constexpr int NRT = 7;                           // taps whose fragments live in registers (9: 144 + 32 accumulators + the rest spilled 29 registers)
constexpr int WREG_LDS = WL_OFF + (9 - NRT) * 8 * 1024;
constexpr int NST = 4;                           // 16-byte stores per lane and tile
constexpr int WCHUNKS = 64 * 9 * 8;              // 16-byte chunks of the (zero-padded) weight tensor [64][9][64]
static_assert(WCHUNKS * 16 <= NB * WBYTES && WCHUNKS % (NW * 64) == 0, "the weights pass through the window buffers once");

// Diagnostic builds only (timing, wrong results): bit 0 no MFMAs, bit 1 no output stores, bit 2 no window DMA, bit 3 no weight loads,
// bit 4 no per-tile DMA address set-up (the first tile's offsets for every tile).  The product is WREG_VAR 0.
#ifndef WREG_VAR
#define WREG_VAR 0
#endif

template <int N>
__device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

typedef int v4i __attribute__((ext_vector_type(4)));

template <bool STATS>                                   // false: an input gradient - no batch statistics (their sums were a fifth of the epilogue)
__global__ __launch_bounds__(NW * 64) void conv_wreg_kernel(const IgemmArgs p, const FastDiv d_wp, const FastDiv d_h, const int n_tiles, const int per_xcd, const int sgn) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int frow = lane & 15, fq = lane >> 4;
    const int pb = wave >> 1, nh = wave & 1;                   // the wave's pixel block (64 pixels) and channel half (32 channels)
    const int W = p.GW, H = p.GH;
    const auto src_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.src), 0, p.src_bytes, 0x00020000);
    const auto wt_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.wt), 0, p.wt_bytes, 0x00020000);
    const auto dst_rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<bf16*>(p.dst), 0, p.dst_bytes, 0x00020000);

    // this workgroup's tiles: XCD x (workgroups are dealt round-robin over the XCDs) owns [x * per_xcd, (x + 1) * per_xcd), its
    // workgroups walk them interleaved - the tiles in flight on an XCD at any time are neighbours
    const int xcd = blockIdx.x & 7, wj = blockIdx.x >> 3, wpx = gridDim.x >> 3;
    const int t_lo = xcd * per_xcd, t_hi = min(t_lo + per_xcd, n_tiles);
    const int my_tiles = t_lo + wj < t_hi ? (t_hi - t_lo - wj + wpx - 1) / wpx : 0;
    if (my_tiles == 0) return;                                // (uniform; nothing was issued yet)

    // ---- the weights, once per CU: a coalesced copy of the tensor into LDS in FRAGMENT order, then every lane reads its fragments.
    // Fragment (tap t, k half h, channel block cb of 16) is 64 lanes x 16 B: lane (j = frow, fq) holds channel
    // ch(cb, j) = (cb >> 1) * 32 + 8 (j >> 2) + 4 (cb & 1) + (j & 3), k = h * 32 + fq * 8 .. + 7 of the tap's weight slot.
    // (Every lane fetching its own fragments from memory - 16 rows 4.6 KB apart per instruction, 288 KB per CU from the same few L2
    // channels on all CUs at once - cost 9 us at the head of every launch.)
    bf16x8 Wf[NRT][2][2];
    {
        constexpr int CPT = WCHUNKS / (NW * 64);              // 9 chunks per thread
        v4i wv[CPT];
        const int kc8 = p.K >> 3;
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int g = u * (NW * 64) + tid;                 // logical chunk (ch, slot, c) of [64][9][8]
            const int ch = g / 72, rem = g - ch * 72, slot = rem >> 3, c = rem & 7;
            const int okm = -(int)(ch < p.N && c < kc8);
            const int off = ((int)((((long)ch * p.WT + slot) * p.K + c * 8) * 2) & okm) | (OOB & ~okm);
            wv[u] = __builtin_amdgcn_raw_buffer_load_b128(wt_rsrc, (WREG_VAR & 8) ? OOB : off, 0, 0);
        }
        // slot -> tap (the tap tables are a permutation of the nine slots: checked on the host)
        unsigned long long tap_of = 0ull;
        for (int t = 0; t < 9; ++t) tap_of |= (unsigned long long)t << (4 * (int)((p.tap_slot >> (4 * t)) & 15ull));
#pragma unroll
        for (int u = 0; u < CPT; ++u) {
            const int g = u * (NW * 64) + tid;
            const int ch = g / 72, rem = g - ch * 72, slot = rem >> 3, c = rem & 7;
            const int t = (int)((tap_of >> (4 * slot)) & 15ull);
            const int cb = ((ch >> 5) << 1) | ((ch >> 2) & 1), j = (((ch & 31) >> 3) << 2) | (ch & 3);
            const int dst = (((t * 2 + (c >> 2)) * 4 + cb) * 64 + (c & 3) * 16 + j) * 16;
            *reinterpret_cast<v4i*>(smem + (t < NRT ? dst : dst + (WL_OFF - NRT * 8 * 1024))) = wv[u];      // (the last taps: to where they stay)
        }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NRT; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq)
                    Wf[t][h][qq] = *reinterpret_cast<const bf16x8*>(smem + (((t * 2 + h) * 4 + nh * 2 + qq) * 64 + lane) * 16);
        __syncthreads();                                       // (the window DMA below overwrites the staging area)
    }

    // ---- DMA side: per tile and lane the byte offset of its NI slots for tap row 1 and which of the three tap rows exist for them
    const int lchunk = (lane & 7) ^ ((lane >> 3) & 7);
    const int rowb = sgn * (int)(W * p.ld_src * 2);          // sgn = -1: an input gradient - tap t sits at (-(t / 3 - 1), -(t % 3 - 1)), the windows come bottom row first
    const int BH = p.B * H;
    int off0[NI];
    unsigned vm = 0u;
    auto dma_setup = [&](int tile) {
        const int m0 = tile * TM;
        const int n0 = fdiv(m0, p.d_plane);
        const int rem = m0 - n0 * (H * W);
        const int y0 = fdiv(rem, p.d_gw), x0 = rem - y0 * W;
        const int yg0 = n0 * H + y0;
        vm = 0u;
        // slot (j * NW + wave) * 8 + lane / 8: the first by division, the others 8 NW = 64 padded positions further each (W + 2 >= 34:
        // at most two row wraps per step)
        const int F0 = x0 + wave * 8 + (lane >> 3);
        const int yrel0 = fdiv(F0, d_wp);
        int xp = F0 - yrel0 * (W + 2);
        int yg = yg0 + yrel0;
        int y = yg - fdiv(yg, d_h) * H;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            if (j > 0) {
                xp += 8 * NW;
#pragma unroll
                for (int wr = 0; wr < 2; ++wr) {
                    const bool c = xp >= W + 2;
                    xp -= c ? W + 2 : 0; yg += c ? 1 : 0; y += c ? 1 : 0;
                    y = y >= H ? y - H : y;
                }
            }
            const bool ok = xp >= 1 && xp <= W && yg < BH && lchunk * 8 < p.K;
            off0[j] = (int)((((long)yg * W + xp - 1) * p.ld_src + lchunk * 8) * 2);
            const bool up = y >= 1, down = y <= H - 2;                     // bit dyi: the row of tap row dyi, sgn * (dyi - 1), exists
            const unsigned b = ok ? (((sgn > 0 ? up : down) ? 1u : 0u) | 2u | ((sgn > 0 ? down : up) ? 4u : 0u)) : 0u;
            vm |= b << (3 * j);
        }
    };
    auto issue = [&](auto dyi_c) {                            // window of tap row dyi of the DMA side's tile into buffer dyi
        constexpr int dyi = decltype(dyi_c)::value;
        char* buf = smem + dyi * WBYTES;
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int m = -(int)((vm >> (3 * j + dyi)) & 1u);
            const int vo = ((off0[j] + (dyi - 1) * rowb) & m) | (OOB & ~m);
            if constexpr (!(WREG_VAR & 4)) __builtin_amdgcn_raw_ptr_buffer_load_lds(src_rsrc, (lptr_t)(buf + (j * NW + wave) * 1024), 16, vo, 0, 0, 0);
            else { int k_ = vo; asm volatile("" :: "v"(k_)); }
        }
    };

    // ---- compute side: the window slots of a lane's four pixels (pixel block i: pixel pb * 64 + i * 16 + frow of the tile)
    int ad[4][3];                                            // LDS byte offset of (pixel block i, tap column d), k half 0; half 1 = ^ 64
    int c_m0 = 0;
    auto cmp_setup = [&](int tile) {
        const int m0 = tile * TM;
        c_m0 = m0;
        const int n0 = fdiv(m0, p.d_plane);
        const int rem = m0 - n0 * (H * W);
        const int y0 = fdiv(rem, p.d_gw), x0 = rem - y0 * W;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = pb * 64 + i * 16 + frow;
            const int Fr = x0 + r;
            const int yrel = fdiv(Fr, p.d_gw), xr = Fr - yrel * W;
            const int slot = yrel * (W + 2) + xr + 1 - x0;
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const int sl = slot + sgn * (d - 1);
                ad[i][d] = sl * 128 + ((fq ^ (sl & 7)) << 4);
            }
        }
    };

    f32x4 acc[4][2];                                          // [pixel block][channel block]: channels nh * 32 + 8 fq + 4 qq + r of pixel i * 16 + frow
    float s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
    auto compute = [&](auto dyi_c) {
        constexpr int dyi = decltype(dyi_c)::value;
        const char* buf = smem + dyi * WBYTES;
        if constexpr (dyi == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) acc[i][qq] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int t = dyi * 3 + d;
                bf16x8 fx[4], fw[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) fx[i] = *reinterpret_cast<const bf16x8*>(buf + (ad[i][d] ^ (h * 64)));
#pragma unroll
                for (int qq = 0; qq < 2; ++qq) {
                    if (t < NRT) fw[qq] = Wf[t < NRT ? t : 0][h][qq];
                    else fw[qq] = *reinterpret_cast<const bf16x8*>(smem + WL_OFF + ((((t - NRT) * 2 + h) * 4 + nh * 2 + qq) * 64 + lane) * 16);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int qq = 0; qq < 2; ++qq)
                        if constexpr (!(WREG_VAR & 1)) acc[i][qq] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[qq], fx[i], acc[i][qq], 0, 0, 0);
                        else asm volatile("" : "+v"(acc[i][qq]) : "v"(fx[i]), "v"(fw[qq]));
            }
        }
    };
    const int cbyte = (nh * 32 + fq * 8) * 2;
    const bool cok = nh * 32 + fq * 8 < p.N;
    auto epilogue = [&](auto full_c) {                        // full: every pixel of the tile exists (uniform; only the launch's last tile can hold pixels past M)
        constexpr bool full = decltype(full_c)::value;
        const int mw = c_m0 + pb * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mw + i * 16 + frow;
            const bool live = full || m < (int)p.M;
            bf16x8 o;
#pragma unroll
            for (int qq = 0; qq < 2; ++qq)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = acc[i][qq][r];
                    if constexpr (STATS) {
                        const float vs = live ? v : 0.f;
                        s1[qq * 4 + r] += vs; s2[qq * 4 + r] = fmaf(vs, vs, s2[qq * 4 + r]);
                    }
                    o[qq * 4 + r] = (bf16)v;
                }
            const int so = (live && cok) ? m * ((int)p.ld_dst * 2) + cbyte : OOB;      // every store is issued (counted waits)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i, o), dst_rsrc, (WREG_VAR & 2) ? OOB : so, 0, 0);
        }
    };

    // ---- the stream of steps: step (i, dyi) multiplies window dyi of the workgroup's i-th tile; its DMA was issued two steps
    // earlier.  Per step: this wave's DMAs of the step have landed (the younger window and the last tile's stores may stay in flight),
    // then everybody's (barrier) - which also says that every wave is through with the step before, whose buffer the window issued
    // now takes.
    const int tile0 = t_lo + wj;
    dma_setup(tile0);
    cmp_setup(tile0);
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    for (int i = 0; i < my_tiles; ++i) {
        const bool more = i + 1 < my_tiles;
        const int next = tile0 + (i + 1) * wpx;
        // tap row 0: younger than its window are window 1 (NI) and - from the second tile on - the stores of the tile before (NST)
        if (i == 0) wait_vm<NI>(); else wait_vm<NI + NST>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue(std::integral_constant<int, 2>{});
        compute(std::integral_constant<int, 0>{});
        if (more && !(WREG_VAR & 16)) dma_setup(next);
        // tap row 1: younger are window 2 (NI) and the stores of the tile before, which were issued between them
        if (i == 0) wait_vm<NI>(); else wait_vm<NI + NST>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) issue(std::integral_constant<int, 0>{});
        compute(std::integral_constant<int, 1>{});
        // tap row 2: younger is the next tile's window 0, if there is one
        if (more) wait_vm<NI>(); else wait_vm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (more) issue(std::integral_constant<int, 1>{});
        compute(std::integral_constant<int, 2>{});
        if (c_m0 + TM <= (int)p.M) epilogue(std::true_type{}); else epilogue(std::false_type{});
        if (more) cmp_setup(next);
    }

    if (STATS && p.stats) {
        // a lane's sums are over its pixels (frow, the four blocks, every tile): fold the 16 pixel lanes, then the four pixel-block
        // waves of a channel half through LDS, in a fixed order
        float* red = reinterpret_cast<float*>(smem + RED_OFF);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float a = s1[e], b = s2[e];
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); b += __shfl_xor(b, o, 64); }
            if (frow == 0) {
                red[(wave * 2 + 0) * 32 + fq * 8 + e] = a;
                red[(wave * 2 + 1) * 32 + fq * 8 + e] = b;
            }
        }
        __syncthreads();
        long long* st = p.stats + (long)(blockIdx.x % p.stats_replicas) * 2 * p.N;
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) v += red[(((r << 1) | (c >> 5)) * 2 + which) * 32 + (c & 31)];
            if (c < p.N) atomicAdd((unsigned long long*)(st + (long)which * p.N + c), (unsigned long long)to_fix(v));
        }
    }
}

}  // namespace

namespace ep24_igemm {

// The layers this kernel takes: 3x3 stride-1 pad-1 in the standard tap order, at most 64 channels on either side, a plain first-writer
// bf16 destination with 16-byte rows, no bias / inference epilogue / fused reduce, and enough pixels to give every CU a tile.
bool launch_wreg(const IgemmArgs& a, hipStream_t stream, bool dry, int* rc) {
    *rc = EP24_OK;
    if (a.T != 9 || a.sy != 1 || a.sx != 1 || a.GH != a.SH || a.GW != a.SW || a.WT != 9) return false;
    const int sgn = a.oy[0] < 0 ? 1 : -1;                    // forward / input gradient (the tap tables of conv_fwd_impl / conv_dgrad_impl)
    unsigned seen = 0u;
    for (int t = 0; t < 9; ++t) {
        if (a.oy[t] != sgn * (t / 3 - 1) || a.ox[t] != sgn * (t % 3 - 1) || a.wslot[t] < 0 || a.wslot[t] > 8) return false;
        seen |= 1u << a.wslot[t];
    }
    if (seen != 0x1FFu) return false;                         // the nine slots, each once
    const bool plain_dst = a.dsy == 1 && a.dsx == 1 && a.dy0 == 0 && a.dx0 == 0 && a.DW == a.GW && a.dp0 == 0 && a.dbs == (long)a.GH * a.GW;
    const long dst_b = ((a.M - 1) * a.ld_dst + a.N) * 2;
    if (!plain_dst || a.K > 64 || a.K % 8 || a.N > 64 || a.N % 8 || a.ld_dst % 8 || (reinterpret_cast<unsigned long long>(a.dst) & 15) || dst_b >= 0x7FFF0000L) return false;
    if (a.bias || a.epi_infer || a.accumulate || a.bnr_z || a.narrow_epi) return false;
    if (a.GW < 32 || a.M < 256L * TM || a.M != (long)a.B * a.GH * a.GW) return false;
    if (dry) return true;
    *rc = [&]() -> int {
        static std::atomic<unsigned long long> done{0};      // per-device attribute, set once
        int dev = 0;
        EP24_REQUIRE(hipGetDevice(&dev) == hipSuccess, EP24_E_LAUNCH, "conv_wreg: hipGetDevice failed");
        const unsigned long long bit = 1ull << (dev & 63);
        if (!(done.load(std::memory_order_acquire) & bit)) {
            hipError_t e = hipFuncSetAttribute((const void*)conv_wreg_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, WREG_LDS);
            if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_wreg_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, WREG_LDS);
            EP24_REQUIRE(e == hipSuccess, EP24_E_LAUNCH, "conv_wreg: hipFuncSetAttribute(MaxDynamicSharedMemorySize, %d) failed on device %d: %s", WREG_LDS, dev, hipGetErrorString(e));
            done.fetch_or(bit, std::memory_order_release);
        }
        IgemmArgs b = a;
        b.dst_bytes = (unsigned)dst_b;
        const int n_tiles = (int)((a.M + TM - 1) / TM);
        const int per_xcd = (n_tiles + 7) / 8;
        if (a.stats) hipLaunchKernelGGL(conv_wreg_kernel<true>, dim3(256), dim3(NW * 64), WREG_LDS, stream, b, make_fastdiv((unsigned)(a.GW + 2)), make_fastdiv((unsigned)a.GH), n_tiles, per_xcd, sgn);
        else hipLaunchKernelGGL(conv_wreg_kernel<false>, dim3(256), dim3(NW * 64), WREG_LDS, stream, b, make_fastdiv((unsigned)(a.GW + 2)), make_fastdiv((unsigned)a.GH), n_tiles, per_xcd, sgn);
        return EP24_OK;
    }();
    return true;
}

}  // namespace ep24_igemm
