// ep24 - bf16 weight-gradient convolution on CDNA4 MFMA.
//
//   dW[co][t][ci] += sum_p dY[p][co] * X[pix(p) + off(t)][ci]        p over the B*OH*OW output pixels
//
// Both operands are NHWC (channel-contiguous) while the reduction runs over pixels, so both MFMA operands
// are needed "k-strided".  The tiles are staged [pixel][channel] in LDS exactly as they sit in HBM (coalesced
// 256-B rows) and read back through gfx950's transposing LDS read ds_read_b64_tr_b16, which hands every lane
// 4 pixels of one channel; two of them make the 8-deep k fragment of v_mfma_f32_16x16x32_bf16.
// LDS rows are 256 B = one bank row, so 32-B (16-channel) blocks are XOR-permuted by a row key
// ((row&3) | ((row>>3)&1)<<2) that makes the 8 rows a half-wave touches in one transposed read hit disjoint
// banks; 16-B staging writes stay whole because the permutation moves 32-B units.
// Workgroup = 128 co x 128 ci of one tap, 4 waves (2x2 of 64x64), 64 pixels per K-step; pixels are split
// over blockIdx.y and combined with fp32 atomics (device scope, one 64-B segment per 16 lanes).
#include <stdlib.h>
#include <atomic>
#include <type_traits>
#include "common.h"

namespace {

struct WgradArgs {
    const bf16* x; long ld_x; const bf16* dy; long ld_dy; float* dw; long ld_dw;
    float* slab; long slab_stride;   // non-null: pixel split `s` stores its partial dW at slab + s*slab_stride (no atomics)
    int cout_valid, cin_valid;
    int B, H, W, OH, OW, Cin, Cout, ksize, stride, pad;
    long M;          // B*OH*OW
    long chunk;      // pixels per split (multiple of 64)
    int tiles_ci, tiles_co, T, tiles, xcd_remap;
    FastDiv d_plane, d_ow;      // pixel -> (n, oh, ow)
    unsigned x_bytes, dy_bytes; // extents for the buffer descriptors
    int c_n, c_oh, c_ow, c_pix; // byte strides of X per image / output row / output column / input pixel
    int s_dys, s_dxs, s_doff;   // what a 64-pixel step adds to (input row, input column, byte offset) before the wraps
};

// Row keys of the 32-byte-block XOR permutation.  256-B rows (128 channels): one row per bank row, the 8 rows a
// half-wave touches in one transposed read get 8 distinct keys.  128-B rows (64 channels): two rows per bank
// row, rows of equal parity get 4 distinct keys.
template <int TW> __device__ __forceinline__ int rkey(int row);
template <> __device__ __forceinline__ int rkey<128>(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
template <> __device__ __forceinline__ int rkey<64>(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
// 384-B rows (192 channels = the stem's 12 tap slots x 16): a row starts 32 banks after the one above it, so the 8 rows of a
// half-wave's transposed read sit at slot (block ^ key) ^ 4 * (row & 1) of the 8 32-byte slots of a bank row - distinct with the
// 64-channel key; the key stays below 4, so a permuted block stays inside its group of four of the 12 blocks
template <> __device__ __forceinline__ int rkey<192>(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
// byte offset of 16-B chunk c of row r in a [64][TW] bf16 tile
template <int TW> __device__ __forceinline__ int wr_off(int r, int c) {
    return r * (TW * 2) + ((((c >> 1) ^ rkey<TW>(r)) << 1 | (c & 1)) << 4);
}
// byte offset of 4 channels starting at 16-channel block cb, sub-slot pp of row r
template <int TW> __device__ __forceinline__ int tr_off(int r, int cb, int pp) {
    return r * (TW * 2) + ((cb ^ rkey<TW>(r)) << 5) + (pp << 3);
}

typedef bf16x4 __attribute__((address_space(3))) * lds_bf16x4_ptr;

template <int TW> __device__ __forceinline__ bf16x8 tr_frag(const char* tile, int row_lo, int cb, int pp) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + tr_off<TW>(row_lo, cb, pp)));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(tile + tr_off<TW>(row_lo + 4, cb, pp)));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

typedef __attribute__((address_space(3))) void* lptr_t;

// two transposed 8-byte reads (rows r..r+3 and r+4..r+7 of a 16-channel block) = one 8-deep MFMA k fragment
template <int OFF_LO, int OFF_HI> __device__ __forceinline__ bf16x8 tr_pair(unsigned addr) {
    bf16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(lo) : "v"(addr), "n"(OFF_LO));
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi) : "v"(addr), "n"(OFF_HI));
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}

// TCO x TCI output tile of one tap per workgroup (128 or 64 each), 4 waves as 2x2.
// Staging is LDS-DMA (buffer_load ... lds, 16 B per lane): no VGPR round trip and no ds_write - the VGPR->LDS
// store path (~79 B/clk/CU) was the busiest pipe of the register-staged version.  The DMA writes LDS linearly, so
// the 32-B-block permutation is applied to the per-lane SOURCE channel; rows past the pixel range, padding taps
// and channel tails use an out-of-range buffer offset (the DMA then writes zeros).  Two stages, one barrier per
// 64-pixel step.
//
// VTAP (the Focus stem, round 3): X is the space-to-depth image [B][H][W][16] bf16 and the "channels" of an X tile are tap slots
// x 16 channels of the 3x3 neighbourhood (the taps past the ninth and the channels past the twelfth are zeros / dropped).  Since
// round 5 the stem runs as ONE tile of TCI = 192 = 12 tap slots (launch_stem below; as three 64-wide tiles g = taps 4 g .. 4 g + 3
// every dY row was read three times).  The offset table has one entry per (tile row, tap slot) - 64 * TCI / 16 per step, TCI / 64
// per thread - and a DMA lane picks the entry of its chunk's tap.  dW leaves as [Cout][tap * 12 + channel], the layout of the im2col form it replaces
// (which re-read 224 bytes per pixel here: 0.40 ms at B = 20).
template <int TCO, int TCI, bool VTAP = false>
__device__ __forceinline__ void wgrad_body(const WgradArgs& p, const int vblock) {
    constexpr int Y_BYTES = 64 * TCO * 2, X_BYTES = 64 * TCI * 2, STAGE = Y_BYTES + X_BYTES;
    constexpr int FM = TCO / 32, FN = TCI / 32;          // 16x16 fragments per wave along co / ci
    constexpr int YCH = TCO / 8, XCH = TCI / 8;          // 16-B chunks per tile row
    constexpr int YI = 64 * YCH / 256, XI = 64 * XCH / 256;   // DMA instructions per wave (4 or 2)
    constexpr int OOB = 0x7FFFFFF0;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // 1-D grid; workgroup L runs on XCD L % 8.  Every XCD takes one contiguous run of the (split, tile) sequence
    // (split-major), so the tiles of a pixel split - which all re-read that split's dY / X rows, 9 taps x the other
    // operand's tile count - run on one XCD (two at a run boundary) and the rows cross the fabric once or twice
    // instead of once per XCD: FETCH_SIZE showed 6x the algorithmic bytes with the splits dealt round-robin.
    // (the remap itself lives in the kernels below: vblock is this workgroup's place in the problem's (split, tile) sequence)
    int bx = vblock % p.tiles;
    const int by = vblock / p.tiles;
    const int tap = bx % p.T; bx /= p.T;
    const int ci0 = (bx % p.tiles_ci) * TCI;
    const int co0 = (bx / p.tiles_ci) * TCO;
    const int kh = tap / p.ksize, kw = tap % p.ksize;
    const long p_begin = (long)by * p.chunk;
    const long p_end = min(p.M, p_begin + p.chunk);
    const int n_iter = (int)((p_end - p_begin + 63) / 64);      // uniform over the workgroup

    const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.dy), 0, p.dy_bytes, 0x00020000);
    const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.x), 0, p.x_bytes, 0x00020000);
    // lane -> (tile row, logical 16-B chunk) of DMA instruction j: LDS unit U = (wave*NI + j)*64 + lane
    int y_row[YI], y_off[YI];
#pragma unroll
    for (int j = 0; j < YI; ++j) {
        const int U = (wave * YI + j) * 64 + lane;
        const int r = U / YCH, pc = U % YCH;
        const int c = ((((pc >> 1) ^ rkey<TCO>(r)) << 1) | (pc & 1));
        y_row[j] = r;
        y_off[j] = co0 + c * 8 < p.Cout ? (int)((((p_begin + r) * p.ld_dy) + co0 + c * 8) * 2) : OOB;
    }
    // X rows.  Which input pixel a tile row reads (and whether it is padding) depends only on (step, row), not on the
    // lane's channel chunk, so the 64 row offsets of a step are produced ONCE per workgroup - 16 rows per wave, one row
    // per lane, carried from step to step by constant deltas plus at most one wrap of ow and of oh - into a small LDS
    // table two steps ahead; a DMA instruction then costs one ds_read_b32 and one add.  (Decoding per DMA instruction,
    // two divisions and 64-bit multiplies each, kept the VALU busier than the MFMAs: 90 -> 62 us per 3x3 256-ch layer.)
    int* xtab = reinterpret_cast<int*>(smem + 2 * STAGE);          // [2][64] byte offsets, XOOB = padding / past the end (VTAP: [2][64][4])
    constexpr int XOOB = 0x7FFF0000;
    constexpr int XT = VTAP ? 64 * (TCI / 16) : 64;                 // table entries per step (VTAP: a row has TCI / 16 tap slots)
    constexpr int NT = VTAP ? TCI / 64 : 1;                         // entries a thread produces per step (VTAP: tap slots tid % 4 + 4 k of its row)
    int x_row[XI], x_colb[XI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
        const int U = (wave * XI + j) * 64 + lane;
        const int r = U / XCH, pc = U % XCH;
        const int c = ((((pc >> 1) ^ rkey<TCI>(r)) << 1) | (pc & 1));
        if constexpr (VTAP) {                                       // chunk c = tap c / 2 of the tile, half c & 1 of the pixel
            x_row[j] = r * (TCI / 16) + (c >> 1);
            x_colb[j] = (c & 1) * 16;
        } else {
            x_row[j] = r;
            x_colb[j] = ci0 + c * 8 < p.Cin ? (ci0 + c * 8) * 2 : XOOB;
        }
    }
    const int p_end_i = (int)p_end;
    // producer state of tile row wave*16 + (lane & 15)   (VTAP: of (row tid / 4, tap tid % 4 of the tile))
    const int t_row = VTAP ? tid >> 2 : wave * 16 + (lane & 15);
    const int vt = (ci0 >> 6) * 4 + (tid & 3);                     // VTAP: this thread's tap; 9 .. 11 do not exist
    const int pkh = VTAP ? (vt * 11) >> 5 : kh, pkw = VTAP ? vt - 3 * ((vt * 11) >> 5) : kw, ppad = VTAP ? 1 : p.pad;
    int t_p = (int)p_begin + t_row, t_ys, t_xs, t_off;
    {
        const int pd = t_p < p.M ? t_p : 0;
        const int n = fdiv(pd, p.d_plane);
        const int rem = pd - n * (p.OH * p.OW);
        const int oh = fdiv(rem, p.d_ow), ow = rem - oh * p.OW;
        t_ys = oh * p.stride; t_xs = ow * p.stride;                 // input row / column of tap (pad, pad)
        t_off = n * p.c_n + oh * p.c_oh + ow * p.c_ow + ((pkh - ppad) * p.W + (pkw - ppad)) * p.c_pix;
    }
    const int lo_y = ppad - pkh, lo_x = ppad - pkw, wrap_x = p.OW * p.stride, wrap_y = p.OH * p.stride;
    // VTAP with more than one tap slot per thread: slot k is tap vt + 4 k, t_off is the first one's and td[k] what the k-th adds
    int lo_yk[NT], lo_xk[NT], td[NT];
    if constexpr (VTAP) {
#pragma unroll
        for (int k = 0; k < NT; ++k) {
            const int v = vt + 4 * k, h = (v * 11) >> 5, w = v - 3 * h;
            lo_yk[k] = 1 - h; lo_xk[k] = 1 - w;
            td[k] = v < 9 ? ((h - pkh) * p.W + (w - pkw)) * p.c_pix : -1;
        }
    }
    auto produce = [&](int slot) {
        if constexpr (VTAP) {
#pragma unroll
            for (int k = 0; k < NT; ++k) {
                const bool ok = t_p < p_end_i && (unsigned)(t_ys - lo_yk[k]) < (unsigned)p.H && (unsigned)(t_xs - lo_xk[k]) < (unsigned)p.W && td[k] != -1;
                xtab[slot * XT + t_row * (TCI / 16) + (tid & 3) + 4 * k] = ok ? t_off + td[k] : XOOB;
            }
        } else {
            const bool ok = t_p < p_end_i && (unsigned)(t_ys - lo_y) < (unsigned)p.H && (unsigned)(t_xs - lo_x) < (unsigned)p.W;
            if (lane < 16) xtab[slot * 64 + t_row] = ok ? t_off : XOOB;
        }
        t_p += 64; t_ys += p.s_dys; t_xs += p.s_dxs; t_off += p.s_doff;
        if (t_xs >= wrap_x) { t_xs -= wrap_x; t_ys += p.stride; t_off += p.c_oh - p.OW * p.c_ow; }
        if (t_ys >= wrap_y) { t_ys -= wrap_y; t_off += p.c_n - p.OH * p.c_oh; }
    };
    const int y_step = (int)(64 * p.ld_dy * 2);

    auto issue = [&](int it, int buf) {
        char* stage = smem + buf * STAGE;
        const int rows_left = p_end_i - (int)p_begin - it * 64;
        const int* tab = xtab + (it & 1) * XT;
        int xo[XI];
#pragma unroll
        for (int j = 0; j < XI; ++j) xo[j] = tab[x_row[j]];
#pragma unroll
        for (int j = 0; j < YI; ++j) {
            const bool ok = y_row[j] < rows_left && y_off[j] != OOB;
            int vo = ok ? y_off[j] + it * y_step : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (lptr_t)(stage + (wave * YI + j) * 1024), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < XI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lptr_t)(stage + Y_BYTES + (wave * XI + j) * 1024), 16,
                                                     (int)((unsigned)xo[j] + (unsigned)x_colb[j]), 0, 0, 0);
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fq = lane >> 4;            // k group: pixels 8*fq .. 8*fq+7 of a 32-pixel step
    const int q = (lane & 15) >> 2;      // row inside the 4-row transposed block this lane addresses
    const int pp4 = lane & 3;            // 4-channel sub-slot this lane addresses

    // LDS byte addresses of this lane's fragment reads in stage 0, rows (fq*8 + q) / +4, 32-pixel half 0; the other
    // half, the +4 rows and the second stage are immediate / uniform offsets (they do not change the row key)
    const unsigned lds0 = (unsigned)(size_t)(lptr_t)smem;
    unsigned ya[FM], xa[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) ya[i] = lds0 + tr_off<TCO>(fq * 8 + q, wm * FM + i, pp4);
#pragma unroll
    for (int j = 0; j < FN; ++j) xa[j] = lds0 + Y_BYTES + tr_off<TCI>(fq * 8 + q, wn * FN + j, pp4);

    produce(0);
    produce(1);
    __syncthreads();
    if (n_iter > 0) issue(0, 0);
    for (int it = 0; it < n_iter; ++it) {
        __syncthreads();                 // vmcnt(0) + barrier: step `it` landed, the other stage is free
        produce(it & 1);                 // offsets of step it+2; slot it&1 was last read when step `it` was issued
        if (it + 1 < n_iter) issue(it + 1, (it + 1) & 1);
        // Fragment reads are inline asm: the compiler puts an s_waitcnt vmcnt(0) in front of the
        // ds_read_tr16_b64 BUILTIN (it treats it as a possible LDS store that must order behind the LDS-DMA just
        // issued), which serialised every step's DMA round trip with its MFMAs.  LDS returns in order and lgkmcnt
        // saturates at 15, so once all 32 reads are issued the 16 of the first 32-pixel half have landed.
        const unsigned sb = (unsigned)((it & 1) * STAGE);
        bf16x8 fa0[FM], fb0[FN], fa1[FM], fb1[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) fa0[i] = tr_pair<0, 4 * TCO * 2>(ya[i] + sb);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb0[j] = tr_pair<0, 4 * TCI * 2>(xa[j] + sb);
#pragma unroll
        for (int i = 0; i < FM; ++i) fa1[i] = tr_pair<32 * TCO * 2, 36 * TCO * 2>(ya[i] + sb);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb1[j] = tr_pair<32 * TCI * 2, 36 * TCI * 2>(xa[j] + sb);
        // wait until only the second half's 2*(FM+FN) reads are outstanding (the counter saturates at 15)
        if constexpr (2 * (FM + FN) >= 16) asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
        else if constexpr (2 * (FM + FN) == 12) asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
        static_assert(2 * (FM + FN) == 16 || 2 * (FM + FN) == 12 || 2 * (FM + FN) == 8, "fragment read count");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa0[i], fb0[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);          // keep the first half's MFMAs in front of the wait
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa1[i], fb1[j], acc[i][j], 0, 0, 0);
    }

    // D[row = co][col = ci]: row = 4*fq + r, col = lane & 15
    const int fr = lane & 15;
    // Epilogue: the accumulators go through LDS so that every atomic wave-instruction adds 64 CONSECUTIVE floats of
    // one dW row (256 contiguous bytes - the shape the memory-side atomic units take at full rate) instead of
    // four 64-byte pieces in four rows.
    __syncthreads();
    float* tile = reinterpret_cast<float*>(smem);            // [TCO][TCI] fp32 <= 64 KB = the two staging stages
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                tile[(wm * (TCO / 2) + i * 16 + 4 * fq + r) * TCI + wn * (TCI / 2) + j * 16 + fr] = acc[i][j][r];
    __syncthreads();
    // slab form with whole 4-channel groups: 16-byte stores, a wave covers 8 rows x 128 B (64 x 64 tile) or 4 rows x 256 B per
    // instruction - a quarter of the store instructions of the 4-byte form below (the store tail is issue bound)
    if constexpr (VTAP) {                                    // column c of tile g = (tap 4 g + c / 16, channel c % 16) -> dW column tap * 12 + channel
        for (int row = wave; row < TCO; row += 4) {
            const int co = co0 + row;
            if (co >= p.cout_valid) break;
#pragma unroll
            for (int c = lane; c < TCI; c += 64) {
                const int vtap = (ci0 >> 6) * 4 + (c >> 4), cc = c & 15;
                if (vtap < 9 && cc < 12) p.slab[(long)by * p.slab_stride + (long)co * p.ld_dw + vtap * 12 + cc] = tile[row * TCI + c];
            }
        }
        return;
    }
    if (p.slab && (p.cin_valid & 3) == 0 && (p.ld_dw & 3) == 0 && ((long)tap * p.cin_valid & 3) == 0 &&
        (((unsigned long long)p.slab | (unsigned long long)(p.slab_stride * 4)) & 15) == 0) {
        constexpr int C4 = TCI / 4;                                // float4 groups per tile row
        float* base = p.slab + (long)by * p.slab_stride + (long)tap * p.cin_valid + ci0;
        for (int u = tid; u < TCO * C4; u += 256) {
            const int row = u / C4, c4 = u - row * C4;
            const int co = co0 + row;
            if (co < p.cout_valid && ci0 + c4 * 4 < p.cin_valid)
                *reinterpret_cast<f32x4*>(base + (long)co * p.ld_dw + c4 * 4) = *reinterpret_cast<const f32x4*>(tile + row * TCI + c4 * 4);
        }
        return;
    }
    for (int row = wave; row < TCO; row += 4) {
        const int co = co0 + row;
        if (co >= p.cout_valid) break;
        const long e = (long)co * p.ld_dw + (long)tap * p.cin_valid + ci0;
        if (p.slab) {
            float* d = p.slab + (long)by * p.slab_stride + e;
#pragma unroll
            for (int h = 0; h < TCI / 64; ++h) {
                const int c = h * 64 + lane;
                if (ci0 + c < p.cin_valid) d[c] = tile[row * TCI + c];
            }
        } else {
            float* d = p.dw + e;
#pragma unroll
            for (int h = 0; h < TCI / 64; ++h) {
                const int c = h * 64 + lane;
                if (ci0 + c < p.cin_valid) atomicAdd(d + c, tile[row * TCI + c]);
            }
        }
    }
}

// place of workgroup blockIdx.x in the launch's sequence when every XCD takes one contiguous run of it
__device__ __forceinline__ int xcd_run_index() {
    const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
    return (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
}

template <int TCO, int TCI, bool VTAP = false>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    wgrad_body<TCO, TCI, VTAP>(p, p.xcd_remap ? xcd_run_index() : (int)blockIdx.x);
}

// Up to 16 weight gradients of ONE tile class in one launch (round 5, VERDICT r3 / r4: "one grouped launch per shape class").
// A weight gradient is needed by nobody before the optimizer, every layer keeps its own dz and its input activation, so the
// gradients of the layers of a backward segment can wait for each other - and as ONE launch they do not need to be cut into
// pixel splits to fill the chip: 15 1x1 layers of the 40 x 40 level were 15 launches of 16 tiles x 32 splits (16 steps of 64
// pixels per workgroup behind a prologue and a 64-KB epilogue, 32 fp32 slabs per layer for the reduce launch to fold); grouped
// they are 240 tiles x 2 splits of 250 steps.  The workgroups walk (problem, split, tile) in order, an XCD takes a contiguous
// run.  Same arithmetic per (tile, split): bit-reproducible as before (the split count is part of the plan).
constexpr int WG_MAX = 16;
struct WgradGroup {
    WgradArgs a[WG_MAX];
    int prefix[WG_MAX + 1];
    int n;
};
static_assert(sizeof(WgradGroup) <= 4096, "kernel arguments are passed by value");

template <int TCO, int TCI>
__global__ __launch_bounds__(256) void wgrad_group_kernel(const WgradGroup g) {
    const int v = xcd_run_index();
    int i = 0;
#pragma unroll
    for (int k = 1; k < WG_MAX; ++k)
        if (k < g.n && v >= g.prefix[k]) i = k;
    i = __builtin_amdgcn_readfirstlane(i);
    wgrad_body<TCO, TCI, false>(g.a[i], v - g.prefix[i]);
}

// ---------------------------------------------------------------------------------------------------------------------------
// Loader / consumer ring form of the weight gradient for layers with Cout >= 256 (round 4; the structure of conv_ring.hip).
// BUILT, MEASURED, OFF BY DEFAULT: correct (tests/test_gpu_conv.py runs it on every hot shape) and 10 % SLOWER than wgrad_kernel on
// every layer it takes, 23.1 against 22.5 ms per step (profiles/r04_wgrad_ring_ab.txt).  The reasoning below holds for the LDS
// traffic; what it leaves out is that wgrad_kernel's two workgroups per CU hide each other's epilogue (64 KB of fp32 slab stores
// per workgroup, store-issue bound) and prologue, while one ring workgroup per CU exposes 128 KB of them at the end of its one round,
// and that 48 transposing reads per step are 48 issue slots of a single wave.
//
// wgrad_kernel<128,128> is LDS-bound: per 64-pixel step its four 64 x 64 waves read 64 transposed fragments (32 KB each wave) for 32
// MFMAs each, two workgroups per CU - 256 KB of LDS reads and 64 KB of DMA writes per 256 MFMAs, ~1 500 LDS cycles against 1 024
// matrix cycles (its measured step pair: ~3 300 cycles).  Here a workgroup owns 256 co x 128 ci of one tap; four CONSUMER waves
// (one per SIMD) hold 128 co x 64 ci each (8 x 4 accumulator tiles in AGPRs): 48 transposed reads per 64 MFMAs, 96 KB of LDS
// reads per 256 MFMAs - 2.7 times less - and four LOADER waves issue every LDS-DMA (dY tile [64 px][256 co] 32 KB + X tile
// [64 px][128 ci] 16 KB per step, three stages, two steps in flight) and carry the X row offsets in registers: every lane
// follows the four tile rows its DMA instructions touch from step to step by constant deltas and at most one wrap of ow / oh, so
// there is no offset table in LDS and no barrier.  FULL / FREE counters, bounded spins and the step schedule (16 groups of 4 MFMAs,
// fragments requested 8 groups ahead) are conv_ring.hip's.  Products and the order of the sums inside a pixel split are those of
// wgrad_kernel; the number of splits differs (one workgroup per CU here), so the folded gradient agrees with wgrad_kernel's to fp32
// summation order, and with itself bit for bit from run to run (plain slab stores, ordered reduce).
#ifdef EP24_AB_VARIANTS      // the weight gradient as a ring: only in the A/B library (make variants)
namespace wring {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
constexpr int NCW = 4, NLW = 4, NS = 3, LA = 8;
constexpr int TCO = 256, TCI = 128;
constexpr int Y_BYTES = 64 * TCO * 2, X_BYTES = 64 * TCI * 2, STAGE = Y_BYTES + X_BYTES;       // 32 KB + 16 KB
constexpr int YI = 64 * (TCO / 8) / 64 / NLW, XI = 64 * (TCI / 8) / 64 / NLW;                  // DMA instructions per loader and step: 8, 4
constexpr int FM = 8, FN = 4;
constexpr int SPIN_LIMIT = 1 << 14;
constexpr int OOB = 0x7FFFFFF0, XOOB = 0x7FFF0000;

__device__ unsigned g_timeouts;

template <int N> __device__ __forceinline__ void wait_vmcnt_c() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void cbar() { asm volatile("" ::: "memory"); }
// the counters are LDS (address space 3) objects: through generic pointers every look at them was a FLAT load that drained vmcnt and
// lgkmcnt - a loader's poll waited for every DMA it had in flight (conv_ring.hip has the story)
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) v4u lds_v4u;
typedef __attribute__((address_space(3))) void* lds_void_p;
__device__ __forceinline__ void st_flag(lds_u32* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ v4u ld_flags4(const lds_u32* f) { return *reinterpret_cast<const volatile lds_v4u*>(f); }
__device__ __forceinline__ unsigned min4(const lds_u32* f) {
    const v4u v = ld_flags4(f);
    return __builtin_amdgcn_readfirstlane(min(min(v.x, v.y), min(v.z, v.w)));
}
__device__ __forceinline__ void spin_until(const lds_u32* f, unsigned need) {
    int tries = 0;
    while (min4(f) < need) {
        __builtin_amdgcn_s_sleep(1);
        if (++tries > SPIN_LIMIT) {
            if ((threadIdx.x & 63) == 0) atomicAdd(&g_timeouts, 1u);
            break;
        }
    }
    cbar();
}
__device__ __forceinline__ void mfma_acc(f32x4& c, const bf16x8& a, const bf16x8& b) {
    asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// transposed fragment by LDS byte offset (the kernel's only LDS object starts at offset 0)
__device__ __forceinline__ bf16x8 tr_at(int off_lo, int off_hi) {
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(unsigned long)(unsigned)off_lo);
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4_ptr)(unsigned long)(unsigned)off_hi);
    return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
}
// row key of the 32-byte-block permutation for 512-byte rows: the block index has four bits, the key touches the low three (a
// fragment read addresses ONE block column in eight rows: the eight keys spread it over the eight 32-byte slots of a 256-byte bank row)
__device__ __forceinline__ int rkey256(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
__device__ __forceinline__ int tr_off256(int r, int cb, int pp) { return r * (TCO * 2) + ((cb ^ rkey256(r)) << 5) + (pp << 3); }

__global__ __launch_bounds__((NCW + NLW) * 64) void wgrad_ring_kernel(const WgradArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int bx, by;
    {
        const int nwg = gridDim.x, xcd = blockIdx.x & 7, qq = nwg >> 3, rr = nwg & 7;
        const int v = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (blockIdx.x >> 3);
        bx = v % p.tiles;
        by = v / p.tiles;
    }
    const int tap = bx % p.T; bx /= p.T;
    const int ci0 = (bx % p.tiles_ci) * TCI;
    const int co0 = (bx / p.tiles_ci) * TCO;
    const int kh = tap / p.ksize, kw = tap % p.ksize;
    const long p_begin = (long)by * p.chunk;
    const long p_end = min(p.M, p_begin + p.chunk);
    const unsigned S = p_end > p_begin ? (unsigned)((p_end - p_begin + 63) / 64) : 0u;      // steps, uniform over the workgroup
    unsigned* const flags = reinterpret_cast<unsigned*>(smem + NS * STAGE);
    lds_u32* const f_full = (lds_u32*)(lds_void_p)flags;
    lds_u32* const f_free = f_full + 4;
    if (reinterpret_cast<unsigned long>((lptr_t)smem) != 0ul) {
        if (tid == 0) atomicAdd(&g_timeouts, 1u << 16);
        return;
    }
    if (tid < 12) flags[tid] = 0u;
    __syncthreads();

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    auto loader_main = [&]() {
        const int lw = wave - NCW;
        const auto y_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.dy), 0, p.dy_bytes, 0x00020000);
        const auto x_rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(p.x), 0, p.x_bytes, 0x00020000);
        const int p_end_i = (int)p_end;
        // dY: instruction j covers LDS units (lw * YI + j) * 64 + lane: two 512-byte rows, 32 chunks each
        int y_row[YI], y_off[YI];
#pragma unroll
        for (int j = 0; j < YI; ++j) {
            const int U = (lw * YI + j) * 64 + lane;
            const int r = U >> 5, pc = U & 31;
            const int c = ((((pc >> 1) ^ rkey256(r)) << 1) | (pc & 1));
            y_row[j] = r;
            y_off[j] = co0 + c * 8 < p.Cout ? (int)((((p_begin + r) * p.ld_dy) + co0 + c * 8) * 2) : OOB;
        }
        // X: instruction j covers four 256-byte rows; the lane follows ITS row (lw * XI + j) * 4 + lane / 16 through the steps
        int x_colb[XI], t_p[XI], t_ys[XI], t_xs[XI], t_off[XI];
        const int lo_y = p.pad - kh, lo_x = p.pad - kw, wrap_x = p.OW * p.stride, wrap_y = p.OH * p.stride;
#pragma unroll
        for (int j = 0; j < XI; ++j) {
            const int U = (lw * XI + j) * 64 + lane;
            const int r = U >> 4, pc = U & 15;
            const int c = ((((pc >> 1) ^ rkey<TCI>(r)) << 1) | (pc & 1));
            x_colb[j] = ci0 + c * 8 < p.Cin ? (ci0 + c * 8) * 2 : XOOB;
            t_p[j] = (int)p_begin + r;
            const int pd = t_p[j] < p.M ? t_p[j] : 0;
            const int n = fdiv(pd, p.d_plane);
            const int rem = pd - n * (p.OH * p.OW);
            const int oh = fdiv(rem, p.d_ow), ow = rem - oh * p.OW;
            t_ys[j] = oh * p.stride; t_xs[j] = ow * p.stride;
            t_off[j] = n * p.c_n + oh * p.c_oh + ow * p.c_ow + ((kh - p.pad) * p.W + (kw - p.pad)) * p.c_pix;
        }
        const int y_step = (int)(64 * p.ld_dy * 2);
        int st = 0;
        for (unsigned it = 0; it < S; ++it) {
            if (it >= (unsigned)NS) spin_until(f_free, it - (NS - 1));       // stage st was last read in step it - 3
            char* stage = smem + st * STAGE;
            const int rows_left = p_end_i - (int)p_begin - (int)it * 64;
#pragma unroll
            for (int j = 0; j < YI; ++j) {
                const bool ok = y_row[j] < rows_left && y_off[j] != OOB;
                const int vo = ok ? y_off[j] + (int)it * y_step : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(y_rsrc, (lptr_t)(stage + (lw * YI + j) * 1024), 16, vo, 0, 0, 0);
            }
#pragma unroll
            for (int j = 0; j < XI; ++j) {
                const bool ok = t_p[j] < p_end_i && (unsigned)(t_ys[j] - lo_y) < (unsigned)p.H && (unsigned)(t_xs[j] - lo_x) < (unsigned)p.W;
                const int xo = ok ? t_off[j] : XOOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(x_rsrc, (lptr_t)(stage + Y_BYTES + (lw * XI + j) * 1024), 16,
                                                         (int)((unsigned)xo + (unsigned)x_colb[j]), 0, 0, 0);
                t_p[j] += 64; t_ys[j] += p.s_dys; t_xs[j] += p.s_dxs; t_off[j] += p.s_doff;
                if (t_xs[j] >= wrap_x) { t_xs[j] -= wrap_x; t_ys[j] += p.stride; t_off[j] += p.c_oh - p.OW * p.c_ow; }
                if (t_ys[j] >= wrap_y) { t_ys[j] -= wrap_y; t_off[j] += p.c_n - p.OH * p.c_oh; }
            }
            st = st == NS - 1 ? 0 : st + 1;
            if (it > 0) {                                     // two steps in flight: this wait retires step it - 1
                wait_vmcnt_c<YI + XI>();
                if (lane == 0) st_flag(f_full + lw, it);
                cbar();
            }
        }
        wait_vmcnt_c<0>();
        if (lane == 0) st_flag(f_full + lw, S);
        cbar();
    };

    auto consumer_main = [&]() {
        const int cw = wave, wm = cw >> 1, wn = cw & 1;
        const int fq = lane >> 4, q = (lane & 15) >> 2, pp4 = lane & 3;
        // fragment addresses in stage 0, 32-pixel half 0, rows fq * 8 + q (the +4 rows, the other half and the stage are added below:
        // none of them changes the row key)
        int ya[FM], xa[FN];
#pragma unroll
        for (int i = 0; i < FM; ++i) ya[i] = tr_off256(fq * 8 + q, wm * FM + i, pp4);
#pragma unroll
        for (int j = 0; j < FN; ++j) xa[j] = Y_BYTES + tr_off<TCI>(fq * 8 + q, wn * FN + j, pp4);
        auto y_frag = [&](int i, int h, int so) { return tr_at(ya[i] + so + h * (32 * TCO * 2), ya[i] + so + h * (32 * TCO * 2) + 4 * TCO * 2); };
        auto x_frag = [&](int j, int h, int so) { return tr_at(xa[j] + so + h * (32 * TCI * 2), xa[j] + so + h * (32 * TCI * 2) + 4 * TCI * 2); };
        auto mma_row = [&](int i, const bf16x8& fa, const bf16x8 (&fb)[FN]) {
#pragma unroll
            for (int j = 0; j < FN; ++j) mfma_acc(acc[i][j], fa, fb[j]);
        };
        if (S == 0) return;
        spin_until(f_full, 1u);
        bf16x8 pa[LA], fb0[FN], fb1[FN];
#pragma unroll
        for (int g = 0; g < LA; ++g) pa[g] = y_frag(g % FM, g / FM, 0);
#pragma unroll
        for (int j = 0; j < FN; ++j) fb0[j] = x_frag(j, 0, 0);
        unsigned jdone = 0;
        int st = 0;
#pragma unroll 1
        for (unsigned it = 0; it < S; ++it) {
            const int stn = st == NS - 1 ? 0 : st + 1;
            const bool last = it + 1 == S;
            const int so = st * STAGE, son = stn * STAGE;
            bf16x8 F[16];
            v4u fl = {0u, 0u, 0u, 0u};
            constexpr int G_CHECK = 8, G_FLAGS = 4;
            auto group = [&](auto gc) {
                constexpr int g = decltype(gc)::value;
                constexpr int gr = g + LA;
                if constexpr (g == G_CHECK) {
                    if (!last) {
                        const unsigned have = __builtin_amdgcn_readfirstlane(min(min(fl.x, fl.y), min(fl.z, fl.w)));
                        if (have < jdone + 2) spin_until(f_full, jdone + 2);       // this is step jdone (0-based): the next one must have landed
                        cbar();
                    }
                }
                if constexpr (gr < 16) F[gr] = y_frag(gr % FM, gr / FM, so);
                else pa[gr - 16] = y_frag((gr - 16) % FM, (gr - 16) / FM, son);    // behind the last step: bytes nobody uses
                if constexpr (g < 4) fb1[g] = x_frag(g, 1, so);
                if constexpr (g >= G_CHECK && g < G_CHECK + 4) fb0[g - G_CHECK] = x_frag(g - G_CHECK, 0, son);
                if constexpr (g == G_FLAGS) fl = ld_flags4(f_full);
                if constexpr (gr == 15) {
                    cbar();                                   // every fragment read of this step is issued: its stage may be refilled
                    if (lane == 0) st_flag(f_free + cw, jdone + 1);
                    cbar();
                }
                if constexpr (g < LA) mma_row(g % FM, pa[g], g < FM ? fb0 : fb1);
                else mma_row(g % FM, F[g], g < FM ? fb0 : fb1);
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
            ++jdone;
            st = stn;
        }
    };

    const bool consumer = wave < NCW;
    if (consumer) consumer_main(); else loader_main();
    asm volatile("s_nop 15\n\ts_nop 3" ::: "memory");         // inline-asm MFMAs: the wait states in front of the first accumulator read (conv_ring.hip)
    // Epilogue, all eight waves: the accumulators go through LDS ([256 co][128 ci] fp32 = 128 KB over the three stages) and leave
    // as 16-byte stores into this split's slab.
    __syncthreads();
    float* tile = reinterpret_cast<float*>(smem);
    if (consumer) {
        const int wm = wave >> 1, wn = wave & 1, fq = lane >> 4, fr = lane & 15;
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    tile[(wm * (TCO / 2) + i * 16 + 4 * fq + r) * TCI + wn * (TCI / 2) + j * 16 + fr] = acc[i][j][r];
    }
    __syncthreads();
    constexpr int C4 = TCI / 4;
    float* base = p.slab + (long)by * p.slab_stride + (long)tap * p.cin_valid + ci0;
    const bool vec = (p.cin_valid & 3) == 0 && (p.ld_dw & 3) == 0 && (reinterpret_cast<unsigned long long>(base) & 15) == 0;
    for (int u = tid; u < TCO * C4; u += (NCW + NLW) * 64) {
        const int row = u / C4, c4 = u - row * C4;
        const int co = co0 + row;
        if (co >= p.cout_valid) continue;
        if (vec) {
            if (ci0 + c4 * 4 < p.cin_valid)
                *reinterpret_cast<f32x4*>(base + (long)co * p.ld_dw + c4 * 4) = *reinterpret_cast<const f32x4*>(tile + row * TCI + c4 * 4);
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (ci0 + c4 * 4 + e < p.cin_valid) base[(long)co * p.ld_dw + c4 * 4 + e] = tile[row * TCI + c4 * 4 + e];
        }
    }
}

// the ring takes a layer when its slab form has whole float4 groups, Cout >= 256 and the (tile, split) grid fills >= 3/4 of the CUs
bool eligible(const WgradArgs& a, long* splits_out) {
    if (a.Cout < TCO || a.T > 9) return false;
    const int tiles = ep24_cdiv(a.Cin, TCI) * ep24_cdiv(a.Cout, TCO) * a.T;
    const long steps = (a.M + 63) / 64;
    long splits = 256 / tiles;                                // one workgroup per CU (its LDS is the CU's)
    if (splits > steps / 8) splits = steps / 8;
    if (splits < 1) splits = 1;
    if (tiles * splits < 192) return false;
    *splits_out = splits;
    return true;
}

int launch(WgradArgs& a, long splits, hipStream_t stream) {
    a.tiles_ci = ep24_cdiv(a.Cin, TCI); a.tiles_co = ep24_cdiv(a.Cout, TCO);
    a.tiles = a.tiles_ci * a.tiles_co * a.T;
    const long steps = (a.M + 63) / 64;
    a.chunk = ((steps + splits - 1) / splits) * 64;
    a.xcd_remap = 1;
    static std::atomic<unsigned long long> done{0};          // more than 64 KB of dynamic LDS: the attribute, once per device
    int dev = 0;
    EP24_REQUIRE(hipGetDevice(&dev) == hipSuccess, EP24_E_LAUNCH, "wgrad_ring: hipGetDevice failed");
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void*)wgrad_ring_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        EP24_REQUIRE(e == hipSuccess, EP24_E_LAUNCH, "wgrad_ring: hipFuncSetAttribute failed on device %d: %s", dev, hipGetErrorString(e));
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL(wgrad_ring_kernel, dim3((unsigned)(a.tiles * splits)), dim3((NCW + NLW) * 64), NS * STAGE + 64, stream, a);
    return EP24_OK;
}

}  // namespace wring
#endif   // EP24_AB_VARIANTS

// Pixel splits: fill the chip's resident-workgroup slots exactly once (LDS allows 2 / 3 / 4 workgroups per CU for
// the 128x128 / mixed / 64x64 tiles): one slot more than a full round costs a whole extra round (576 workgroups on
// 512 slots ran 1.9x slower than 504), fewer leave CUs idle.  A split keeps at least 8 (16 for the small tile)
// 64-pixel steps so that its prologue and epilogue stay amortised.  tools/conv_probe.py sweeps, round-1 profiles.
template <int TCO, int TCI>
long wgrad_splits(const WgradArgs& a) {
    const int tiles = ep24_cdiv(a.Cin, TCI) * ep24_cdiv(a.Cout, TCO) * a.T;
    const bool small = TCO == 64 && TCI == 64, big = TCO == 128 && TCI == 128;
    const long slots = 256L * (small ? 4 : (big ? 2 : 3));
    // small tile: 12 steps where the layer has several tiles (80x80x128 1x1: 33 -> 27 us with its reduce launch, a workgroup is one
    // memory latency per step and more of them hide it), 16 where it has one (160x160x64: the extra slabs cost the reduce launch
    // more than the lane gains) - tools/wgrad_splits_ab.py, round 3
    const long min_steps = small ? (tiles >= 4 ? 12 : 16) : 8;
    const long steps = (a.M + 63) / 64;
    long splits = slots / tiles;
    if (splits > steps / min_steps) splits = steps / min_steps;
    if (splits < 1) splits = 1;
    return splits;                                          // trailing splits may be empty (they contribute zeros)
}

template <int TCO, int TCI, bool VTAP = false>
void launch_wgrad(WgradArgs& a, hipStream_t stream) {
    a.tiles_ci = ep24_cdiv(a.Cin, TCI); a.tiles_co = ep24_cdiv(a.Cout, TCO);
    const int tiles = a.tiles_ci * a.tiles_co * a.T;
    long steps = (a.M + 63) / 64;
    long splits = wgrad_splits<TCO, TCI>(a);
    a.chunk = ((steps + splits - 1) / splits) * 64;
    a.tiles = tiles;
    a.xcd_remap = 1;
    dim3 grid((unsigned)(tiles * splits));
    hipLaunchKernelGGL((wgrad_kernel<TCO, TCI, VTAP>), grid, dim3(256), 2 * 64 * (TCO + TCI) * 2 + (VTAP ? 2048 : 512), stream, a);
}

// The stem as ONE 64 x 192 tile (round 5): all 12 tap slots of a pixel against one read of its dY row - the three 64 x 64 tiles read
// every dY row three times (672 against 416 bytes per pixel through L2).  64 KB of stages + a 6-KB offset table: two workgroups per
// CU, 512 pixel splits.
constexpr int STEM_TCI = 192, STEM_LDS = 2 * 64 * (64 + STEM_TCI) * 2 + 2 * 64 * (STEM_TCI / 16) * 4;
long stem_splits(const WgradArgs& a) {
    const long steps = (a.M + 63) / 64;
    long splits = 512;
    if (splits > steps / 8) splits = steps / 8;
    return splits < 1 ? 1 : splits;
}
int launch_stem(WgradArgs& a, hipStream_t stream) {
    a.tiles_ci = 1; a.tiles_co = 1; a.tiles = 1;
    const long steps = (a.M + 63) / 64, splits = stem_splits(a);
    a.chunk = ((steps + splits - 1) / splits) * 64;
    a.xcd_remap = 1;
    static std::atomic<unsigned long long> done{0};          // more than 64 KB of dynamic LDS: the attribute, once per device
    int dev = 0;
    EP24_REQUIRE(hipGetDevice(&dev) == hipSuccess, EP24_E_LAUNCH, "stem_conv_wgrad: hipGetDevice failed");
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        const hipError_t e = hipFuncSetAttribute((const void*)wgrad_kernel<64, STEM_TCI, true>, hipFuncAttributeMaxDynamicSharedMemorySize, STEM_LDS);
        EP24_REQUIRE(e == hipSuccess, EP24_E_LAUNCH, "stem_conv_wgrad: hipFuncSetAttribute failed on device %d: %s", dev, hipGetErrorString(e));
        done.fetch_or(bit, std::memory_order_release);
    }
    hipLaunchKernelGGL((wgrad_kernel<64, STEM_TCI, true>), dim3((unsigned)splits), dim3(256), STEM_LDS, stream, a);
    return EP24_OK;
}

int fill_args(WgradArgs& a, const void* x, int64_t ld_x, const void* dy, int64_t ld_dy, float* dw, int64_t ld_dw, int cout_valid,
              int cin_valid, int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    EP24_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && ld_x % 8 == 0 && ld_dy % 8 == 0, EP24_E_ARG,
                 "conv_wgrad: channel counts / strides must be multiples of 8 (Cin=%d Cout=%d)", Cin, Cout);
    EP24_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), EP24_E_UNSUPPORTED,
                 "conv_wgrad: k=%d s=%d unsupported", ksize, stride);
    EP24_REQUIRE(cout_valid <= Cout && cin_valid <= Cin, EP24_E_ARG, "conv_wgrad: valid > padded");
    a.x = (const bf16*)x; a.ld_x = ld_x; a.dy = (const bf16*)dy; a.ld_dy = ld_dy; a.dw = dw; a.ld_dw = ld_dw;
    a.cout_valid = cout_valid; a.cin_valid = cin_valid;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ksize = ksize; a.stride = stride; a.pad = (ksize - 1) / 2;
    a.OH = (H + 2 * a.pad - ksize) / stride + 1; a.OW = (W + 2 * a.pad - ksize) / stride + 1;
    a.M = (long)B * a.OH * a.OW;
    a.T = ksize * ksize;
    EP24_REQUIRE(a.M < (1L << 31), EP24_E_UNSUPPORTED, "conv_wgrad: more than 2^31 output pixels");
    a.d_plane = make_fastdiv((unsigned)(a.OH * a.OW)); a.d_ow = make_fastdiv((unsigned)a.OW);
    a.x_bytes = (unsigned)((((long)B * H * W - 1) * ld_x + Cin) * 2);
    EP24_REQUIRE((((long)B * H * W) * ld_x) * 2 < 0x7FFF0000L, EP24_E_UNSUPPORTED, "conv_wgrad: input larger than 2 GiB");
    a.c_pix = (int)(ld_x * 2); a.c_ow = stride * a.c_pix; a.c_oh = stride * W * a.c_pix; a.c_n = H * W * a.c_pix;
    { const int plane = a.OH * a.OW, dn = 64 / plane, rem = 64 % plane;
      const int doh = rem / a.OW, dow = rem % a.OW;
      a.s_dys = doh * stride; a.s_dxs = dow * stride;
      a.s_doff = dn * a.c_n + doh * a.c_oh + dow * a.c_ow; }
    a.dy_bytes = (unsigned)(((a.M - 1) * ld_dy + Cout) * 2);
#if defined(EP24_STAMPS) || defined(EP24_DIAG_NOLOAD)
    // diagnostic builds only (make stamps / make noload): EP24_WGRAD_NOLOAD=1 makes every tile load out of range (zero fill, same instruction stream) - what
    // the kernel would cost with no fill traffic at all
    if (getenv("EP24_WGRAD_NOLOAD")) a.x_bytes = a.dy_bytes = 0;
#endif
    return EP24_OK;
}

// tile shape: 1x1 layers are tall-skinny (huge pixel count, small dW): 64x64 tiles quarter the epilogue bytes per workgroup
void tile_choice(const WgradArgs& a, bool& co64, bool& ci64) {
    co64 = a.Cout <= 64 || (a.ksize == 1 && a.Cout <= 256 && a.Cin <= 256);
    ci64 = a.Cin <= 64 || (a.ksize == 1 && a.Cout <= 256 && a.Cin <= 256);
}

constexpr int WOPT_RING = 1;               // kernel_opts of the _ex entry points, bit 0: the loader / consumer ring where the layer is eligible (an A/B
                                           // option: 10 % slower than wgrad_kernel on every layer, +0.6 ms per step - profiles/r04_wgrad_ring_ab.txt)

long splits_of(const WgradArgs& a, int opts = 0) {
#ifdef EP24_AB_VARIANTS
    long rs = 0;
    if ((opts & WOPT_RING) && wring::eligible(a, &rs)) return rs;
#else
    (void)opts;
#endif
    bool co64, ci64;
    tile_choice(a, co64, ci64);
    if (co64 && ci64) return wgrad_splits<64, 64>(a);
    if (co64) return wgrad_splits<64, 128>(a);
    if (ci64) return wgrad_splits<128, 64>(a);
    return wgrad_splits<128, 128>(a);
}

int dispatch(WgradArgs& a, hipStream_t stream, int opts = 0) {
#ifdef EP24_AB_VARIANTS
    long rs = 0;
    if (a.slab && (opts & WOPT_RING) && wring::eligible(a, &rs)) return wring::launch(a, rs, stream);
#else
    EP24_REQUIRE(!(opts & WOPT_RING), EP24_E_UNSUPPORTED, "conv_wgrad: kernel_opts bit 0 (the ring form: measured, lost) lives in the A/B library: make variants");
#endif
    bool co64, ci64;
    tile_choice(a, co64, ci64);
    if (co64 && ci64) launch_wgrad<64, 64>(a, stream);
    else if (co64) launch_wgrad<64, 128>(a, stream);
    else if (ci64) launch_wgrad<128, 64>(a, stream);
    else launch_wgrad<128, 128>(a, stream);
    return EP24_OK;
}

// gflat[off + i] += sum_s slab[slab_off + s * numel + i] for a list of layers: desc rows (off, numel, splits, slab_off)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const long* desc, float* grad, const float* slab) {
    const long* d = desc + 4 * blockIdx.y;
    const long off = d[0], numel = d[1], splits = d[2], soff = d[3];
    float* g = grad + off;
    const float* sl = slab + soff;
    if (((off | numel | soff) & 3) == 0 && splits >= 128 && numel <= (1L << 16)) {
        // Few elements, very many splits (the stem: 7 168 weights x 768 splits): the plain form below leaves the sum to a handful of
        // blocks, each thread walking all splits (55 us on the step's critical tail).  Here 16 lanes share an element quad, each
        // sums a contiguous sixteenth of the splits in order, and the sixteen partial sums are added in a fixed order: the same
        // result every run.
        __shared__ f32x4 part[16][16];
        const long n4 = numel >> 2;
        const int e = threadIdx.x & 15, c = threadIdx.x >> 4;
        const long per = (splits + 15) / 16;
        for (long i0 = (long)blockIdx.x * 16; i0 < n4; i0 += (long)gridDim.x * 16) {
            const long i = i0 + e;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            if (i < n4) {
                const long lo = c * per, hi = lo + per < splits ? lo + per : splits;
                for (long s2 = lo; s2 < hi; s2 += 8) {
                    f32x4 v[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = reinterpret_cast<const f32x4*>(sl + (s2 + k < hi ? s2 + k : lo) * numel)[i];
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (s2 + k < hi) { acc[0] += v[k][0]; acc[1] += v[k][1]; acc[2] += v[k][2]; acc[3] += v[k][3]; }
                }
            }
            part[c][e] = acc;
            __syncthreads();
            if (c == 0 && i < n4) {
                f32x4 t = part[0][e];
#pragma unroll
                for (int k = 1; k < 16; ++k) { t[0] += part[k][e][0]; t[1] += part[k][e][1]; t[2] += part[k][e][2]; t[3] += part[k][e][3]; }
                float4 o = reinterpret_cast<float4*>(g)[i];
                o.x += t[0]; o.y += t[1]; o.z += t[2]; o.w += t[3];
                reinterpret_cast<float4*>(g)[i] = o;
            }
            __syncthreads();
        }
    } else if (((off | numel | soff) & 3) == 0) {
        const long n4 = numel >> 2;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
            float4 acc = {0.f, 0.f, 0.f, 0.f};
            for (long s2 = 0; s2 < splits; s2 += 8) {              // 8 loads in flight, summed in split order
                f32x4 v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (s2 + k < splits) v[k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(sl + (s2 + k) * numel) + i);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (s2 + k < splits) { acc.x += v[k].x; acc.y += v[k].y; acc.z += v[k].z; acc.w += v[k].w; }
            }
            float4 o = reinterpret_cast<float4*>(g)[i];
            o.x += acc.x; o.y += acc.y; o.z += acc.z; o.w += acc.w;
            reinterpret_cast<float4*>(g)[i] = o;
        }
    } else {
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < numel; i += (long)gridDim.x * 256) {
            float acc = 0.f;
            for (long s2 = 0; s2 < splits; s2 += 8) {              // 8 loads in flight, summed in split order
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = s2 + k < splits ? sl[(s2 + k) * numel + i] : 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (s2 + k < splits) acc += v[k];
            }
            g[i] += acc;
        }
    }
}

}  // namespace

extern "C" int ep24_conv_wgrad_bf16(const void* x, int64_t ld_x, const void* dy, int64_t ld_dy, float* dw, int64_t ld_dw,
                                    int cout_valid, int cin_valid, int B, int H, int W, int Cin, int Cout, int ksize,
                                    int stride, void* stream) {
    EP24_REQUIRE(x && dy && dw, EP24_E_ARG, "conv_wgrad: null pointer");
    WgradArgs a{};
    if (int rc = fill_args(a, x, ld_x, dy, ld_dy, dw, ld_dw, cout_valid, cin_valid, B, H, W, Cin, Cout, ksize, stride)) return rc;
    if (int rc = dispatch(a, (hipStream_t)stream)) return rc;
    EP24_LAUNCH_CHECK("ep24_conv_wgrad");
    return EP24_OK;
}

extern "C" int ep24_conv_wgrad_splits_ex(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int kernel_opts) {
    WgradArgs a{};
    if (int rc = fill_args(a, nullptr, 8, nullptr, 8, nullptr, 0, Cout, Cin, B, H, W, Cin, Cout, ksize, stride)) return rc;
    return (int)splits_of(a, kernel_opts);
}

extern "C" int ep24_conv_wgrad_splits(int B, int H, int W, int Cin, int Cout, int ksize, int stride) {
    return ep24_conv_wgrad_splits_ex(B, H, W, Cin, Cout, ksize, stride, 0);
}

extern "C" int ep24_conv_wgrad_slab_bf16_ex(const void* x, int64_t ld_x, const void* dy, int64_t ld_dy, float* slab,
                                            int64_t slab_floats, int64_t ld_dw, int cout_valid, int cin_valid, int B, int H,
                                            int W, int Cin, int Cout, int ksize, int stride, int kernel_opts, void* stream) {
    EP24_REQUIRE(x && dy && slab, EP24_E_ARG, "conv_wgrad_slab: null pointer");
    WgradArgs a{};
    if (int rc = fill_args(a, x, ld_x, dy, ld_dy, nullptr, ld_dw, cout_valid, cin_valid, B, H, W, Cin, Cout, ksize, stride)) return rc;
    a.slab = slab;
    a.slab_stride = (long)cout_valid * ld_dw;
    EP24_REQUIRE(splits_of(a, kernel_opts) * a.slab_stride <= slab_floats, EP24_E_ARG, "conv_wgrad_slab: slab holds %ld floats, %ld needed",
                 (long)slab_floats, splits_of(a, kernel_opts) * a.slab_stride);
    if (int rc = dispatch(a, (hipStream_t)stream, kernel_opts)) return rc;
    EP24_LAUNCH_CHECK("ep24_conv_wgrad_slab");
    return EP24_OK;
}

extern "C" int ep24_conv_wgrad_slab_bf16(const void* x, int64_t ld_x, const void* dy, int64_t ld_dy, float* slab,
                                         int64_t slab_floats, int64_t ld_dw, int cout_valid, int cin_valid, int B, int H,
                                         int W, int Cin, int Cout, int ksize, int stride, void* stream) {
    return ep24_conv_wgrad_slab_bf16_ex(x, ld_x, dy, ld_dy, slab, slab_floats, ld_dw, cout_valid, cin_valid, B, H, W, Cin, Cout, ksize, stride, 0, stream);
}

/* Grouped launch (see wgrad_group_kernel).  desc: HOST array [n][17] of int64 -
 * x, ld_x, dy, ld_dy, slab, slab_floats, ld_dw, cout_valid, cin_valid, B, H, W, Cin, Cout, ksize, stride, splits. */
template <int TCO, int TCI>
static int launch_group(const int64_t* desc, int n, hipStream_t stream) {
    WgradGroup g{};
    g.n = n;
    for (int i = 0; i < n; ++i) {
        const int64_t* d = desc + 17 * i;
        WgradArgs& a = g.a[i];
        EP24_REQUIRE(d[0] && d[2] && d[4], EP24_E_ARG, "conv_wgrad_group: null pointer in problem %d", i);
        if (int rc = fill_args(a, (const void*)d[0], d[1], (const void*)d[2], d[3], nullptr, d[6], (int)d[7], (int)d[8], (int)d[9], (int)d[10], (int)d[11],
                               (int)d[12], (int)d[13], (int)d[14], (int)d[15])) return rc;
        bool co64, ci64;
        tile_choice(a, co64, ci64);
        EP24_REQUIRE((co64 ? 64 : 128) == TCO && (ci64 ? 64 : 128) == TCI, EP24_E_ARG, "conv_wgrad_group: problem %d is not of the group's tile class", i);
        a.slab = (float*)d[4];
        a.slab_stride = (long)a.cout_valid * a.ld_dw;
        const long splits = d[16], steps = (a.M + 63) / 64;
        EP24_REQUIRE(splits >= 1 && splits * a.slab_stride <= d[5], EP24_E_ARG, "conv_wgrad_group: problem %d: %ld splits need %ld slab floats, %ld given", i,
                     splits, splits * a.slab_stride, (long)d[5]);
        a.tiles_ci = ep24_cdiv(a.Cin, TCI); a.tiles_co = ep24_cdiv(a.Cout, TCO);
        a.tiles = a.tiles_ci * a.tiles_co * a.T;
        a.chunk = ((steps + splits - 1) / splits) * 64;
        a.xcd_remap = 1;
        g.prefix[i + 1] = g.prefix[i] + (int)(a.tiles * splits);
    }
    hipLaunchKernelGGL((wgrad_group_kernel<TCO, TCI>), dim3((unsigned)g.prefix[n]), dim3(256), 2 * 64 * (TCO + TCI) * 2 + 512, stream, g);
    return EP24_OK;
}

extern "C" int ep24_conv_wgrad_tile_class(int Cin, int Cout, int ksize) {
    WgradArgs a{};
    a.Cin = Cin; a.Cout = Cout; a.ksize = ksize;
    bool co64, ci64;
    tile_choice(a, co64, ci64);
    return (co64 ? 1 : 0) | (ci64 ? 2 : 0);
}

extern "C" int ep24_conv_wgrad_group_bf16(const int64_t* desc, int n, void* stream) {
    EP24_REQUIRE(desc && n >= 1 && n <= WG_MAX, EP24_E_ARG, "conv_wgrad_group: 1 .. %d problems per launch", WG_MAX);
    const int cls = ep24_conv_wgrad_tile_class((int)desc[12], (int)desc[13], (int)desc[14]);
    int rc;
    if (cls == 3) rc = launch_group<64, 64>(desc, n, (hipStream_t)stream);
    else if (cls == 1) rc = launch_group<64, 128>(desc, n, (hipStream_t)stream);
    else if (cls == 2) rc = launch_group<128, 64>(desc, n, (hipStream_t)stream);
    else rc = launch_group<128, 128>(desc, n, (hipStream_t)stream);
    if (rc) return rc;
    EP24_LAUNCH_CHECK("ep24_conv_wgrad_group");
    return EP24_OK;
}

// bounded waits of the ring kernels of this translation unit that gave up (added to ep24_conv_ring_timeouts by conv_ring.hip)
namespace ep24_igemm {
int wgrad_ring_timeouts() {
#ifdef EP24_AB_VARIANTS
    unsigned v = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(wring::g_timeouts), sizeof(v)) != hipSuccess) return -1;
    return (int)v;
#else
    return 0;
#endif
}
}

// Weight gradient of the Focus stem straight from the space-to-depth image (VTAP above): slab[s][Cout][108], column tap * 12 + channel
static int stem_wgrad_args(WgradArgs& a, const void* f16, const void* dy, int64_t ld_dy, int B, int FH, int FW, int Cout) {
    EP24_REQUIRE(Cout % 8 == 0 && Cout > 0 && Cout <= 64, EP24_E_UNSUPPORTED, "stem_conv_wgrad: Cout=%d (a multiple of 8, at most 64)", Cout);
    // one virtual channel tile of 192 = 12 tap slots x 16 channels; 1x1 addressing over the 32-byte pixels
    if (int rc = fill_args(a, f16, 16, dy, ld_dy, nullptr, 108, Cout, 108, B, FH, FW, 192, Cout, 1, 1)) return rc;
    a.x_bytes = (unsigned)((long)B * FH * FW * 32);
    return EP24_OK;
}

extern "C" int ep24_stem_conv_wgrad_splits(int B, int FH, int FW, int Cout) {
    WgradArgs a{};
    if (int rc = stem_wgrad_args(a, nullptr, nullptr, (Cout + 7) / 8 * 8, B, FH, FW, Cout)) return rc;
    return (int)stem_splits(a);
}

extern "C" int ep24_stem_conv_wgrad_slab_bf16(const void* f16, const void* dy, int64_t ld_dy, float* slab, int64_t slab_floats, int B,
                                              int FH, int FW, int Cout, void* stream) {
    EP24_REQUIRE(f16 && dy && slab, EP24_E_ARG, "stem_conv_wgrad_slab: null pointer");
    WgradArgs a{};
    if (int rc = stem_wgrad_args(a, f16, dy, ld_dy, B, FH, FW, Cout)) return rc;
    a.slab = slab;
    a.slab_stride = (long)Cout * 108;
    const long need = stem_splits(a) * a.slab_stride;
    EP24_REQUIRE(need <= slab_floats, EP24_E_ARG, "stem_conv_wgrad_slab: slab holds %ld floats, %ld needed", (long)slab_floats, need);
    if (int rc = launch_stem(a, (hipStream_t)stream)) return rc;
    EP24_LAUNCH_CHECK("ep24_stem_conv_wgrad_slab");
    return EP24_OK;
}

extern "C" int ep24_wgrad_reduce(const int64_t* desc, int n_layers, int64_t max_numel, float* grad, const float* slab, void* stream) {
    EP24_REQUIRE(desc && grad && slab && n_layers > 0 && max_numel > 0, EP24_E_ARG, "wgrad_reduce: bad arguments");
    long bx = (max_numel / 4 + 255) / 256;
    bx = bx < 128 ? 128 : (bx > 512 ? 512 : bx);              // at least 128: the many-splits form of a small layer spreads over them
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)bx, (unsigned)n_layers), dim3(256), 0, (hipStream_t)stream, (const long*)desc,
                       grad, slab);
    EP24_LAUNCH_CHECK("ep24_wgrad_reduce");
    return EP24_OK;
}
