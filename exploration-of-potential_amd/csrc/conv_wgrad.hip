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
#include "common.h"

namespace {

struct WgradArgs {
    const bf16* x; long ld_x; const bf16* dy; long ld_dy; float* dw; long ld_dw;
    int cout_valid, cin_valid;
    int B, H, W, OH, OW, Cin, Cout, ksize, stride, pad;
    long M;          // B*OH*OW
    long chunk;      // pixels per split (multiple of 64)
    int tiles_ci, tiles_co, T;
    FastDiv d_plane, d_ow;      // pixel -> (n, oh, ow)
};

// Row keys of the 32-byte-block XOR permutation.  256-B rows (128 channels): one row per bank row, the 8 rows a
// half-wave touches in one transposed read get 8 distinct keys.  128-B rows (64 channels): two rows per bank
// row, rows of equal parity get 4 distinct keys.
template <int TW> __device__ __forceinline__ int rkey(int row);
template <> __device__ __forceinline__ int rkey<128>(int row) { return (row & 3) | (((row >> 3) & 1) << 2); }
template <> __device__ __forceinline__ int rkey<64>(int row) { return ((row >> 1) & 1) | (((row >> 3) & 1) << 1); }
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

// TCO x TCI output tile of one tap per workgroup (128 or 64 each), 4 waves as 2x2.
template <int TCO, int TCI>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradArgs p) {
    constexpr int Y_BYTES = 64 * TCO * 2, X_BYTES = 64 * TCI * 2;
    constexpr int FM = TCO / 32, FN = TCI / 32;          // 16x16 fragments per wave along co / ci
    constexpr int YCH = TCO / 8, XCH = TCI / 8;          // 16-B chunks per tile row
    constexpr int YP = YCH / 4, XP = XCH / 4;            // pieces per thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const lds_y = smem;                            // one stage: <= 32 KB, 4 workgroups per CU
    char* const lds_x = smem + Y_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x;
    const int tap = bx % p.T; bx /= p.T;
    const int ci0 = (bx % p.tiles_ci) * TCI;
    const int co0 = (bx / p.tiles_ci) * TCO;
    const int kh = tap / p.ksize, kw = tap % p.ksize;
    const long p_begin = (long)blockIdx.y * p.chunk;
    const long p_end = min(p.M, p_begin + p.chunk);
    const int n_iter = (int)((p_end - p_begin + 63) / 64);      // uniform over the workgroup

    const int yc = tid % YCH, yr = tid / YCH;            // dY pieces: rows yr + (256/YCH) i
    const int xc = tid % XCH, xr = tid / XCH;
    const bool y_col_ok = co0 + yc * 8 < p.Cout;
    const bool x_col_ok = ci0 + xc * 8 < p.Cin;
    bf16x8 ry[YP], rx[XP];

    auto load_tile = [&](int it) {
        const long base = p_begin + (long)it * 64;
#pragma unroll
        for (int i = 0; i < YP; ++i) {
            const long pp = base + yr + (256 / YCH) * i;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pp < p_end && y_col_ok) v = *reinterpret_cast<const bf16x8*>(p.dy + pp * p.ld_dy + co0 + yc * 8);
            ry[i] = v;
        }
#pragma unroll
        for (int i = 0; i < XP; ++i) {
            const long pp = base + xr + (256 / XCH) * i;
            bf16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
            if (pp < p_end && x_col_ok) {
                const int n = fdiv((int)pp, p.d_plane);
                const int rem = (int)pp - n * (p.OH * p.OW);
                const int oh = fdiv(rem, p.d_ow), ow = rem - oh * p.OW;
                const int iy = oh * p.stride + kh - p.pad, ix = ow * p.stride + kw - p.pad;
                if (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                    v = *reinterpret_cast<const bf16x8*>(p.x + ((long)(n * p.H + iy) * p.W + ix) * p.ld_x + ci0 + xc * 8);
            }
            rx[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < YP; ++i) *reinterpret_cast<bf16x8*>(lds_y + wr_off<TCO>(yr + (256 / YCH) * i, yc)) = ry[i];
#pragma unroll
        for (int i = 0; i < XP; ++i) *reinterpret_cast<bf16x8*>(lds_x + wr_off<TCI>(xr + (256 / XCH) * i, xc)) = rx[i];
    };

    f32x4 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int fq = lane >> 4;            // k group: pixels 8*fq .. 8*fq+7 of a 32-pixel step
    const int q = (lane & 15) >> 2;      // row inside the 4-row transposed block this lane addresses
    const int pp4 = lane & 3;            // 4-channel sub-slot this lane addresses

    if (n_iter > 0) load_tile(0);
    for (int it = 0; it < n_iter; ++it) {
        if (it) __syncthreads();
        store_tile();
        __syncthreads();
        if (it + 1 < n_iter) load_tile(it + 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int row_lo = ks * 32 + fq * 8 + q;
            bf16x8 fa[FM], fb[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) fa[i] = tr_frag<TCO>(lds_y, row_lo, wm * FM + i, pp4);
#pragma unroll
            for (int j = 0; j < FN; ++j) fb[j] = tr_frag<TCI>(lds_x, row_lo, wn * FN + j, pp4);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    // D[row = co][col = ci]: row = 4*fq + r, col = lane & 15
    const int fr = lane & 15;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int co = co0 + wm * (TCO / 2) + i * 16 + 4 * fq + r;
            if (co >= p.cout_valid) continue;
#pragma unroll
            for (int j = 0; j < FN; ++j) {
                const int ci = ci0 + wn * (TCI / 2) + j * 16 + fr;
                if (ci < p.cin_valid) atomicAdd(p.dw + (long)co * p.ld_dw + (long)tap * p.cin_valid + ci, acc[i][j][r]);
            }
        }
}

template <int TCO, int TCI>
void launch_wgrad(WgradArgs& a, hipStream_t stream) {
    a.tiles_ci = ep24_cdiv(a.Cin, TCI); a.tiles_co = ep24_cdiv(a.Cout, TCO);
    const int tiles = a.tiles_ci * a.tiles_co * a.T;
    // ~3-4 workgroups per CU, at least 8 K-steps (512 pixels) per split so the fp32 atomic epilogue amortises
    long steps = (a.M + 63) / 64;
    long splits = (896 + tiles - 1) / tiles;
    if (splits > steps / 8) splits = steps / 8;
    if (splits < 1) splits = 1;
    a.chunk = ((steps + splits - 1) / splits) * 64;
    splits = (a.M + a.chunk - 1) / a.chunk;
    dim3 grid(tiles, (unsigned)splits);
    hipLaunchKernelGGL((wgrad_kernel<TCO, TCI>), grid, dim3(256), 64 * (TCO + TCI) * 2, stream, a);
}

}  // namespace

extern "C" int ep24_conv_wgrad_bf16(const void* x, int64_t ld_x, const void* dy, int64_t ld_dy, float* dw, int64_t ld_dw,
                                    int cout_valid, int cin_valid, int B, int H, int W, int Cin, int Cout, int ksize,
                                    int stride, void* stream) {
    EP24_REQUIRE(x && dy && dw, EP24_E_ARG, "conv_wgrad: null pointer");
    EP24_REQUIRE(Cin % 8 == 0 && Cout % 8 == 0 && ld_x % 8 == 0 && ld_dy % 8 == 0, EP24_E_ARG,
                 "conv_wgrad: channel counts / strides must be multiples of 8 (Cin=%d Cout=%d)", Cin, Cout);
    EP24_REQUIRE((ksize == 1 || ksize == 3) && (stride == 1 || stride == 2), EP24_E_UNSUPPORTED,
                 "conv_wgrad: k=%d s=%d unsupported", ksize, stride);
    EP24_REQUIRE(cout_valid <= Cout && cin_valid <= Cin, EP24_E_ARG, "conv_wgrad: valid > padded");
    WgradArgs a{};
    a.x = (const bf16*)x; a.ld_x = ld_x; a.dy = (const bf16*)dy; a.ld_dy = ld_dy; a.dw = dw; a.ld_dw = ld_dw;
    a.cout_valid = cout_valid; a.cin_valid = cin_valid;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.ksize = ksize; a.stride = stride; a.pad = (ksize - 1) / 2;
    a.OH = (H + 2 * a.pad - ksize) / stride + 1; a.OW = (W + 2 * a.pad - ksize) / stride + 1;
    a.M = (long)B * a.OH * a.OW;
    a.T = ksize * ksize;
    EP24_REQUIRE(a.M < (1L << 31), EP24_E_UNSUPPORTED, "conv_wgrad: more than 2^31 output pixels");
    a.d_plane = make_fastdiv((unsigned)(a.OH * a.OW)); a.d_ow = make_fastdiv((unsigned)a.OW);
    const bool co64 = Cout <= 64, ci64 = Cin <= 64;
    if (co64 && ci64) launch_wgrad<64, 64>(a, (hipStream_t)stream);
    else if (co64) launch_wgrad<64, 128>(a, (hipStream_t)stream);
    else if (ci64) launch_wgrad<128, 64>(a, (hipStream_t)stream);
    else launch_wgrad<128, 128>(a, (hipStream_t)stream);
    EP24_LAUNCH_CHECK("ep24_conv_wgrad");
    return EP24_OK;
}
