/* ep24 - C ABI of the MI355X-native YOLOX-24p training path (libep24.so, gfx950 only).
 *
 * The reference (IN2-ViAUn/Exploration-of-Potential, /root/reference) has NO native/FFI boundary on this
 * path: it is plain Python over torch ops.  The drop-in surface is therefore the reference's Python API
 * (exploration-of-potential_amd/yolox_24p mirrors it) and this C ABI sits one level below it: every entry
 * point replaces the torch-op sequence of one row of SURVEY.md section 8(a), cited per function as
 * file:line under /root/reference.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host; the caller owns every buffer
 *   - `stream` is a hipStream_t (passed as void*); nothing here allocates, frees or synchronises
 *   - activations are NHWC bf16 with an explicit row stride `ld_*` (elements between consecutive pixels),
 *     so a tensor may be a channel slice of a wider (concat) buffer
 *   - conv weights are [Cout][KH*KW][Cin] ("KRSC"); fp32 masters, bf16 packed copies
 *   - return 0 on success, a negative EP24_E_* otherwise; text via ep24_last_error()
 */
#ifndef EP24_H
#define EP24_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EP24_OK 0
#define EP24_E_ARG (-1)          /* bad shape / alignment / null pointer (reference raises IndexError/assert) */
#define EP24_E_LAUNCH (-2)       /* HIP launch error */
#define EP24_E_UNSUPPORTED (-3)

#define EP24_MAX_GT 50           /* max_labels, yolox_24p/datasets/data_augment.py:131 */
#define EP24_RAYS 24
#define EP24_LABEL_COLS 51
#define EP24_NUM_SUMS 32         /* per-step loss accumulators, see ep24_loss_* */

const char* ep24_last_error(void);
/* Bumped whenever an exported signature or the meaning of an argument changes.  2 (round 4): ep24_bn_act_bwd_reduce / _apply /
 * _apply_acc and ep24_pack_weights_batched took new arguments in round 3, ep24_conv_set_patch went away, the BatchNorm-backward sums
 * became 2^-36 fixed point, ep24_circle_lens is new.  A caller built against another version must not call in. */
#define EP24_ABI_VERSION 3
int ep24_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * a1  conv + BN + SiLU blocks  (yolox_24p/models/network_blocks.py:29-54 BaseConv.forward and its autograd)
 * ------------------------------------------------------------------------------------------------ */

/* y[B,OH,OW,Cout] = conv(x[B,H,W,Cin], w[Cout][k*k][Cin]), pad (k-1)/2, bf16 MFMA implicit GEMM, fp32
 * accumulate.  y is bf16 (y_f32=0) or fp32 (y_f32=1, + bias) with row stride ld_y.  If `stats` is non-null
 * the per-channel sum and sum-of-squares of the fp32 results are atomically added, as 2^-20 fixed point
 * int64 (order independent => bitwise reproducible), into stats[replica][0][c] / stats[replica][1][c]
 * (replica = block % stats_replicas): the BN batch statistics (torch.nn.BatchNorm2d training mode,
 * network_blocks.py:47).  Requires Cin % 8 == 0.
 * Output pixel (n,oh,ow) lands on row n*y_batch_rows + y_row0 + oh*OW + ow of y (y_batch_rows = 0 means
 * dense OH*OW): lets the head write straight into its slice of the [B,8400,107] tensor. */
int ep24_conv_fwd_bf16(const void* x, int64_t ld_x, const void* w, void* y, int64_t ld_y, int y_f32,
                       int64_t y_batch_rows, int64_t y_row0, const float* bias, int64_t* stats, int stats_replicas,
                       int B, int H, int W, int Cin, int Cout, int ksize, int stride, void* stream);

/* The same two entry points with an explicit kernel choice PER CALL (A/B timing, tests that compare two kernels on one shape;
 * there is no process-wide switch).  kernel_opts bit 0: the 3x3 stride-1 layers run in the generic tiled kernel instead of the
 * halo-patch kernels; bit 1: the tiled kernels store their output 8 bytes per lane instead of staging it for 16-byte stores;
 * bit 2: a stride-2 input gradient runs as one launch per parity class (four) instead of one merged launch; bit 3: the 3x3
 * stride-1 layers run in the 8-wave lockstep halo-patch kernel (csrc/conv_patch.hip) instead of the loader / consumer ring
 * (csrc/conv_ring.hip, the default since round 4); bit 4: layers of the tiled kernel that fill the chip with 256 x 128 tiles run in
 * the ring without a patch (measured slower on every layer of YOLOX-l at B = 20: off by default); bit 5: the ring's consumers
 * multiply with v_mfma_f32_32x32x16_bf16 instead of v_mfma_f32_16x16x32_bf16 (same products; the fp32 sum of a 64-channel chunk in
 * four steps of 16 instead of two of 32, so results differ from the other kernels in the last bit; fewer cycles, lower clock: an
 * A/B option); bit 6: the ring with its NARROW tile (256 pixels x 64 channels, four consumers of 64 x 64, a weight-ring stage per
 * tap and loaders five steps ahead) for every 3x3 stride-1 layer with at least 128 such tiles - also the 20 x 20 level and N = 64,
 * which the 256 x 128 ring does not take (bit-identical results; measured slower than the tiled kernel it would replace: an A/B
 * option); bit 7: the tiled kernel without its three-stage form (round 4: layers whose 128-wide tiles leave at most one workgroup per
 * CU - the 20 x 20 level at B = 20 - run in 128-wide tiles with three LDS stages and two tiles in flight; with bit 7 they run in 64-wide
 * two-stage tiles as before; bit-identical results); bit 8: 1x1 stride-1 layers with 128 < K <= 256 and fewer than 100 000 pixels run
 * in the tiled kernel, as they did before the streaming kernel's weight tile was requested in one batch (round 5: the streaming kernel
 * is the default for every 1x1 stride-1 layer with K <= 256 now); bit 9: 3x3 stride-1 layers with at most 64 channels on either side
 * run in the tiled kernel instead of conv_wreg_kernel (round 5, csrc/conv_wreg.hip: every weight fragment of the layer in registers,
 * persistent workgroups, one LDS window per tap row; bit-identical outputs, the batch statistics equal to fp32 summation order; the
 * default for such layers with at least 65 536 pixels, an image at least 32 wide and a plain first-writer destination).  kernel_opts = 0 is exactly ep24_conv_fwd_bf16 / ep24_conv_dgrad_bf16, and every default kernel of a 3x3 stride-1
 * layer (ring, 8-wave halo patch, tiled) gives bit-identical results. */
/* Round 5: bits 4, 5 and 6 (and bit 0 of the weight gradient's kernel_opts) select variants that were built, measured and lost in
 * rounds 3 - 4; they left the product library, which answers EP24_E_UNSUPPORTED for them.  `make -C csrc variants` builds
 * libep24_variants.so with them (EP24_LIB selects it); ep24_ab_variants() says which kind of library is loaded. */
int ep24_ab_variants(void);
int ep24_conv_fwd_bf16_ex(const void* x, int64_t ld_x, const void* w, void* y, int64_t ld_y, int y_f32,
                          int64_t y_batch_rows, int64_t y_row0, const float* bias, int64_t* stats, int stats_replicas,
                          int B, int H, int W, int Cin, int Cout, int ksize, int stride, int kernel_opts, void* stream);
int ep24_conv_dgrad_bf16_ex(const void* dy, int64_t ld_dy, const void* wt, void* dx, int64_t ld_dx, int accumulate,
                            int B, int H, int W, int Cin, int Cout_k, int ksize, int stride, int kernel_opts, void* stream);

/* The 3x3 stride-1 layers run as a loader / consumer ring (csrc/conv_ring.hip: 4 MFMA waves fed by 4 LDS-DMA waves through counters
 * in LDS).  Every wait on a counter is a bounded spin; this returns how many of them have given up since the library was loaded
 * (reads a device word: synchronises with the device; the weight-gradient ring of csrc/conv_wgrad.hip counts in).  Always 0 unless
 * the hand-off protocol is broken - tests assert it. */
int ep24_conv_ring_timeouts(void);

/* Input gradient of a stride-1 conv that is the ONLY consumer of the conv-BN-act unit below it (a Bottleneck's 3x3 over its 1x1,
 * network_blocks.py:54-79): dx IS that unit's dy, so the epilogue also takes the unit's two BatchNorm-backward sums - what
 * ep24_bn_act_bwd_reduce would compute from dx and z - from the tile on its way out.  z [B*H*W][ld_z]: the unit's conv output;
 * mean / invstd / gamma / beta [Cin]; dgamma / dbeta: fixed-point sums, replica r at + r * rep_stride.  First writer only (dx is
 * overwritten); Cin, ld_dx, ld_z multiples of 8. */
int ep24_conv_dgrad_bnr_bf16(const void* dy, int64_t ld_dy, const void* wt, void* dx, int64_t ld_dx, int B, int H, int W, int Cin,
                             int Cout_k, int ksize, const void* z, int64_t ld_z, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, int64_t* dgamma, int64_t* dbeta, int64_t rep_stride, int reps,
                             int act, void* stream);

/* Which device kernel ep24_conv_fwd_bf16 (dgrad = 0) / ep24_conv_dgrad_bf16 (dgrad = 1, stride 1) launches for a shape - the
 * library's own dispatch rule, for reports (bench.py attributes launch times to kernels with it).  Returns 0 =
 * igemm_dma_kernel (generic tiled), 1 = conv_patch_kernel (halo patch, 8 waves in lockstep), 2 = igemm_stream_kernel (1x1
 * streaming), 3 = conv_ring_kernel (halo patch as a loader / consumer ring), 4 = conv_ring_generic_kernel (the ring without a
 * patch; only with kernel_opts bit 4), 5 = conv_ring_kernel with the narrow tile (only with kernel_opts bit 6), 6 = conv_wreg_kernel
 * (3x3 stride-1, at most 64 channels on either side: weights in registers), < 0 on error.
 * y_f32 / has_bias as in ep24_conv_fwd_bf16 (both 0 for dgrad). */
int ep24_conv_kernel_for(int dgrad, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int y_f32, int has_bias);
/* ... and for the _ex entry points with the given kernel_opts. */
int ep24_conv_kernel_for_ex(int dgrad, int B, int H, int W, int Cin, int Cout, int ksize, int stride, int y_f32, int has_bias,
                            int kernel_opts);

/* dx[B,H,W,Cin] (+)= conv_transpose(dy[B,OH,OW,Cout_k], wt[Cin][k*k][Cout_k]); Cout_k % 8 == 0 (zero padded).
 * wt is the pure transpose of w (no tap flip).  Replaces autograd's conv input gradient. */
int ep24_conv_dgrad_bf16(const void* dy, int64_t ld_dy, const void* wt, void* dx, int64_t ld_dx, int accumulate,
                         int B, int H, int W, int Cin, int Cout_k, int ksize, int stride, void* stream);

/* dw[co][t][ci] += sum_pixels dy[.,co] * x[.@t,ci]   fp32, row stride ld_dw between co rows (= taps*cin_valid
 * when dense), only co < cout_valid and ci < cin_valid are written.  Split over pixels with fp32 atomics.
 * Replaces autograd's conv weight gradient. */
int ep24_conv_wgrad_bf16(const void* x, int64_t ld_x, const void* dy, int64_t ld_dy, float* dw, int64_t ld_dw,
                         int cout_valid, int cin_valid,
                         int B, int H, int W, int Cin, int Cout, int ksize, int stride, void* stream);

/* The same weight gradient without atomics (bitwise reproducible): pixel split s of the launch stores its partial
 * dW (layout of dw above, cout_valid * ld_dw floats) at slab + s * cout_valid * ld_dw with plain stores; every
 * element of every split is written.  ep24_conv_wgrad_splits returns the number of splits that launch uses (>= 1, a
 * pure function of the shape), so the caller can size the slab; ep24_wgrad_reduce then does, for n_layers rows
 * desc[i] = (grad offset, numel, splits, slab offset) (all in floats), grad[off + j] += sum_s slab[soff + s*numel + j]
 * in a fixed order.  Together they replace autograd's conv weight gradient for the training engine. */
int ep24_conv_wgrad_splits(int B, int H, int W, int Cin, int Cout, int ksize, int stride);
int ep24_conv_wgrad_slab_bf16(const void* x, int64_t ld_x, const void* dy, int64_t ld_dy, float* slab, int64_t slab_floats,
                              int64_t ld_dw, int cout_valid, int cin_valid,
                              int B, int H, int W, int Cin, int Cout, int ksize, int stride, void* stream);
/* The same two with an explicit kernel choice per call (A/B timing, tests): kernel_opts bit 0 = the loader / consumer ring form
 * (layers with Cout >= 256 whose (tile, split) grid fills the chip; csrc/conv_wgrad.hip, round 4: measured 10 % slower than
 * wgrad_kernel, off by default).  The split count - and with it the slab size - depends on the kernel: ask with the same options. */
int ep24_conv_wgrad_splits_ex(int B, int H, int W, int Cin, int Cout, int ksize, int stride, int kernel_opts);
int ep24_conv_wgrad_slab_bf16_ex(const void* x, int64_t ld_x, const void* dy, int64_t ld_dy, float* slab, int64_t slab_floats,
                                 int64_t ld_dw, int cout_valid, int cin_valid, int B, int H, int W, int Cin, int Cout, int ksize,
                                 int stride, int kernel_opts, void* stream);
/* Round 5: up to 16 weight gradients of ONE tile class in one launch (a weight gradient is needed by nobody before the optimizer and
 * every layer keeps its dz and its input, so the layers of a backward segment wait for each other; grouped they fill the chip without
 * being cut into many short pixel splits).  desc is a HOST array [n][17] of int64 - x, ld_x, dy, ld_dy, slab, slab_floats, ld_dw,
 * cout_valid, cin_valid, B, H, W, Cin, Cout, ksize, stride, splits - read during the call; `splits` (>= 1) is the caller's: slab s
 * of problem i receives the partial sums of its s-th run of ceil(steps / splits) 64-pixel steps, ep24_wgrad_reduce folds them.
 * ep24_conv_wgrad_tile_class: bit 0 = 64-wide tiles over Cout (else 128), bit 1 = 64-wide tiles over Cin; one class per launch. */
int ep24_conv_wgrad_tile_class(int Cin, int Cout, int ksize);
int ep24_conv_wgrad_group_bf16(const int64_t* desc, int n, void* stream);
int ep24_wgrad_reduce(const int64_t* desc, int n_layers, int64_t max_numel, float* grad, const float* slab, void* stream);

/* fp32 master [Cout][T][Cin] (row stride ld_w) -> bf16 w_fwd [Cout][T][Cin_pad] and bf16 w_dgrad
 * [Cin][T][Cout_pad] (either may be null).  Only real elements are written; the caller zero-initialises the
 * padded buffers once. */
int ep24_pack_weights(const float* w, int64_t ld_w, void* w_fwd, void* w_dgrad, int Cout, int T, int Cin,
                      int Cin_pad, int Cout_pad, void* stream);

/* The same for every conv segment of a model in one launch.  desc [n_seg][8] int64 = {master offset, w_fwd offset,
 * w_dgrad offset or -1, Cout, T, Cin, Cin_pad, Cout_pad} (element offsets into flat / w_fwd / w_dgrad); prefix
 * [n_seg+1] int64 = running sum of Cout*T*Cin (total = prefix[n_seg]); tile_prefix [n_seg+1] = running sum of
 * T*ceil(Cout/64)*ceil(Cin/64), the 64x64 tiles of the LDS transpose that writes w_dgrad coalesced.
 * chunk_seg [ceil(total/4096)] int32 = segment of element 4096*c and tile_seg [total_tiles] int32 = segment of tile t are
 * optional lookup tables built once by the host (NULL: the kernels bisect the prefix tables, eight dependent loads per chunk).
 * which: 0 both copies, 1 w_fwd only, 2 w_dgrad only (backward is the first to read it). */
int ep24_pack_weights_batched(const float* flat, const int64_t* desc, const int64_t* prefix, const int64_t* tile_prefix,
                              int n_seg, void* w_fwd, void* w_dgrad, int64_t total, int64_t total_tiles,
                              const int32_t* chunk_seg, const int32_t* tile_seg, int which, void* stream);

/* y = silu(bn(z)) (+ residual), training-mode BatchNorm with batch statistics taken from `stats`
 * ([replicas][2][C] fixed-point sums over the M rows, as written by ep24_conv_fwd_bf16).  Also writes save[0][c]=mean,
 * save[1][c]=invstd for the backward and updates running_mean / running_var (unbiased) / num_batches_tracked
 * (momentum, eps: yolox_24p/exp/yolox_base.py:58-62).  act: 1 = SiLU, 0 = identity. */
int ep24_bn_act_fwd(const void* z, int64_t ld_z, const int64_t* stats, int stats_replicas, const float* gamma,
                    const float* beta, float* running_mean, float* running_var, int64_t* num_batches, int64_t* num_batches2,
                    float* save, void* y, int64_t ld_y, const void* residual, int64_t ld_res,
                    int64_t M, int C, float eps, float momentum, int act, void* stream);   /* num_batches2: the counter of the
                    second BatchNorm module when two units share the launch (merged CSP / head pairs), else NULL */

/* pass 1 of the backward: dgamma[c] += sum du*zhat, dbeta[c] += sum du, du = dy * silu'(bn(z)); the sums are
 * 2^-36 fixed-point int64 (NOT the forward statistics' 2^-20: gradient sums of the head are 1e-5 .. 1e-4 per workgroup; a workgroup's
 * partial sum must stay below 2^20 in magnitude - one that does not, or is NaN, makes the channel's folded sum NaN), kept in `reps` replicas (replica r of either sum 2*C*r elements behind
 * the pointer, i.e. [reps][2][C] when dbeta = dgamma + C; a workgroup adds to replica blockIdx % reps): the memory-side atomic
 * units serialise the adds to one address, and with a single copy every workgroup of the launch hit the same 2 C addresses. */
int ep24_bn_act_bwd_reduce(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save,
                           const float* gamma, const float* beta, int64_t* dgamma, int64_t* dbeta,
                           int64_t M, int C, int act, int reps, void* stream);
/* pass 2: dz = gamma*invstd*(du - dbeta/M - zhat*dgamma/M)  -> bf16 [M,C] (ld_dz).  dgamma/dbeta are this
 * call's sums (scratch, zeroed by the caller before pass 1); if gamma_grad/beta_grad are non-null they
 * receive += of those sums (the parameter .grad accumulators).  The `reps` replicas of the sums (layout as in pass 1) are
 * folded exactly (integers) by every workgroup.  save, gamma, beta, gamma_grad and beta_grad are read / updated 8 channels at
 * a time as 16-byte vectors: they must be 16-byte aligned (C % 8 == 0 already). */
int ep24_bn_act_bwd_apply(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save,
                          const float* gamma, const float* beta, const int64_t* dgamma, const int64_t* dbeta,
                          float* gamma_grad, float* beta_grad, void* dz, int64_t ld_dz, int64_t M, int C, int act,
                          int reps, void* stream);

/* The input gradient of a 1x1 stride-1 conv of the streaming kernel (Cout_k <= 256) that ALSO takes the BatchNorm-backward sums of the
 * unit below (round 5): what it stores - with `accumulate`, old + new: the complete gradient over a shortcut - is that unit's dy, so
 * sum(du) and sum(du * zhat) are added to dgamma / dbeta ([reps][..] 2^-36 fixed point, replica stride rep_stride, zeroed by the caller)
 * from the rounded rows on their way out, with ep24_bn_act_bwd_reduce's expressions (the fp32 order of the partial sums differs); that
 * unit's reduce launch is not needed.  dx is bit-identical to ep24_conv_dgrad_bf16's.  z / mean / invstd / gamma / beta: the unit below
 * (act = 1, SiLU).  EP24_E_UNSUPPORTED for a shape the streaming kernel does not take. */
int ep24_conv1x1_dgrad_bnr_bf16(const void* dy, int64_t ld_dy, const void* wt, void* dx, int64_t ld_dx, int accumulate, int B, int H, int W,
                                int Cin, int Cout_k, const void* z, int64_t ld_z, const float* mean, const float* invstd,
                                const float* gamma, const float* beta, int64_t* dgamma, int64_t* dbeta, int64_t rep_stride, int reps,
                                int act, void* stream);

/* Depthwise 3x3 convolution (groups = channels) of the DWConv blocks (yolox_24p/models/network_blocks.py:57-76: a depthwise BaseConv
 * followed by a 1x1 BaseConv; `depthwise=True` of CSPDarknet / Bottleneck / YOLOPAFPN / YOLOXHead, darknet.py:107, network_blocks.py:92,
 * yolo_pafpn.py:30, yolo_head_24p.py:45 - in no BASELINE configuration).  HBM-bound elementwise kernels (csrc/dwconv.hip): NHWC bf16
 * activations, w = the fp32 master [C][3][3] read in place, fp32 accumulation, stride 1 or 2, pad 1; C % 8 == 0.
 *   fwd:   z = conv(x, w); `stats` as ep24_conv_fwd_bf16 ([replicas][2][C] fixed-point batch statistics of z), or NULL.
 *   dgrad: dx (+)= the input gradient from dz [B,OH,OW,C].
 *   wgrad: slab [splits][C][9] fp32 = per-workgroup partial sums of dw, splits = ep24_dwconv_wgrad_splits(...) (fixed by the shape);
 *          ep24_wgrad_reduce folds them in order into the flat gradient, as for every other weight gradient (no atomics).
 * BatchNorm + activation of the unit are the ep24_bn_act_* entry points. */
int ep24_dwconv_fwd_bf16(const void* x, int64_t ld_x, const float* w, void* z, int64_t ld_z, int64_t* stats, int stats_replicas,
                         int B, int H, int W, int C, int ksize, int stride, void* stream);
int ep24_dwconv_dgrad_bf16(const void* dz, int64_t ld_dz, const float* w, void* dx, int64_t ld_dx, int accumulate, int B, int H,
                           int W, int C, int ksize, int stride, void* stream);
int ep24_dwconv_wgrad_splits(int B, int H, int W, int C, int stride);
int ep24_dwconv_wgrad_slab_bf16(const void* x, int64_t ld_x, const void* dz, int64_t ld_dz, float* slab, int64_t slab_floats, int B,
                                int H, int W, int C, int ksize, int stride, void* stream);

/* A 1x1 stride-1 conv unit TOGETHER WITH the BatchNorm pass in front of it (round 5; csrc/conv_igemm.hip igemm_stream_kernel<XF>).
 * The streaming 1x1 kernel is the one conv kernel whose A operand passes through registers, so the pass that would have produced
 * that operand as a launch of its own runs on the rows in flight: one dependent launch and one read of the rows less, the same
 * values bit for bit (the rows the MFMAs read are the bf16 values that are stored).  Shapes: Cin, Cout <= 128 (one K pass, one N
 * tile); ep24_conv1x1_xf_ok says whether a shape is taken.  SiLU units only (act = 1).
 *   ep24_conv1x1_bnin_bf16: y_in = silu(bn(z_in)) (+ residual) exactly as ep24_bn_act_fwd (statistics fold, save, running
 *     statistics, counters), then z_out = conv1x1(y_in, w) with its batch statistics exactly as ep24_conv_fwd_bf16
 *     (network_blocks.py:50-51 of the producing unit and :38-48 of this one; Bottleneck.forward :95-99).
 *   ep24_conv1x1_dgrad_bnbwd_bf16: dz = ep24_bn_act_bwd_apply(dy, z, ...) - stored, the weight gradient reads it; the sums are
 *     published into gamma_grad / beta_grad - then dx (+)= dz . wt exactly as ep24_conv_dgrad_bf16 of the unit's 1x1 conv
 *     (Cin / Cout_k as there: the forward conv's input channels / its output channels padded to 8). */
int ep24_conv1x1_xf_ok(int B, int H, int W, int Cin, int Cout);
int ep24_conv1x1_bnin_bf16(const void* z_in, int64_t ld_zin, const int64_t* stats_in, int reps_in, const float* gamma,
                           const float* beta, float* running_mean, float* running_var, int64_t* num_batches,
                           int64_t* num_batches2, float* save, void* y_in, int64_t ld_yin, const void* residual, int64_t ld_res,
                           float eps, float momentum, int act, const void* w, void* z_out, int64_t ld_zout, int64_t* stats_out,
                           int reps_out, int B, int H, int W, int Cin, int Cout, void* stream);
int ep24_conv1x1_dgrad_bnbwd_bf16(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save,
                                  const float* gamma, const float* beta, const int64_t* dgamma, const int64_t* dbeta,
                                  float* gamma_grad, float* beta_grad, void* dz, int64_t ld_dz, int act, int reps, const void* wt,
                                  void* dx, int64_t ld_dx, int accumulate, int B, int H, int W, int Cin, int Cout_k, void* stream);

/* ------------------------------------------------------------------------------------------------
 * fp32 PARITY MODE of the conv graph (csrc/f32path.hip).  The reference trains in fp32 (train_24p.py:86-104: no AMP,
 * network_blocks.py:50-51); ep24.engine.Engine(dtype=torch.float32) runs the same launch plan on fp32 activations through
 * these entry points so that images -> SimOTA indices / loss / gradients can be compared with the CPU oracle at fp32
 * accuracy (tests/test_gpu_fp32.py).  One thread per output element, reductions in double: not a fast path.
 * All tensors NHWC fp32 with a row stride in elements; weights are read in place from the flat fp32 master:
 * element (co, tap, ci) at w[co * w_co_stride + tap * w_tap_stride + ci].
 * ------------------------------------------------------------------------------------------------ */
int ep24_f32_stem_pack(const float* images, float* rows, int64_t ld, int B, int H, int W, void* stream);            /* Focus + 3x3 im2col */
/* transposed = 0: y = conv(x, w) (+ bias) with the row mapping of ep24_conv_fwd_bf16 (network_blocks.py:38-51);
 * transposed = 1: x is dy [B,OH,OW,Cout], y is dx [B,H,W,Cin] (+)= the input gradient.  B,H,W,Cin,Cout,ksize,stride always
 * describe the FORWARD convolution. */
int ep24_f32_conv(const float* x, int64_t ld_x, const float* w, int64_t w_co_stride, int64_t w_tap_stride, float* y, int64_t ld_y,
                  int64_t y_batch_rows, int64_t y_row0, const float* bias, int accumulate, int B, int H, int W, int Cin, int Cout,
                  int ksize, int stride, int transposed, void* stream);
int ep24_f32_conv_wgrad(const float* x, int64_t ld_x, const float* dy, int64_t ld_dy, float* dw, int64_t w_co_stride,
                        int64_t w_tap_stride, int B, int H, int W, int Cin, int Cout, int ksize, int stride, void* stream);   /* dw += */
/* training-mode BatchNorm + activation (+ residual): batch statistics (two-pass, double) -> save[0]=mean, save[1]=invstd,
 * running statistics, num_batches; then y = act(bn(z)) + residual. */
int ep24_f32_bn_act_fwd(const float* z, int64_t ld_z, const float* gamma, const float* beta, float* running_mean, float* running_var,
                        int64_t* num_batches, float* save, float* y, int64_t ld_y, const float* residual, int64_t ld_res, int64_t M,
                        int C, float eps, float momentum, int act, void* stream);
/* both passes of its backward: sums [2C] double scratch; gamma_grad / beta_grad += ; dz written. */
int ep24_f32_bn_act_bwd(const float* dy, int64_t ld_dy, const float* z, int64_t ld_z, const float* save, const float* gamma,
                        const float* beta, double* sums, float* gamma_grad, float* beta_grad, float* dz, int64_t ld_dz, int64_t M,
                        int C, int act, void* stream);
int ep24_f32_spp_fwd(const float* x, int64_t ld_x, float* y5, float* y9, float* y13, int64_t ld_y, int32_t* idx, int B, int H, int W,
                     int C, void* stream);                                        /* idx [3][B*H*W*C]: argmax pixel (y*W+x) */
int ep24_f32_spp_bwd(const float* dy5, const float* dy9, const float* dy13, int64_t ld_dy, const int32_t* idx, float* dx,
                     int64_t ld_dx, int accumulate, int B, int H, int W, int C, void* stream);
int ep24_f32_upsample2_fwd(const float* x, int64_t ld_x, float* y, int64_t ld_y, int B, int H, int W, int C, void* stream);
int ep24_f32_upsample2_bwd(const float* dy, int64_t ld_dy, float* dx, int64_t ld_dx, int accumulate, int B, int H, int W, int C,
                           void* stream);
int ep24_f32_rows_copy(const float* src, int64_t ld_src, float* dst, int64_t ld_dst, int accumulate, int64_t M, int C, void* stream);
int ep24_f32_head_decode_bwd(const float* dout, const float* out, float* d_regobj, float* d_cls, int B, int A, int a0, int H, int W,
                             float stride, int ncols, const float* d_origin, void* stream);   /* d_regobj [cells,32], d_cls [cells,ceil8(C)] */
int ep24_f32_colsum(const float* g, int64_t ld, float* db, int64_t M, int N, void* stream);    /* db[c] += sum over rows */

/* ------------------------------------------------------------------------------------------------
 * a1/a2  glue ops of the graph
 * ------------------------------------------------------------------------------------------------ */
/* Focus + im2col for the 3x3 stem: images [B,3,S,S] fp32 NCHW -> rows [B*(S/2)^2][ld] bf16, columns
 * (kh,kw,c4) with c4 = TL,BL,TR,BR x 3 channels, 108 real + zero padding to ld (network_blocks.py:188-210). */
int ep24_stem_pack(const float* images, void* rows, int64_t ld, int B, int H, int W, void* stream);

/* The Focus stem without an im2col buffer (round 3; network_blocks.py:188-210, Focus.forward + its BaseConv 3x3 over 12 channels).
 * ep24_focus_pack: images [B,3,S,S] fp32 NCHW -> f16 [B][S/2][S/2][16] bf16, channel (x parity * 2 + y parity) * 3 + c (the
 * reference's cat(top-left, bottom-left, top-right, bottom-right)), channels 12..15 zero.
 * ep24_stem_conv_fwd_bf16: y[B*FH*FW][ld_y] = conv3x3(f16, w), w = [Cout][ld_w] bf16 with column tap * 12 + channel (the packed
 * forward copy of the [Cout][3][3][12] master); stats as ep24_conv_fwd_bf16.  _infer: y = act(acc + bias) (BatchNorm folded).
 * ep24_stem_conv_wgrad_slab_bf16: slab[s][Cout][108] = partial dW of pixel split s (column tap * 12 + channel), folded by
 * ep24_wgrad_reduce; ep24_stem_conv_wgrad_splits = the number of splits.  Cout <= 64 for the weight gradient. */
int ep24_focus_pack(const float* images, void* f16, int B, int H, int W, void* stream);
int ep24_stem_conv_fwd_bf16(const void* f16, const void* w, int64_t ld_w, void* y, int64_t ld_y, int64_t* stats, int stats_replicas,
                            int B, int FH, int FW, int Cout, void* stream);
int ep24_stem_conv_fwd_infer_bf16(const void* f16, const void* w, int64_t ld_w, const float* bias, int act, void* y, int64_t ld_y,
                                  int B, int FH, int FW, int Cout, void* stream);
int ep24_stem_conv_wgrad_slab_bf16(const void* f16, const void* dy, int64_t ld_dy, float* slab, int64_t slab_floats, int B, int FH,
                                   int FW, int Cout, void* stream);
int ep24_stem_conv_wgrad_splits(int B, int FH, int FW, int Cout);

/* SPP max pools k = 5, 9, 13, stride 1, pad k/2 over x[B,H,W,C] (network_blocks.py:131-143).  Writes the
 * three pooled maps into y5/y9/y13 (row stride ld_y) and the winning window offset (dy*16+dx biased by 8)
 * into idx [3][B*H*W][C] uint8 for the backward (first maximum in row-major window order, as ATen).
 * scratch: 9*B*H*W*C bytes for the separable two-pass form (row maxima + their column offsets); null = one pass. */
int ep24_spp_fwd(const void* x, int64_t ld_x, void* y5, void* y9, void* y13, int64_t ld_y, uint8_t* idx,
                 int B, int H, int W, int C, void* scratch, void* stream);
/* dx (+)= routed gradients of the three pools. */
int ep24_spp_bwd(const void* dy5, const void* dy9, const void* dy13, int64_t ld_dy, const uint8_t* idx,
                 void* dx, int64_t ld_dx, int accumulate, int B, int H, int W, int C, void* stream);

/* nearest x2 upsample into a slice (yolo_pafpn.py:32) and its backward (sum of the 2x2 block). */
int ep24_upsample2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, int B, int H, int W, int C, void* stream);
int ep24_upsample2_bwd(const void* dy, int64_t ld_dy, void* dx, int64_t ld_dx, int accumulate,
                       int B, int H, int W, int C, void* stream);

/* hipMemsetAsync(p, 0, bytes) on the stream (step-start clearing of gradient / statistics buffers). */
int ep24_memset_zero(void* p, int64_t bytes, void* stream);
/* fp32 <-> bf16 casts of n (multiple of 4) contiguous elements: the bf16 wire format of the data-parallel gradient buckets
 * (ep24.dp.GradReducer(comm_dtype=torch.bfloat16): half the xGMI bytes of the reference's fp32 DDP buckets, core/trainer.py:163). */
int ep24_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
int ep24_cast_bf16_f32(const void* src, float* dst, int64_t n, void* stream);

/* strided row copy / add:  dst[m, 0:C] (=|+=) src[m, 0:C]  (bf16). */
int ep24_rows_copy(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, int accumulate, int64_t M, int C,
                   void* stream);

/* ------------------------------------------------------------------------------------------------
 * a3  head decode (yolox_24p/models/yolo_head_24p.py:212-237)
 * ------------------------------------------------------------------------------------------------ */
/* in place on out[B,A,107] fp32 rows of one level (anchor offset a0, HxW cells, stride s):
 * xy=(t+grid)*s, r=exp(t)*s, obj/cls logits untouched.  origin [B,A,26] (nullable) receives the raw regression
 * outputs first - the head's origin_preds under use_l1 (yolo_head_24p.py:179-188). */
int ep24_head_decode_fwd(float* out, int B, int A, int a0, int H, int W, float stride, int ncols, float* origin,
                         void* stream);
/* backward through the decode for one level + split into the padded bf16 gradients the prediction convs
 * consume: d_regobj [B*H*W][32] (26 reg + 1 obj + zeros), d_cls [B*H*W][round8(C)] (C classes + zeros).
 * d_origin [B,A,26] (nullable) is added to the raw regression gradient (the L1 branch's path around the decode). */
int ep24_head_decode_bwd(const float* dout, const float* out, void* d_regobj, void* d_cls, int B, int A, int a0,
                         int H, int W, float stride, int ncols, const float* d_origin, void* stream);
/* bias gradient: db[n] += sum_m g[m][n] for n < N (bf16 rows of stride ld). */
int ep24_colsum(const void* g, int64_t ld, float* db, int64_t M, int N, void* stream);
/* the same without atomics: ep24_colsum_splits(M) partial rows slab[s*N + n], folded in order by ep24_wgrad_reduce
 * with a row (offset, N, splits, slab offset). */
int ep24_colsum_splits(int64_t M);
int ep24_colsum_slab(const void* g, int64_t ld, float* slab, int64_t M, int N, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a4-a8  SimOTA assignment, batched over images, no host synchronisation
 *        (yolox_24p/models/losses.py:360-592, yolox_24p/utils/boxes.py:102-243)
 * ------------------------------------------------------------------------------------------------ */
/* a4+a5 (pts_in_poly + get_in_boxes_info, losses.py:497-592).  labels [B,50,51] fp32; xs/ys/strides [A].
 * Out: num_gt[B] int32 (rows with sum>0, losses.py:190), in_box[B*A], in_ctr[B*A] uint64 bitmasks over GTs.
 * Safe on any stream next to any other kernel.  (Round 2 measured different angle sums in lanes 48..63 of a wave when MFMA
 * kernels of another stream shared the SIMDs; the cause was a packed fp32 multiply with swapped operand halves that the
 * vectoriser had emitted - tools/hazard_probe.hip - and the library is built without such instructions since round 3:
 * DESIGN.md section 4, tests/test_abi.py::test_no_half_swapped_packed_fp32, tests/test_gpu_hazard.py.) */
int ep24_assign_candidates(const float* labels, const float* xs, const float* ys, const float* strides,
                           int32_t* num_gt, uint64_t* in_box, uint64_t* in_ctr, int B, int A, void* stream);
/* a6+a7 (bboxes_iou + class cost + total cost, boxes.py:166-243, losses.py:396-424) for every candidate
 * anchor: pw[B,50,A] and cost[B,50,A] fp32 (entries of non-candidate anchors / g >= num_gt are untouched). */
int ep24_assign_cost(const float* outputs, int ncols, const float* labels, const int32_t* num_gt,
                     const uint64_t* in_box, const uint64_t* in_ctr, float* pw, float* cost, int B, int A,
                     int num_classes, void* stream);
/* The same for anchors [a_lo, a_hi) of every image (round 5).  A pair's values do not depend on the launch's extent: launches over
 * disjoint ranges that cover [0, A) write exactly what ep24_assign_cost writes, so a head level's rows can be computed as soon as
 * that level's outputs exist (ep24.train: on the forward lane that produced them, off the loss path). */
int ep24_assign_cost_range(const float* outputs, int ncols, const float* labels, const int32_t* num_gt,
                           const uint64_t* in_box, const uint64_t* in_ctr, float* pw, float* cost, int B, int A,
                           int num_classes, int a_lo, int a_hi, void* stream);
/* a8 part 1 (dynamic_k_matching, losses.py:449-464): per (image, gt) top-10 pw sum -> k, the k cheapest
 * candidates are OR-ed into match[B*A] (uint64 bit g).  match must be zeroed by the caller; ks[B,50] out. */
int ep24_dynamic_k(const float* pw, const float* cost, const int32_t* num_gt, const uint64_t* in_box,
                   const uint64_t* in_ctr, uint64_t* match, int32_t* ks, int B, int A, void* stream);
/* a8 part 2 (losses.py:471-493): conflicts -> argmin cost, final fg mask, matched gt index (-1 = bg),
 * matched pairwise value. */
int ep24_assign_resolve(const uint64_t* match, const float* pw, const float* cost, const int32_t* num_gt,
                        int32_t* matched_gt, float* matched_iou, int B, int A, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a9+a10  losses and their gradient (yolox_24p/models/losses.py:80-157, :283-357)
 * ------------------------------------------------------------------------------------------------ */
/* per-block partial sums of the 24 circle-GIoU terms (matched rows), obj BCE (all anchors), cls BCE
 * (matched rows), num_fg; partials [nblocks][EP24_NUM_SUMS]; nblocks = ep24_loss_blocks(B, A). */
int ep24_loss_blocks(int B, int A);
/* L1 branch (use_l1, losses.py:197-198,255-262,304-309,594-604): origin [B,A,26] = the raw regression
 * outputs before the decode (null = branch off), xs/ys/strides [A] the anchor grid; adds sum |origin - l1_target| of
 * the matched rows as accumulator 27. */
int ep24_loss_terms(const float* outputs, int ncols, const float* labels, const int32_t* matched_gt,
                    const float* matched_iou, float* partials, int B, int A, int num_classes, const float* origin,
                    const float* xs, const float* ys, const float* strides, void* stream);
/* fixed-order reduction of the partials, dynamic task weights and state update (losses.py:286-345).
 * state[26] = last_{iou[24],obj,cls} (initialise to 1.0).  result[64]:
 *   [0] loss  [1..24] reg_w*loss_iou  [25] loss_obj  [26] loss_cls  [27] num_fg (clamped >= 1)
 *   [28] num_gts  [29..52] reg_w  [53] obj_w  [54] cls_w  [55] num_fg raw  [56] loss_l1 (added to [0] unweighted) */
int ep24_loss_finalize(const float* partials, int nblocks, const int32_t* num_gt, int B, float* state, float* result,
                       void* stream);
/* d loss / d outputs [B,A,ncols] fp32, scaled by *grad_scale (device scalar, may be null = 1).  With the L1 branch
 * (d_origin non-null) also d loss / d origin [B,A,26] = sign(origin - l1_target) / num_fg on matched rows, 0 elsewhere. */
int ep24_loss_grad(const float* outputs, int ncols, const float* labels, const int32_t* matched_gt,
                   const float* matched_iou, const float* result, const float* grad_scale, float* dout, int B, int A,
                   int num_classes, const float* origin, const float* xs, const float* ys, const float* strides,
                   float* d_origin, void* stream);

/* Round 5: ep24_loss_grad and ep24_head_decode_bwd of every level in ONE pass (the captured step, not the L1 branch): instead of the dense
 * fp32 [B,A,27+C] gradient it writes, per head level, the bf16 rows the prediction convs' backward consumes - reg+obj [B*cells][32] and
 * classes [B*cells][ld_cls = C rounded up to 8] - with the decode's chain rule applied (xy * stride, radii * decoded radius;
 * yolo_head_24p.py:212-237).  levels: HOST array [n_levels][4] of int64 = (cells per image, the stride's float32 bits, d_regobj, d_cls)
 * in anchor order; bit-identical to the two-launch form. */
int ep24_loss_grad_decode(const float* outputs, int ncols, const float* labels, const int32_t* matched_gt, const float* matched_iou,
                          const float* result, int B, int A, int num_classes, int n_levels, const int64_t* levels, void* stream);

/* stand-alone forms behind utils.bboxes_iou (boxes.py:166-243) and IOUloss.forward (losses.py:80-157) */
int ep24_circle_pairwise(const float* gt50, const float* pred26, float* out, int G, int P, void* stream);
int ep24_circle_matched_fwd(const float* pred26, const float* target50, float* loss24, int N, void* stream);
int ep24_circle_matched_bwd(const float* pred26, const float* target50, const float* dloss24, float* dpred26, int N,
                            void* stream);
/* circle_inter itself (IOUloss.circle_inter, losses.py:23-78; module-level utils.boxes.circle_inter, boxes.py:102-163):
 * intersection area of the k-th gt circle (radius gt_r[g][k]) with the k-th predicted circle and the centre distance,
 * res_inter / dist [pairs][24].  pairwise = 0: G == P rows matched one to one (the method); pairwise = 1: every gt row against
 * every pred row, pair index g * P + p (the module function's repeat_interleave / repeat order).  Case order as the reference:
 * |r1 - r2| >= d -> pi * rmin^2; d >= r1 + r2 -> 0 (overrides); else the lens with cosines clipped to +-0.99. */
int ep24_circle_lens(const float* gt_cx, const float* gt_cy, const float* gt_r, const float* pd_cx, const float* pd_cy,
                     const float* pd_r, float* res_inter, float* dist, int G, int P, int pairwise, void* stream);

/* ------------------------------------------------------------------------------------------------
 * a11  optimizer (yolox_24p/exp/yolox_base.py:120-124: SGD momentum 0.9 nesterov, no decay)
 * ------------------------------------------------------------------------------------------------ */
/* flat fp32 p/g/buf of n elements: buf = first ? g : m*buf+g ; p -= lr*(g + m*buf); g is scaled by
 * grad_scale first (1/world for data parallel means).  `first` is read from *first_flag (device int32),
 * which is cleared afterwards. */
int ep24_sgd_nesterov(float* p, const float* g, float* buf, int64_t n, float lr, float momentum, float grad_scale,
                      int32_t* first_flag, void* stream);
/* The same update with its hyper-parameters read from DEVICE memory, hp[8] = {lr, momentum, grad_scale, ema decay d,
 * 1 - d, ...}: a captured step follows an LRScheduler (utils/lr_scheduler.py:33-34) without being
 * re-captured.  ema (nullable, same layout as p) receives ModelEMA.update of the freshly written parameters in the
 * same pass: ema = ema*d + (1-d)*p, each product and the sum rounded to fp32 (utils/ema.py:47-60). */
int ep24_sgd_nesterov_hp(float* p, const float* g, float* buf, int64_t n, const float* hp, int32_t* first_flag, float* ema,
                         void* stream);
/* The same update restricted to elements [first, first + n) (first % 4 == 0): parameters whose gradients are complete are
 * updated while the tail of backward still produces the rest.  last != 0 on the call that finishes the step (it clears
 * first_flag); ema, if given, is the EMA copy's flat buffer (same layout). */
int ep24_sgd_nesterov_hp_range(float* p, const float* g, float* buf, int64_t first, int64_t n, const float* hp,
                               int32_t* first_flag, float* ema, int last, void* stream);
/* ... and the packed bf16 FORWARD copy of the conv weights kept current by the same pass (round 5; ABI 3): the flat buffer holds a
 * conv weight as [Cout][kh][kw][Cin], which for Cin % 8 == 0 is the layout of its packed copy, and segments start at multiples of 64
 * elements; wf_delta[e >> 6] = (offset of element e's segment in wf) - (its offset in the flat buffer), INT32_MIN for groups without
 * such a copy (those still go through ep24_pack_weights_batched).  The step no longer re-reads the masters to pack them. */
int ep24_sgd_nesterov_hp_range_pack(float* p, const float* g, float* buf, int64_t first, int64_t n, const float* hp,
                                    int32_t* first_flag, float* ema, int last, const int32_t* wf_delta, void* wf, void* stream);
/* ModelEMA.update over one flat buffer (parameters, or the BatchNorm running statistics); hp non-null overrides
 * decay / one_minus_decay with hp[3] / hp[4]. */
int ep24_ema_update(float* ema, const float* src, int64_t n, float decay, float one_minus_decay, const float* hp,
                    void* stream);
/* writes hp[0..4] on the stream (by-value arguments: no host buffer has to outlive the call). */
int ep24_set_hparams(float* hp, float lr, float momentum, float grad_scale, float ema_decay, float one_minus_decay,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * a13  fisheye sector warp (yolox/demo_featuremap.py:244-328)
 * ------------------------------------------------------------------------------------------------ */
/* The host mirror computes the 1-D tables exactly as the reference does (np.linspace / np.cos / np.sin,
 * demo_featuremap.py:258-283) and the crop box; the O(13200*T) part runs here.
 * winner[y*canvas_w+x] = max over (angle a, radius r) pairs landing on canvas pixel (y,x) of a*T+r, i.e. the
 * LAST writer of the reference's fancy-index scatter (:295-298).  Caller initialises winner to -1.
 * cos_tab/sin_tab [n_ang] and rho [T] are device float64. */
int ep24_sector_map(const double* cos_tab, const double* sin_tab, int n_ang, const double* rho, int T, int canvas_w,
                    int canvas_h, int32_t* winner, void* stream);
/* Crop [y0:y0+out_h, x0:x0+out_w] of the canvas (:301-306): dst [out_h][out_w][3] uint8 gathered from
 * src [T][n_ang][3] through the winner map, `fill` where nothing landed (114 image / 0 mask).  If src_index
 * is non-null the flat source index (row*n_ang+col, -1 = fill) is written too. */
int ep24_sector_gather(const uint8_t* src, const int32_t* winner, int canvas_w, int y0, int x0, int out_h, int out_w,
                       int T, int n_ang, uint8_t* dst, int fill, int32_t* src_index, void* stream);
/* bounding box of the non-zero pixels of channel 0 (:309-326): box = {xmin,ymin,xmax,ymax}; caller initialises
 * it to {INT_MAX,INT_MAX,-1,-1}. */
int ep24_mask_bbox(const uint8_t* mask3, int out_h, int out_w, int32_t* box, void* stream);

/* uint8 HWC bilinear resize with OpenCV's INTER_LINEAR fixed-point arithmetic (cv2.resize(image, (13200, T)),
 * demo_featuremap.py:285).  src [sh][sw][3] -> dst [dh][dw][3]. */
int ep24_resize_linear_u8(const uint8_t* src, int sh, int sw, uint8_t* dst, int dh, int dw, void* stream);
/* gather + resize in one pass (no [T, n_ang] intermediate): dst[oy][ox] = bilinear sample of src at the texel the
 * winner map selects, with the arithmetic of ep24_resize_linear_u8 / ep24_sector_gather (bit-identical results). */
int ep24_sector_warp_u8(const uint8_t* src, int sh, int sw, const int32_t* winner, int canvas_w, int y0, int x0,
                        int out_h, int out_w, int T, int n_ang, uint8_t* dst, int fill, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N3  inference path: eval-mode network + postprocess
 * ------------------------------------------------------------------------------------------------ */
/* BaseConv in eval mode (network_blocks.py:50-51 with BatchNorm2d.eval()): y = act(z*scale + shift) (+ residual),
 * scale = gamma / sqrt(running_var + eps), shift = beta - running_mean*scale. */
int ep24_bn_act_infer(const void* z, int64_t ld_z, const float* gamma, const float* beta, const float* running_mean,
                      const float* running_var, void* y, int64_t ld_y, const void* residual, int64_t ld_res,
                      int64_t M, int C, float eps, int act, void* stream);
/* eval head (yolo_head_24p.py:190-191, 239-256): decode xy / radii in place and apply sigmoid to obj and class logits. */
int ep24_head_decode_eval(float* out, int B, int A, int a0, int H, int W, float stride, int ncols, void* stream);
/* BatchNorm folding for inference, all conv units in two launches: w' = w * gamma / sqrt(running_var + eps) (packed bf16
 * [Cout][T][Cin_pad]) and bias = beta - running_mean * scale.  flat = the fp32 parameter buffer, bstat = the running
 * statistics buffer; desc[n_seg][12] int64 = {master offset, packed offset, Cout, T, Cin, Cin_pad, gamma offset, beta offset
 * (flat), running-mean offset, running-var offset (bstat), bias offset, 0}; prefix / cprefix [n_seg+1] = element / channel
 * prefix sums; eps[n_seg]. */
int ep24_fold_bn(const float* flat, const float* bstat, const int64_t* desc, const int64_t* prefix, const int64_t* cprefix,
                 const float* eps, int n_seg, int64_t total, int64_t total_channels, void* w_folded, float* bias, void* stream);
/* conv -> BN(running statistics) -> act (+ residual) of the eval-mode network as ONE launch: the conv over the folded weights
 * with y = act(acc + bias) + residual in its epilogue (act: 0 none, 1 SiLU, 2 ReLU; res nullable). */
int ep24_conv_fwd_infer_bf16(const void* x, int64_t ld_x, const void* w, const float* bias, int act, const void* res,
                             int64_t ld_res, void* y, int64_t ld_y, int B, int H, int W, int Cin, int Cout, int ksize,
                             int stride, void* stream);
/* postprocess (utils/boxes.py:29-99) in three launches.  prepare: per row best class (first maximum), class_conf,
 * score = obj*class_conf or -1 when below conf_thre, bounding rectangle of the 24 points (with the reference's
 * theta*cos(theta) factors, passed in as the host computes them).  nms: per image (one workgroup) candidates sorted by score (ties: lower row first), greedy
 * suppression of IoU > nms_thre within a class (torchvision batched_nms) or across classes (class_agnostic); keep
 * [B][A] receives the kept rows in order, keep_count[B] their number; sort_key / sort_idx / dead are [B][P] scratch,
 * P a power of two >= A.  gather: det[n][29] = (pred[:, :27], class_conf, class_pred) of one image's kept rows. */
int ep24_post_prepare(const float* pred, int ncols, int num_classes, int64_t n_rows, float conf_thre,
                      const float* ray_factors /* [48]: theta_k*cos(theta_k), then theta_k*sin(theta_k) */, float* score,
                      float* conf, int32_t* cls, float* rect, void* stream);
int ep24_post_nms(const float* score, const int32_t* cls, const float* rect, int B, int A, float nms_thre,
                  int class_agnostic, float* sort_key, int32_t* sort_idx, uint8_t* dead, int32_t* keep,
                  int32_t* keep_count, int P, void* stream);
int ep24_post_gather(const float* pred, int ncols, const float* conf, const int32_t* cls, const int32_t* keep, int n,
                     float* det, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N4  24-point label generation (yolox_24p/datasets/2+24_labels_create.py:61-116, :175-180)
 * ------------------------------------------------------------------------------------------------ */
/* rotation_for_24p for n objects (n <= 65535 per call).  masks: uint8 instance masks (non-zero = object) somewhere in
 * one device buffer; desc[n][6] int64 = {byte offset of the mask, H, W, L = int(sqrt(H^2 + W^2)), number of ray samples
 * = len(arange(0, L, 0.2)), row stride in bytes}; centre[n][2] double = box centre (x, y) as the reference forms it
 * (:167-168); rot[24][2] double = cos / sin of k*15 degrees as numpy computes them.  H + W + L must stay below 32768
 * (the reference holds the coordinates in int16).  Out: out_pts[n][24][2] int32 (x, y), out_r[n][24] double; a ray
 * without any candidate pixel (np.argmin of an empty array in the reference) gives (-1,-1) and +inf. */
int ep24_ray24(const uint8_t* masks, const int64_t* desc, const double* centre, const double* rot, int n,
               int32_t* out_pts, double* out_r, void* stream);
/* cv2.contourArea(cv2.convexHull(points)) of each object's 24 integer points (:175-176): area[n] double. */
int ep24_hull_area24(const int32_t* pts, int n, double* area, void* stream);

/* ------------------------------------------------------------------------------------------------
 * N1  input pipeline (yolox_24p/datasets/data_augment.py:109-174)
 * ------------------------------------------------------------------------------------------------ */
/* preproc for n images (n <= 65535): images = raw uint8 HWC sources in one device buffer; desc[n][6] int64 =
 * {byte offset, h, w, row stride in bytes, rh = int(h*r), rw = int(w*r)} with r = min(S_h/h, S_w/w) (:118-123);
 * scales[n][2] double = {1/(rw/w), 1/(rh/h)} (OpenCV's scale_x, scale_y).  out [n,3,S_h,S_w] fp32: the image resized
 * with cv2.resize's INTER_LINEAR fixed-point arithmetic in the top-left corner, 114 elsewhere, channels in source order. */
int ep24_preproc_u8(const uint8_t* images, const int64_t* desc, const double* scales, int n, float* out, int S_h, int S_w,
                    void* stream);
/* TrainTransform's label half (:145-173): rows [total][51] double = (class, 50 normalised coordinates) of all images,
 * row_off[n+1] int64, whr[n][3] double = (width, height, r).  out [n][max_labels][51] fp32: x columns (v*width)*r,
 * y columns (v*height)*r, first max_labels rows, zero padded. */
int ep24_preproc_labels(const double* rows, const int64_t* row_off, const double* whr, int n, float* out, int max_labels,
                        void* stream);

/* ------------------------------------------------------------------------------------------------
 * C4  swapped backbones (yolox_24p/models/darknet.py:179-429, yolox/models/yolo_pafpn.py:31-38; BASELINE config 4)
 *     The conv / BN+act entry points above carry them (act = 2 is ReLU); these are the remaining pieces.
 * ------------------------------------------------------------------------------------------------ */
/* im2col of fp32 NCHW images for a k x k / stride / pad conv (the ResNet stem: 7, 2, 3): rows [B*OH*OW][ld] bf16, column
 * (kh*k + kw)*C + c, zero in the padding and in columns >= k*k*C; the conv itself is then a 1x1 GEMM over the rows. */
int ep24_im2col_bf16(const float* images, void* rows, int64_t ld, int B, int C, int H, int W, int k, int stride, int pad,
                     void* stream);
/* Bottleneck tail "out += identity; out = relu(out)" (darknet.py:266-268): the sum comes out of ep24_bn_act_fwd (act 0 +
 * residual); relu_fwd clamps it in place, relu_bwd masks the incoming gradient in place with the stored output (y > 0). */
int ep24_relu_fwd(void* y, int64_t ld, int64_t M, int C, void* stream);
int ep24_relu_bwd(void* dy, int64_t ld_dy, const void* y, int64_t ld_y, int64_t M, int C, void* stream);
/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) (darknet.py:303) on NHWC bf16; idx [B*OH*OW*C] uint8 = winning tap
 * (first maximum in scan order, as ATen); the backward gathers (no atomics), accumulate = 1 adds to dx. */
int ep24_maxpool3s2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, uint8_t* idx, int B, int H, int W, int C,
                        void* stream);
int ep24_maxpool3s2_bwd(const void* dy, int64_t ld_dy, const uint8_t* idx, void* dx, int64_t ld_dx, int accumulate, int B,
                        int H, int W, int C, void* stream);

/* DenseNet pieces (darknet.py:515-674).  Its pre-activation blocks (BatchNorm -> ReLU -> conv over a growing concatenation)
 * run as ep24_bn_act_fwd on the shared input with per-layer statistics gathered from the block's, the input gradient of
 * every consumer accumulating through ep24_bn_act_bwd_apply_acc. */
/* per-channel sum / sum of squares of x [M][C] (row stride ld) added to stats[0][0..1][c] with channel stride ld_stats
 * (the conv epilogue's fixed-point layout [rep][2][ld_stats]). */
int ep24_colstats(const void* x, int64_t ld, int64_t* stats, int64_t ld_stats, int64_t M, int C, void* stream);
/* dst[rep][2][C] = the first C channels of src[rep][2][ld_src]. */
int ep24_stats_gather(const int64_t* src, int64_t ld_src, int64_t* dst, int C, int reps, void* stream);
/* ep24_bn_act_bwd_apply with dz += instead of dz = (same arguments). */
/* Round 5: pass 1 and pass 2 as ONE launch (at most one 256-thread workgroup per CU; a grid-wide arrive / wait on *barrier between the
 * passes - an int32 the caller zeroes before the launch; a bounded spin whose give-ups ep24_conv_ring_timeouts counts).  Same sums,
 * same dz, same gradient publication as the two launches. */
int ep24_bn_act_bwd_fused(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save, const float* gamma,
                          const float* beta, int64_t* dgamma, int64_t* dbeta, float* gamma_grad, float* beta_grad, void* dz,
                          int64_t ld_dz, int64_t M, int C, int act, int reps, int32_t* barrier, void* stream);
int ep24_bn_act_bwd_apply_acc(const void* dy, int64_t ld_dy, const void* z, int64_t ld_z, const float* save,
                              const float* gamma, const float* beta, const int64_t* dgamma, const int64_t* dbeta,
                              float* gamma_grad, float* beta_grad, void* dz, int64_t ld_dz, int64_t M, int C, int act, int reps,
                              void* stream);
/* nn.AvgPool2d(2, 2) of the Transition blocks, NHWC bf16. */
int ep24_avgpool2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, int B, int H, int W, int C, void* stream);
int ep24_avgpool2_bwd(const void* dy, int64_t ld_dy, void* dx, int64_t ld_dx, int accumulate, int B, int H, int W, int C,
                      void* stream);
/* nn.MaxPool2d(kernel_size=2, stride=2) of the VGG backbone (darknet.py:481), NHWC bf16; idx = winning tap per output element. */
int ep24_maxpool2_fwd(const void* x, int64_t ld_x, void* y, int64_t ld_y, uint8_t* idx, int B, int H, int W, int C,
                      void* stream);
int ep24_maxpool2_bwd(const void* dy, int64_t ld_dy, const uint8_t* idx, void* dx, int64_t ld_dx, int accumulate, int B,
                      int H, int W, int C, void* stream);
/* *p += 1 on the stream: num_batches_tracked of a BatchNorm module whose launch is shared with a neighbour (the merged
 * conv1 / conv2 unit of a CSP layer). */
int ep24_incr_i64(int64_t* p, void* stream);
/* nn.Dropout2d as data: x[n, :, :, c] *= keep[n*C + c] in place (keep = 0 or 1/(1-p), drawn by the host once per step);
 * the same call is its backward. */
int ep24_chanscale(void* x, int64_t ld, const float* keep, int B, int64_t HW, int C, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EP24_H */
