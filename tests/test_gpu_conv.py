"""GPU numerics of the conv-graph kernels (through the C ABI) against plain torch fp32 on the same
bf16-rounded operands.  Tolerances: the kernels accumulate in fp32 and round once to bf16 (2^-8 relative)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _abi():
    from ep24._lib import call, ptr, stream_ptr
    return call, ptr, stream_ptr


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def nhwc(x_nchw):
    return x_nchw.permute(0, 2, 3, 1).contiguous()


def close(got, want, rel=1.2e-2):
    got, want = got.float().cpu(), want.float().cpu()
    err = (got - want).abs().max().item()
    ref = want.abs().max().item()
    if not err <= rel * ref + 1e-6:                       # say WHERE (a protocol error of a ring kernel shows as whole tiles / channel groups)
        bad = ((got - want).abs() > rel * ref + 1e-6).nonzero()
        raise AssertionError("max err %.4g vs ref max %.4g; %d of %d elements off, first at %s, last at %s" % (
            err, ref, len(bad), got.numel(), bad[0].tolist(), bad[-1].tolist()))


CONV_CASES = [  # B, H, Cin, Cout, k, s
    (2, 20, 64, 64, 3, 1), (3, 16, 128, 128, 3, 1), (2, 16, 64, 128, 3, 2), (2, 10, 256, 256, 1, 1),
    (2, 12, 16, 24, 3, 1), (1, 24, 8, 16, 3, 2), (5, 9, 72, 200, 3, 1), (2, 8, 512, 64, 1, 1),
    # 1x1, K <= 256: the streaming kernel (row / channel / K tails, several row groups per workgroup)
    (3, 13, 128, 48, 1, 1), (2, 24, 112, 64, 1, 1), (1, 33, 24, 200, 1, 1), (4, 40, 64, 64, 1, 1), (20, 40, 256, 128, 1, 1),
]


@pytest.mark.parametrize("B,H,Cin,Cout,k,s", CONV_CASES)
def test_conv_fwd_dgrad_wgrad(B, H, Cin, Cout, k, s):
    call, ptr, sp = _abi()
    W = H
    pad = (k - 1) // 2
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, s, pad)
    OH, OW = y_ref.shape[2:]
    gy = rnd(B, Cout, OH, OW, seed=3)
    y_ref.backward(gy.float())

    xd = nhwc(x).to(DEV)
    wf = w.permute(0, 2, 3, 1).contiguous().to(DEV)                    # [Cout][kh][kw][Cin]
    wd = w.permute(1, 2, 3, 0).contiguous().to(DEV)                    # [Cin][kh][kw][Cout]
    y = torch.zeros(B, OH, OW, Cout, dtype=BF, device=DEV)
    R = 4
    stats = torch.zeros(R, 2, Cout, dtype=torch.int64, device=DEV)
    call("conv_fwd_bf16", ptr(xd), Cin, ptr(wf), ptr(y), Cout, 0, 0, 0, None, ptr(stats), R, B, H, W, Cin, Cout, k, s, sp())
    close(y.permute(0, 3, 1, 2), y_ref.detach())
    st = stats.sum(0).cpu().double().div(2 ** 20).float()               # 2^-20 fixed point
    yr = y_ref.detach()
    close(st[0], yr.sum((0, 2, 3)), rel=2e-3)
    close(st[1], (yr * yr).sum((0, 2, 3)), rel=2e-3)

    gyd = nhwc(gy).to(DEV)
    dx = torch.full((B, H, W, Cin), 7.0, dtype=BF, device=DEV)          # must be overwritten (accumulate=0)
    call("conv_dgrad_bf16", ptr(gyd), Cout, ptr(wd), ptr(dx), Cin, 0, B, H, W, Cin, Cout, k, s, sp())
    close(dx.permute(0, 3, 1, 2), xr.grad)
    base = rnd(B, H, W, Cin, seed=5).to(DEV)
    dx2 = base.clone()
    call("conv_dgrad_bf16", ptr(gyd), Cout, ptr(wd), ptr(dx2), Cin, 1, B, H, W, Cin, Cout, k, s, sp())
    close(dx2.permute(0, 3, 1, 2), xr.grad + base.float().cpu().permute(0, 3, 1, 2))

    dw = torch.zeros(Cout, k * k, Cin, device=DEV)
    call("conv_wgrad_bf16", ptr(xd), Cin, ptr(gyd), Cout, ptr(dw), k * k * Cin, Cout, Cin, B, H, W, Cin, Cout, k, s, sp())
    close(dw.view(Cout, k, k, Cin).permute(0, 3, 1, 2), wr.grad, rel=5e-3)
    call("conv_wgrad_bf16", ptr(xd), Cin, ptr(gyd), Cout, ptr(dw), k * k * Cin, Cout, Cin, B, H, W, Cin, Cout, k, s, sp())
    close(dw.view(Cout, k, k, Cin).permute(0, 3, 1, 2), 2 * wr.grad, rel=5e-3)      # += semantics

    # the atomic-free form: per-split slabs + ordered reduce; bitwise reproducible, same += semantics
    from ep24 import _lib
    splits = _lib.lib().fn["ep24_conv_wgrad_splits"](B, H, W, Cin, Cout, k, s)
    assert splits >= 1
    numel = Cout * k * k * Cin
    pad = 24                                                             # odd offsets: the reducer's scalar path
    slab = torch.full((pad + splits * numel,), float("nan"), device=DEV)  # every element must be overwritten
    grads = []
    for rep in range(2):
        g = torch.zeros(8 + numel, device=DEV)
        desc = torch.tensor([[8, numel, splits, pad]], dtype=torch.int64, device=DEV)
        call("conv_wgrad_slab_bf16", ptr(xd), Cin, ptr(gyd), Cout, ptr(slab, pad), splits * numel, k * k * Cin, Cout, Cin,
             B, H, W, Cin, Cout, k, s, sp())
        call("wgrad_reduce", ptr(desc), 1, numel, ptr(g), ptr(slab), sp())
        call("wgrad_reduce", ptr(desc), 1, numel, ptr(g), ptr(slab), sp())
        grads.append(g.clone())
    assert float(grads[0][:8].abs().max()) == 0.0
    close(grads[0][8:].view(Cout, k, k, Cin).permute(0, 3, 1, 2), 2 * wr.grad, rel=5e-3)
    assert torch.equal(grads[0], grads[1])


# The layers that carry ~80 % of YOLOX-l-24p's FLOPs at the BASELINE size (B = 20, 640 x 640: SURVEY.md 8d shape list), plus the
# shapes that select every production instantiation: the halo-patch kernel (one tile shape, 256 x 128; the 20x20x512 and 160x160x64
# cases fall back to the tiled kernel, so both kernel settings run the same code there), the wide and narrow tiled kernel (stride 2,
# K > 256 1x1), the streaming kernel with one and two K halves, the 112-column stem.  An entry may carry W as a 7th value.
HOT_SHAPES = [  # B, H, Cin, Cout, k, s[, W]
    (20, 40, 256, 256, 3, 1), (20, 80, 128, 128, 3, 1), (20, 20, 512, 512, 3, 1), (20, 80, 256, 256, 3, 1), (4, 160, 64, 64, 3, 1),
    (20, 40, 512, 1024, 3, 2), (20, 80, 256, 512, 3, 2), (20, 20, 2048, 1024, 1, 1), (20, 20, 1024, 512, 1, 1),
    (20, 40, 512, 256, 1, 1), (20, 80, 128, 128, 1, 1), (20, 80, 256, 256, 1, 1), (2, 320, 112, 64, 1, 1),
    # not on the YOLOX-l path: the halo-patch kernel with a K tail (96 = 64 + 32 channels) and an N tail (192 = 128 + 64)
    (20, 80, 96, 192, 3, 1), (10, 80, 160, 136, 3, 1),
    # non-square inputs whose pixel count is NOT a multiple of the 256-row tile (multiscale / rectangular training sizes produce
    # them: 480 x 672 -> 60 x 84): rows m >= M of the last tile, the `live` guard of the statistics in the wide epilogue,
    # out-of-range patch pieces, with two patch buffers (K = 128, LDS exactly 160 KB) and with one (K = 64)
    (20, 60, 128, 128, 3, 1, 84), (12, 52, 64, 128, 3, 1, 100),
    # the weights-in-registers kernel (round 5: 3x3 stride-1, <= 64 channels on either side, >= 256 tiles of 256 pixels; the
    # 160 x 160 x 64 shape above takes it too): a last tile with rows past M on a non-square image, a K tail (48 channels: the
    # window's last two chunks are zeros) with an N tail (40 channels), the narrowest image it takes (W = 32: nine image rows per tile)
    (3, 150, 64, 64, 3, 1, 160), (5, 120, 48, 40, 3, 1, 120), (9, 230, 64, 64, 3, 1, 32),
    # the streaming 1x1 kernel's tails (its loads and stores are unconditional; what does not exist is an out-of-range buffer
    # offset): rows past M in the last block and a whole block past the end (odd unit count), an N tail, K = 64 (two K steps),
    # a K tail inside a step (K = 120), two K halves with a tail (K = 200, M large enough for the streaming form)
    (3, 50, 64, 72, 1, 1, 50), (5, 36, 120, 200, 1, 1, 36), (17, 80, 200, 136, 1, 1, 79),
    # the ring without a patch (round 4; kernel_opts bit 4; an 8th value is the kernel id the shape must dispatch to with it): the 4-step 1x1 layer of the
    # 40x40 level, a 1x1 layer with M, N and K tails, a stride-2 3x3 layer with a K tail
    (20, 40, 256, 256, 1, 1, 40, 4), (13, 50, 200, 264, 1, 1, 47, 4), (20, 90, 72, 136, 3, 2, 86, 4),
    # the NARROW ring (256 x 64 tiles, kernel_opts bit 6, an A/B option; a 9th value = the kernel_opts the id is asked under): the 20 x 20
    # level of YOLOX-l, which 256 x 128 tiles do not spread over the chip; M and N tails; a K tail over two patch buffers; N = 64 with one chunk and the widest patch
    # (W = 160: three patch pieces per loader and step); N = 64 with an M tail
    (20, 20, 512, 512, 3, 1, 20, 5, 64), (21, 20, 256, 200, 3, 1, 19, 5, 64), (21, 20, 72, 256, 3, 1, 19, 5, 64), (4, 160, 64, 64, 3, 1, 160, 5, 64),
    (7, 52, 64, 64, 3, 1, 100, 5, 64), (20, 40, 256, 256, 3, 1, 40, 5, 64),
]


def _torch_conv_ref(x, w, gy, s, pad):
    """fp32 conv forward / input gradient / weight gradient on the bf16-rounded operands (device if it works, else host)."""
    for dev in (DEV, "cpu"):
        try:
            xr = x.float().to(dev).requires_grad_(True)
            wr = w.float().to(dev).requires_grad_(True)
            y = F.conv2d(xr, wr, None, s, pad)
            y.backward(gy.float().to(dev))
            return y.detach().cpu(), xr.grad.cpu(), wr.grad.cpu()
        except RuntimeError:
            continue
    raise RuntimeError("no fp32 reference")


@pytest.mark.parametrize("shape", HOT_SHAPES, ids=lambda t: "x".join(str(v) for v in t))
def test_conv_hot_shapes(shape):
    B, H, Cin, Cout, k, s = shape[:6]
    W = shape[6] if len(shape) > 6 else H
    call, ptr, sp = _abi()
    from ep24 import _lib
    fn = _lib.lib().fn
    pad = (k - 1) // 2
    x = rnd(B, Cin, H, W, seed=1)
    w = rnd(Cout, Cin, k, k, seed=2, scale=(Cin * k * k) ** -0.5)
    OH, OW = (H + 2 * pad - k) // s + 1, (W + 2 * pad - k) // s + 1
    gy = rnd(B, Cout, OH, OW, seed=3)
    y_ref, dx_ref, dw_ref = _torch_conv_ref(x, w, gy, s, pad)
    M = B * OH * OW
    # Round 5: the variants that lost twice (ring without a patch = bit 4, 32x32x16 consumers = bit 5, narrow ring = bit 6, the weight
    # gradient as a ring) left the product library; `make variants` + EP24_LIB=.../libep24_variants.so brings them back for this test
    have_variants = fn["ep24_ab_variants"]() == 1
    if len(shape) > 7 and not have_variants:
        # the product library refuses the bits loudly (also in the dry-run query)
        assert fn["ep24_conv_kernel_for_ex"](0, B, H, W, Cin, Cout, k, s, 0, 0, shape[8] if len(shape) > 8 else 16) < 0
        assert "make" in _lib.lib().last_error() and "variants" in _lib.lib().last_error()
        pytest.skip("variant shape: needs the A/B library (make variants)")
    if len(shape) > 7:
        assert fn["ep24_conv_kernel_for_ex"](0, B, H, W, Cin, Cout, k, s, 0, 0, shape[8] if len(shape) > 8 else 16) == shape[7]
        if shape[7] == 5:                                                  # ... which no layer takes by default
            assert fn["ep24_conv_kernel_for_ex"](0, B, H, W, Cin, Cout, k, s, 0, 0, 0) in (0, 1, 3)
    elif len(shape) > 6:
        want = (6 if max(Cin, Cout) <= 64 else 3) if k == 3 else 2       # 3: the loader / consumer ring (round 4), 6: weights in registers (round 5), 2: streaming
        assert M % 256 != 0 and fn["ep24_conv_kernel_for"](0, B, H, W, Cin, Cout, k, s, 0, 0) == want, "meant to exercise the ring / streaming kernel's tails"

    xd = nhwc(x).to(DEV)
    wf = w.permute(0, 2, 3, 1).contiguous().to(DEV)
    wd = w.permute(1, 2, 3, 0).contiguous().to(DEV)
    gyd = nhwc(gy).to(DEV)
    R = 8
    outs = {}
    # kernel_opts of the _ex entry points (per call, no global switch): 0 = the shape's default (3x3 stride-1: the loader / consumer
    # ring of conv_ring.hip where it fits), bit 3 = the 8-wave halo-patch kernel instead, bit 0 = the generic tiled kernel
    # other layers: bit 4 sends those that fill the chip with 256 x 128 tiles to the ring without a patch (kernel id 4; an A/B option)
    # 3x3 stride-1: the default is the ring with 16x16x32 consumers; bit 5 = its 32x32x16 form (key 3), bit 3 = the 8-wave kernel
    # key 4: the narrow ring where it fits; key 5 (bits 0 + 7): the tiled kernel WITHOUT its three-stage form (the 20 x 20 level)
    # key 6 (bit 9): a 3x3 stride-1 layer with <= 64 channels in the kernel it ran in before the weights-in-registers kernel
    variants = ((0, 1), (2, 0), (3, 32), (1, 8), (4, 64), (5, 129), (6, 512)) if (k == 3 and s == 1) else ((0, 128), (2, 0), (5, 16))
    if not have_variants:
        variants = tuple(v for v in variants if not (v[1] & (16 | 32 | 64)))
    for patch, ko in variants:
        y = torch.zeros(B, OH, OW, Cout, dtype=BF, device=DEV)
        stats = torch.zeros(R, 2, Cout, dtype=torch.int64, device=DEV)
        call("conv_fwd_bf16_ex", ptr(xd), Cin, ptr(wf), ptr(y), Cout, 0, 0, 0, None, ptr(stats), R, B, H, W, Cin, Cout, k, s, ko, sp())
        dx = torch.full((B, H, W, Cin), 7.0, dtype=BF, device=DEV)
        if not (k == 1 and s == 2):
            call("conv_dgrad_bf16_ex", ptr(gyd), Cout, ptr(wd), ptr(dx), Cin, 0, B, H, W, Cin, Cout, k, s, ko, sp())
        base = rnd(B, H, W, Cin, seed=5).to(DEV)
        dx2 = base.clone()
        call("conv_dgrad_bf16_ex", ptr(gyd), Cout, ptr(wd), ptr(dx2), Cin, 1, B, H, W, Cin, Cout, k, s, ko, sp())
        torch.cuda.synchronize()
        close(y.permute(0, 3, 1, 2), y_ref)
        st = stats.sum(0).cpu().double().div(2 ** 20).float()
        close(st[0], y_ref.sum((0, 2, 3)), rel=2e-3)
        close(st[1], (y_ref * y_ref).sum((0, 2, 3)), rel=2e-3)
        close(dx.permute(0, 3, 1, 2), dx_ref)
        close(dx2.permute(0, 3, 1, 2), dx_ref + base.float().cpu().permute(0, 3, 1, 2))
        outs[patch] = (y, dx, dx2, stats.clone())
    # same products, same fp32 accumulation order per output element (channel chunks outer, taps inner): bit-equal outputs and input
    # gradients (plain and accumulated), whichever kernel ran; the batch statistics are sums of per-tile fp32 partial sums, and the
    # tiles differ (256 rows against 128): equal to fp32 summation order
    for other in outs:
        if other == 3 and have_variants and fn["ep24_conv_kernel_for_ex"](0, B, H, W, Cin, Cout, k, s, 0, 0, 32) == 3:
            # The 32x32x16 consumers add the SAME products, the fp32 sum of a 64-channel chunk in four k-steps of 16 instead of two of
            # 32: fp32 sums agree to rounding, so after the one rounding to bf16 an element differs by at most one bf16 ulp (2^-8
            # relative; an element that is itself a cancelled sum by 1e-5 of the tensor's range), and only a few per cent do at all.
            for a_, b_ in zip(outs[0][:3], outs[3][:3]):
                a32, b32 = a_.float(), b_.float()
                d = (a32 - b32).abs()
                assert bool((d <= 2.0 ** -8 * b32.abs() + 1e-5 * float(b32.abs().max())).all()), float(d.max())
                assert float((d > 0).float().mean()) < 0.05, float((d > 0).float().mean())
        else:
            for a_, b_ in zip(outs[0][:3], outs[other][:3]):
                assert torch.equal(a_, b_), other
        sa, sb = outs[0][3].sum(0).double(), outs[other][3].sum(0).double()
        assert float((sa - sb).abs().max()) <= 1e-5 * float(sb.abs().max()), other
    if k == 3 and s == 1 and max(Cin, Cout) <= 64 and M >= 65536:
        # the weights-in-registers kernel is what the default ran, the tiled kernel what bit 9 ran; its statistics (a workgroup's sums
        # over all of its tiles) are reproducible bit for bit
        assert fn["ep24_conv_kernel_for"](0, B, H, W, Cin, Cout, k, s, 0, 0) == 6 and fn["ep24_conv_kernel_for"](1, B, H, W, Cin, Cout, k, s, 0, 0) == 6
        assert fn["ep24_conv_kernel_for_ex"](0, B, H, W, Cin, Cout, k, s, 0, 0, 512) == 0
        y_again = torch.zeros(B, OH, OW, Cout, dtype=BF, device=DEV)
        st_again = torch.zeros(R, 2, Cout, dtype=torch.int64, device=DEV)
        call("conv_fwd_bf16_ex", ptr(xd), Cin, ptr(wf), ptr(y_again), Cout, 0, 0, 0, None, ptr(st_again), R, B, H, W, Cin, Cout, k, s, 0, sp())
        assert torch.equal(y_again, outs[2][0]) and torch.equal(st_again, outs[2][3])
    if 3 in outs:                                                          # ... and is bitwise reproducible run to run
        y_again = torch.zeros(B, OH, OW, Cout, dtype=BF, device=DEV)
        call("conv_fwd_bf16_ex", ptr(xd), Cin, ptr(wf), ptr(y_again), Cout, 0, 0, 0, None, None, R, B, H, W, Cin, Cout, k, s, 32, sp())
        assert torch.equal(y_again, outs[3][0])
    # no bounded wait of the ring kernel ever gave up (a protocol error would show here even if the numbers happened to agree)
    assert fn["ep24_conv_ring_timeouts"]() == 0

    # weight gradient: wgrad_kernel (the default) and, where the layer is eligible (Cout >= 256, the grid fills the chip), its loader /
    # consumer ring form (kernel_opts bit 0, an A/B option).  The kernels may split the pixels differently: equal to fp32 summation
    # order; each is bitwise reproducible.
    numel = Cout * k * k * Cin
    grads = {}
    for wo in ((0, 1) if have_variants else (0,)):
        splits = fn["ep24_conv_wgrad_splits_ex"](B, H, W, Cin, Cout, k, s, wo)
        reps = []
        for rep in range(2):
            slab = torch.full((splits * numel,), float("nan"), device=DEV)
            g = torch.zeros(numel, device=DEV)
            desc = torch.tensor([[0, numel, splits, 0]], dtype=torch.int64, device=DEV)
            call("conv_wgrad_slab_bf16_ex", ptr(xd), Cin, ptr(gyd), Cout, ptr(slab), splits * numel, k * k * Cin, Cout, Cin, B, H, W, Cin, Cout, k, s, wo, sp())
            call("wgrad_reduce", ptr(desc), 1, numel, ptr(g), ptr(slab), sp())
            reps.append(g)
        close(reps[0].view(Cout, k, k, Cin).permute(0, 3, 1, 2), dw_ref, rel=5e-3)
        if len(reps) == 2:
            assert torch.equal(reps[0], reps[1])
        grads[wo] = (reps[0], splits)
    if have_variants:
        assert float((grads[0][0] - grads[1][0]).abs().max()) <= 2e-5 * float(grads[1][0].abs().max()), (grads[0][1], grads[1][1])
    assert fn["ep24_conv_ring_timeouts"]() == 0


@pytest.mark.parametrize("k,shapes,splits", [
    (1, [(4, 40, 40, 256, 256), (4, 40, 40, 128, 256), (3, 20, 24, 256, 200), (4, 40, 40, 256, 256)], [3, 1, 2, 5]),     # 64 x 64 tiles
    (3, [(4, 40, 40, 256, 256), (2, 24, 20, 136, 384), (4, 40, 40, 256, 256)], [2, 1, 4]),                              # 128 x 128 tiles
    (3, [(3, 40, 40, 64, 256), (2, 32, 32, 64, 128)], [2, 3]),                                                          # 128 (co) x 64 (ci)
])
def test_grouped_weight_gradients(k, shapes, splits):
    """Round 5: ep24_conv_wgrad_group_bf16 - several layers' weight gradients (one tile class, different shapes and split counts)
    in ONE launch.  Each problem against torch fp32 on the bf16-rounded operands, bitwise reproducible, and a problem's slabs equal
    bit for bit what the same (shape, splits) gives as a group of one (a workgroup's work does not depend on its neighbours)."""
    call, ptr, sp = _abi()
    from ep24 import _lib
    fn = _lib.lib().fn
    cls = {fn["ep24_conv_wgrad_tile_class"](cin, cout, k) for (_, _, _, cin, cout) in shapes}
    assert len(cls) == 1
    probs = []
    for i, (B, H, W, cin, cout) in enumerate(shapes):
        x = rnd(B, cin, H, W, seed=10 + i)
        gy = rnd(B, cout, H, W, seed=20 + i)
        xr = x.float().requires_grad_(False)
        wr = torch.zeros(cout, cin, k, k, requires_grad=True)
        F.conv2d(xr, wr, None, 1, (k - 1) // 2).backward(gy.float())
        probs.append(dict(x=nhwc(x).to(DEV), gy=nhwc(gy).to(DEV), ref=wr.grad.permute(0, 2, 3, 1).reshape(-1), B=B, H=H, W=W, cin=cin, cout=cout,
                          numel=cout * k * k * cin))

    def run(sel):
        outs = []
        rows, slabs = [], []
        for i in sel:
            pr, sp_ = probs[i], splits[i]
            slab = torch.full((sp_ * pr["numel"],), float("nan"), device=DEV)
            slabs.append(slab)
            rows.append([pr["x"].data_ptr(), pr["cin"], pr["gy"].data_ptr(), pr["cout"], slab.data_ptr(), sp_ * pr["numel"], k * k * pr["cin"],
                         pr["cout"], pr["cin"], pr["B"], pr["H"], pr["W"], pr["cin"], pr["cout"], k, 1, sp_])
        desc = torch.tensor(rows, dtype=torch.int64)                    # a HOST table
        call("conv_wgrad_group_bf16", desc.data_ptr(), len(rows), sp())
        for i, slab in zip(sel, slabs):
            pr = probs[i]
            g = torch.zeros(pr["numel"], device=DEV)
            d = torch.tensor([[0, pr["numel"], splits[i], 0]], dtype=torch.int64, device=DEV)
            call("wgrad_reduce", ptr(d), 1, pr["numel"], ptr(g), ptr(slab), sp())
            outs.append((slab.clone(), g))
        torch.cuda.synchronize()
        return outs

    whole = run(list(range(len(shapes))))
    again = run(list(range(len(shapes))))
    for i, ((slab, g), (slab2, g2)) in enumerate(zip(whole, again)):
        assert not torch.isnan(slab).any(), "a slab element nobody wrote"
        assert torch.equal(slab, slab2) and torch.equal(g, g2)
        close(g, probs[i]["ref"], rel=5e-3)
        alone = run([i])[0]
        assert torch.equal(alone[0], slab), i
    with pytest.raises(_lib.Ep24Error):                                 # another tile class in the same launch is refused
        other = (512, 512, 1) if k == 1 else (256, 256, 1)
        assert fn["ep24_conv_wgrad_tile_class"](*other) not in cls
        B, H, W, cin, cout = shapes[0]
        x2 = torch.zeros(B * H * W, other[0], dtype=BF, device=DEV)
        g2 = torch.zeros(B * H * W, other[1], dtype=BF, device=DEV)
        sl = torch.zeros(other[0] * other[1] * other[2] ** 2, device=DEV)
        pr = probs[0]
        rows = [[pr["x"].data_ptr(), pr["cin"], pr["gy"].data_ptr(), pr["cout"], whole[0][0].data_ptr(), splits[0] * pr["numel"], k * k * pr["cin"], pr["cout"],
                 pr["cin"], pr["B"], pr["H"], pr["W"], pr["cin"], pr["cout"], k, 1, splits[0]],
                [x2.data_ptr(), other[0], g2.data_ptr(), other[1], sl.data_ptr(), sl.numel(), other[2] ** 2 * other[0], other[1], other[0], B, H, W,
                 other[0], other[1], other[2], 1, 1]]
        call("conv_wgrad_group_bf16", torch.tensor(rows, dtype=torch.int64).data_ptr(), 2, sp())


@pytest.mark.parametrize("B,IH,IW,Cout", [(2, 32, 32, 64), (3, 96, 80, 64), (1, 64, 132, 24), (2, 48, 40, 8), (5, 160, 160, 64)])
def test_focus_stem_without_im2col(B, IH, IW, Cout):
    """Focus + 3x3 conv over the space-to-depth image (network_blocks.py:188-210): pack bit-exact, forward / statistics / eval-mode
    form / weight gradient against torch on the same bf16 operands; row, column and channel tails, several pixel splits."""
    call, ptr, sp = _abi()
    from ep24._lib import lib
    img = torch.rand(B, 3, IH, IW, generator=torch.Generator().manual_seed(61)) * 255 - 100
    FH, FW = IH // 2, IW // 2
    f16 = torch.full((B, FH, FW, 16), 9.0, dtype=BF, device=DEV)
    imgd = img.to(DEV)
    call("focus_pack", ptr(imgd), ptr(f16), B, IH, IW, sp())
    foc = torch.cat([img[..., 0::2, 0::2], img[..., 1::2, 0::2], img[..., 0::2, 1::2], img[..., 1::2, 1::2]], 1)   # Focus.forward
    assert torch.equal(f16[..., :12].float().cpu(), nhwc(foc).to(BF).float())
    assert float(f16[..., 12:].abs().sum()) == 0
    w = rnd(Cout, 12, 3, 3, seed=62, scale=108 ** -0.5 / 50)
    xr = foc.to(BF).float()
    wr = w.float().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, 1, 1)
    wrow = torch.zeros(Cout, 112, dtype=BF)
    wrow[:, :108] = w.permute(0, 2, 3, 1).reshape(Cout, 108)                # column tap * 12 + channel
    wrow = wrow.to(DEV)
    R = 3
    y = torch.full((B, FH, FW, Cout), 5.0, dtype=BF, device=DEV)
    stats = torch.zeros(R, 2, Cout, dtype=torch.int64, device=DEV)
    call("stem_conv_fwd_bf16", ptr(f16), ptr(wrow), 112, ptr(y), Cout, ptr(stats), R, B, FH, FW, Cout, sp())
    close(y.permute(0, 3, 1, 2), y_ref.detach())
    st = stats.sum(0).cpu().double().div(2 ** 20).float()
    yr = y_ref.detach()
    close(st[0], yr.sum((0, 2, 3)), rel=2e-3)
    close(st[1], (yr * yr).sum((0, 2, 3)), rel=2e-3)
    # eval mode: y = silu(acc + bias)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(63))
    y2 = torch.zeros(B, FH, FW, Cout, dtype=BF, device=DEV)
    bd = bias.to(DEV)
    call("stem_conv_fwd_infer_bf16", ptr(f16), ptr(wrow), 112, ptr(bd), 1, ptr(y2), Cout, B, FH, FW, Cout, sp())
    close(y2.permute(0, 3, 1, 2), F.silu(yr + bias.view(1, -1, 1, 1)))
    # weight gradient: slabs of the pixel splits, folded by wgrad_reduce
    gy = rnd(B, Cout, FH, FW, seed=64)
    y_ref.backward(gy.float())
    gyd = nhwc(gy).to(DEV)
    splits = lib().fn["ep24_stem_conv_wgrad_splits"](B, FH, FW, Cout)
    assert splits >= 1
    slab = torch.full((splits, Cout, 108), float("nan"), device=DEV)
    call("stem_conv_wgrad_slab_bf16", ptr(f16), ptr(gyd), Cout, ptr(slab), slab.numel(), B, FH, FW, Cout, sp())
    dw = slab.sum(0).view(Cout, 3, 3, 12).permute(0, 3, 1, 2)
    close(dw, wr.grad, rel=5e-3)
    slab2 = torch.zeros_like(slab)
    call("stem_conv_wgrad_slab_bf16", ptr(f16), ptr(gyd), Cout, ptr(slab2), slab.numel(), B, FH, FW, Cout, sp())
    assert torch.equal(slab, slab2)                                         # plain stores, fixed order: the same bits every run


@pytest.mark.parametrize("B,H,W,Cin,Cout,act", [(2, 20, 20, 64, 64, 1), (3, 16, 24, 128, 128, 1), (5, 9, 9, 72, 200, 2), (2, 12, 12, 16, 24, 1),
                                                  (2, 40, 40, 256, 256, 1), (1, 7, 13, 8, 40, 3)])
def test_dgrad_with_fused_bn_reduce(B, H, W, Cin, Cout, act):
    """ep24_conv_dgrad_bnr_bf16 = ep24_conv_dgrad_bf16 followed by ep24_bn_act_bwd_reduce on its result: the same dx bit for bit,
    the same two sums up to the order of fp32 partial sums (halo-patch and tiled kernels, row / channel tails, three activations)."""
    call, ptr, sp = _abi()
    M = B * H * W
    gy = nhwc(rnd(B, Cout, H, W, seed=71)).to(DEV)
    wd = rnd(Cin, 3, 3, Cout, seed=72, scale=(Cout * 9) ** -0.5).to(DEV)
    z = nhwc(rnd(B, Cin, H, W, seed=73)).to(DEV)
    g = torch.Generator().manual_seed(74)
    mean, inv = torch.randn(Cin, generator=g) * 0.3, torch.rand(Cin, generator=g) + 0.5
    save = torch.cat([mean, inv]).to(DEV)
    gamma, beta = (torch.rand(Cin, generator=g) + 0.5).to(DEV), (torch.randn(Cin, generator=g) * 0.2).to(DEV)
    R = 4
    dx = torch.full((B, H, W, Cin), 3.0, dtype=BF, device=DEV)
    call("conv_dgrad_bf16", ptr(gy), Cout, ptr(wd), ptr(dx), Cin, 0, B, H, W, Cin, Cout, 3, 1, sp())
    sums = torch.zeros(R, 2, Cin, dtype=torch.int64, device=DEV)
    call("bn_act_bwd_reduce", ptr(dx), Cin, ptr(z), Cin, ptr(save), ptr(gamma), ptr(beta), ptr(sums), ptr(sums) + 8 * Cin, M, Cin, act, R, sp())
    dx2 = torch.full((B, H, W, Cin), 5.0, dtype=BF, device=DEV)
    sums2 = torch.zeros(R, 2, Cin, dtype=torch.int64, device=DEV)
    call("conv_dgrad_bnr_bf16", ptr(gy), Cout, ptr(wd), ptr(dx2), Cin, B, H, W, Cin, Cout, 3, ptr(z), Cin, ptr(save), ptr(save) + 4 * Cin,
         ptr(gamma), ptr(beta), ptr(sums2), ptr(sums2) + 8 * Cin, 2 * Cin, R, act, sp())
    assert torch.equal(dx, dx2)
    a, b = sums.sum(0).cpu().double() / 2 ** 36, sums2.sum(0).cpu().double() / 2 ** 36
    scale = a.abs().max().item()
    assert (a - b).abs().max().item() <= 2e-5 * scale + 1e-6, ((a - b).abs().max().item(), scale)
    sums3 = torch.zeros_like(sums2)
    call("conv_dgrad_bnr_bf16", ptr(gy), Cout, ptr(wd), ptr(dx2), Cin, B, H, W, Cin, Cout, 3, ptr(z), Cin, ptr(save), ptr(save) + 4 * Cin,
         ptr(gamma), ptr(beta), ptr(sums3), ptr(sums3) + 8 * Cin, 2 * Cin, R, act, sp())
    assert torch.equal(sums2.sum(0), sums3.sum(0))                           # fixed-point sums: the same bits every run


def test_conv_refuses_operands_beyond_2gib():
    """32-bit tile addressing: an activation of 2 GiB or more is refused, not wrapped (pointers are never dereferenced)."""
    call, ptr, sp = _abi()
    from ep24 import _lib
    t = torch.zeros(64, dtype=BF, device=DEV)
    fn = _lib.lib().fn
    rc = fn["ep24_conv_fwd_bf16"](ptr(t), 64, ptr(t), ptr(t), 64, 0, 0, 0, None, None, 1, 24, 1280, 1280, 64, 64, 3, 1, sp())
    assert rc != 0 and "2 GiB" in _lib.lib().last_error()
    rc = fn["ep24_conv_dgrad_bf16"](ptr(t), 64, ptr(t), ptr(t), 64, 0, 24, 1280, 1280, 64, 64, 3, 1, sp())
    assert rc != 0 and "2 GiB" in _lib.lib().last_error()


def test_conv_slices_fp32_out_bias_and_row_mapping():
    """1x1 predictor form: input is a channel slice of a wider buffer, output fp32 + bias into rows n*A + a0 + hw."""
    call, ptr, sp = _abi()
    B, H, W, Cin, N, ld_x, A, a0, ncols, col0 = 3, 5, 5, 64, 27, 96, 40, 7, 107, 0
    xb = rnd(B * H * W, ld_x, seed=11).to(DEV)
    w = rnd(N, Cin, seed=12, scale=0.1)
    bias = torch.linspace(-1, 1, N)
    out = torch.full((B, A, ncols), -5.0, device=DEV)
    wd_, bd_ = w.to(DEV), bias.to(DEV)          # keep the device copies alive across the launch
    call("conv_fwd_bf16", ptr(xb, 16), ld_x, ptr(wd_), ptr(out, col0), ncols, 1, A, a0, ptr(bd_), None, 1,
         B, H, W, Cin, N, 1, 1, sp())
    ref = xb.float().cpu()[:, 16:16 + Cin] @ w.float().t() + bias
    got = out.cpu()[:, a0:a0 + H * W, :N].reshape(-1, N)
    close(got, ref, rel=2e-3)
    assert float(out.cpu()[:, :a0].min()) == -5.0 and float(out.cpu()[:, a0 + H * W:].max()) == -5.0
    assert float(out.cpu()[:, :, N:].max()) == -5.0                      # columns beyond N untouched


def test_wgrad_padded_valid_region():
    call, ptr, sp = _abi()
    B, H, W, Cin, Cv, Cp = 2, 6, 6, 64, 27, 32
    x = rnd(B * H * W, Cin, seed=21).to(DEV)
    dy = torch.zeros(B * H * W, Cp, dtype=BF)
    dy[:, :Cv] = rnd(B * H * W, Cv, seed=22)
    dw = torch.zeros(Cv, Cin, device=DEV)
    dyd = dy.to(DEV)
    call("conv_wgrad_bf16", ptr(x), Cin, ptr(dyd), Cp, ptr(dw), Cin, Cv, Cin, B, H, W, Cin, Cp, 1, 1, sp())
    close(dw, dy.float()[:, :Cv].t() @ x.float().cpu(), rel=5e-3)


@pytest.mark.parametrize("M,C,res", [(400, 64, False), (1000, 24, True), (130, 512, True), (33, 8, False), (32000, 256, False),
                                     (128000, 128, True)])
def test_bn_silu_fwd_bwd(M, C, res):
    call, ptr, sp = _abi()
    z = rnd(M, C, seed=31, scale=2.0)
    gamma = torch.rand(C) + 0.5
    beta = torch.rand(C) - 0.5
    r = rnd(M, C, seed=32) if res else None
    zr = z.float().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    rm, rv = torch.zeros(C), torch.ones(C)
    u = F.batch_norm(zr, rm, rv, g_, b_, True, 0.03, 1e-3)
    y_ref = F.silu(u) + (r.float() if res else 0)
    dy = rnd(M, C, seed=33)
    y_ref.backward(dy.float())

    zd = z.to(DEV)
    R = 2
    stats = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
    stats[0, 0] = (z.double().sum(0) * 2 ** 20).round().long().to(DEV)
    stats[0, 1] = ((z.double() ** 2).sum(0) * 2 ** 20).round().long().to(DEV)
    save = torch.zeros(2, C, device=DEV)
    rmd, rvd, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    y = torch.zeros(M, C, dtype=BF, device=DEV)
    gd, bd = gamma.to(DEV), beta.to(DEV)
    rd = r.to(DEV) if res else None
    nbt2 = torch.full((), 5, dtype=torch.int64, device=DEV)
    call("bn_act_fwd", ptr(zd), C, ptr(stats), R, ptr(gd), ptr(bd), ptr(rmd), ptr(rvd), ptr(nbt), ptr(nbt2), ptr(save), ptr(y), C,
         ptr(rd), C if res else 0, M, C, 1e-3, 0.03, 1, sp())
    close(y, y_ref.detach())
    close(rmd, rm, rel=1e-4)
    close(rvd, rv, rel=1e-4)
    assert int(nbt) == 1 and int(nbt2) == 6
    dyd = dy.to(DEV)
    sums = torch.zeros(3, 2, C, dtype=torch.int64, device=DEV)          # three replicas of the two sums: [reps][2][C]
    ggrad, bgrad = torch.ones(C, device=DEV), torch.ones(C, device=DEV)
    call("bn_act_bwd_reduce", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), M, C, 1, 3, sp())
    dz = torch.zeros(M, C, dtype=BF, device=DEV)
    call("bn_act_bwd_apply", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), ptr(ggrad),
         ptr(bgrad), ptr(dz), C, M, C, 1, 3, sp())
    close(dz, zr.grad, rel=2e-2)
    close(ggrad - 1, g_.grad, rel=5e-3)
    close(bgrad - 1, b_.grad, rel=5e-3)


@pytest.mark.parametrize("bad", [float("nan"), float("inf"), -float("inf")])
def test_bn_backward_keeps_a_diverged_gradient_visible(bad):
    """one NaN / Inf in dy of ONE block's rows, with net-NEGATIVE sums everywhere else (the usual start of a
    divergence): the channel's gamma / beta gradients must read NaN, not a finite value decoded from 'marker minus the rest' -
    and every other channel stays finite and right."""
    call, ptr, sp = _abi()
    M, C = 32000, 256
    z = rnd(M, C, seed=41, scale=2.0)
    dy = -(rnd(M, C, seed=42).float().abs() + 0.5).to(BF)              # every du sum negative and large
    dy[12345, 7] = bad
    gamma, beta = torch.rand(C) + 0.5, torch.rand(C) - 0.5
    zd, dyd, gd, bd = z.to(DEV), dy.to(DEV), gamma.to(DEV), beta.to(DEV)
    R = 2
    stats = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
    stats[0, 0] = (z.double().sum(0) * 2 ** 20).round().long().to(DEV)
    stats[0, 1] = ((z.double() ** 2).sum(0) * 2 ** 20).round().long().to(DEV)
    save = torch.zeros(2, C, device=DEV)
    rmd, rvd, nbt = torch.zeros(C, device=DEV), torch.ones(C, device=DEV), torch.zeros((), dtype=torch.int64, device=DEV)
    y = torch.zeros(M, C, dtype=BF, device=DEV)
    call("bn_act_fwd", ptr(zd), C, ptr(stats), R, ptr(gd), ptr(bd), ptr(rmd), ptr(rvd), ptr(nbt), ptr(None), ptr(save), ptr(y), C,
         ptr(None), 0, M, C, 1e-3, 0.03, 1, sp())
    for reps in (1, 8):
        sums = torch.zeros(reps, 2, C, dtype=torch.int64, device=DEV)
        ggrad, bgrad = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        call("bn_act_bwd_reduce", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), M, C, 1, reps, sp())
        dz = torch.zeros(M, C, dtype=BF, device=DEV)
        call("bn_act_bwd_apply", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), ptr(ggrad),
             ptr(bgrad), ptr(dz), C, M, C, 1, reps, sp())
        g, b = ggrad.cpu(), bgrad.cpu()
        assert torch.isnan(g[7]) and torch.isnan(b[7]), (reps, float(g[7]), float(b[7]))
        ok = torch.ones(C, dtype=torch.bool)
        ok[7] = False
        assert torch.isfinite(g[ok]).all() and torch.isfinite(b[ok]).all()
        assert float(b[ok].max()) < 0                                  # the negative rest is really there


@pytest.mark.parametrize("M,C", [(400, 64), (32000, 256), (128000, 128), (8000, 1024), (2048000 // 8, 64)])
def test_bn_backward_one_launch(M, C):
    """Round 5: ep24_bn_act_bwd_fused = reduce + grid-wide wait + apply in one launch.  Against torch fp32 like the two-launch form, against the
    two-launch form itself (its sums are folded from other per-block partials: equal to fp32 rounding, not bit for bit), reproducible run to
    run, the wait's counter reaching the grid size, and no give-up of the bounded spin."""
    call, ptr, sp = _abi()
    from ep24 import _lib
    z = rnd(M, C, seed=51, scale=2.0)
    dy = rnd(M, C, seed=52)
    gamma, beta = torch.rand(C) + 0.5, torch.rand(C) - 0.5
    zr = z.float().requires_grad_(True)
    g_, b_ = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    y_ref = F.silu(F.batch_norm(zr, torch.zeros(C), torch.ones(C), g_, b_, True, 0.03, 1e-3))
    y_ref.backward(dy.float())
    zd, dyd, gd, bd = z.to(DEV), dy.to(DEV), gamma.to(DEV), beta.to(DEV)
    R = 8
    stats = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
    stats[0, 0] = (z.double().sum(0) * 2 ** 20).round().long().to(DEV)
    stats[0, 1] = ((z.double() ** 2).sum(0) * 2 ** 20).round().long().to(DEV)
    save = torch.zeros(2, C, device=DEV)
    y = torch.zeros(M, C, dtype=BF, device=DEV)
    call("bn_act_fwd", ptr(zd), C, ptr(stats), R, ptr(gd), ptr(bd), ptr(torch.zeros(C, device=DEV)), ptr(torch.ones(C, device=DEV)),
         ptr(torch.zeros((), dtype=torch.int64, device=DEV)), None, ptr(save), ptr(y), C, None, 0, M, C, 1e-3, 0.03, 1, sp())

    def two():
        sums = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
        gg, bg = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dz = torch.zeros(M, C, dtype=BF, device=DEV)
        call("bn_act_bwd_reduce", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), M, C, 1, R, sp())
        call("bn_act_bwd_apply", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), ptr(gg), ptr(bg), ptr(dz), C, M, C, 1, R, sp())
        return dz, gg, bg

    def one():
        sums = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
        gg, bg = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
        dz = torch.zeros(M, C, dtype=BF, device=DEV)
        bar = torch.zeros(2, dtype=torch.int32, device=DEV)
        call("bn_act_bwd_fused", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), ptr(gg), ptr(bg), ptr(dz), C, M, C, 1, R,
             ptr(bar), sp())
        torch.cuda.synchronize()
        return dz, gg, bg, int(bar[0])

    dz2, gg2, bg2 = two()
    dz1, gg1, bg1, arrived = one()
    assert 1 <= arrived <= 256                                        # every workgroup of the (one per CU at most) grid arrived
    close(dz1, zr.grad, rel=2e-2)
    close(gg1, g_.grad, rel=5e-3)
    close(bg1, b_.grad, rel=5e-3)
    close(gg1, gg2.cpu(), rel=1e-4)
    close(bg1, bg2.cpu(), rel=1e-4)
    assert float((dz1.float() - dz2.float()).abs().max()) <= 2.0 ** -7 * float(dz2.float().abs().max())      # a bf16 ulp of the tensor's range
    again = one()
    assert torch.equal(again[0], dz1) and torch.equal(again[1], gg1) and torch.equal(again[2], bg1)
    assert _lib.lib().fn["ep24_conv_ring_timeouts"]() == 0


def test_spp_fwd_bwd():
    call, ptr, sp = _abi()
    B, H, W, C = 2, 20, 20, 16
    x = rnd(B, C, H, W, seed=41)
    x = (x.float() * 4).round().div(4).to(BF)                            # many exact ties: exercises first-max routing
    xr = x.float().requires_grad_(True)
    pools = [F.max_pool2d(xr, k, 1, k // 2) for k in (5, 9, 13)]
    gys = [rnd(B, C, H, W, seed=42 + i) for i in range(3)]
    sum((p * g.float()).sum() for p, g in zip(pools, gys)).backward()
    cat = torch.zeros(B * H * W, 4 * C, dtype=BF, device=DEV)
    cat[:, :C] = nhwc(x).reshape(-1, C).to(DEV)
    idx = torch.zeros(3 * B * H * W * C, dtype=torch.uint8, device=DEV)
    call("spp_fwd", ptr(cat), 4 * C, ptr(cat, C), ptr(cat, 2 * C), ptr(cat, 3 * C), 4 * C, ptr(idx), B, H, W, C, None, sp())
    # the separable two-pass form must give the same maps and the same winner codes
    cat2 = cat.clone()
    cat2[:, C:] = 0
    idx2 = torch.zeros_like(idx)
    scratch = torch.zeros(9 * B * H * W * C, dtype=torch.uint8, device=DEV)
    call("spp_fwd", ptr(cat2), 4 * C, ptr(cat2, C), ptr(cat2, 2 * C), ptr(cat2, 3 * C), 4 * C, ptr(idx2), B, H, W, C, ptr(scratch), sp())
    assert torch.equal(cat2, cat) and torch.equal(idx2, idx)
    for i, p in enumerate(pools):
        got = cat[:, (i + 1) * C:(i + 2) * C].reshape(B, H, W, C).permute(0, 3, 1, 2)
        assert torch.equal(got.float().cpu(), p.detach())
    dcat = torch.zeros(B * H * W, 4 * C, dtype=BF, device=DEV)
    for i, g in enumerate(gys):
        dcat[:, (i + 1) * C:(i + 2) * C] = nhwc(g).reshape(-1, C).to(DEV)
    call("spp_bwd", ptr(dcat, C), ptr(dcat, 2 * C), ptr(dcat, 3 * C), 4 * C, ptr(idx), ptr(dcat), 4 * C, 0, B, H, W, C, sp())
    close(dcat[:, :C].reshape(B, H, W, C).permute(0, 3, 1, 2), xr.grad)


def test_upsample_stem_decode_sgd():
    call, ptr, sp = _abi()
    # nearest x2
    B, H, W, C = 2, 5, 5, 16
    x = rnd(B, C, H, W, seed=51)
    y = torch.zeros(B, 2 * H, 2 * W, C, dtype=BF, device=DEV)
    xd = nhwc(x).to(DEV)
    call("upsample2_fwd", ptr(xd), C, ptr(y), C, B, H, W, C, sp())
    assert torch.equal(y.permute(0, 3, 1, 2).float().cpu(), F.interpolate(x.float(), scale_factor=2, mode="nearest"))
    gy = rnd(B, 2 * H, 2 * W, C, seed=52).to(DEV)
    gx = torch.zeros(B, H, W, C, dtype=BF, device=DEV)
    call("upsample2_bwd", ptr(gy), C, ptr(gx), C, 0, B, H, W, C, sp())
    ref = gy.float().cpu().reshape(B, H, 2, W, 2, C).sum((2, 4))
    close(gx, ref)
    # stem: Focus + 3x3 im2col
    S = 16
    img = torch.rand(2, 3, S, S, generator=torch.Generator().manual_seed(53)) * 255
    rows = torch.zeros(2 * 8 * 8, 112, dtype=BF, device=DEV)
    imgd = img.to(DEV)
    call("stem_pack", ptr(imgd), ptr(rows), 112, 2, S, S, sp())
    foc = torch.cat([img[..., 0::2, 0::2], img[..., 1::2, 0::2], img[..., 0::2, 1::2], img[..., 1::2, 1::2]], 1)
    cols = F.unfold(foc, 3, padding=1).reshape(2, 12, 9, 64).permute(0, 3, 2, 1).reshape(-1, 108)   # (kh,kw) major, c minor
    assert torch.equal(rows[:, :108].float().cpu(), cols.to(BF).float())
    assert float(rows[:, 108:].abs().sum()) == 0
    # decode fwd / bwd of one level
    Bz, A, a0, Hh, Ww, s, nc = 2, 30, 5, 4, 4, 16.0, 107
    raw = torch.randn(Bz, A, nc, generator=torch.Generator().manual_seed(54))
    out = raw.clone().to(DEV)
    call("head_decode_fwd", ptr(out), Bz, A, a0, Hh, Ww, s, nc, None, sp())
    yv, xv = torch.meshgrid(torch.arange(Hh), torch.arange(Ww), indexing="ij")
    ref = raw.clone()
    lvl = ref[:, a0:a0 + 16]
    lvl[..., 0] = (lvl[..., 0] + xv.reshape(-1)) * s
    lvl[..., 1] = (lvl[..., 1] + yv.reshape(-1)) * s
    lvl[..., 2:26] = torch.exp(lvl[..., 2:26]) * s
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-6)
    dout = torch.randn(Bz, A, nc, generator=torch.Generator().manual_seed(55)).to(DEV)
    dro = torch.zeros(Bz * 16, 32, dtype=BF, device=DEV)
    dcl = torch.zeros(Bz * 16, 80, dtype=BF, device=DEV)
    call("head_decode_bwd", ptr(dout), ptr(out), ptr(dro), ptr(dcl), Bz, A, a0, Hh, Ww, s, nc, None, sp())
    d = dout.cpu()[:, a0:a0 + 16]
    o = out.cpu()[:, a0:a0 + 16]
    want = torch.cat([d[..., :2] * s, d[..., 2:26] * o[..., 2:26], d[..., 26:27]], -1).reshape(-1, 27)
    close(dro[:, :27], want, rel=5e-3)
    close(dcl, d[..., 27:].reshape(-1, 80), rel=5e-3)
    assert float(dro[:, 27:].abs().sum()) == 0
    # SGD nesterov, two steps vs torch.optim.SGD
    n = 1003
    p0 = torch.randn(n, generator=torch.Generator().manual_seed(56))
    pt = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([pt], lr=0.01, momentum=0.9, nesterov=True)
    p, buf, first = torch.zeros(1004, device=DEV), torch.zeros(1004, device=DEV), torch.ones(1, dtype=torch.int32, device=DEV)
    p[:n] = p0.to(DEV)
    for step in range(2):
        g = torch.randn(n, generator=torch.Generator().manual_seed(57 + step))
        pt.grad = g.clone()
        opt.step()
        gd = torch.zeros(1004, device=DEV)
        gd[:n] = g.to(DEV)
        call("sgd_nesterov", ptr(p), ptr(gd), ptr(buf), 1004, 0.01, 0.9, 1.0, ptr(first), sp())
        torch.testing.assert_close(p[:n].cpu(), pt.detach(), rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("B,H,W,Cin,Cout,k,act,with_res", [(20, 40, 40, 256, 256, 3, 1, True), (20, 80, 80, 128, 128, 3, 1, False), (12, 52, 100, 64, 128, 3, 1, True),
                                                            (20, 20, 20, 512, 512, 3, 1, True), (4, 40, 40, 256, 256, 1, 1, True)])
def test_conv_infer_unit(B, H, W, Cin, Cout, k, act, with_res):
    """The eval-mode unit y = act(conv(x) + bias) + residual (ep24_conv_fwd_infer_bf16) on shapes of every kernel that
    carries its epilogue - the loader / consumer ring (round 4: 3x3 stride-1 with >= 200 tiles of 256 x 128), the tiled kernel (the
    20 x 20 level) and the 1x1 path - against torch on the bf16-rounded operands."""
    call, ptr, sp = _abi()
    pad = (k - 1) // 2
    x = rnd(B, Cin, H, W, seed=11)
    w = rnd(Cout, Cin, k, k, seed=12, scale=(Cin * k * k) ** -0.5)
    bias = torch.randn(Cout, generator=torch.Generator().manual_seed(13))
    res = rnd(B, Cout, H, W, seed=14) if with_res else None
    ref = F.conv2d(x.float().to(DEV), w.float().to(DEV), bias.to(DEV), 1, pad)
    ref = ref * torch.sigmoid(ref) if act == 1 else ref
    if with_res:
        ref = ref + res.float().to(DEV)
    xd, wf = nhwc(x).to(DEV), w.permute(0, 2, 3, 1).contiguous().to(DEV)
    rd = nhwc(res).to(DEV) if with_res else None
    y = torch.zeros(B, H, W, Cout, dtype=BF, device=DEV)
    call("conv_fwd_infer_bf16", ptr(xd), Cin, ptr(wf), ptr(bias.to(DEV)), act, ptr(rd) if with_res else None, Cout, ptr(y), Cout, B, H, W, Cin, Cout, k, 1, sp())
    torch.cuda.synchronize()
    close(y.permute(0, 3, 1, 2), ref.cpu())
    from ep24 import _lib
    assert _lib.lib().fn["ep24_conv_ring_timeouts"]() == 0
