"""The streaming 1x1 kernel with a transformed A operand (round 5: ep24_conv1x1_bnin_bf16, ep24_conv1x1_dgrad_bnbwd_bf16) against
the two launches it replaces, through the C ABI.  The claim is bit identity - the fused kernel evaluates the BatchNorm kernels'
expressions on the rows in flight and multiplies the bf16 values it stores - so every comparison here is torch.equal; what the two
launches themselves compute is pinned against torch fp32 in tests/test_gpu_conv.py."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def _abi():
    from ep24._lib import call, ptr, stream_ptr
    return call, ptr, stream_ptr


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF)


def _stats_of(z, R):
    C = z.shape[1]
    stats = torch.zeros(R, 2, C, dtype=torch.int64, device=DEV)
    # spread over the replicas as the conv epilogues do (any split gives the same fold: integers)
    half = z.shape[0] // 2
    for r, part in enumerate((z[:half], z[half:])):
        stats[r % R, 0] += (part.double().sum(0) * 2 ** 20).round().long().to(DEV)
        stats[r % R, 1] += ((part.double() ** 2).sum(0) * 2 ** 20).round().long().to(DEV)
    return stats


# B, H, W, Cin, Cout, residual, strided rows (operands as channel slices of wider buffers)
FWD_CASES = [(20, 80, 80, 128, 128, True, False), (2, 160, 160, 64, 64, True, False), (3, 13, 17, 128, 48, False, False),
             (2, 24, 24, 112, 64, True, True), (1, 33, 9, 24, 128, False, True), (4, 40, 40, 64, 128, True, False),
             (1, 5, 5, 8, 8, False, False)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,res,strided", FWD_CASES)
def test_bnin_forward_is_the_two_launches(B, H, W, Cin, Cout, res, strided):
    call, ptr, sp = _abi()
    from ep24._lib import lib
    assert lib().fn["ep24_conv1x1_xf_ok"](B, H, W, Cin, Cout) == 1
    M, R = B * H * W, 8
    ldz = Cin + 16 if strided else Cin                     # z / y / residual rows inside wider buffers, the conv output too
    ldo = Cout + 8 if strided else Cout
    c0 = 8 if strided else 0
    zbuf = torch.zeros(M, ldz, dtype=BF)
    zbuf[:, c0:c0 + Cin] = rnd(M, Cin, seed=1, scale=2.0)
    rbuf = torch.zeros(M, ldz, dtype=BF)
    rbuf[:, c0:c0 + Cin] = rnd(M, Cin, seed=2)
    zd, rd = zbuf.to(DEV), rbuf.to(DEV)
    z = zd[:, c0:c0 + Cin]
    gamma, beta = (torch.rand(Cin) + 0.5).to(DEV), (torch.rand(Cin) - 0.5).to(DEV)
    w = rnd(Cout, Cin, seed=3, scale=Cin ** -0.5).to(DEV)
    stats_in = _stats_of(z.float().cpu(), R)

    def run(fused):
        save = torch.zeros(2, Cin, device=DEV)
        rm, rv = torch.full((Cin,), 0.25, device=DEV), torch.full((Cin,), 1.5, device=DEV)
        nbt, nbt2 = torch.zeros((), dtype=torch.int64, device=DEV), torch.full((), 5, dtype=torch.int64, device=DEV)
        y = torch.full((M, ldz), 7.0, dtype=BF, device=DEV)                 # the fill must survive outside the slice
        out = torch.full((M, ldo), 3.0, dtype=BF, device=DEV)
        stats_out = torch.zeros(R, 2, Cout, dtype=torch.int64, device=DEV)
        yp, zp, rp, op = ptr(y, c0), ptr(zd, c0), (ptr(rd, c0) if res else None), ptr(out, 4 if strided else 0)
        if fused:
            call("conv1x1_bnin_bf16", zp, ldz, ptr(stats_in), R, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), ptr(nbt2), ptr(save),
                 yp, ldz, rp, ldz if res else 0, 1e-3, 0.03, 1, ptr(w), op, ldo, ptr(stats_out), R, B, H, W, Cin, Cout, sp())
        else:
            call("bn_act_fwd", zp, ldz, ptr(stats_in), R, ptr(gamma), ptr(beta), ptr(rm), ptr(rv), ptr(nbt), ptr(nbt2), ptr(save),
                 yp, ldz, rp, ldz if res else 0, M, Cin, 1e-3, 0.03, 1, sp())
            call("conv_fwd_bf16", yp, ldz, ptr(w), op, ldo, 0, 0, 0, None, ptr(stats_out), R, B, H, W, Cin, Cout, 1, 1, sp())
        torch.cuda.synchronize()
        return y, out, stats_out.sum(0), save, rm, rv, int(nbt), int(nbt2)

    a, b = run(False), run(True)
    names = ("y", "conv output", "output statistics", "save", "running_mean", "running_var", "num_batches", "num_batches2")
    for name, u, v in zip(names, a, b):
        ok = torch.equal(u, v) if torch.is_tensor(u) else u == v
        assert ok, "%s differs between the fused launch and the two launches" % name
    assert a[6] == 1 and a[7] == 6
    assert float(b[1].float().abs().max()) > 0.1                              # the comparison is not between two empty results


# B, H, W, Cin (forward input channels = N of the input gradient), Cout (K), accumulate, strided
BWD_CASES = [(20, 80, 80, 128, 128, 1, False), (20, 80, 80, 128, 128, 0, False), (2, 160, 160, 64, 64, 1, False),
             (3, 13, 17, 48, 128, 0, True), (2, 24, 24, 64, 112, 1, True), (1, 5, 5, 8, 8, 0, False)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,acc,strided", BWD_CASES)
def test_dgrad_bnbwd_is_the_two_launches(B, H, W, Cin, Cout, acc, strided):
    call, ptr, sp = _abi()
    M, R = B * H * W, 8
    ld = Cout + 16 if strided else Cout
    ldx = Cin + 8 if strided else Cin
    c0 = 8 if strided else 0
    dybuf, zbuf = torch.zeros(M, ld, dtype=BF), torch.zeros(M, ld, dtype=BF)
    dybuf[:, c0:c0 + Cout] = rnd(M, Cout, seed=11)
    zbuf[:, c0:c0 + Cout] = rnd(M, Cout, seed=12, scale=2.0)
    dyd, zd = dybuf.to(DEV), zbuf.to(DEV)
    gamma, beta = (torch.rand(Cout) + 0.5).to(DEV), (torch.rand(Cout) - 0.5).to(DEV)
    wt = rnd(Cin, Cout, seed=13, scale=Cout ** -0.5).to(DEV)               # [Cin][1][Cout]: the input gradient's weight copy
    dx0 = rnd(M, ldx, seed=14).to(DEV)
    # forward statistics -> save, then the reduce pass: both exactly as the step runs them
    z = zd[:, c0:c0 + Cout]
    stats = _stats_of(z.float().cpu(), R)
    save = torch.zeros(2, Cout, device=DEV)
    ytmp = torch.zeros(M, Cout, dtype=BF, device=DEV)
    call("bn_act_fwd", ptr(zd, c0), ld, ptr(stats), R, ptr(gamma), ptr(beta), None, None, None, None, ptr(save), ptr(ytmp), Cout,
         None, 0, M, Cout, 1e-3, 0.03, 1, sp())
    sums = torch.zeros(R, 2, Cout, dtype=torch.int64, device=DEV)
    call("bn_act_bwd_reduce", ptr(dyd, c0), ld, ptr(zd, c0), ld, ptr(save), ptr(gamma), ptr(beta), ptr(sums), ptr(sums, Cout), M, Cout, 1, R, sp())

    def run(fused):
        gg, bg = torch.full((Cout,), 0.5, device=DEV), torch.full((Cout,), -0.25, device=DEV)
        dz = torch.full((M, Cout), 9.0, dtype=BF, device=DEV)
        dx = dx0.clone()
        if fused:
            call("conv1x1_dgrad_bnbwd_bf16", ptr(dyd, c0), ld, ptr(zd, c0), ld, ptr(save), ptr(gamma), ptr(beta), ptr(sums), ptr(sums, Cout),
                 ptr(gg), ptr(bg), ptr(dz), Cout, 1, R, ptr(wt), ptr(dx, 4 if strided else 0), ldx, acc, B, H, W, Cin, Cout, sp())
        else:
            call("bn_act_bwd_apply", ptr(dyd, c0), ld, ptr(zd, c0), ld, ptr(save), ptr(gamma), ptr(beta), ptr(sums), ptr(sums, Cout),
                 ptr(gg), ptr(bg), ptr(dz), Cout, M, Cout, 1, R, sp())
            call("conv_dgrad_bf16", ptr(dz), Cout, ptr(wt), ptr(dx, 4 if strided else 0), ldx, acc, B, H, W, Cin, Cout, 1, 1, sp())
        torch.cuda.synchronize()
        return dz, dx, gg, bg

    a, b = run(False), run(True)
    for name, u, v in zip(("dz", "dx", "gamma gradient", "beta gradient"), a, b):
        assert torch.equal(u, v), "%s differs between the fused launch and the two launches" % name
    assert not torch.equal(a[1], dx0)


def test_xf_refuses_what_it_does_not_take():
    from ep24._lib import lib, ptr, stream_ptr
    fn = lib().fn
    assert fn["ep24_conv1x1_xf_ok"](20, 40, 40, 256, 256) == 0 and fn["ep24_conv1x1_xf_ok"](20, 80, 80, 128, 256) == 0
    t = torch.zeros(64, dtype=BF, device=DEV)
    f = torch.zeros(64, device=DEV)
    i = torch.zeros(64, dtype=torch.int64, device=DEV)
    rc = fn["ep24_conv1x1_bnin_bf16"](ptr(t), 256, ptr(i), 1, ptr(f), ptr(f), None, None, None, None, ptr(f), ptr(t), 256, None, 0, 1e-3, 0.03, 1,
                                      ptr(t), ptr(t), 256, None, 1, 1, 4, 4, 256, 256, stream_ptr())
    assert rc != 0 and "transformed-A" in lib().last_error()
    rc = fn["ep24_conv1x1_bnin_bf16"](ptr(t), 64, ptr(i), 1, ptr(f), ptr(f), None, None, None, None, ptr(f), ptr(t), 64, None, 0, 1e-3, 0.03, 2,
                                      ptr(t), ptr(t), 64, None, 1, 1, 1, 1, 64, 64, stream_ptr())
    assert rc != 0 and "SiLU" in lib().last_error()


# the input gradient of a 1x1 conv of the streaming kernel with the reduce pass of the unit below in its epilogue (ep24_conv1x1_dgrad_bnr_bf16)
# B, H, W, Cin (= channels of the unit below), Cout (K), accumulate
BNR_CASES = [(20, 40, 40, 256, 256, 1), (20, 40, 40, 256, 256, 0), (4, 20, 20, 256, 256, 1), (2, 24, 24, 64, 192, 0), (3, 13, 17, 48, 160, 1), (20, 80, 80, 128, 128, 1)]


@pytest.mark.parametrize("B,H,W,Cin,Cout,acc", BNR_CASES)
def test_dgrad_with_the_reduce_of_the_unit_below(B, H, W, Cin, Cout, acc):
    """dx must be what ep24_conv_dgrad_bf16 stores, bit for bit (same products, same order); the two sums must be what
    ep24_bn_act_bwd_reduce makes of that dx and the unit's z, up to the fp32 order of the partial sums (2^-36 fixed point both)."""
    call, ptr, sp = _abi()
    M, R = B * H * W, 8
    dz = rnd(M, Cout, seed=21).to(DEV)
    wt = rnd(Cin, Cout, seed=22, scale=Cout ** -0.5).to(DEV)
    dx0 = rnd(M, Cin, seed=23).to(DEV)
    zb = rnd(M, Cin, seed=24, scale=2.0).to(DEV)                              # the unit below: its raw conv output and BatchNorm
    gamma, beta = (torch.rand(Cin) + 0.5).to(DEV), (torch.rand(Cin) - 0.5).to(DEV)
    stats = _stats_of(zb.float().cpu(), R)
    save = torch.zeros(2, Cin, device=DEV)
    ytmp = torch.zeros(M, Cin, dtype=BF, device=DEV)
    call("bn_act_fwd", ptr(zb), Cin, ptr(stats), R, ptr(gamma), ptr(beta), None, None, None, None, ptr(save), ptr(ytmp), Cin, None, 0, M, Cin, 1e-3, 0.03, 1, sp())
    # the two launches
    dx_a = dx0.clone()
    call("conv_dgrad_bf16", ptr(dz), Cout, ptr(wt), ptr(dx_a), Cin, acc, B, H, W, Cin, Cout, 1, 1, sp())
    sums_a = torch.zeros(R, 2, Cin, dtype=torch.int64, device=DEV)
    call("bn_act_bwd_reduce", ptr(dx_a), Cin, ptr(zb), Cin, ptr(save), ptr(gamma), ptr(beta), ptr(sums_a), ptr(sums_a, Cin), M, Cin, 1, R, sp())
    # the one launch
    dx_b = dx0.clone()
    sums_b = torch.zeros(R, 2, Cin, dtype=torch.int64, device=DEV)
    call("conv1x1_dgrad_bnr_bf16", ptr(dz), Cout, ptr(wt), ptr(dx_b), Cin, acc, B, H, W, Cin, Cout, ptr(zb), Cin, ptr(save), ptr(save, Cin), ptr(gamma), ptr(beta),
         ptr(sums_b), ptr(sums_b, Cin), 2 * Cin, R, 1, sp())
    torch.cuda.synchronize()
    assert torch.equal(dx_a, dx_b)
    a, b = sums_a.sum(0).double() / 2 ** 36, sums_b.sum(0).double() / 2 ** 36
    assert float(a.abs().max()) > 1.0
    err = float((a - b).abs().max() / a.abs().max())
    assert err < 2e-5, err


def test_dgrad_bnr_refuses_what_the_streaming_kernel_does_not_take():
    from ep24._lib import lib, ptr, stream_ptr
    fn = lib().fn
    t = torch.zeros(4096, dtype=BF, device=DEV)
    f = torch.zeros(4096, device=DEV)
    i = torch.zeros(4096, dtype=torch.int64, device=DEV)
    rc = fn["ep24_conv1x1_dgrad_bnr_bf16"](ptr(t), 512, ptr(t), ptr(t), 64, 0, 1, 2, 2, 64, 512, ptr(t), 64, ptr(f), ptr(f), ptr(f), ptr(f), ptr(i), ptr(i), 128, 1, 1, stream_ptr())
    assert rc != 0 and "streaming" in lib().last_error()
