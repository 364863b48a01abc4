"""The geometry kernels next to MFMA kernels (round 2: profiles/r02_candidate_mask_hazard.txt; root cause round 3:
profiles/r03_packed_fp32_hazard.txt, tools/hazard_probe.hip).

The SimOTA candidate-mask kernel (pts_in_poly, yolox_24p/models/losses.py:555-592: 24 atan2 per anchor and GT) returned
different masks in lanes 48..63 of a wave when a conv kernel of another stream shared its SIMDs.  The cause was a packed fp32
multiply with swapped operand halves (v_pk_mul_f32 op_sel) that the SLP vectoriser had made of the cross / dot products; the
library is now built without it.  This test runs the kernel - and the cost kernel with its acos / sin chains - on a second
stream WHILE conv kernels run, once, and requires the results of the quiet run bit for bit (no retry loop: the old build
deviated in a few launches out of a hundred)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_candidate_and_cost_kernels_beside_mfma_kernels_equal_the_quiet_run():
    from ep24 import loss as eloss, synth
    from ep24._lib import call, ptr
    B, S, G = 20, 640, 10
    labels = synth.make_labels(B, G, size=S, seed=1000).to(DEV)
    xs, ys, st = [], [], []
    for s in (8, 16, 32):
        h = S // s
        yv, xv = torch.meshgrid(torch.arange(h), torch.arange(h), indexing="ij")
        xs.append(xv.reshape(-1).float()); ys.append(yv.reshape(-1).float()); st.append(torch.full((h * h,), float(s)))
    xs, ys, st = torch.cat(xs).to(DEV), torch.cat(ys).to(DEV), torch.cat(st).to(DEV)
    A = xs.numel()
    lf = eloss.Loss_Function(80)
    ws = lf.workspace(B, A, torch.device(DEV))
    g = torch.Generator().manual_seed(5)
    outputs = torch.randn(B, A, 107, generator=g).to(DEV)
    outputs[..., 0:2] = torch.rand(B, A, 2, generator=g).to(DEV) * S
    outputs[..., 2:26] = 20 + 60 * torch.rand(B, A, 24, generator=g).to(DEV)

    def geometry():
        eloss.assign_candidates(ws, labels, xs, ys, st)
        call("assign_cost", ptr(outputs), 107, ptr(labels), ptr(ws.num_gt), ptr(ws.masks[0]), ptr(ws.masks[1]), ptr(ws.pw), ptr(ws.cost),
             B, A, 80, torch.cuda.current_stream().cuda_stream)
        return ws.masks[0].clone(), ws.masks[1].clone(), ws.pw.clone(), ws.cost.clone()

    quiet = geometry()
    torch.cuda.synchronize()
    assert int((quiet[0] != 0).sum()) > 1000                      # the masks are not trivially empty

    # the neighbour: the 3x3 conv of the 40x40 level (halo-patch MFMA kernel, a one-round grid on every CU), back to back
    Bc, H, C = 20, 40, 256
    x = torch.randn(Bc * H * H, C, device=DEV).to(torch.bfloat16)
    w = (torch.randn(C, 9, C, device=DEV) * 0.05).to(torch.bfloat16)
    y = torch.zeros(Bc * H * H, C, device=DEV, dtype=torch.bfloat16)
    conv_stream, geo_stream = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(conv_stream):
        for _ in range(400):                                        # ~20 ms of MFMA work
            call("conv_fwd_bf16", ptr(x), C, ptr(w), ptr(y), C, 0, 0, 0, None, None, 1, Bc, H, H, C, C, 3, 1, conv_stream.cuda_stream)
    bad = 0
    with torch.cuda.stream(geo_stream):
        runs = [geometry() for _ in range(60)]                      # ~15 ms: inside the conv stream's busy time
    torch.cuda.synchronize()
    for r in runs:
        bad += sum(int(not torch.equal(a, b)) for a, b in zip(r, quiet))
    assert bad == 0, "%d of %d result tensors differ from the quiet run" % (bad, 4 * len(runs))
