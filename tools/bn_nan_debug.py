"""Debug helper (round 5): raw fixed-point sums of bn_act_bwd_reduce with one NaN in dy."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24 import _lib
call, ptr, sp = _lib.call, _lib.ptr, _lib.stream_ptr
DEV, BF = "cuda:0", torch.bfloat16
M, C = 32000, 256
g = torch.Generator().manual_seed(41)
z = (torch.randn(M, C, generator=g) * 2).to(BF)
dy = -(torch.randn(M, C, generator=g).abs() + 0.5).to(BF)
for bad in (None, float("nan"), float("inf")):
    d = dy.clone()
    if bad is not None:
        d[12345, 7] = bad
    print("bad", bad, "dy[12345,7] =", float(d[12345, 7]), "isnan count", int(torch.isnan(d.float()).sum()))
    zd, dyd = z.to(DEV), d.to(DEV)
    print("  on device isnan count", int(torch.isnan(dyd.float()).sum()), "isinf", int(torch.isinf(dyd.float()).sum()))
    gd, bd = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    save = torch.zeros(2, C, device=DEV)
    save[1] = 1.0
    sums = torch.zeros(1, 2, C, dtype=torch.int64, device=DEV)
    call("bn_act_bwd_reduce", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), M, C, 1, 1, sp())
    torch.cuda.synchronize()
    s = sums.cpu()
    print("  raw dgamma[7] = %d (2^%.2f), dbeta[7] = %d; dbeta[6] = %d" % (int(s[0, 0, 7]), __import__("math").log2(abs(int(s[0, 0, 7])) + 1), int(s[0, 1, 7]), int(s[0, 1, 6])))
    ggrad, bgrad = torch.zeros(C, device=DEV), torch.zeros(C, device=DEV)
    dz = torch.zeros(M, C, dtype=BF, device=DEV)
    call("bn_act_bwd_apply", ptr(dyd), C, ptr(zd), C, ptr(save), ptr(gd), ptr(bd), ptr(sums), ptr(sums, C), ptr(ggrad), ptr(bgrad), ptr(dz), C, M, C, 1, 1, sp())
    torch.cuda.synchronize()
    print("  ggrad[6:9]", ggrad[6:9].tolist(), "bgrad[6:9]", bgrad[6:9].tolist())
