import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch
from ep24._lib import call, ptr, stream_ptr
DEV = "cuda:0"
torch.manual_seed(0)
B, H, W, C = 20, 20, 20, 512
M = B * H * W
x = torch.randn(M, C, device=DEV).to(torch.bfloat16)
cat = torch.zeros(M, 4 * C, device=DEV, dtype=torch.bfloat16)
idx = torch.zeros(3, M, C, dtype=torch.uint8, device=DEV)
scratch = torch.zeros(9 * M * C // 1, dtype=torch.uint8, device=DEV)
y5, y9, y13 = cat[:, C:2 * C], cat[:, 2 * C:3 * C], cat[:, 3 * C:]
call("spp_fwd", ptr(x), C, ptr(y5), ptr(y9), ptr(y13), 4 * C, ptr(idx), B, H, W, C, ptr(scratch), stream_ptr())
g = torch.randn(M, 4 * C, device=DEV).to(torch.bfloat16)
dx = torch.randn(M, C, device=DEV).to(torch.bfloat16)
for acc in (0, 1):
    d = dx.clone()
    call("spp_bwd", ptr(g[:, C:2 * C]), ptr(g[:, 2 * C:3 * C]), ptr(g[:, 3 * C:]), 4 * C, ptr(idx), ptr(d), C, acc, B, H, W, C, stream_ptr())
    torch.cuda.synchronize()
    print("acc", acc, "checksum", int(d.view(torch.int16).sum(dtype=torch.int64)), float(d.float().abs().sum()))
    if len(sys.argv) > 1:
        torch.save(d.cpu(), sys.argv[1] + ".%d.pt" % acc)
import time
torch.cuda.synchronize(); t0 = time.time()
for _ in range(50):
    call("spp_bwd", ptr(g[:, C:2 * C]), ptr(g[:, 2 * C:3 * C]), ptr(g[:, 3 * C:]), 4 * C, ptr(idx), ptr(dx), C, 0, B, H, W, C, stream_ptr())
torch.cuda.synchronize(); print("us per launch %.1f" % ((time.time() - t0) / 50 * 1e6))
