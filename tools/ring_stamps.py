"""GPU box helper: where the time goes INSIDE the loader / consumer ring kernel (csrc/conv_ring.hip).  Needs the diagnostic library
(make -C csrc stamps), which records s_memtime stamps in consumer waves 0 and 3 of the first 32 workgroups: prologue / main loop /
epilogue cycles, the cycles a consumer spent waiting for the loaders' FULL counters (and how many steps had to wait at all), and the
in-kernel clock (s_memtime over s_memrealtime, the latter ticks at 100 MHz).
usage: EP24_LIB=.../libep24_stamps.so ring_stamps.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "exploration-of-potential_amd"))
import torch  # noqa: E402
from ep24 import _lib  # noqa: E402
from ep24._lib import call, ptr, stream_ptr  # noqa: E402

DEV = "cuda:0"


def main():
    L = _lib.lib()
    rd = L.cdll.ep24_debug_read_ring_stamps
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    print("shape                steps  prologue   loop  epilogue | cycles per step  of which waiting for FULL  steps that waited | clock GHz | kernel us | MFMA issue share of the loop")
    # the last three: the narrow tile (256 x 64; its MFMA issue per step is 512 cycles, so the last column reads half of the real share)
    for B, H, Cin, Cout in [(20, 40, 256, 256), (20, 80, 128, 128), (20, 80, 256, 256), (20, 40, 256, 512), (20, 20, 512, 512), (20, 20, 256, 256), (20, 160, 64, 64)]:
        W = H
        x = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
        w = (torch.randn(Cout, 9, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        y = torch.zeros(B * H * W, Cout, device=DEV, dtype=torch.bfloat16)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        for _ in range(50):                                  # sustained load: the clock under load, not the idle clock
            call("conv_fwd_bf16_ex", ptr(x), Cin, ptr(w), ptr(y), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, 3, 1,
                 int(os.environ.get("EP24_STAMP_OPTS", "0")), stream_ptr())
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 512)()
        assert rd(buf, 512) == 0
        for which, name in ((0, "consumer 0"), (1, "consumer 3")):
            rows = [buf[(i * 2 + which) * 8:(i * 2 + which + 1) * 8] for i in range(32)]
            med = [sorted(r[k] for r in rows)[16] for k in range(8)]
            n = max(med[7], 1)
            clk = med[6] / max(med[5], 1) * 0.1
            print("%-20s %5d %9d %6d %9d | %15.0f %25.0f %18d | %9.2f | %9.1f | %.2f  %s" % (
                "%d,%d,%d,%d" % (B, H, Cin, Cout), n, med[0], med[1], med[2], med[1] / n, med[3] / n, med[4], clk, med[5] / 100.0,
                1024.0 * n / max(med[1], 1), name), flush=True)
    print("ring timeouts:", L.fn["ep24_conv_ring_timeouts"]())


main()
