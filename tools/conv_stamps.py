"""GPU box helper: where the time goes INSIDE the halo-patch kernel.  Needs the diagnostic library (make -C csrc stamps),
which records s_memtime stamps in wave 0 of the first 64 workgroups: prologue / main loop / epilogue cycles, the cycles
of the loop spent in the per-step wait + barrier and in the rest of the step, and the in-kernel clock (s_memtime over
s_memrealtime, the latter ticks at 100 MHz).
usage: EP24_LIB=.../libep24_stamps.so conv_stamps.py"""
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
    rd = L.cdll.ep24_debug_read_stamps
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    print("shape                steps  prologue   loop  epilogue | per step: wait+barrier  work | clock GHz | kernel us")
    for B, H, Cin, Cout in [(20, 40, 256, 256), (20, 80, 128, 128), (20, 20, 512, 512), (20, 80, 256, 256), (20, 160, 64, 64)]:
        W = H
        x = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
        w = (torch.randn(Cout, 9, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        y = torch.zeros(B * H * W, Cout, device=DEV, dtype=torch.bfloat16)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        for _ in range(50):                                  # sustained load: the clock under load, not the idle clock
            call("conv_fwd_bf16", ptr(x), Cin, ptr(w), ptr(y), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, 3, 1, stream_ptr())
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 512)()
        assert rd(buf, 512) == 0
        for which, name in ((0, "wave 0"), (1, "wave 7")):       # the oldest and the youngest wave of the workgroup
            rows = [buf[(i * 2 + which) * 8:(i * 2 + which + 1) * 8] for i in range(32)]
            med = [sorted(r[k] for r in rows)[16] for k in range(8)]
            n = max(med[7], 1)
            clk = med[6] / max(med[5], 1) * 0.1
            print("%-20s %5d %9d %6d %9d | %22.0f %5.0f | %9.2f | %9.1f  %s" % ("%d,%d,%d,%d" % (B, H, Cin, Cout), n, med[0], med[1], med[2],
                                                                          med[3] / n, med[4] / n, clk, med[5] / 100.0, name), flush=True)


main()
