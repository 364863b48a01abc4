"""GPU box helper: the ring kernel's EPILOGUE piece by piece (csrc/conv_ring.hip, diagnostic library `make -C csrc stamps`): s_memtime stamps
of wave 0 (a consumer) and wave 4 (a loader) of the first 32 workgroups - entry, the barrier behind the main loop, the staging writes, the
barrier behind them, the 16-byte stores' issue, the barrier before the statistics, the statistics and their atomics.  Medians, cycles.
usage: EP24_LIB=.../libep24_stamps.so ring_estamps.py"""
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
    rd = L.cdll.ep24_debug_read_ring_estamps
    rd.argtypes = [ctypes.c_void_p, ctypes.c_int]
    rd0 = L.cdll.ep24_debug_read_ring_stamps
    rd0.argtypes = [ctypes.c_void_p, ctypes.c_int]
    print("shape               wave      | barrier 1  staging  barrier 2  store issue  barrier 3  statistics   rest |  epilogue  (whole: prologue  loop  epilogue)")
    for B, H, Cin, Cout in [(20, 40, 256, 256), (20, 80, 128, 128), (20, 80, 256, 256)]:
        W = H
        x = torch.randn(B * H * W, Cin, device=DEV).to(torch.bfloat16)
        w = (torch.randn(Cout, 9, Cin, device=DEV) * 0.05).to(torch.bfloat16)
        y = torch.zeros(B * H * W, Cout, device=DEV, dtype=torch.bfloat16)
        stats = torch.zeros(8, 2, Cout, dtype=torch.int64, device=DEV)
        for _ in range(50):
            call("conv_fwd_bf16_ex", ptr(x), Cin, ptr(w), ptr(y), Cout, 0, 0, 0, None, ptr(stats), 8, B, H, W, Cin, Cout, 3, 1, 0, stream_ptr())
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 512)()
        assert rd(buf, 512) == 0
        b0 = (ctypes.c_ulonglong * 512)()
        assert rd0(b0, 512) == 0
        for which, name in ((0, "consumer 0"), (1, "loader 4")):
            rows = [buf[(i * 2 + which) * 8:(i * 2 + which + 1) * 8] for i in range(32)]
            rows = [[r[0]] + [r[k] if r[k] else 0 for k in range(1, 8)] for r in rows]
            for r in rows:                                   # a stamp that no longer exists (the barrier before the statistics went in round 5) reads 0
                for k in range(1, 8):
                    if r[k] == 0:
                        r[k] = r[k - 1]
            d = [[r[k + 1] - r[k] for k in range(7)] for r in rows]
            med = [sorted(x_[k] for x_ in d)[16] for k in range(7)]
            tot = sorted(r[7] - r[0] for r in rows)[16]
            r0 = [b0[(i * 2) * 8:(i * 2 + 1) * 8] for i in range(32)]
            m0 = [sorted(r[k] for r in r0)[16] for k in range(3)]
            print("%-19s %-10s| %9d %8d %10d %12d %10d %11d %6d | %9d  (%d %d %d)" % ("%d,%d,%d,%d" % (B, H, Cin, Cout), name, *med, tot, *m0), flush=True)
    print("ring timeouts:", L.fn["ep24_conv_ring_timeouts"]())


main()
