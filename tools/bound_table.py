"""Per-class bounds of one training step from a layer table (bench.py with EP24_LAYER_TABLE=...): launches, serial ms (HIP events
round every launch of the instrumented serial pass), algorithmic bytes / 6.3 TB/s (the achievable HBM rate, MI355X_MICROARCH.md) and
conv FLOPs over BOTH the chip's dense bf16 MFMA peak (2 500 TFLOP/s: the roofline) and 900 TFLOP/s (what a 128 x 128, two-barrier tile
structure reaches on this chip - a property of that structure, not a bound).  "x over" columns are against the CHIP's bounds.
usage: bound_table.py LAYER_TABLE.txt"""
import collections
import sys

HBM, MFMA, MFMA_OLD = 6.3e12, 2500e12, 900e12
conv = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])      # class -> launches, ms, bytes, flops
other = collections.defaultdict(lambda: [0, 0.0, 0.0])
sec = 0
for line in open(sys.argv[1]):
    if line.startswith("other kernels"):
        sec = 1
        continue
    if ":" not in line or line.startswith("kernel ") or line.startswith("#"):
        continue
    left, right = line.split(":")
    f = left.split()
    n, ms = int(right.split()[0]), float(right.split()[1])
    if sec == 0 and f[0].startswith("conv_"):
        B, H, W, Ci, Co, k, s = map(int, f[1:8])
        OH, OW = (H - 1) // s + 1, (W - 1) // s + 1
        flops = 2.0 * B * OH * OW * Ci * Co * k * k * n
        byts = 2.0 * (B * H * W * Ci + B * OH * OW * Co + Co * k * k * Ci) * n
        if f[0].startswith("conv_wgrad"):
            byts += 0  # slab stores / reads are the split scheme's, not the algorithm's
        c = conv[("1x1" if k == 1 else "3x3 s%d" % s)]
        c[0] += n; c[1] += ms; c[2] += byts; c[3] += flops
    elif f[0].startswith("conv1x1_"):
        # a 1x1 unit together with the BatchNorm pass in front of it (round 5): forward reads z (+ shortcut, not counted), stores y and
        # the conv output; backward reads dy and z, stores dz and the input gradient
        if len(f) < 6:          # a table written before bench.py printed this entry's shape (r05d): YOLOX-l's twelve launches are the 40x40x256 Bottlenecks
            f = [f[0], "20", "40", "40", "256", "256"]
        B, H, W, Ci, Co = map(int, f[1:6])
        M = B * H * W
        K, N = (Ci, Co) if "bnin" in f[0] else (Co, Ci)
        if "dgrad_bnr" in f[0]:        # input gradient + the reduce pass of the unit below: reads dz and that unit's z, stores dx (old dx not counted)
            byts = 2.0 * (M * K + 2 * M * N + N * K) * n
        else:
            byts = 2.0 * ((2 if "bnin" in f[0] else 3) * M * K + M * N + N * K) * n
        c = conv["1x1 + BN pass"]
        c[0] += n; c[1] += ms; c[2] += byts; c[3] += 2.0 * M * N * K * n
    elif sec == 1:
        name, a = f[0], [int(v) for v in f[1:]]
        o = other[name]
        o[0] += n; o[1] += ms
        if name == "bn_act_fwd":
            o[2] += 4.0 * a[0] * a[1] * n
        elif name == "bn_act_bwd_reduce":
            o[2] += 4.0 * a[0] * a[1] * n
        elif name in ("bn_act_bwd_apply", "bn_act_bwd_apply_acc"):
            o[2] += 6.0 * a[1] * a[2] * n
        elif name == "wgrad_reduce":
            o[2] += 0.0
print("| class | launches | serial ms | algorithmic GB | HBM bound ms (6.3 TB/s) | MFMA bound ms (2 500 TF) | serial / larger bound | (MFMA ms at 900 TF) |")
print("|---|---|---|---|---|---|---|---|")
tot = [0, 0.0, 0.0]
for k in ("1x1", "1x1 + BN pass", "3x3 s1", "3x3 s2"):
    n, ms, by, fl = conv[k]
    if not n:
        continue
    hb, mb = by / HBM * 1e3, fl / MFMA * 1e3
    print("| conv %s (%s) | %d | %.2f | %.2f | %.2f | %.2f | %.2fx | %.2f |" % (k, "fwd / dgrad with the BatchNorm fwd / bwd-apply pass in the launch" if "BN" in k else "fwd + dgrad + wgrad",
                                                                            n, ms, by / 1e9, hb, mb, ms / max(hb, mb), fl / MFMA_OLD * 1e3))
    tot[0] += n; tot[1] += ms; tot[2] += max(hb, mb)
bn = [0, 0.0, 0.0]
for k in ("bn_act_fwd", "bn_act_bwd_reduce", "bn_act_bwd_apply"):
    n, ms, by = other[k]
    print("| %s | %d | %.2f | %.2f | %.2f | - | %.2fx | - |" % (k, n, ms, by / 1e9, by / HBM * 1e3, ms / (by / HBM * 1e3)))
    tot[0] += n; tot[1] += ms; tot[2] += by / HBM * 1e3
STEM = ("focus_pack", "stem_conv_fwd_bf16", "stem_conv_wgrad_slab_bf16")
st_n = sum(other[k][0] for k in STEM if k in other)
st_ms = sum(other[k][1] for k in STEM if k in other)
if st_n:
    # per pixel of the space-to-depth grid: pack 48 B in + 32 B out, conv 32 B in + 128 B out, weight gradient 128 + 32 B in
    Mst = [int(l.split()[1]) * int(l.split()[2]) * int(l.split()[3]) // 4 for l in open(sys.argv[1]) if l.startswith("focus_pack")][0]
    by = 400.0 * Mst
    print("| Focus stem (focus_pack + gathering conv + weight gradient; no im2col buffer) | %d | %.2f | %.2f | %.2f | 0.02 | %.2fx | 0.06 |" % (
        st_n, st_ms, by / 1e9, by / HBM * 1e3, st_ms / (by / HBM * 1e3)))
    tot[0] += st_n; tot[1] += st_ms; tot[2] += by / HBM * 1e3
rest_n = sum(v[0] for k, v in other.items() if not k.startswith("bn_act") and k not in STEM)
rest_ms = sum(v[1] for k, v in other.items() if not k.startswith("bn_act") and k not in STEM)
print("| everything else in the table (wgrad_reduce, SPP, upsample, decode, copies) | %d | %.2f | - | - | - | - | - |" % (rest_n, rest_ms))
print("| sum | %d | %.2f | | %.2f (sum of the larger bounds) | | %.2fx | |" % (tot[0] + rest_n, tot[1] + rest_ms, tot[2], (tot[1]) / tot[2]))
