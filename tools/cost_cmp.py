import torch, sys
a, b = torch.load(sys.argv[1]), torch.load(sys.argv[2])
print("loss", float(a["loss"][0]), float(b["loss"][0]), "cand equal", bool((a["cand"] == b["cand"]).all()), "matched equal", bool((a["mg"] == b["mg"]).all()))
cand = a["cand"]                       # [B, A]
B, G, A = a["pw"].shape
ng = a["num_gt"]
for name in ("pw", "cost"):
    x, y = a[name], b[name]
    nd, first = 0, None
    for bi in range(B):
        for g in range(int(ng[bi])):
            m = cand[bi]
            d = (x[bi, g][m] != y[bi, g][m])
            if d.any():
                nd += int(d.sum())
                if first is None:
                    idx = m.nonzero().flatten()[d.nonzero().flatten()[0]]
                    first = (bi, g, int(idx), float(x[bi, g, idx]), float(y[bi, g, idx]))
    print(name, "differing candidate entries:", nd, first)
