import torch, sys, itertools
names = sys.argv[1:]
d = {n: torch.load(n) for n in names}
for a, b in itertools.combinations(names, 2):
    res = []
    for x, y in zip(d[a], d[b]):
        res.append("eq" if torch.equal(x, y) else f"diff(max {float((x.float()-y.float()).abs().max()):.3e})")
    print(a.split('/')[-1], b.split('/')[-1], res)
