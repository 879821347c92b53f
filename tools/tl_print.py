import collections, sys
ev = collections.defaultdict(dict)
for l in open(sys.argv[1]):
    p = l.split()
    if len(p) != 6: continue
    ev[(int(p[1]), int(p[2]), int(p[3]))][int(p[4])] = int(p[5])
for net in (0, 1):
    ks = [k for k in ev if k[0] == net]
    if not ks: continue
    base = min(ev[k][0] for k in ks)
    print("net", net, "per row tile [start | MFMA block | epilogue]; first 4 = layer 0, then the hidden layers; gap = idle between layers")
    for k in sorted(ks, key=lambda k: (k[2], k[1])):
        v = [ev[k][i] - base for i in range(36)]
        tri = [(v[i], v[i + 1] - v[i], v[i + 2] - v[i + 1]) for i in range(0, 36, 3)]
        gaps = [tri[i][0] - (tri[i - 1][0] + tri[i - 1][1] + tri[i - 1][2]) for i in (4, 8)]
        per_layer = [tri[i + 3][0] + tri[i + 3][1] + tri[i + 3][2] - tri[i][0] for i in (0, 4, 8)]
        print(f" simd {k[2]} wave {k[1]}: " + " ".join(f"[{b}|{c}]" for a, b, c in tri) + f"  gaps {gaps} layers {per_layer}")
