import sys
import torch
a = torch.load(sys.argv[1]).view(torch.float32)
b = torch.load(sys.argv[2]).view(torch.float32)
stride, nrows = int(sys.argv[3]), int(sys.argv[4])
a, b = a[: stride * nrows].view(nrows, stride), b[: stride * nrows].view(nrows, stride)
neq = a.view(torch.int32) != b.view(torch.int32)
print("rows with differences:", neq.any(1).nonzero().flatten().tolist())
cols = neq.any(0).nonzero().flatten()
print("columns with differences:", cols.numel(), cols[:10].tolist(), "...", cols[-10:].tolist())
# runs of consecutive differing columns
if cols.numel():
    brk = (cols[1:] != cols[:-1] + 1).nonzero().flatten() + 1
    starts = torch.cat([cols[:1], cols[brk]]); ends = torch.cat([cols[brk - 1], cols[-1:]])
    for s0, e0 in list(zip(starts.tolist(), ends.tolist()))[:40]:
        rws = neq[:, s0:e0 + 1].any(1).nonzero().flatten()
        print(f"  cols {s0}..{e0} ({e0 - s0 + 1}) rows {rws[:6].tolist()}..{rws[-3:].tolist()} n={rws.numel()}  e.g. {a[rws[0], s0].item():.6e} vs {b[rws[0], s0].item():.6e}")
