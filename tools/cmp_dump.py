import torch,sys
a=torch.load(sys.argv[1]); b=torch.load(sys.argv[2])
for k in ("store","work"):
    if k not in a: print(k,"absent"); continue
    d=(a[k]!=b[k]).nonzero().flatten()
    print(k, a[k+"_bytes"], "MiB chunks:", a[k].numel(), "differing:", d.numel(), d[:20].tolist(), d[-5:].tolist())
