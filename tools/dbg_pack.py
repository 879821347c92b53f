import sys, os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from nerfca_amd import synthetic, _capi
from nerfca_amd.model.CPPN import CPPN
dev = torch.device("cuda:0")
sdef, tdef = synthetic.net_definitions(dev, F=32, early=1, L=4)
m = CPPN(sdef).to(dev); m.update_freq_mask_alpha(5,10)
opt = torch.optim.Adam(m.parameters(), lr=1e-2, fused=True)
x = torch.rand(64,3,device=dev)
b = m._binding
for it in range(3):
    y = m(x); 
    key = (b.flat._version, tuple(p._version for p in b.params()))
    opt.zero_grad(); y.sum().backward(); opt.step()
    print(it, float(y.sum()), key, b._is_flat(), len(os.sched_getaffinity(0)), os.cpu_count())
