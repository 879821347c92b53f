import sys, torch, time
sys.path.insert(0, '.')
import nerfca_amd
from nerfca_amd import synthetic, _capi
from nerfca_amd.model.CPPN import CPPN
from nerfca_amd.model.Temporal import Temporal
from nerfca_amd.train import model_helpers as MH
dev = torch.device('cuda:0')
prec = sys.argv[1] if len(sys.argv) > 1 else 'bf16'
torch.manual_seed(1)
sdef, tdef = synthetic.net_definitions(dev)
s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
nerfca_amd.set_precision(prec, s, t)
for m in (s, t): m.update_freq_mask_alpha(75000, 150000)
R, S = 65536, 192
g = torch.Generator(device=dev).manual_seed(0)
o = torch.randn(R, 3, dtype=torch.float64, device=dev, generator=g) * 0.1 + torch.tensor([0, 0, -4.5], dtype=torch.float64, device=dev)
d = torch.randn(R, 3, dtype=torch.float64, device=dev, generator=g) * 0.05 + torch.tensor([0, 0, 1.0], dtype=torch.float64, device=dev)
ph = torch.randint(0, 10, (R,), device=dev, generator=g)
I0 = torch.full((R,), 2.16, device=dev)
z = torch.linspace(3.4259, 5.5741, S, device=dev)
dists = MH._interval_lengths(z, d)
with torch.no_grad():
    for _ in range(3): out = nerfca_amd.render_rays(s, t, o, d, ph, I0, z, dists)
    torch.cuda.synchronize()
    _capi.timing_reset(); _capi.timing_enable(True)
    for _ in range(10): out = nerfca_amd.render_rays(s, t, o, d, ph, I0, z, dists)
    torch.cuda.synchronize()
    ms, n = _capi.timing_read('fwd')
print(prec, 'plain forward', round(ms / n, 3), 'ms', n, 'launches', float(out[0].double().sum()))
