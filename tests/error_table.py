"""Developer aid (GPU box): per-tensor error table of the fused render (outputs and every gradient) against
the oracle.  usage: python tests/error_table.py R S F [bf16|f32]   (bf16 compares with the emulating oracle)"""
import sys, os
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE); sys.path[:0] = [ROOT, HERE]
import torch
from oracle import nerfca_oracle as O
from test_hip_parity import make_static, make_dynamic, _oracle_render_grads, grads_of
from conftest import rel_err
from nerfca_amd import render_rays, set_precision
dev = torch.device("cuda:0")
R, S, F = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
prec = sys.argv[4] if len(sys.argv) > 4 else "bf16"
gen = torch.Generator().manual_seed(1)
emu = prec == 'bf16'
ss = O.NetSpec(num_filters=F, num_early_layers=2, num_time_dim=0, emulate_bf16=emu)
sd = O.NetSpec(num_filters=F, num_early_layers=2, num_time_dim=8, emulate_bf16=emu)
ps, pd = O.init_params(ss, gen), O.init_params(sd, gen)
win = O.freq_mask_alpha(12, 75000, 150000, 1)[0]
o = (torch.rand(R, 3, generator=gen) * 0.2 + torch.tensor([3.0, -2.0, 2.5])).double()
d = (torch.rand(R, 3, generator=gen) - 0.5).double(); d = d / d.norm(dim=-1, keepdim=True)
ph = torch.randint(0, 10, (R,), generator=gen)
z = O.stratified_depths(O.depth_values(3.4259, 5.5741, S), torch.rand(S, generator=gen))
I0 = torch.full((R,), 2.15991)
cp, cs, cd = torch.randn(R, generator=gen).double(), torch.randn(R, S, generator=gen), torch.randn(R, S, generator=gen)
pix, a, b, dists, ps32, pd32 = _oracle_render_grads(ps, ss, pd, sd, win, o, d, ph, I0, z, cp, cs, cd, torch.float32)
s = make_static(ps, dev, F=F, early=2, late=0); t = make_dynamic(pd, dev, F=F, early=2, late=0, T=8)
set_precision(prec, s, t)
s.update_freq_mask_alpha(75000, 150000); t.update_freq_mask_alpha(75000, 150000)
pix2, a2, b2 = render_rays(s, t, o.to(dev), d.to(dev), ph.to(dev), I0.to(dev), z.to(dev), dists.to(dev))
print("fwd", rel_err(pix2.cpu(), pix), rel_err(a2.cpu(), a), rel_err(b2.cpu(), b))
((pix2 * cp.to(dev)).sum() + (a2 * cs.to(dev)).sum() * 50 + (b2 * cd.to(dev)).sum() * 50).backward()
for name, got, p32 in (("s", grads_of(s), ps32), ("d", grads_of(t), pd32)):
    for k in p32:
        g, r = got[k], p32[k].grad
        extra = ""
        if g.dim() == 2 and g.shape[1] > 8:
            ce = ((g - r).abs().max(0)[0] / r.abs().max()).tolist()
            bad = [i for i, e in enumerate(ce) if e > 0.06]
            extra = f" badcols={bad[:20]}{'...' if len(bad) > 20 else ''} n={len(bad)}"
        print(name, k, f"{rel_err(g, r):.3e}", extra)
