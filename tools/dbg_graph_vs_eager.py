import os, sys, torch
sys.path.insert(0, os.getcwd())
import nerfca_amd
from nerfca_amd import synthetic, _capi, fused
from nerfca_amd.model.CPPN import CPPN
from nerfca_amd.model.Temporal import Temporal
from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
dev = torch.device("cuda", 0)
rays = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
data = synthetic.make_dataset(256, 192, dev, views=synthetic.TRAIN_VIEWS)
def mk():
    torch.manual_seed(1)
    sdef, tdef = synthetic.net_definitions(dev)
    s, t = CPPN(sdef).to(dev), Temporal(tdef).to(dev)
    nerfca_amd.set_precision("bf16", s, t)
    return CompositeTrainer(TrainConfig(depth_samples_per_ray_coarse=192, img_sample_size=rays), s, t, data, dev, seed=0)
# keep the tensors of the capture alive for inspection
kept = []
orig = fused.render_forward_raw
def fwd(*a, **k):
    r = orig(*a, **k); kept.append(r[:3]); return r
fused.render_forward_raw = fwd
batches = []
oinit = fused._RayBatch.__init__
def binit(self, *a, **k):
    oinit(self, *a, **k); batches.append(self)
fused._RayBatch.__init__ = binit
tr = mk()
out = tr.step_graph(75000)
torch.cuda.synchronize()
print("graph terms", [f"{float(x):.4e}" for x in out[2]])
pix, ss, sd = kept[-1]
print("graph pix", float(pix.min()), float(pix.max()), bool(torch.isfinite(pix).all()), "sig_s", float(ss.min()), float(ss.max()), bool(torch.isfinite(ss).all()), "sig_d", float(sd.min()), float(sd.max()), bool(torch.isfinite(sd).all()))
b = batches[-1]
print("graph batch dists", b.dists.dtype, b.dists[:4].tolist(), b.dists[-2:].tolist(), "z", b.z[:3].tolist(), "ptr", hex(b.dists.data_ptr()), "n batches", len(batches))
for i, bb in enumerate(batches): print("  batch", i, hex(bb.dists.data_ptr()), bb.dists[:2].tolist(), hex(bb.z.data_ptr()))
print("plan", _capi.last_plan())
tr2 = mk()
terms, gs, gd = tr2.fused_gradients(75000)
print("eager terms", [f"{float(x):.4e}" for x in terms])
pix2, ss2, sd2 = kept[-1]
print("eager pix", float(pix2.min()), float(pix2.max()), "sig_s", float(ss2.min()), float(ss2.max()), "sig_d", float(sd2.min()), float(sd2.max()))
print("pix equal", torch.equal(pix, pix2), "sig_s equal", torch.equal(ss, ss2), "sig_d equal", torch.equal(sd, sd2))
if not torch.equal(ss, ss2):
    bad = (ss != ss2).nonzero()
    print("sig_s differs at", bad.shape[0], bad[:5].tolist(), bad[-3:].tolist())
if not torch.equal(sd, sd2):
    bad = (sd != sd2).nonzero()
    print("sig_d differs at", bad.shape[0], bad[:5].tolist(), bad[-3:].tolist())
if not torch.equal(pix, pix2):
    bad = (pix != pix2).nonzero()
    print("pix differs at", bad.shape[0], bad[:5].flatten().tolist(), bad[-3:].flatten().tolist())
