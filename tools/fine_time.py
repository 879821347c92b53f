"""Timing of the hierarchical (fine) pass step with and without the reference's through-depth gradient (run on the GPU box)."""
import sys, time, torch
sys.path.insert(0, '/root/repo')
import nerfca_amd
from nerfca_amd import synthetic
from nerfca_amd.model.CPPN import CPPN
from nerfca_amd.model.Temporal import Temporal
from nerfca_amd.train.trainer import CompositeTrainer, TrainConfig
dev = torch.device('cuda', 0)
S, NF, R = 192, 64, 4096
data = synthetic.make_dataset(64, S, dev, views=synthetic.TRAIN_VIEWS[:2], n_phases=3)
for prec in ('f32', 'bf16'):
    for dg in (False, True):
        torch.manual_seed(0)
        sdef, tdef = synthetic.net_definitions(dev)
        fs, ft = synthetic.net_definitions(dev, F=64)
        nets = [CPPN(sdef).to(dev), Temporal(tdef).to(dev), CPPN(fs).to(dev), Temporal(ft).to(dev)]
        nerfca_amd.set_precision(prec, *nets)
        cfg = TrainConfig(depth_samples_per_ray_coarse=S, depth_samples_per_ray_fine=NF, img_sample_size=R, fine_depth_gradients=dg)
        tr = CompositeTrainer(cfg, nets[0], nets[1], data, dev, seed=1, static_model_fine=nets[2], temp_model_fine=nets[3])
        for i in range(3): tr.step(1000 + i)
        per = []
        for i in range(10):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            tr.step(2000 + i)
            torch.cuda.synchronize(); per.append(round((time.perf_counter() - t0) * 1e3, 1))
        tot = sum(per) / 10
        print('   per step (ms):', per, flush=True)
        # split: forward (local_loss) / backward
        tr.update_windows(3000)
        ids = tr.draw_ray_ids_device(3000); tj = tr.draw_jitter(3000)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        loss, _, _ = tr.local_loss(3000, ids, tj)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        tr.opt.zero_grad(set_to_none=True); loss.backward()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        print(prec, 'depth_gradients', dg, f'{tot:.2f} ms/step  (forward {1e3*(t1-t0):.2f} ms, backward {1e3*(t2-t1):.2f} ms)', flush=True)
        t = {}
        for name, fn in (("windows", lambda: tr.update_windows(3001)), ("ids", lambda: tr.draw_ray_ids_device(3001)), ("jitter", lambda: tr.draw_jitter(3001)),
                         ("opt", lambda: tr.opt.step()), ("sched", lambda: tr.sched.step())):
            torch.cuda.synchronize(); a = time.perf_counter(); fn(); torch.cuda.synchronize(); t[name] = round(1e3 * (time.perf_counter() - a), 2)
        print('   pieces (ms):', t, 'loss', float(loss), 'finite grads', all(bool(torch.isfinite(p.grad).all()) for p in tr.params), flush=True)
