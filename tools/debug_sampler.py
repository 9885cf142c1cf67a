import numpy as np, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morpheus_amd import synth, ops
from oracle import field as of
o, d, t, rid = synth.frame_rays(25, 64, 64)
o, d = o[0], d[0]
o = torch.cat([o, torch.tensor([[3.0, 3.0, 3.0], [0.0, 0.0, 2.0], [0.5, 0.5, 2.0]])])
d = torch.cat([d, torch.tensor([[1.0, 0.0, 0.0], [0.0, 0.0, -1.0], [0.0, 0.0, 1.0]])])
N = o.shape[0]; S = 64
jit = synth.ray_jitter(N)
ri_o, ts_o, te_o = of.uniform_samples(o, d, jit, S, 1.01)
ri, ts, te, xyz, rs, rc = ops.sample_uniform(o.cuda(), d.cuda(), jit.cuda(), S, 1.01, with_xyz=True)
ts = ts.cpu()
bad = (ts != ts_o).nonzero().flatten()
print("mismatches", bad.numel(), "of", ts.numel())
f = np.float32
on, dn, un = o.numpy(), d.numpy(), jit.numpy()
b = f(1.01)
np.seterr(all='ignore'); ta = (-b - on) / dn; tb = (b - on) / dn
tmin = np.maximum(np.minimum(ta, tb).max(-1), f(0)); tmax = np.maximum(ta, tb).min(-1)
dt = (tmax - tmin) / f(S + 1)
for k in bad[:5].tolist():
    r, i = k // S, k % S
    a = f(i) + un[r]; m = a * dt[r]; s = tmin[r] + m
    print(k, r, i, "gpu", ts[k].item().hex(), "cpu", ts_o[k].item().hex(), "np", float(s).hex(), "tmin", float(tmin[r]).hex(), "dt", float(dt[r]).hex(), "u", float(un[r]).hex())
    # what would fma give
    print("   fma:", float(np.float32(np.float64(a) * np.float64(dt[r]) + np.float64(tmin[r]))).hex())
    # reciprocal-multiply division variant
    dt2 = (tmax[r] - tmin[r]) * (f(1) / f(S + 1)); print("   dt recip:", float(dt2).hex(), " s:", float(tmin[r] + a * dt2).hex())
