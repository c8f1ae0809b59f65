import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np
import edm_amd.hip as H
from oracle import binding as B
import test_gpu_fuzz as F
name=sys.argv[1]
sc=[s for s in F.scenarios() if s['name']==name][0]
print({k:sc[k] for k in ('lo','hi','sp','per','sg','bnd','batches')})
lib=B.load("oracle"); dim=sc['dim']
g = H.Gauss.create(sc["lo"], sc["hi"], sc["sp"], sc["per"], 1, sc["sg"])
o = B.Gauss.create(lib, sc["lo"], sc["hi"], sc["sp"], sc["per"], 1, sc["sg"])
if sc["bnd"]:
    g.set_boundary(*sc["bnd"]); o.set_boundary(*sc["bnd"])
print("number", list(g.number), "minisize", g.minisize)
rng = np.random.default_rng(sc["seed"])
lo, hi = np.array(sc["lo"]), np.array(sc["hi"])
n=np.array([int(v) for v in g.number])
prev=np.zeros(int(np.prod(n)))
for nh in sc["batches"]:
    hx = np.zeros((nh, 3))
    centre = lo + rng.uniform(0.1, 0.9, dim) * (hi - lo)
    spread = rng.uniform(0.05, 0.6)
    hx[:, :dim] = np.where(rng.random((nh, 1)) < 0.5, lo + (rng.uniform(-0.05, 1.05, (nh, dim))) * (hi - lo),
                           centre + rng.normal(0, spread, (nh, dim)) * (hi - lo) * 0.2)
    hh = rng.uniform(-0.3, 1.0, nh)
    g.add_values(hx, hh)
    for x, h in zip(hx, hh): o.add_value(x[:dim], float(h))
    v,_=g.download(); ov=o.grid.values
    bad=np.where(np.abs(v-ov)>1e-9*np.abs(ov)+1e-11)[0]
    print("batch",nh,"bad nodes",len(bad))
    if len(bad):
        idx=np.stack([bad%n[0],(bad//n[0])%n[1],bad//(n[0]*n[1])],axis=1) if dim==3 else None
        print("bad idx min",idx.min(axis=0),"max",idx.max(axis=0))
        print("sample", idx[:10].tolist(), v[bad[:5]], ov[bad[:5]])
        # which hills are near?
        dxs=np.array(sc['sp'])
        gdx=(np.array(g.max)-np.array(g.min))/1.0
        hc=np.floor((hx[:,:dim]-lo)/np.array([float(t) for t in g.dx])).astype(int)
        for b0 in idx[:3]:
            d=np.abs(hc-b0); near=np.where((d<=np.array(g.minisize)+1).all(axis=1))[0]
            print(" node",b0.tolist(),"near hills",[(int(i),hc[i].tolist(),float(hh[i])) for i in near[:5]])
        break
