import sys; sys.path.insert(0,'.')
import numpy as np
import edm_amd.hip as H, edm_amd.workloads as W
from oracle import binding as B
c=dict(lo=[0.0], hi=[2.8], sp=[0.00025], per=[0], sg=[0.025])
g = H.Gauss.create(c["lo"], c["hi"], c["sp"], c["per"], 1, c["sg"])
o = B.Gauss.create(B.load("oracle"), c["lo"], c["hi"], c["sp"], c["per"], 1, c["sg"])
hx = np.zeros((200, 3)); hx[:, 0] = W.pair_distances(200, 5)
g.add_values(hx, 0.01)
for x in hx: o.add_value(x[:1], 0.01)
og=o.grid; dx=float(og.dx[0])
ks = np.concatenate([np.arange(0, 11201, 37), [0, 1, 11198, 11199, 11200]]).astype(np.float64)
base = ks * dx
r = np.concatenate([base, np.nextafter(base, 10), np.nextafter(base, -10), [2.8, 2.8 - 1e-16, 0.0, -0.0, 2.8 + dx, -1e-300]])
e,f=g.pair_forces(r)
E2,D2=g.get_value_deriv(r.reshape(-1,1))
ref=np.array([-o.get_value_deriv([x])[1][0] for x in r])
d=np.abs(f-ref); i=np.argsort(-d)[:8]
for j in i: print(j, repr(r[j]), f[j], ref[j], -D2[j,0], d[j], "k=",r[j]/dx)
