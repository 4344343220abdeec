import sys; sys.path.insert(0,'/root/repo')
import numpy as np, flux_amd
from flux_amd.render import debug_fastmath
rng=np.random.default_rng(1)
x=np.concatenate([rng.uniform(1e-3,1e3,200000), 10.0**rng.uniform(-20,20,200000)])*rng.choice([-1.0,1.0],400000)
r=debug_fastmath(9,x)
rel=np.abs(r*x-1.0)
print("raw v_rcp_f64: max rel err %.3e = 2^%.1f ; mean %.3e" % (rel.max(), np.log2(rel.max()), rel.mean()))
r=debug_fastmath(8,np.abs(x))
rel=np.abs(r*r*np.abs(x)-1.0)/2
print("raw v_rsq_f64: max rel err %.3e = 2^%.1f" % (rel.max(), np.log2(rel.max())))
