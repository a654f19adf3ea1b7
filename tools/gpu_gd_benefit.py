import sys; sys.path.insert(0,'/root/repo')
import numpy as np
from gradientdomain_mitsuba_amd import gpt as G, scenes
W,H,spp=160,120,8
S=G.Scene(scenes.cornell_box(W,H,"diffuse"))
r=G.GradientPathIntegrator(maxDepth=8,reconstructL1=False,reconstructL2=False).render(S,64*spp,seed=99); ref=r["-throughput"]+r["-direct"]
for name,kw in (("L1",dict(reconstructL1=True)),("L2",dict(reconstructL1=False,reconstructL2=True))):
    o=G.GradientPathIntegrator(maxDepth=8,**kw).render(S,spp); p=o["-throughput"]+o["-direct"]
    rel=lambda im: float(np.mean((im-ref)**2/(ref**2+1e-2)))
    print(name, "primal relMSE %.4g  reconstructed %.4g  ratio %.3f"%(rel(p),rel(o["-final"]),rel(o["-final"])/rel(p)))
