import sys, os
sys.path.insert(0,'cuda-path-tracer_amd'); sys.path.insert(0,'tests')
import numpy as np, nori_hip as NH
g=np.load('tests/golden/cbox.npz')
hs=NH.HostScene('scenes/cbox/cbox_path_mis.xml',64,64,1); ctx=hs.preprocess()
hits,full=ctx.trace(g['rays'],full=True)
d=np.abs(full-g['full'])
print('col maxdiff', d.max(0))
bad=np.where(d.max(1)>0)[0]
print(len(bad),'rows; shapes', np.unique(g['full'][bad,11], return_counts=True))
i=bad[0]; print(full[i], g['full'][i])
