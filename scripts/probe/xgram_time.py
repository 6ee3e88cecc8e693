import sys, os
sys.path.insert(0, '/root/repo')
import numpy as np
from pybo_amd._lib import Engine
N=8192; d=8; M=1<<18
rng=np.random.RandomState(1)
X=rng.rand(N,d); y=-((X-0.5)**2).sum(1)+1e-3*rng.randn(N)
ell=0.25*np.ones(d); rho=float(np.var(y)); bias=float(y.mean()); sn2=1e-4*rho
e=Engine(0); e.fit(X,y,sys.argv[1] if len(sys.argv)>1 else 'se',ell,rho,sn2,bias)
Z=rng.rand(M,d)
e.sweep('ei',0.0,Z,k=4,want_all=False)
e.timers(reset=True)
for _ in range(3): e.sweep('ei',0.0,Z,k=4,want_all=False)
t=e.timers()
print('GPX_XRT=%s %s cross_gram per 2^20 candidates: %.3f ms' % (os.environ.get('GPX_XRT','-'), sys.argv[1] if len(sys.argv)>1 else 'se', t['cross_gram']/3*4))
