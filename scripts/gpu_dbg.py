import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybo_amd._lib import Engine
from oracle import gp_ref
for (N,d,kern,sn2) in [(129,1,'matern5',1e-4),(129,1,'se',1e-2),(129,3,'se',1e-3),(200,2,'matern5',1e-4),(257,1,'matern5',1e-4)]:
    rng=np.random.RandomState(0)
    X=rng.rand(N,d); y=np.sin(3*X.sum(1))+0.1*rng.randn(N)
    ell=0.3+0.2*rng.rand(d)
    e=Engine(0); e.fit(X,y,kern,ell,1.3,sn2,0.2)
    L=e.get_matrix('L'); T=e.get_matrix('T')
    E=T@L-np.eye(N); i,j=np.unravel_index(np.argmax(np.abs(E)),E.shape)
    Ti=np.linalg.inv(L)
    D=np.abs(T-Ti); i2,j2=np.unravel_index(np.argmax(D),D.shape)
    print(N,d,kern,sn2,'max|TL-I|',np.abs(E).max(),'at',(i,j),' max|T-inv(L)|',D.max(),'at',(i2,j2),'max|T|',np.abs(Ti).max(), 'cond L', np.linalg.cond(L))
    e.close()
