"""The Thompson sweep of config D (64 draws x 100 features, d = 32) on a slice of the grid, for counter passes:
    scripts/pmc_cmd.sh rff1 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" scripts/pmc_rff.py
    scripts/pmc_cmd.sh rff2 "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" scripts/pmc_rff.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pybo_amd._lib import Engine
d, S, n, M = 32, 64, 100, 1 << 18
rng = np.random.RandomState(3)
e = Engine(0)
W, b, th = rng.randn(S, n, d), rng.rand(S, n) * 6.28, rng.randn(S, n)
Z = rng.rand(M, d) * 2 - 1
for _ in range(2):
    e.rff_sweep(W, b, th, 0.1, Z, k=1, want_all=False)
tm = e.timers()
print('rff stage ms (2 sweeps of %d candidates x %d draws): %.3f' % (M, S, tm['rff']))
