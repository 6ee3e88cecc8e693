"""North-star step through the HOST-buffer entry points (gpx_fit + gpx_sweep): PCIe-inclusive wall-clock."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
sys.argv = ['bench.py']
import bench
from pybo_amd._lib import Engine
w = bench.make_workload('ns', 1 << 20)
e = Engine(0)
for it in range(3):
    t0 = time.perf_counter()
    e.fit(w['X'], w['y'], w['kernel'], w['ell'], w['rho'], w['sn2'], w['bias'])
    _, mx = e.mean_at_obs()
    r = e.sweep('ei', mx, w['Xc'], k=10, want_all=False)
    t1 = time.perf_counter()
    tm = e.timers(reset=True)
    print(f"host-buffer step {t1-t0:.4f} s  (copies {tm['copies']:.2f} ms, sweep_trmm {tm['sweep_trmm']:.1f} ms)  selected {r['top_idx'][0]}")
