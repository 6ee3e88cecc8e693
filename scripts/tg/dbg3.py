import sys
from sim3 import *
c = dict(potrf=30, hop=2.5, trsm=5, updq=5, ovh=4, kblk=32, kblk_u=18, fast_d=2, claim=3)
nP = int(sys.argv[1]); lo = int(sys.argv[2]); hi = int(sys.argv[3])
s = Sim(nP, parse_D('1,1,2,4*', nP), 506, 4, c)
log = []
orig_start, orig_complete = s.start, s.complete
def st(t, crit):
    if lo <= t[1] <= hi and (t[0] != 'upd' or t[2] - t[1] <= 1): log.append((round(s.t, 1), 'start', t))
    orig_start(t, crit)
def cp(t):
    orig_complete(t)
    if lo <= t[1] <= hi and (t[0] != 'upd' or t[2] - t[1] <= 1): log.append((round(s.t, 1), 'done ', t))
s.start, s.complete = st, cp
s.run()
for l in log: print(l)
