"""profiles/r06_roofline.json: kernel -> algorithmic units per launch, average launch duration (rocprofv3 kernel stats of the
same bench.py command), achieved rate, peak, fraction -- so that every fraction quoted in DESIGN.md / README.md can be
recomputed without reading prose.   python scripts/r06_roofline.py gpurun_out/r06 > profiles/r06_roofline.json"""
import csv, json, os, sys

O = sys.argv[1]
PEAK_TF, PEAK_LANE, HBM = 78.6, 256 * 4 * 16 * 2.4e9 / 1e12, 8000.0


def stats(tag):
    path = os.path.join(O, '%s_kernel_stats.csv' % tag)
    out = {}
    if not os.path.exists(path):
        return out
    for r in csv.DictReader(open(path)):
        out[r['Name']] = dict(calls=int(r['Calls']), avg_ns=float(r['AverageNs']), total_ns=float(r['TotalDurationNs']))
    return out


def pick(st, key):
    hits = [(k, v) for k, v in st.items() if key in k]
    if not hits:
        return None, None
    k, v = max(hits, key=lambda kv: kv[1]['total_ns'])
    return k, v


def load(name):
    p = os.path.join(O, 'bench_%s.json' % name)
    return json.load(open(p)) if os.path.exists(p) and os.path.getsize(p) else None


rows = []
ns, d_, e_, b_, ens_ = load('ns'), load('d'), load('e'), load('b'), load('ns_ens10')
st = stats('ns')
N, M, cols = 8192, 1 << 20, 65536
k, v = pick(st, 'k_sweep_trmm')
if v:
    flop = float(N) * N * cols
    rows.append(dict(kernel=k[:60], workload='ns', bound='mfma', unit='TFLOP/s', algorithmic='N^2 flop per candidate x 65536 candidates per launch',
                     units_per_launch=flop, avg_ns=v['avg_ns'], calls=v['calls'], achieved=flop / v['avg_ns'] / 1e3, peak=PEAK_TF,
                     frac=flop / v['avg_ns'] / 1e3 / PEAK_TF))
k, v = pick(st, 'k_cross_gram')
if v:
    byt = 8.0 * N * cols
    rows.append(dict(kernel=k[:60], workload='ns', bound='hbm', unit='GB/s', algorithmic='8 Np bytes written per candidate x 65536 per launch',
                     units_per_launch=byt, avg_ns=v['avg_ns'], calls=v['calls'], achieved=byt / v['avg_ns'], peak=HBM, frac=byt / v['avg_ns'] / HBM))
for tag, Nn in (('ns', 8192), ('d', 16384), ('e', 8192)):
    s2 = stats(tag)
    k, v = pick(s2, 'k_chol_tg')
    if v:
        flop = float(Nn) ** 3 / 3.0
        rows.append(dict(kernel=k[:60], workload=tag, bound='mfma', unit='TFLOP/s', algorithmic='N^3/3 flop per factorisation (one launch)',
                         units_per_launch=flop, avg_ns=v['avg_ns'], calls=v['calls'], achieved=flop / v['avg_ns'] / 1e3, peak=PEAK_TF,
                         frac=flop / v['avg_ns'] / 1e3 / PEAK_TF))
    k, v = pick(s2, 'k_rff_mfma')
    if v:
        S, n, dd = (64, 100, 32) if tag == 'd' else (8, 100, 6)
        ops = float(S) * n * (dd + 20) * M
        rows.append(dict(kernel=k[:60], workload=tag, bound='fp64 lanes (MFMA + VALU share them)', unit='T lane-operations/s',
                         algorithmic='draws x features x (d + 20) x candidates per launch', units_per_launch=ops, avg_ns=v['avg_ns'],
                         calls=v['calls'], achieved=ops / v['avg_ns'] / 1e3, peak=PEAK_LANE, frac=ops / v['avg_ns'] / 1e3 / PEAK_LANE))
# the triangular inverse: the sum of its launches per fit
trt = [(k, v) for k, v in st.items() if 'trtri' in k]
if trt and ns:
    tot = sum(v['total_ns'] for _, v in trt)
    fits = max(v['calls'] for k, v in trt if 'diag128' in k) if any('diag128' in k for k, _ in trt) else 1
    flop = float(N) ** 3 / 3.0
    rows.append(dict(kernel='k_trtri_* (all launches of one inverse)', workload='ns', bound='mfma', unit='TFLOP/s', algorithmic='N^3/3 flop per inverse',
                     units_per_launch=flop, avg_ns=tot / fits, calls=fits, achieved=flop / (tot / fits) / 1e3, peak=PEAK_TF,
                     frac=flop / (tot / fits) / 1e3 / PEAK_TF))
out = {'note': 'avg_ns from rocprofv3 --kernel-trace --stats of `python bench.py [--workload w] --steps 2 --warmup 1` (scripts/trace.sh); '
               'achieved = units_per_launch / avg_ns; peaks: fp64 MFMA 78.6 TFLOP/s, fp64 lanes 39.3 T lane-op/s, HBM 8000 GB/s (datasheet)',
       'kernels': rows,
       'bench_lines': {w: {kk: b[kk] for kk in ('ms_per_step', 'roofline', 'roofline_fit', 'roofline_rff', 'parity') if kk in b}
                       for w, b in (('ns', ns), ('b', b_), ('d', d_), ('e', e_), ('ns_ens10', ens_)) if b}}
print(json.dumps(out, indent=1))
