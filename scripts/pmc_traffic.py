"""Reduce the two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace only) of
`bench.py --steps 1 --warmup 0 --candidates 131072` to HBM bytes per k_sweep_trmm launch, with the gfx950 corrections of
MI355X_MICROARCH.md (HBM section): the counters are in KiB; FETCH_SIZE reports half of the bytes of wide (16 B/lane)
coalesced reads, so fetch is doubled; WRITE_SIZE is calibrated on k_cross_gram, whose store volume is known exactly.
Usage: python scripts/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> [N=8192] [cols per launch=65536] > profiles/rNN_pmc_traffic.json"""
import csv, json, sys, collections


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    return acc


fetch, write = per_kernel(sys.argv[1], 'FETCH_SIZE'), per_kernel(sys.argv[2], 'WRITE_SIZE')
pick = lambda acc, key: next(v for k, v in acc.items() if key in k)         # noqa: E731
N = Np = int(sys.argv[3]) if len(sys.argv) > 3 else 8192
cols = int(sys.argv[4]) if len(sys.argv) > 4 else 65536
tf, tw = pick(fetch, 'k_sweep_trmm'), pick(write, 'k_sweep_trmm')
xw = pick(write, 'k_cross_gram')
alg_x = Np * cols * 8.0
calib = (sum(xw) / len(xw)) * 1024.0 / alg_x
fetch_b = 2.0 * 1024.0 * sum(tf) / len(tf)
write_b = 1024.0 * sum(tw) / len(tw) / calib
out = {
    'note': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only); KiB counters; FETCH_SIZE '
            'doubled (gfx950 wide coalesced reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE calibrated on k_cross_gram',
    'config': {'N': N, 'Np': Np, 'd': 8, 'cols_per_launch': cols, 'tile_order': 19, 'flop_per_candidate': 'N^2 (algorithmic)'},
    'k_sweep_trmm': {'launches': len(tf), 'fetch_kib_raw_mean': sum(tf) / len(tf), 'fetch_bytes_corrected': fetch_b,
                     'write_bytes': write_b, 'traffic_bytes_per_launch': fetch_b + write_b,
                     'algorithmic_input_bytes_per_launch': Np * Np * 8 // 2 + Np * cols * 8},
    'k_cross_gram': {'write_bytes': (sum(xw) / len(xw)) * 1024.0, 'algorithmic_write_bytes': alg_x,
                     'write_size_calibration': calib},
}
print(json.dumps(out, indent=1))
