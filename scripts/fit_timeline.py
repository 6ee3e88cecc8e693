"""Timeline of the LAST fit in a rocprofv3 kernel trace (scripts/trace.sh): start, end, duration, kernel, grid, stream."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lo, hi = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (0.0, 1e18)
idx = [i for i, r in enumerate(rows) if 'k_gram_sym' in r['Kernel_Name']]
seg = rows[idx[-1]:]
t0 = int(seg[0]['Start_Timestamp'])
names = ('k_gram_sym', 'k_potrf16', 'k_potrf_diag', 'k_panel_solve16', 'k_panel_trsm', 'k_row_update64', 'k_syrk_update',
         'k_trtri_diag128', 'k_trtri_gemm', 'k_tri_matvec', 'k_rff', 'k_cross_gram', 'k_sweep_trmm')
for r in seg:
    short = [k for k in names if k in r['Kernel_Name']]
    if not short:
        continue
    s, e = (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3
    if lo <= s <= hi:
        print('%9.1f %9.1f %7.1f  %-16s grid %6s q %s' % (s, e, e - s, short[0], r['Grid_Size_X'], r['Queue_Id']))
