"""Fill the @@NAME@@ placeholders of DESIGN.md / README.md / profiles/README.md from the round's evidence files
(profiles/r05_bench_*.json, r05_pmc_traffic.json): python scripts/fill_docs.py [--check]   (run after scripts/collect_round5.sh)"""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def line(name):
    path = os.path.join(P, 'r05_bench_%s.json' % name)
    return json.loads(open(path).read().strip().splitlines()[-1])


ns, b, c, d, e = (line(w) for w in ('ns', 'b', 'c', 'd', 'e'))
ens = line('ns_ens10')
tr = json.load(open(os.path.join(P, 'r05_pmc_traffic.json')))
v = {}
v['NS_MS'] = '%.1f' % ns['ms_per_step']
v['NS_FRAC'] = '%.3f' % ns['roofline']['frac']
v['NS_FRACCLK'] = '%.3f' % ns['roofline']['frac_at_measured_clock']
v['NS_SCLK'] = '%.0f' % ns['roofline']['sclk_mhz']
v['NS_TF'] = '%.1f' % ns['roofline']['achieved']
v['NS_LAUNCH_MS'] = '%.2f' % ns['roofline']['avg_launch_ms']
v['NS_TRAFFIC'] = '%.1f' % (tr['k_sweep_trmm']['traffic_bytes_per_launch'] / 1e9)
v['CHOL_NS'] = '%.2f' % ns['roofline_fit']['cholesky']['ms']
v['CHOL_NS_FRAC'] = '%.2f' % ns['roofline_fit']['cholesky']['frac']
v['TRTRI_NS'] = '%.2f' % ns['roofline_fit']['trtri']['ms']
v['SERIAL2'] = '%.2f' % (ns['roofline_fit']['cholesky']['ms'] + ns['roofline_fit']['trtri']['ms'])
v['SERIAL'] = '%.1f' % (ns['roofline_fit']['cholesky']['ms'] + ns['roofline_fit']['trtri']['ms'] + 0.25)
v['CHOL_D'] = '%.1f' % d['roofline_fit']['cholesky']['ms']
v['CHOL_D_FRAC'] = '%.3f' % d['roofline_fit']['cholesky']['frac']
v['D_MS'] = '%.1f' % d['ms_per_step']
v['E_MS'] = '%.2f' % e['ms_per_step']
v['CHOL_E'] = '%.2f' % e['roofline_fit']['cholesky']['ms']
v['D_RFF_MS'] = '%.1f' % d['roofline_rff']['ms']
v['D_RFF_FRAC'] = '%.2f' % d['roofline_rff']['frac']
v['D_RFF_FRACCLK'] = '%.2f' % d['roofline_rff']['frac_at_measured_clock']
v['E_RFF_MS'] = '%.2f' % e['roofline_rff']['ms']
v['E_RFF_FRAC'] = '%.2f' % e['roofline_rff']['frac']
v['B_MS'] = '%.1f' % b['ms_per_step']
v['B_FRAC'] = '%.3f' % b['roofline']['frac']
v['C_MS'] = '%.1f' % c['ms_per_step']
v['ENS_MS'] = '%.0f' % ens['ms_per_step']
v['ENS_FRAC'] = '%.3f' % ens['roofline']['frac']
v['XGRAM_MS'] = '%.1f' % ns['stage_ms_per_step_rank0']['cross_gram']
v['CPU_S'] = '%.0f' % ns['cpu_baseline']['seconds_per_step']
v['WARM_MS'] = '%.1f' % ns['warm_step']['ms_per_step'] if 'warm_step' in ns else '?'
v['PLUGIN_WARM_MS'] = '%.1f' % ns['plugin_step']['warm_ms'] if 'plugin_step' in ns else '?'
v['P8_MS'] = '%.0f' % ((ns['stage_ms_per_step_rank0']['cross_gram'] + ns['stage_ms_per_step_rank0']['sweep_trmm'] + ns['stage_ms_per_step_rank0']['acq_topk']) / 8.0
                      + ns['roofline_fit']['cholesky']['ms'] + ns['roofline_fit']['trtri']['ms'] + 0.25 + 0.5)
check = '--check' in sys.argv
for name in ('DESIGN.md', 'README.md', os.path.join('profiles', 'README.md')):
    path = os.path.join(ROOT, name)
    s = open(path).read()
    keys = set(re.findall(r'@@([A-Z0-9_]+)@@', s))
    missing = [k for k in keys if k not in v]
    if missing:
        print('%s: no value for %s' % (name, missing))
    if not check:
        for k in keys:
            if k in v:
                s = s.replace('@@%s@@' % k, v[k])
        open(path, 'w').write(s)
    print('%s: %d placeholders %s' % (name, len(keys), 'found' if check else 'filled'))
print(json.dumps(v, indent=1))
