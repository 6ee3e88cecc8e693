"""Write the numbers of DESIGN.md / README.md from the round's evidence files: every value sits between <!--K:NAME--> and <!--/K-->
markers and is REPLACED on every run (idempotent), so the documents always say what profiles/r06_* say.
    python scripts/fill_docs.py            rewrite the documents
    python scripts/fill_docs.py --check    exit 1 if a document is stale or names an unknown key
Run after scripts/collect_round6.sh."""
import json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, 'profiles')


def line(name):
    path = os.path.join(P, 'r06_bench_%s.json' % name)
    if not os.path.exists(path):
        return None
    return json.loads(open(path).read().strip().splitlines()[-1])


def text(name):
    path = os.path.join(P, name)
    return open(path).read() if os.path.exists(path) else ''


v = {}
ns, b, c, d, e, ens, sh8 = (line(w) for w in ('ns', 'b', 'c', 'd', 'e', 'ns_ens10', 'ns_share8'))
if ns:
    st = ns['stage_ms_per_step_rank0']
    v['NS_MS'] = '%.1f' % ns['ms_per_step']
    v['NS_SWEEP_MS'] = '%.1f' % st['sweep_trmm']
    v['NS_FRAC'] = '%.3f' % ns['roofline']['frac']
    v['NS_FRACCLK'] = '%.3f' % ns['roofline']['frac_at_measured_clock']
    v['NS_SCLK'] = '%.0f' % ns['roofline']['sclk_mhz']
    v['NS_TF'] = '%.1f' % ns['roofline']['achieved']
    v['NS_LAUNCH_MS'] = '%.2f' % ns['roofline']['avg_launch_ms']
    v['XGRAM_MS'] = '%.1f' % st['cross_gram']
    v['XGRAM_TBS'] = '%.1f' % (8.0 * 8192 * (1 << 20) / (st['cross_gram'] * 1e-3) / 1e12)
    v['ACQ_MS'] = '%.2f' % st['acq_topk']
    v['CHOL_NS'] = '%.2f' % ns['roofline_fit']['cholesky']['ms']
    v['CHOL_NS_FRAC'] = '%.2f' % ns['roofline_fit']['cholesky']['frac']
    v['TRTRI_NS'] = '%.2f' % ns['roofline_fit']['trtri']['ms']
    v['FI_NS'] = '%.2f' % (ns['roofline_fit']['cholesky']['ms'] + ns['roofline_fit']['trtri']['ms'])
    v['SERIAL'] = '%.1f' % (ns['roofline_fit']['cholesky']['ms'] + ns['roofline_fit']['trtri']['ms'] + 0.25)
    v['P8_MS'] = '%.0f' % ((st['cross_gram'] + st['sweep_trmm'] + st['acq_topk']) / 8.0 + float(v['SERIAL']) + 0.5)
    v['CPU_S'] = '%.0f' % ns['cpu_baseline']['seconds_per_step']
    v['CPU_CORES'] = '%d' % ns['cpu_baseline']['cores']
    if 'warm_step' in ns:
        v['WARM_MS'] = '%.1f' % ns['warm_step']['ms_per_step']
    if 'plugin_step' in ns and 'warm_ms' in ns['plugin_step']:
        v['PLUGIN_WARM_MS'] = '%.1f' % ns['plugin_step']['warm_ms']
if b:
    v['B_MS'] = '%.2f' % b['ms_per_step']
    v['B_FRAC'] = '%.3f' % b['roofline']['frac']
if c:
    v['C_MS'] = '%.0f' % c['ms_per_step']
if d:
    v['D_MS'] = '%.1f' % d['ms_per_step']
    v['CHOL_D'] = '%.1f' % d['roofline_fit']['cholesky']['ms']
    v['CHOL_D_FRAC'] = '%.2f' % d['roofline_fit']['cholesky']['frac']
    v['D_RFF_MS'] = '%.1f' % d['roofline_rff']['ms']
    v['D_RFF_FRAC'] = '%.2f' % d['roofline_rff']['frac']
if e:
    v['E_MS'] = '%.2f' % e['ms_per_step']
    v['CHOL_E'] = '%.2f' % e['roofline_fit']['cholesky']['ms']
    v['E_RFF_MS'] = '%.2f' % e['roofline_rff']['ms']
    v['E_RFF_FRAC'] = '%.2f' % e['roofline_rff']['frac']
if ens:
    v['ENS_MS'] = '%.0f' % ens['ms_per_step']
    v['ENS_FRAC'] = '%.3f' % ens['roofline']['frac']
if sh8:
    v['SHARE8_MS'] = '%.0f' % sh8['ms_per_step']
tr = text('r06_pmc_traffic.json')
if tr:
    v['NS_TRAFFIC'] = '%.0f' % (json.loads(tr)['k_sweep_trmm']['traffic_bytes_per_launch'] / 1e9)
# MFMA busy of the default schedule: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), section "tile_order 19"
sq = text('r06_pmc_sq_summary.txt')
m = re.search(r'## tile_order 19(.*)', sq, flags=re.S)
if m:
    busy = re.search(r'SQ_VALU_MFMA_BUSY_CYCLES\s+n=\s*\d+\s+mean=([0-9.e+]+)', m.group(1))
    act = re.search(r'GRBM_GUI_ACTIVE\s+n=\s*\d+\s+mean=([0-9.e+]+)', m.group(1))
    if busy and act:
        v['MFMA_BUSY'] = '%.1f' % (100.0 * float(busy.group(1)) / (1024.0 * float(act.group(1)) / 8.0))
# the factorisation series: "N=8192 chol_tg=1   median 3.921 min ..." and the inverse alone (trtri_ahead=0)
ct = text('r06_chol_taskgraph.txt')
for n in (2048, 8192, 12288, 16384):
    m = re.search(r'N=%d chol_tg=1\s+median ([0-9.]+)' % n, ct)
    if m:
        v['CHOL_%d' % n] = ('%.2f' if n < 16384 else '%.1f') % float(m.group(1))
for n in (8192, 16384):
    m = re.search(r'N=%d trtri_ahead=0 trtri median ([0-9.]+)' % n, ct)
    if m:
        v['TRTRI_%d' % n] = ('%.2f' if n < 16384 else '%.1f') % float(m.group(1))
su = text('r06_gpu_suite.log')
m = re.search(r'(\d+) passed', su)
if m:
    v['GPU_TESTS'] = m.group(1)

check = '--check' in sys.argv
stale = 0
pat = re.compile(r'<!--K:([A-Z0-9_]+)-->(.*?)<!--/K-->', flags=re.S)
for name in ('DESIGN.md', 'README.md'):
    path = os.path.join(ROOT, name)
    if not os.path.exists(path):
        continue
    s = open(path, encoding='utf-8').read()
    keys = set(k for k, _ in pat.findall(s))
    missing = sorted(k for k in keys if k not in v)
    if missing:
        print('%s: no evidence for %s (left as they are)' % (name, missing))

    def sub(mo):
        k = mo.group(1)
        return '<!--K:%s-->%s<!--/K-->' % (k, v.get(k, mo.group(2)))
    s2 = pat.sub(sub, s)
    if s2 != s:
        stale += 1
        if not check:
            open(path, 'w', encoding='utf-8').write(s2)
    print('%s: %d keys, %s' % (name, len(keys), 'stale' if (check and s2 != s) else ('rewritten' if s2 != s else 'up to date')))
if check and stale:
    sys.exit(1)
