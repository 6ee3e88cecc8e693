"""The documents say what the evidence says: every number of DESIGN.md / README.md sits between <!--K:NAME--> markers and is written
by scripts/fill_docs.py from profiles/r06_*; a stale document (a profile refreshed without re-running the script, a hand-edited
number) fails here.  Also: the size budget of DESIGN.md and the evidence pipeline's syntax."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_documents_are_filled_from_the_committed_evidence():
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'scripts', 'fill_docs.py'), '--check'], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'no evidence for' not in r.stdout, r.stdout          # every key of the documents has its evidence file


def test_design_is_a_current_state_document_within_its_budget():
    lines = open(os.path.join(ROOT, 'DESIGN.md'), encoding='utf-8').read().split('\n')
    assert len(lines) <= 400 and max(len(ln) for ln in lines) <= 160
    # the pipeline that produces the evidence parses
    for sh in ('profile_round6.sh', 'collect_round6.sh'):
        assert subprocess.run(['bash', '-n', os.path.join(ROOT, 'scripts', sh)]).returncode == 0
