#!/bin/bash
# Round 3, call 16: full GPU suite + smoke + default bench at the current commit.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
O=$ROOT/gpurun_out/${1:-r03_call16}
mkdir -p $O
cd $ROOT
( time timeout 1100 python -m pytest tests -m gpu -q ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout 500 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json; tail -3 $O/bench.err
python - $O/bench.json <<'P'
import json,sys
b=json.load(open(sys.argv[1]))
print(b['value'], b['roofline']['frac'], b['cpu_baseline'])
for k,v in b['secondary'].items():
    print(k, {kk:vv for kk,vv in v.items() if not isinstance(vv,(dict,list))})
    if 'cpu_baseline' in v: print('  cpu', v['cpu_baseline'])
P
