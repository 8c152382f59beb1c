# the whole-file block of bench.py with the other blocks cut down (small similarity problem, no Butina, no synthetic set, no cfg1, no CPU
# samples): is its ETKDG slower inside the full line because of what ran before it in the process?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_alone2; mkdir -p $O
timeout 900 python bench.py --n-query 20000 --n-ref 20000 --butina-n 0 --conformer-mols 200 --cfg1 0 --cpu-seconds 0 --chembl 0 > $O/bench_small.json 2> $O/bench_small.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_alone2/bench_small.json'))
b=d['secondary'].get('conformers_chembl_all')
print({k: b[k] for k in ('value','etkdg_seconds','mmff_seconds','table_assembly_host_seconds')} if b else list(d['secondary']))
PY
tail -3 $O/bench_small.err
