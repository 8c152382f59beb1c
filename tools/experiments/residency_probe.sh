set -u
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
O=gpurun_out/r04_residency
mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DUBENCH_THREADS=64 tools/ubench_hess.hip -o /tmp/ubh64 2>/dev/null
: > $O/ubench_resident.jsonl
# one wave per system, the whole triangle resident in LDS: 4 per CU (39.5 KiB) and 2 per CU (79 KiB); full chip and one per CU
for ARGS in "64 16384 39.5 2 40" "80 16384 39.5 2 40" "96 8192 79 2 40" "120 8192 79 2 40" "96 256 79 2 40" "144 4096 159 2 40" "144 256 159 2 40" "144 16384 19.5 2 40" "144 8192 39.5 2 40"; do
  timeout 60 /tmp/ubh64 $ARGS >> $O/ubench_resident.jsonl
done
cat $O/ubench_resident.jsonl
CACHE=/tmp/nvmk_lib_cache
for L in auto 26 40 auto 40; do
  echo "lds=$L" >> $O/ab_lds_budget.txt
  NVMK_BFGS_LDS=$L timeout 300 python tools/bench_conformers.py --mols 10000 --cache $CACHE 2>/dev/null | grep '^{' | python -c "
import sys,json
for line in sys.stdin:
    d=json.loads(line); print({k:(round(v,4) if isinstance(v,float) else v) for k,v in d.items() if k in ('etkdg_s','mmff_s','mols_per_s_etkdg_plus_mmff','etkdg_conformers')})" >> $O/ab_lds_budget.txt
done
cat $O/ab_lds_budget.txt
