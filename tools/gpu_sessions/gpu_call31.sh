# merged MMFF non-bonded table: parity subset, A/B throughput, phase clocks
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "forcefield or bfgs or ff_ or config_size or mmff or uff or device_chain or constraints or batched" 2>&1 | tail -2
for m in 0 1; do echo "merge $m"; NVMK_MMFF_MERGE=$m python tools/bench_conformers.py --mols 10000 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['mmff_s'], b['etkdg_s'], b['mols_per_s_etkdg_plus_mmff'])"; done
NVMK_BFGS_PROFILE=1 python tools/bench_conformers.py --mols 400 2>&1 >/dev/null | grep "kind 2 systems 39"
