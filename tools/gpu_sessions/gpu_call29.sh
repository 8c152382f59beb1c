# XCD-grouped hand-out of BFGS systems (conformers of a molecule on one XCD): A/B
cd $GRAFT_REPO_ROOT
for g in 1 16 1 16 32; do echo "group $g"; NVMK_BFGS_XCD_GROUP=$g python tools/bench_conformers.py --mols 10000 2>/dev/null | python -c "import json,sys; b=json.loads(sys.stdin.read()); print(b['mmff_s'], b['etkdg_s'], b['mols_per_s_etkdg_plus_mmff'])"; done
