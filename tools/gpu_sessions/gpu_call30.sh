# H pass: permlane swaps instead of LDS shuffles, 32-bit row offsets — parity, microbenchmark, throughput, phase clocks
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x -k "forcefield or bfgs or ff_ or etkdg or config_size or mmff or uff or embed" 2>&1 | tail -2
for n in 144 192; do tools/ubench_hess $n 4096 79 2; done
python tools/bench_conformers.py --mols 10000 2>/dev/null
NVMK_BFGS_PROFILE=1 python tools/bench_conformers.py --mols 400 2>&1 >/dev/null | grep "systems 4096\|systems 40[0-9][0-9]\|systems 39[0-9][0-9]"
