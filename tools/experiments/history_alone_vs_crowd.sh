# the history product of one 250-atom system alone on the chip and of 128 / 512 of them: phase clocks per iteration (NVMK_BFGS_PROFILE=1)
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_alone}; mkdir -p $O
for A in 250 500; do
NVMK_BFGS_PROFILE=1 timeout 600 python tools/bench_large_systems.py --kind dg --atoms $A --copies 1,128,512 --iters 400 --repeat 1 2>&1 | grep -E "^\{|profile" | cut -c1-330 | tee -a $O/alone_vs_crowd.txt
done
