#!/bin/bash
# The ONE runner handed to gpurun (round 4 on; the per-call scripts of rounds 2-3 are summarised in tools/gpu_sessions/README.md):
#   gpurun --timeout 1500 -- bash tools/gpu_session.sh <name> <step> [<step> ...]
# Every step writes into gpurun_out/<name>/ and is wrapped in its own timeout.  Steps:
#   ubench_hess         inverse-Hessian pass alone, full chip, one / two / four waves per system
#   ubench_hess_pmc     FETCH_SIZE / WRITE_SIZE of the same (separate --pmc passes)
#   parity_conformers   BFGS / ETKDG / force-field GPU parity tests
#   ab_conformers       tools/bench_conformers.py --mols 10000 with every nvmolkit_amd/lib/libnvmolkit_amd_<variant>.so beside the product
#   ab_sched            bench_conformers with NVMK_BFGS_SCHED=hw and queue, alternating
#   conformer_traffic_seq  the PMC passes with the size classes one after the other (NVMK_BFGS_OVERLAP=0)
#   chembl_tests        tests/test_chembl_conformers_gpu.py + tests/test_chembl_whole_file_gpu.py
#   sq_counters         SQ PMC counters of the BFGS kernels and the row-panel count kernel -> sq_counters.json (tools/sq_summary.py)
#   timeline            BFGS per-system timeline of one 10 000-molecule run (NVMK_BFGS_PROFILE=1 NVMK_BFGS_TIMELINE)
#   conformer_traffic   tools/profile_conformer_traffic.sh 2000
#   table_tests         the table builder's GPU tests + the suites that build batches through it
#   ab_batch            bench_conformers with the default 16384 attempts per batch against equalised batches (16667 x 6, 14286 x 7)
#   chembl256_timeline  the ChEMBL topologies of up to 256 atoms with the per-system BFGS timeline, summarised per size class
#   wave8_tests         BFGS parity + inverse-Hessian update tests (the eight-wave class of round 5)
#   ab_wave8            ChEMBL topologies of up to 256 atoms with NVMK_BFGS_WAVE8=0 (four waves for every large system) and the default, alternating
#   chembl_all_timeline the whole file with the per-system BFGS timeline, summarised per class and team width
#   chembl_all          every molecule of the ChEMBL file (up to 1063 atoms)
#   conf10k             tools/bench_conformers.py --mols 10000 (resident tables, and end to end from the host arrays)
#   batch_ab            the same candidates end to end (table assembly inside the clock), alternating, twice
#   batch_sweep         attempts per ETKDG batch (BATCHES="-1 33334 ...", EXTRA_ENV=...) on the whole ChEMBL file and on the synthetic set, resident tables
#   pytest_gpu          the whole -m gpu suite (stops at the first failure; pytest_gpu_all: runs on)
#   smoke               __graft_entry__.smoke()
#   bench               python bench.py (default flags), plain
#   bench_stats         python bench.py under rocprofv3 --kernel-trace --stats
#   bench_traffic       tools/profile_bench_traffic.sh (PMC of the dense launches)
#   markers             rocprofv3 --marker-trace --kernel-trace of one ETKDG + MMFF run (roctx ranges of the library)
#   chembl              the conformer block on the ChEMBL topologies (tools/bench_conformers.py --set chembl)
#   strong              bench.py strong-scaling conformer mode as one rank over RCCL
#   butina_tests        the clustering, full-size, benchmark-molecule and sharded-Butina GPU tests
#   diag_chembl         tools/diag_chembl_energy.py (which conformers end an MMFF minimisation above their starting energy)
#   ab_half_prune       (with tools/experiments/panel_half_k_prune.patch applied) tools/bench_butina.py at 1M on the product library and on
#                       lib/libnvmolkit_amd_noprune.so (NVMK_EXTRA_HIPCC_FLAGS=-DNVMK_PANEL_NO_HALF_PRUNE NVMK_BUILD_VARIANT=noprune)
#   panel_fetch         tools/profile_panel_fetch.sh: FETCH_SIZE of the row-panel count kernel, product and variant libraries
#   team_tests          the cooperative BFGS class's tests (tests/test_bfgs_parity_gpu.py -k team)
#   history_ab          large systems, 400 iterations, inverse Hessian as the triangle and as the history of its updates
#   chembl_all_history  the whole ChEMBL file with NVMK_BFGS_HISTORY=auto and 0
#   trajectory_depth    tools/probe_history_depth.py: how far the two forms of a team's inverse Hessian and the oracle agree, by depth
#   ubench_team_pass    tools/ubench_team_pass.hip: the team pass alone, chip-wide, by size / width / threads
#   team_sweep          bench_large_systems over team widths x threads per workgroup
#   final               PMC traffic files (dense launches, conformers) and then the bench line that quotes them
#   large_profile       the same with NVMK_BFGS_PROFILE=1: the kernels' phase clocks per size
#   large_systems       tools/bench_large_systems.py: microseconds per BFGS iteration of 300 ... 1063-atom systems, 1 ... 256 copies
#   butina_bench        tools/bench_butina.py 1000000 --repeat 3 on the planted clusters and on the wide-popcount-spread set
#   butina              tools/bench_butina.py + clustering tests (ab_butina: the bench alone, tile against panel kernel)
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
NAME=${1:?session name}
shift
O=$ROOT/gpurun_out/$NAME
mkdir -p $O
cd $ROOT
export TMPDIR=/tmp
export NVMK_ROOT=$ROOT
CACHE=/tmp/nvmk_lib_cache
pick() { python -c "import sys,json
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); print('$1', {k: (round(d[k],4) if isinstance(d[k],float) else d[k]) for k in d if k in ('mols_per_s_etkdg_plus_mmff','etkdg_s','mmff_s','etkdg_conformers','mmff_converged_frac','mols','mean_atoms','mols_per_s_end_to_end','end_to_end_s','table_assembly_host_s','mmff_tables_wait_s','table_assembly_steps')})"; }

for STEP in "$@"; do
  echo "==== $STEP ($(date +%T))"
  case $STEP in
    ubench_hess)
      for T in 64 128 256; do
        hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DUBENCH_THREADS=$T tools/ubench_hess.hip -o /tmp/ubh_${T}_0 2>/dev/null &
      done
      wait
      : > $O/ubench_hess.jsonl
      for N in 96 144 176; do timeout 60 /tmp/ubh_64_0 $N 16384 19.5 2 40 >> $O/ubench_hess.jsonl; done
      timeout 60 /tmp/ubh_64_0 144 256 19.5 2 40 >> $O/ubench_hess.jsonl     # one system per CU: latency, no contention
      for N in 200 256; do timeout 60 /tmp/ubh_128_0 $N 8192 39.5 2 40 >> $O/ubench_hess.jsonl; done
      for N in 192 300 384; do timeout 60 /tmp/ubh_256_0 $N 4096 79 2 40 >> $O/ubench_hess.jsonl; done
      cat $O/ubench_hess.jsonl
      ;;
    ubench_hess_pmc)
      cd /tmp
      for C in FETCH_SIZE WRITE_SIZE; do
        timeout 120 rocprofv3 --pmc $C -f csv -d $O/ubh_pmc_$C -- /tmp/ubh_64_0 144 16384 19.5 2 40 > $O/ubh_pmc_$C.log 2>&1
      done
      cd $ROOT
      python - "$O" <<'PY' | tee $O/ubench_hess_pmc.json
import csv, glob, json, sys
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    out[c + "_KiB"] = sum(float(r["Counter_Value"]) for f in glob.glob(f"{sys.argv[1]}/ubh_pmc_{c}/**/*_counter_collection.csv", recursive=True)
                          for r in csv.DictReader(open(f)) if "pass_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c)
tri = (144 * 144 // 2) * 8 * 16384 * 42   # warm-up launch 2 passes + timed launch 40
out["read_ratio"] = 2 * out["FETCH_SIZE_KiB"] * 1024 / tri   # gfx950: FETCH_SIZE counts half the bytes of 128-byte requests
out["write_ratio"] = out["WRITE_SIZE_KiB"] * 1024 / tri
print(json.dumps(out, indent=1))
PY
      ;;
    parity_conformers)
      ( time timeout 900 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_etkdg_gpu.py tests/test_forcefield_gpu.py tests/test_etkdg_driver_gpu.py -m gpu -q -x ) > $O/parity_conformers.log 2>&1
      tail -4 $O/parity_conformers.log
      ;;
    ab_conformers)
      : > $O/ab_conformers.txt
      for i in 1 2; do
        for L in "" $(ls nvmolkit_amd/lib/ | sed -n 's/^libnvmolkit_amd_\(.*\)\.so$/\1/p'); do
          LIBP=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd${L:+_$L}.so
          NVMOLKIT_AMD_LIB=$LIBP timeout 300 python tools/bench_conformers.py --mols 10000 --repeat 2 --cache $CACHE 2>/dev/null | pick "${L:-product}" | tee -a $O/ab_conformers.txt
        done
      done
      ;;
    ab_sched)
      : > $O/ab_sched.txt
      for i in 1 2; do
        for M in hw queue; do
          NVMK_BFGS_SCHED=$M timeout 300 python tools/bench_conformers.py --mols 10000 --repeat 2 --cache $CACHE 2>/dev/null | pick "sched=$M" | tee -a $O/ab_sched.txt
        done
      done
      ;;
    ab_prune)
      : > $O/ab_prune.txt
      for i in 1 2; do for M in 0 1; do
        NVMK_ETKDG_PRUNE=$M timeout 300 python tools/bench_conformers.py --mols 10000 --repeat 2 --cache $CACHE 2>/dev/null | pick "prune=$M" | tee -a $O/ab_prune.txt
      done; done
      for M in 0 1; do
        NVMK_ETKDG_PRUNE=$M timeout 600 python tools/bench_conformers.py --set chembl --mols 10000 --repeat 2 --cache $CACHE 2>/dev/null | pick "chembl prune=$M" | tee -a $O/ab_prune.txt
      done
      ;;
    sweep_conformers)
      : > $O/sweep_conformers.txt
      run() { env "$@" timeout 300 python tools/bench_conformers.py --mols 10000 --repeat 2 --cache $CACHE $EXTRA 2>/dev/null | pick "$* $EXTRA" | tee -a $O/sweep_conformers.txt; }
      EXTRA=""
      run NVMK_X=0
      for G in 8 32 64; do run NVMK_BFGS_XCD_GROUP=$G; done
      for W in 144 160 200; do run NVMK_BFGS_WAVE=$W; done
      for W in 224 288; do run NVMK_BFGS_WAVE2=$W; done
      for B in 12288 24576 32768; do EXTRA="--batch-size $B"; run NVMK_X=0; done
      EXTRA=""
      run NVMK_X=0
      ;;
    ab_workers)
      : > $O/ab_workers.txt
      for W in "16384 1" "16384 2" "8192 2" "8192 3" "4096 4"; do
        set -- $W
        timeout 300 python tools/bench_conformers.py --mols 10000 --repeat 2 --batch-size $1 --batches-per-gpu $2 --cache $CACHE 2>/dev/null | pick "batch=$1 workers=$2" | tee -a $O/ab_workers.txt
      done
      ;;
    new_abi_tests)
      ( time timeout 900 python -m pytest tests/test_multi_gpu_abi_gpu.py tests/test_cxx_example.py tests/test_etkdg_gpu.py tests/test_config_size_gpu.py -m gpu -q -x ) > $O/new_abi_tests.log 2>&1
      tail -8 $O/new_abi_tests.log
      ;;
    conformer_traffic_seq)
      rm -rf gpurun_out/pmc_traffic/conf_fetch gpurun_out/pmc_traffic/conf_write
      NVMK_BFGS_OVERLAP=0 timeout 900 bash tools/profile_conformer_traffic.sh 2000 > $O/conformer_traffic_seq.log 2>&1
      cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json $O/pmc_hbm_traffic_conformers_classes_one_after_the_other.json 2>/dev/null
      tail -30 $O/conformer_traffic_seq.log
      ;;
    chembl256_team)
      : > $O/chembl256_team.txt
      for T in ${TEAM_MINS:-1068 656}; do
        echo "== NVMK_BFGS_TEAM=$T" | tee -a $O/chembl256_team.txt
        NVMK_BFGS_TEAM=$T timeout 600 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 256 --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | cut -c1-420 | tee -a $O/chembl256_team.txt
      done
      ;;
    chembl256)
      timeout 600 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 256 --cache $CACHE 2> $O/chembl256.err | tee $O/chembl256.json | cut -c1-700
      ;;
    chembl_tests)
      ( time timeout 1500 python -m pytest tests/test_chembl_conformers_gpu.py tests/test_chembl_whole_file_gpu.py -m gpu -q ) > $O/chembl_tests.log 2>&1
      tail -15 $O/chembl_tests.log
      ;;
    sq_counters)
      # SQ counters of the BFGS kernels (2000 molecules of the benchmark set) and of the row-panel count kernel (1M rows), two passes
      # of eight SQ counters each (MI355X_MICROARCH.md: 8 SQ slots per pass), never combined with tracing
      cd /tmp
      CONF="python $ROOT/tools/bench_conformers.py --mols 2000 --cache $CACHE"
      PANEL="python $ROOT/tools/bench_butina.py 1000000 --skip-butina"
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES -f csv -d $O/sq_conf_1 -- $CONF > $O/sq_conf_1.log 2>&1
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -f csv -d $O/sq_conf_2 -- $CONF > $O/sq_conf_2.log 2>&1
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES -f csv -d $O/sq_panel_1 -- $PANEL > $O/sq_panel_1.log 2>&1
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE -f csv -d $O/sq_panel_2 -- $PANEL > $O/sq_panel_2.log 2>&1
      # (round 6: the Morgan kernel once, so that "LDS / latency bound" has a number: LDS instructions, bank conflicts, waits)
      MORGAN="python $ROOT/tools/bench_morgan.py --mols 1000000"
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES -f csv -d $O/sq_morgan_1 -- $MORGAN > $O/sq_morgan_1.log 2>&1
      timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD -f csv -d $O/sq_morgan_2 -- $MORGAN > $O/sq_morgan_2.log 2>&1
      cd $ROOT
      python tools/sq_summary.py $O > $O/sq_counters.json && head -c 3500 $O/sq_counters.json
      rm -rf $O/sq_conf_1 $O/sq_conf_2 $O/sq_panel_1 $O/sq_panel_2 $O/sq_morgan_1 $O/sq_morgan_2
      ;;
    timeline)
      rm -f $O/bfgs_timeline.txt
      NVMK_BFGS_PROFILE=1 NVMK_BFGS_TIMELINE=$O/bfgs_timeline.txt timeout 300 python tools/bench_conformers.py --mols 10000 --cache $CACHE > $O/timeline_run.log 2>&1
      tail -2 $O/timeline_run.log | cut -c1-300
      python tools/bfgs_timeline.py $O/bfgs_timeline.txt > $O/bfgs_timeline_summary.json && cat $O/bfgs_timeline_summary.json | head -60
      gzip -f $O/bfgs_timeline.txt
      ;;
    conformer_traffic)
      rm -rf gpurun_out/pmc_traffic/conf_fetch gpurun_out/pmc_traffic/conf_write
      timeout 900 bash tools/profile_conformer_traffic.sh 2000 > $O/conformer_traffic.log 2>&1
      cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json $O/ 2>/dev/null
      # (bench.py quotes the file from profiles/ while the kernel sources' digest matches: in place for the bench steps of this session)
      mkdir -p profiles/r06_conformers && cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json profiles/r06_conformers/ 2>/dev/null
      tail -30 $O/conformer_traffic.log
      ;;
    table_tests)
      ( time timeout 900 python -m pytest tests/test_table_build_gpu.py tests/test_cxx_example.py tests/test_device_chain_gpu.py tests/test_etkdg_gpu.py tests/test_forcefield_gpu.py tests/test_constraints.py -m gpu -q ) > $O/table_tests.log 2>&1
      tail -15 $O/table_tests.log
      ;;
    ab_batch)
      : > $O/ab_batch.txt
      for i in 1 2; do for B in -1 16667 14286; do
        timeout 300 python tools/bench_conformers.py --mols 10000 --repeat 2 --batch-size $B --cache $CACHE 2>/dev/null | pick "batch=$B" | tee -a $O/ab_batch.txt
      done; done
      ;;
    chembl256_timeline)
      rm -f $O/chembl256_timeline.txt
      NVMK_BFGS_PROFILE=1 NVMK_BFGS_TIMELINE=$O/chembl256_timeline.txt timeout 900 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 256 --cache $CACHE > $O/chembl256_timeline_run.log 2>&1
      grep '^{' $O/chembl256_timeline_run.log | tail -1 | cut -c1-600
      python tools/bfgs_timeline.py $O/chembl256_timeline.txt > $O/chembl256_timeline_summary.json && python -c "import json; d=json.load(open('$O/chembl256_timeline_summary.json')); print(json.dumps({k: d[k] for k in ('systems','launch_groups','total_mean_occupancy','total_tail_ms_below_half','wall_ms_first_to_last')})); print(json.dumps(d['by_class'], indent=0))"
      gzip -f $O/chembl256_timeline.txt; rm -f $O/chembl256_timeline.txt.gz
      ;;
    wave8_tests)
      ( time timeout 1200 python -m pytest tests/test_bfgs_parity_gpu.py tests/test_hessian_update_gpu.py -m gpu -q -x ) > $O/wave8_tests.log 2>&1
      tail -8 $O/wave8_tests.log
      ;;
    ab_wave8)
      : > $O/ab_wave8.txt
      for W in 0 656 0 656; do
        echo "NVMK_BFGS_WAVE8=$W" | tee -a $O/ab_wave8.txt
        NVMK_BFGS_WAVE8=$W timeout 600 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 256 --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | cut -c1-700 | tee -a $O/ab_wave8.txt
      done
      ;;
    chembl_all_timeline)
      rm -f $O/chembl_all_timeline.txt
      NVMK_ETKDG_TIMING=1 NVMK_BFGS_PROFILE=1 NVMK_BFGS_TIMELINE=$O/chembl_all_timeline.txt timeout 1500 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 100000 --cache $CACHE > $O/chembl_all_timeline_run.log 2>&1
      grep '^{' $O/chembl_all_timeline_run.log | tail -1 | cut -c1-600
      python tools/bfgs_timeline.py $O/chembl_all_timeline.txt > $O/chembl_all_timeline_summary.json && python -c "import json; d=json.load(open('$O/chembl_all_timeline_summary.json')); print(json.dumps({k: d[k] for k in ('systems','launch_groups','total_mean_occupancy','total_tail_ms_below_half','wall_ms_first_to_last')})); print(json.dumps(d['by_team_width'], indent=0)); print(json.dumps(d['by_class'], indent=0)); print(json.dumps(sorted(d['groups'], key=lambda g: -g['span_ms'])[:6], indent=0)); [print(json.dumps(x)) for x in d.get('by_kind_and_size', [])]"
      gzip -f $O/chembl_all_timeline.txt; rm -f $O/chembl_all_timeline.txt.gz
      ;;
    chembl_all_traffic)
      # FETCH_SIZE / WRITE_SIZE of the BFGS kernels (team kernels included) on the WHOLE benchmark file, separate --pmc passes
      rm -rf gpurun_out/pmc_traffic/conf_fetch gpurun_out/pmc_traffic/conf_write
      BENCH_EXTRA="--set chembl --max-atoms 100000" timeout 1500 bash tools/profile_conformer_traffic.sh 10000 > $O/chembl_all_traffic.log 2>&1
      mkdir -p profiles/r06_conformers
      cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json $O/pmc_hbm_traffic_conformers_chembl_whole_file.json 2>/dev/null && cp $O/pmc_hbm_traffic_conformers_chembl_whole_file.json profiles/r06_conformers/
      tail -30 $O/chembl_all_traffic.log
      ;;
    ab_chembl_all)
      # the whole ChEMBL file on the product library and on every variant library beside it (nvmolkit_amd/lib/libnvmolkit_amd_<variant>.so), alternating
      : > $O/ab_chembl_all.txt
      for i in 1 2; do
        for L in "" $(ls nvmolkit_amd/lib/ | sed -n 's/^libnvmolkit_amd_\(.*\)\.so$/\1/p'); do
          LIBP=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd${L:+_$L}.so
          echo "== ${L:-product}" | tee -a $O/ab_chembl_all.txt
          NVMOLKIT_AMD_LIB=$LIBP timeout 900 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 100000 --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | cut -c1-420 | tee -a $O/ab_chembl_all.txt
        done
      done
      ;;
    chembl_all_share)
      : > $O/chembl_all_share.txt
      for K in ${SHARES:-1024 2048}; do
        echo "== NVMK_BFGS_TEAM_SHARE_KB=$K ${EXTRA_ENV:-}" | tee -a $O/chembl_all_share.txt
        env NVMK_BFGS_TEAM_SHARE_KB=$K ${EXTRA_ENV:-} timeout 900 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 100000 --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | cut -c1-420 | tee -a $O/chembl_all_share.txt
      done
      ;;
    chembl_all)
      timeout 1200 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 100000 --end-to-end --cache $CACHE 2> $O/chembl_all.err | tee $O/chembl_all.json | cut -c1-900
      tail -3 $O/chembl_all.err
      ;;
    batch_sweep)
      : > $O/batch_sweep.txt
      for B in ${BATCHES:--1 33334 50000}; do
        echo "== chembl whole file, --batch-size $B ${EXTRA_ENV:-}" | tee -a $O/batch_sweep.txt
        env ${EXTRA_ENV:-} timeout 900 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 100000 --batch-size $B --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | cut -c1-420 | tee -a $O/batch_sweep.txt
        echo "== synthetic 10000, --batch-size $B ${EXTRA_ENV:-}" | tee -a $O/batch_sweep.txt
        env ${EXTRA_ENV:-} timeout 600 python tools/bench_conformers.py --mols 10000 --repeat 2 --batch-size $B --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | cut -c1-420 | tee -a $O/batch_sweep.txt
      done
      ;;
    batch_ab)
      # end to end (table assembly inside the clock), the candidates alternating on one box
      : > $O/batch_ab.txt
      for i in 1 2; do
        for B in ${BATCHES:--1 33334 50000}; do
          echo "== synthetic 10000 end to end, --batch-size $B" | tee -a $O/batch_ab.txt
          timeout 600 python tools/bench_conformers.py --mols 10000 --repeat 3 --end-to-end --batch-size $B --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | pick b$B | tee -a $O/batch_ab.txt
          echo "== chembl whole file end to end, --batch-size $B" | tee -a $O/batch_ab.txt
          timeout 900 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 100000 --end-to-end --batch-size $B --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | pick b$B | tee -a $O/batch_ab.txt
        done
      done
      ;;
    conf10k)
      timeout 600 python tools/bench_conformers.py --mols 10000 --repeat 2 --end-to-end --cache $CACHE 2> $O/conf10k.err | tee $O/conf10k.json | pick conf10k
      tail -3 $O/conf10k.err
      ;;
    pytest_gpu_all)
      ( time timeout 2400 python -m pytest tests -m gpu -q ) > $O/pytest_gpu.log 2>&1
      tail -25 $O/pytest_gpu.log
      ;;
    pytest_gpu)
      ( time timeout 2400 python -m pytest tests -m gpu -q -x ) > $O/pytest_gpu.log 2>&1
      tail -5 $O/pytest_gpu.log
      ;;
    smoke)
      timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1
      tail -3 $O/smoke.log
      ;;
    bench)
      timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
      tail -c 3000 $O/bench.json
      ;;
    bench_stats)
      cd /tmp
      timeout 1200 rocprofv3 --kernel-trace --stats -f csv -d $O/bench_stats -- python $ROOT/bench.py > $O/bench_under_rocprof.json 2> $O/bench_stats.err
      cd $ROOT
      f=$(find $O/bench_stats -name '*kernel_stats.csv' | head -1)
      [ -n "$f" ] && cp $f $O/kernel_stats_bench_py.csv && head -12 $O/kernel_stats_bench_py.csv | cut -c1-200
      rm -rf $O/bench_stats
      ;;
    bench_traffic)
      rm -rf gpurun_out/pmc_traffic/fetch gpurun_out/pmc_traffic/write
      timeout 900 bash tools/profile_bench_traffic.sh > $O/bench_traffic.log 2>&1
      cp gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json $O/ 2>/dev/null
      mkdir -p profiles/r06_similarity && cp gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json profiles/r06_similarity/ 2>/dev/null
      tail -20 $O/bench_traffic.log
      ;;
    markers)
      cd /tmp
      timeout 600 rocprofv3 --marker-trace --kernel-trace -f csv -d $O/markers -- python $ROOT/tools/bench_conformers.py --mols 2000 --cache $CACHE > $O/markers_run.log 2>&1
      cd $ROOT
      f=$(find $O/markers -name '*marker_api_trace.csv' | head -1)
      [ -n "$f" ] && cp $f $O/marker_trace.csv && python tools/marker_summary.py $O/marker_trace.csv | tee $O/marker_summary.txt | head -60
      rm -rf $O/markers
      ;;
    chembl)
      timeout 900 python tools/bench_conformers.py --set chembl --mols 10000 --repeat 2 --cache $CACHE 2> $O/chembl.err | tee $O/chembl.json | cut -c1-1500
      ;;
    strong)
      NVMK_BENCH_SINGLE_RANK_COLLECTIVES=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
        bench.py --gpus 1 --steps 3 --warmup 1 --conformer-total 100000 > $O/bench_strong_single_rank.json 2> $O/bench_strong.err
      tail -c 2500 $O/bench_strong_single_rank.json
      ;;
    butina_tests)
      ( time timeout 1200 python -m pytest tests/test_clustering_gpu.py tests/test_full_size_gpu.py tests/test_benchmark_molecules_gpu.py tests/test_distributed_gpu.py -m gpu -q -x ) > $O/butina_tests.log 2>&1
      tail -8 $O/butina_tests.log
      ;;
    diag_chembl)
      timeout 900 python tools/diag_chembl_energy.py > $O/diag_chembl.log 2>&1; tail -16 $O/diag_chembl.log
      timeout 900 python tools/diag_chembl_energy.py 16384 > $O/diag_chembl_16384.log 2>&1; tail -6 $O/diag_chembl_16384.log
      ;;
    ab_half_prune)
      : > $O/ab_half_prune.txt
      for i in 1 2; do for L in "" noprune; do
        LIBP=$ROOT/nvmolkit_amd/lib/libnvmolkit_amd${L:+_$L}.so
        echo "== ${L:-product}" | tee -a $O/ab_half_prune.txt
        NVMOLKIT_AMD_LIB=$LIBP timeout 300 python tools/bench_butina.py 1000000 --repeat 3 2>/dev/null | tail -1 | cut -c1-500 | tee -a $O/ab_half_prune.txt
        NVMOLKIT_AMD_LIB=$LIBP timeout 300 python tools/bench_butina.py 1000000 --spread --repeat 3 2>/dev/null | tail -1 | cut -c1-500 | tee -a $O/ab_half_prune.txt
      done; done
      ;;
    panel_fetch)
      bash tools/profile_panel_fetch.sh $O 2>&1 | tail -40
      ;;
    history_ab)
      # the inverse Hessian of a team's system as the packed triangle (0) and as the history of its updates (auto): microseconds per
      # iteration of large systems over a 400-iteration minimisation, alone and 64 at a time
      : > $O/history_ab.txt
      for H in 0 auto; do for K in ${HIST_KINDS:-dg mmff}; do
        echo "== NVMK_BFGS_HISTORY=$H $K ${EXTRA_ENV:-}" | tee -a $O/history_ab.txt
        env ${EXTRA_ENV:-} NVMK_BFGS_HISTORY=$H timeout 600 python tools/bench_large_systems.py --kind $K --atoms ${LARGE_ATOMS:-250,400,700,1063} --copies ${LARGE_COPIES:-1,64} --iters ${HIST_ITERS:-400} --repeat 1 2>/dev/null | cut -c1-220 | tee -a $O/history_ab.txt
      done; done
      ;;
    chembl_all_history)
      : > $O/chembl_all_history.txt
      for H in ${HIST_MODES:-auto 0}; do
        echo "== NVMK_BFGS_HISTORY=$H ${EXTRA_ENV:-}" | tee -a $O/chembl_all_history.txt
        env ${EXTRA_ENV:-} NVMK_BFGS_HISTORY=$H timeout 900 python tools/bench_conformers.py --set chembl --mols 10000 --max-atoms 100000 --cache $CACHE 2>/dev/null | grep '^{' | tail -1 | cut -c1-1400 | tee -a $O/chembl_all_history.txt
      done
      ;;
    trajectory_depth)
      timeout 600 python tools/probe_history_depth.py 2>&1 | tail -40 | tee $O/trajectory_depth.txt
      ;;
    team_tests)
      ( time timeout 1200 python -m pytest tests/test_bfgs_parity_gpu.py -m gpu -q -x -k "team" ) > $O/team_tests.log 2>&1
      tail -15 $O/team_tests.log
      ;;
    ubench_team_ahead)
      for A in ${AHEADS:-0 2 3 4}; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DNVMK_HESS_AHEAD=$A ${UTP_FLAGS:-} tools/ubench_team_pass.hip -o /tmp/utp_a$A 2>/dev/null & done; wait
      : > $O/ubench_team_ahead.jsonl
      for N in ${UTP_N:-1200 2000 4252}; do for W in ${UTP_W:-8 32}; do for A in ${AHEADS:-0 2 3 4}; do
        echo -n "ahead $A " | tee -a $O/ubench_team_ahead.jsonl; timeout 120 /tmp/utp_a$A $N $W 256 10 | tee -a $O/ubench_team_ahead.jsonl
      done; done; done
      ;;
    final)
      # everything a round's closing line is bound to (tests/test_bench_contract.py): the PMC traffic files of the kernel sources as
      # they are, THEN the bench line that quotes them, the rocprofv3 statistics of the same command, the SQ counters
      rm -rf gpurun_out/pmc_traffic
      timeout 900 bash tools/profile_bench_traffic.sh > $O/bench_traffic.log 2>&1
      mkdir -p profiles/r06_similarity profiles/r06_conformers
      cp gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json $O/ 2>/dev/null && cp gpurun_out/pmc_traffic/pmc_hbm_traffic_bench_launch.json profiles/r06_similarity/
      timeout 900 bash tools/profile_conformer_traffic.sh 2000 > $O/conformer_traffic.log 2>&1
      cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json $O/ 2>/dev/null && cp gpurun_out/pmc_traffic/pmc_hbm_traffic_conformers.json profiles/r06_conformers/
      tail -5 $O/bench_traffic.log; tail -5 $O/conformer_traffic.log
      timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err
      tail -c 1500 $O/bench.json
      ;;
    ubench_team_occ)
      hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include tools/ubench_team_pass.hip -o /tmp/utp_occ 2>/dev/null
      : > $O/ubench_team_occ.jsonl
      for N in ${UTP_N:-2000 4252}; do for G in ${UTP_G:-32 64 128 256}; do
        timeout 120 /tmp/utp_occ $N 8 $G 10 | tee -a $O/ubench_team_occ.jsonl
      done; done
      ;;
    ubench_team_pass)
      for T in 512 256; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -DUBENCH_THREADS=$T tools/ubench_team_pass.hip -o /tmp/utp_$T 2>/dev/null & done; wait
      : > $O/ubench_team_pass.jsonl
      for N in ${UTP_N:-1200 2000 4252}; do for W in ${UTP_W:-1 8 32}; do
        timeout 120 /tmp/utp_512 $N $W 256 10 >> $O/ubench_team_pass.jsonl
        timeout 120 /tmp/utp_256 $N $W 512 10 1024 >> $O/ubench_team_pass.jsonl
      done; done
      cat $O/ubench_team_pass.jsonl
      ;;
    team_sweep)
      : > $O/team_sweep.txt
      for T in 512 256; do for W in ${SWEEP_WIDTHS:-4 8 16 32}; do
        echo "== threads $T width $W" | tee -a $O/team_sweep.txt
        NVMK_BFGS_TEAM_THREADS=$T NVMK_BFGS_TEAM_WIDTH=$W timeout 300 python tools/bench_large_systems.py --kind ${SWEEP_KIND:-dg} --atoms ${LARGE_ATOMS:-300,500,800,1063} --copies ${LARGE_COPIES:-1,256} --repeat 1 2>/dev/null | cut -c1-220 | tee -a $O/team_sweep.txt
      done; done
      ;;
    large_profile)
      for K in dg mmff; do
        NVMK_BFGS_PROFILE=1 timeout 600 python tools/bench_large_systems.py --kind $K --atoms ${LARGE_ATOMS:-300,500,1063} --copies ${LARGE_COPIES:-1,64} --repeat 1 2>&1 | grep -E "^\{|profile" | tee -a $O/large_profile.txt
      done
      ;;
    large_systems)
      : > $O/large_systems.jsonl
      for K in dg mmff; do
        timeout 900 python tools/bench_large_systems.py --kind $K --atoms ${LARGE_ATOMS:-300,500,800,1063} --copies ${LARGE_COPIES:-1,8,64,256} 2>> $O/large_systems.err | tee -a $O/large_systems.jsonl
      done
      ;;
    butina_bench)
      timeout 300 python tools/bench_butina.py 1000000 --repeat 3 2>/dev/null | tail -1 | cut -c1-500 | tee $O/butina_bench.txt
      timeout 300 python tools/bench_butina.py 1000000 --spread --repeat 3 2>/dev/null | tail -1 | cut -c1-500 | tee -a $O/butina_bench.txt
      ;;
    ab_butina)
      for T in tile panel tile panel; do
        echo "NVMK_COUNT_KERNEL=$T" | tee -a $O/bench_butina.txt
        NVMK_COUNT_KERNEL=$T timeout 300 python tools/bench_butina.py 2>/dev/null | tail -4 | cut -c1-600 | tee -a $O/bench_butina.txt
      done
      ;;
    butina)
      ( time timeout 900 python -m pytest tests/test_clustering_gpu.py tests/test_full_size_gpu.py tests/test_benchmark_molecules_gpu.py -m gpu -q -x ) > $O/butina_tests.log 2>&1
      tail -3 $O/butina_tests.log
      for T in tile panel tile panel; do
        echo "NVMK_COUNT_KERNEL=$T" | tee -a $O/bench_butina.txt
        NVMK_COUNT_KERNEL=$T timeout 300 python tools/bench_butina.py 2>/dev/null | tail -4 | tee -a $O/bench_butina.txt
      done
      ;;
    *) echo "unknown step $STEP" ;;
  esac
done
echo "==== done ($(date +%T))"
