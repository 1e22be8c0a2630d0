#!/bin/bash
# round 3 evidence run: GEMM bench (hip vs TunableOp torch), GEMM kernel traces + PMC, A2C per-kernel profile, bench lines
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03q
mkdir -p $O
export TMPDIR=/tmp
cd $R
timeout -k 10 400 python tools/bench_gemm_hip.py --tune --rounds 5 > $O/gemm_hip_vs_torch.json 2> $O/gemm.err; echo "gemm rc=$?" | tee -a $O/status.txt
timeout -k 10 200 python tools/bench_rollout_gemm.py > $O/rollout_gemm.json 2> $O/rollout_gemm.err; echo "rollout gemm rc=$?" | tee -a $O/status.txt
timeout -k 10 400 python tools/profile_a2c.py > $O/a2c_profile.txt 2>&1; echo "a2c profile rc=$?" | tee -a $O/status.txt
cd /tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/prof_gemm_run.py > $O/trace.log 2>&1; echo "trace rc=$?" | tee -a $O/status.txt
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/pmc1 -- python3 $R/tools/prof_gemm_run.py > $O/pmc1.log 2>&1; echo "pmc1 rc=$?" | tee -a $O/status.txt
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -- python3 $R/tools/prof_gemm_run.py > $O/pmc2.log 2>&1; echo "pmc2 rc=$?" | tee -a $O/status.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/prof_gemm_run.py > $O/pmc_$c.log 2>&1; echo "pmc $c rc=$?" | tee -a $O/status.txt
done
cd $R
find $O -name "*_kernel_trace.csv" -size +3M -delete; find $O -name "*agent_info.csv" -delete
python3 tools/pmc_digest.py $O > $O/digest.txt 2>&1
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_style.json 2> $O/bench_driver_style.err; echo "bench driver rc=$?" | tee -a $O/status.txt
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?" | tee -a $O/status.txt
cat $O/status.txt
