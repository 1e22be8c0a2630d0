#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03g
mkdir -p $O
export TMPDIR=/tmp
cd /tmp
rocprofv3 -L > $O/counters.txt 2>&1
grep -ciE "mfma" $O/counters.txt
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/tools/prof_gemm_run.py > $O/trace.log 2>&1
echo "trace rc=$?" | tee -a $O/status.txt
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/pmc1 -- python3 $R/tools/prof_gemm_run.py > $O/pmc1.log 2>&1
echo "pmc1 rc=$?" | tee -a $O/status.txt
timeout -k 10 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/pmc2 -- python3 $R/tools/prof_gemm_run.py > $O/pmc2.log 2>&1
echo "pmc2 rc=$?" | tee -a $O/status.txt
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/tools/prof_gemm_run.py > $O/pmc_$c.log 2>&1
  echo "pmc $c rc=$?" | tee -a $O/status.txt
done
cd $R
find $O -name "*_kernel_trace.csv" -size +3M -delete
find $O -name "*agent_info.csv" -delete
python3 tools/pmc_digest.py $O > $O/digest.txt 2>&1
grep -E "gemm|== " $O/digest.txt | cut -c1-200
tail -3 $O/pmc2.log
cat $O/status.txt
