#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r02c
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests/test_hip_parity.py tests/test_step_many_gpu.py tests/test_shim_dropin.py -x -q -m gpu > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/status.txt
tail -15 $O/tests.log
for i in 1 2; do
timeout -k 10 300 python bench.py --n-bs 16 --n-ue 200 --envs 8192 --steps 300 --warmup 30 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/bench_config5_$i.json 2> $O/bench_config5.err
echo "bench config5 rc=$?" | tee -a $O/status.txt
python -c "
import json; d=json.loads(open('$O/bench_config5_$i.json').read().strip().splitlines()[-1]); print('config5: %.1f us/step  %.3g env-steps/s' % (d['roofline']['avg_step_us'], d['value']))"
done
timeout -k 10 300 python bench.py --n-bs 4 --n-ue 200 --envs 8192 --steps 300 --warmup 30 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/bench_4x200.json 2> $O/bench_4x200.err
python -c "
import json; d=json.loads(open('$O/bench_4x200.json').read().strip().splitlines()[-1]); print('4x200: %.1f us/step  %.3g env-steps/s' % (d['roofline']['avg_step_us'], d['value']))"
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $O/c5_pmc_sq -- python3 $R/bench.py --n-bs 16 --n-ue 200 --envs 8192 --steps 40 --warmup 10 --launch eager --no-cpu-baseline --no-a2c --no-alt > $O/c5_pmc_sq.log 2>&1
echo "c5 pmc sq rc=$?" | tee -a $O/status.txt
cd $R
python3 tools/pmc_digest.py $O 2>/dev/null | grep "multipass<16, 2" 
cat $O/status.txt
