#!/bin/bash
# round 5: the tuned two-kernel D-infinity slope pass (rolling window, 48 VGPRs in the candidate kernel) vs one kernel; eager exchanges in the generic sweeps (strip tests of every sweep tool); eight-strip D8 trace again
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for V in two one; do
  unset TDX_DINF_SLOPE_ONE_PASS; if [ $V = one ]; then export TDX_DINF_SLOPE_ONE_PASS=1; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -o r -- $R/taudem_amd/bin/tdxbench dinf -n 32768 -steps 2 -warmup 1 -crc > $R/gpurun_out/r05i_tdxbench_dinf_32768_$V.json 2>/dev/null)
  echo "$V pass: $(tail -n 1 gpurun_out/r05i_tdxbench_dinf_32768_$V.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['dinfflowdir_ms'], d['areadinf_ms'], d.get('crc'))")"
  find gpurun_out/prof_x -name '*kernel_stats.csv' | head -n 1 | xargs grep -E "dinf_slope|dinf_cand" | awk -F'",' '{print substr($1,1,60), $2}'
  rm -rf gpurun_out/prof_x
done 2>&1 | tee gpurun_out/r05i_dinf_slope_two_vs_one.txt
unset TDX_DINF_SLOPE_ONE_PASS
timeout 1500 python -m pytest tests/test_gpu_multigpu.py tests/test_gpu_gridnet.py tests/test_gpu_cli.py tests/test_gpu_dinf.py tests/test_gpu_d8.py tests/test_gpu_large_golden.py -m gpu -q -x -k "strip or Strip or ranks or gpus or eight or three" > gpurun_out/r05i_pytest_strips.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r05i_pytest_strips.txt | tail -3; grep -B30 "short test summary" gpurun_out/r05i_pytest_strips.txt | head -50
