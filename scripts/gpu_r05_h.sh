#!/bin/bash
# round 5: (1) the D-infinity slope pass as two kernels (fp32 candidate mask + fp64 on the candidates) - digests and times; (2) at most 8 sweep rounds between two exchanges - strips tests, eight-strip decay projection
mkdir -p gpurun_out
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_dinf.py tests/test_gpu_large_golden.py tests/test_gpu_multigpu.py tests/test_gpu_pathological.py tests/test_gpu_fullsize.py -m gpu -q -x -k "dinf or Dinf or digest or strips or decay or pathological" > gpurun_out/r05h_pytest.txt 2>&1
echo "pytest rc=$?"; grep -E "passed|failed|error" gpurun_out/r05h_pytest.txt | tail -3; grep -B30 "short test summary" gpurun_out/r05h_pytest.txt | head -60
for V in two one; do
  unset TDX_DINF_SLOPE_ONE_PASS; if [ $V = one ]; then export TDX_DINF_SLOPE_ONE_PASS=1; fi
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_x -o r -- $R/taudem_amd/bin/tdxbench dinf -n 32768 -steps 2 -warmup 1 -crc > $R/gpurun_out/r05h_tdxbench_dinf_32768_$V.json 2>/dev/null)
  echo "$V pass: $(tail -n 1 gpurun_out/r05h_tdxbench_dinf_32768_$V.json | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['dinfflowdir_ms'], d['areadinf_ms'], d.get('crc'))")"
  find gpurun_out/prof_x -name '*kernel_stats.csv' | head -n 1 | xargs grep -E "dinf_slope|dinf_cand" | awk -F'",' '{print substr($1,1,60), $2}'
  rm -rf gpurun_out/prof_x
done
unset TDX_DINF_SLOPE_ONE_PASS
export TDX_COMM_TRACE=1
for E in 8 0; do
  TDX_SWEEP_EAGER_ROUNDS=$E timeout 900 python bench.py --gpus 8 --in-process --workload decay --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/r05h_seg2_decay_eager$E.json > gpurun_out/r05h_8strips_decay_eager$E.json 2> gpurun_out/r05h_8strips_decay_eager$E.err
  echo "decay eager=$E rc=$?"; python scripts/project_8gpu.py gpurun_out/r05h_seg2_decay_eager$E.json | tee gpurun_out/r05h_projection_decay_eager$E.txt | tail -4
  cut -c1-600 gpurun_out/r05h_8strips_decay_eager$E.json
done
