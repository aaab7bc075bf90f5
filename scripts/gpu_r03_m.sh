#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
B=taudem_amd/bin/tdxbench
T=${1:-r03m}
for i in 1 2; do
timeout 90 $B d8 -n 16384 -steps 5 -crc > gpurun_out/${T}_d8_$i.json 2>> gpurun_out/${T}.err
python3 -c "
import json
d=json.load(open('gpurun_out/${T}_d8_$i.json'))
print({k:v for k,v in d.items() if not isinstance(v,dict)}, d['crc'], d['d8flowdir']['ms_class'])
"
done
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$T -o r -- $GRAFT_REPO_ROOT/taudem_amd/bin/tdxbench d8 -n 16384 -steps 5 > /dev/null 2>&1)
find gpurun_out/prof_$T -name "*kernel_stats.csv" | head -n 1 | xargs -I{} cp {} gpurun_out/${T}_kernel_stats_tdxbench_d8.csv; rm -rf gpurun_out/prof_$T
grep -E "classify_stream|slope_kernel|setflow2_stream|flat_stats|flat_count|flat_scan" gpurun_out/${T}_kernel_stats_tdxbench_d8.csv | cut -d, -f1-4 | cut -c1-60,100-260
timeout 600 python -m pytest tests/test_gpu_d8.py tests/test_gpu_large_golden.py tests/test_gpu_multigpu.py tests/test_gpu_cli.py -m gpu -q --no-header -p no:cacheprovider -x --timeout=300 --timeout-method=thread 2>&1 | tail -n 3
