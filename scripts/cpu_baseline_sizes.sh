# the reference's own tools on the GPU box's host cores at 1024^2, 2048^2 and 4096^2 COMPLETE (the default bench line times the 2048^2 sample only)
mkdir -p gpurun_out
for n in 1024 2048 4096; do
  timeout 1500 python -c "
import json, bench
r = bench.cpu_baseline($n, 1234); r.pop('offline_reference', None); print(json.dumps(r))" 2>&1 | tail -1
done > gpurun_out/${1:-r06zzzz}_cpu_baseline_1024_2048_4096.jsonl
cat gpurun_out/${1:-r06zzzz}_cpu_baseline_1024_2048_4096.jsonl | cut -c1-600
