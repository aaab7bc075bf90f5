#!/bin/bash
# round 5, first look: where do the eight strips' AreaD8 milliseconds go (segment trace, both modes) + the comm trace
mkdir -p gpurun_out
export TDX_COMM_TRACE=1
timeout 900 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 --segments 1 --segments-out gpurun_out/r05a_seg1_d8.json > gpurun_out/r05a_8strips_d8_seg1.json 2> gpurun_out/r05a_8strips_d8_seg1.err
echo "seg1 rc=$?"
timeout 900 python bench.py --gpus 8 --in-process --steps 1 --warmup 1 --segments 2 --segments-out gpurun_out/r05a_seg2_d8.json > gpurun_out/r05a_8strips_d8_seg2.json 2> gpurun_out/r05a_8strips_d8_seg2.err
echo "seg2 rc=$?"
python scripts/project_8gpu.py gpurun_out/r05a_seg2_d8.json > gpurun_out/r05a_projection_d8.txt
tail -5 gpurun_out/r05a_8strips_d8_seg1.err
cat gpurun_out/r05a_projection_d8.txt
