#!/bin/bash
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -q --no-header -p no:cacheprovider -x --deselect tests/test_gpu_cli.py::test_bigtiff_above_4gb_round_trip 2>&1 | grep -E "passed|failed|error|Error|assert|differ" | tail -8
timeout 300 python bench.py --cpu-sample 0 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step'], d['kernel_class_ms_per_step'])"
