# reverse tools: lockstep sweeps to the end (default) vs a bounded number followed by walks with the one-slot-per-cell queue
mkdir -p gpurun_out
for sw in "" 0 2 6 16 48; do
  echo "== TDX_D8_BULK_SWEEPS=${sw:-default}"
  if [ -z "$sw" ]; then timeout 300 python scripts/bench_flowalg.py --only dinfrevaccum,dinfupdependence --digest 2>&1 | tail -1
  else TDX_D8_BULK_SWEEPS=$sw timeout 300 python scripts/bench_flowalg.py --only dinfrevaccum,dinfupdependence --digest 2>&1 | tail -1; fi
done > gpurun_out/r06p_reverse_sweeps.txt 2>&1
cat gpurun_out/r06p_reverse_sweeps.txt
