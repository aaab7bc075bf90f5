mkdir -p gpurun_out
for bu in 64 600 2000 6000 1000000; do
  echo "== TDX_D8_BULK_UNTIL=$bu"
  TDX_D8_BULK_UNTIL=$bu timeout 300 python scripts/bench_flowalg.py --only dinfrevaccum,dinfupdependence --digest 2>&1 | tail -1 | cut -c1-330
done > gpurun_out/r06s_reverse_bulk_until.txt 2>&1
cat gpurun_out/r06s_reverse_bulk_until.txt
