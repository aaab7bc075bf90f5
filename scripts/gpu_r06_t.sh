mkdir -p gpurun_out
for sw in 2 4 8 16 32; do
  echo "== inner sweeps per vote=$sw"
  TDX_D8_BULK_SWEEPS=$sw timeout 300 python scripts/bench_flowalg.py --only dinfrevaccum,dinfupdependence --digest 2>&1 | tail -1 | cut -c1-330
done > gpurun_out/r06t_reverse_inner.txt 2>&1
cat gpurun_out/r06t_reverse_inner.txt
