mkdir -p gpurun_out
for sw in 1 2 3 4 6 8 12 16; do
  echo "== TDX_D8_BULK_SWEEPS=$sw"
  TDX_D8_BULK_SWEEPS=$sw timeout 300 python scripts/bench_flowalg.py --only aread8_weighted,gridnet,d8flowpathextremeup,dinfconclimaccum,dinftranslimaccum_cs 2>&1 | tail -1 | cut -c1-300
done > gpurun_out/r06x_forward_bulk_sweeps.txt 2>&1
cat gpurun_out/r06x_forward_bulk_sweeps.txt
