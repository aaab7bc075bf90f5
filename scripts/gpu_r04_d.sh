#!/bin/bash
# Round 4, call D: what bounds the D-infinity bulk rounds?  SQ counters of the sweep kernels (two --pmc passes, no trace domains) and the
# in-kernel phase clocks of the current engine.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04d
mkdir -p $O
B=$R/taudem_amd/bin/tdxbench
cd $R
TDX_DEBUG_ROUNDS=1 timeout 300 $B dinf -n 16384 -steps 1 -warmup 0 > $O/dinf_phases.json 2> $O/dinf_phases.txt; grep -A1 "dinf sweep rounds\|\[rounds" $O/dinf_phases.txt | cut -c1-260 | head -n 60
cd /tmp
i=0
for C in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum"; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $C --output-format csv -d $O/pmc_$i -o p -- $B dinf -n 16384 -steps 1 -warmup 0 > $O/pmc_$i.log 2>&1
  python $R/scripts/pmc_summary.py $O/pmc_$i $O/pmc_dinf_${i}_summary.json | grep -A9 "dsweep32\|dsweep64" | head -n 40
  tail -n 3 $O/pmc_$i.log | cut -c1-300
  rm -rf $O/pmc_$i
done
