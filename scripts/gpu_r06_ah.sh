mkdir -p gpurun_out
( for n in 16384 32768; do for sw in 1 2 3 4 6 8 12 16; do
    echo -n "dinf $n TDX_DINF_BULK_SWEEPS=$sw  "
    TDX_DINF_BULK_SWEEPS=$sw taudem_amd/bin/tdxbench dinf -n $n -steps 2 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['ms_per_step','dinfflowdir_ms','areadinf_ms']})"
done; done
for sw in 2 4 6 12; do echo -n "decay strip TDX_DINF_BULK_SWEEPS=$sw  "; TDX_DINF_BULK_SWEEPS=$sw taudem_amd/bin/tdxbench decay -steps 1 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'])"; done ) > gpurun_out/r06ah_dinf_bulk_sweeps.txt 2>&1
cat gpurun_out/r06ah_dinf_bulk_sweeps.txt
