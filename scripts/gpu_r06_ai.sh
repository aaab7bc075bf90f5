mkdir -p gpurun_out
( for n in 16384 32768; do for bu in 100 400 1600 6400; do
    echo -n "dinf $n TDX_DINF_BULK_UNTIL=$bu  "
    TDX_DINF_BULK_UNTIL=$bu taudem_amd/bin/tdxbench dinf -n $n -steps 2 -crc 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['ms_per_step','dinfflowdir_ms','areadinf_ms']}, d['crc']['sca'])"
done; done
for t in 16 32; do echo -n "dinf 16384 TDX_DINF_BULK_TILE=$t  "; TDX_DINF_BULK_TILE=$t taudem_amd/bin/tdxbench dinf -n 16384 -steps 2 2>/dev/null | tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k:d[k] for k in ['ms_per_step','dinfflowdir_ms','areadinf_ms']})"; done ) > gpurun_out/r06ai_dinf_bulk_until.txt 2>&1
cat gpurun_out/r06ai_dinf_bulk_until.txt
