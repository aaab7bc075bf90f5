#!/bin/bash
# Round 4, call A: canary, the new configuration-scale tests (configs[4] strip under the host checker, eight strips of 65536 columns),
# row-pitch A/B of the pipeline and of the tile-load micro-benchmark (same code, 16384 vs 16448 / 16320 columns), FETCH_SIZE calibration
# on the micro-benchmark's known byte count.  Every command under `timeout`.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04a
mkdir -p $O
B=$R/taudem_amd/bin/tdxbench
cd $R
timeout 120 $B d8 -n 4096 -steps 1 -crc > $O/canary_d8_4096.json 2> $O/canary.err; cut -c1-200 $O/canary_d8_4096.json; grep -o '"crc".*' $O/canary_d8_4096.json
timeout 1500 python -m pytest tests/test_gpu_fullsize.py -q --no-header -p no:cacheprovider --timeout=900 --timeout-method=thread --durations=6 -k "decay_config5 or eight_strips" > $O/pytest_new.txt 2>&1; tail -n 14 $O/pytest_new.txt
for a in "-n 16384" "-nx 16448 -ny 16320" "-nx 16320 -ny 16448" "-n 16384" "-nx 16448 -ny 16320"; do timeout 120 $B d8 $a -steps 3 >> $O/pitch_d8.jsonl 2>> $O/pitch.err; done
for a in "-n 16384" "-nx 16448 -ny 16320" "-n 16384"; do timeout 200 $B dinf $a -steps 1 >> $O/pitch_dinf.jsonl 2>> $O/pitch.err; done
python - <<'PY'
import json
for f in ("pitch_d8.jsonl", "pitch_dinf.jsonl"):
    for l in open("gpurun_out/r04a/" + f):
        try:
            d = json.loads(l)
        except Exception:
            print("unparsed:", l[:200]); continue
        cells = d["nx"] * d["ny"]
        out = {"nx": d["nx"], "ny": d["ny"], "ns_per_kcell": round(d["ms_per_step"] * 1e9 / cells, 2)}
        for k in ("pitremove", "d8flowdir", "aread8", "dinfflowdir", "areadinf"):
            if k in d:
                out[k] = [round(x * 1e9 / cells, 2) for x in d[k]["ms_class"][:6]]   # ns per 1000 cells per kernel class: stencil relax bfs flatdir accum misc
        print(out)
PY
for n in 16384 16448 16320; do timeout 120 scripts/micro/tilebw $n > $O/tilebw_$n.txt 2>&1; head -n 7 $O/tilebw_$n.txt; done
cd /tmp
for pass in fetch write; do
  case $pass in fetch) C="FETCH_SIZE";; write) C="WRITE_SIZE";; esac
  timeout 300 rocprofv3 --pmc $C --output-format csv -d $O/pmc_tilebw_$pass -o p -- $R/scripts/micro/tilebw 16384 > $O/pmc_tilebw_$pass.log 2>&1
  python $R/scripts/pmc_summary.py $O/pmc_tilebw_$pass $O/pmc_tilebw_${pass}_summary.json | head -n 30
  rm -rf $O/pmc_tilebw_$pass
done
