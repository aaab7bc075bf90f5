"""Config-scale digests from the C RESTATEMENT (oracle/taudem_oracle.c, active-list form of the flat loops) at sizes the
real reference cannot reach in reasonable time (its flat resolution is sweeps x flats: days at 16384^2):

    python tests/golden/make_golden_oracle_xl.py 8192 16384

The restatement is pinned to the real reference byte-for-byte by tests/golden/large_digests.json (2048^2, 4096^2, made by the
reference tools themselves) and by the small committed rasters.  Output: tests/golden/xl_digests.json, same layout as
large_digests.json ("source": "restatement").
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden_large import digest  # noqa: E402
from oracle import oracle as O  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "xl_digests.json")


def make(n, seed=1234, dinf=True):
    res = {"n": n, "seed": seed, "source": "restatement", "dx": 30.0, "dy": 30.0, "nodata": -9999.0, "rasters": {}, "oracle_seconds": {}}
    t = time.time()

    def lap(name):
        nonlocal t
        res["oracle_seconds"][name] = time.time() - t
        print(n, name, f"{time.time() - t:.1f} s", flush=True)
        t = time.time()

    dem = O.synth_dem(n, seed)
    res["rasters"]["dem"] = digest(dem)
    fel = O.pitremove(dem, -9999.0)
    del dem
    res["rasters"]["fel"] = digest(fel); lap("pitremove")
    p, sd8, st = O.d8flowdir(fel, -3.0e38, 30.0, 30.0)
    res["d8_stats"] = {k: int(v) for k, v in st.items()}
    res["rasters"]["p"] = digest(p); res["rasters"]["sd8"] = digest(sd8); lap("d8flowdir")
    del sd8
    ad8 = O.aread8(p, -32768)
    res["rasters"]["ad8"] = digest(ad8); lap("aread8")
    del ad8, p
    if dinf:
        ang, slp, st = O.dinfflowdir(fel, -3.0e38, 30.0, 30.0)
        res["rasters"]["ang"] = digest(ang); res["rasters"]["slp"] = digest(slp); lap("dinfflowdir")
        del slp
        sca = O.areadinf(ang, dx=30.0, dy=30.0)
        res["rasters"]["sca"] = digest(sca); lap("areadinf")
    return res


if __name__ == "__main__":
    O.build()
    for n in [int(s) for s in sys.argv[1:]] or [8192]:
        allres = json.load(open(OUT)) if os.path.exists(OUT) else {}
        allres[str(n)] = make(n)
        json.dump(allres, open(OUT, "w"), indent=1)
        print("written", n, flush=True)
