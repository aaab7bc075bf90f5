"""One-off scale check of the multi-GPU path: the 16384^2 pipeline (BASELINE.json configs[1]) through the CLI tools as N row strips
(`--gpus N`, default 2; ranks share the GPU on a 1-GPU box -> peer transport), every output raster against the restatement's digests
(tests/golden/xl_digests.json).  Files go to a scratch directory (about 6 GB).
usage: python scripts/check_strips_16384.py [N] [scratch_dir]"""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import taudem_amd as T
from test_gpu_large_golden import check_digest

n_ranks = int(sys.argv[1]) if len(sys.argv) > 1 else 2
case = json.load(open(os.path.join(ROOT, "tests", "golden", "xl_digests.json")))["16384"]
n, R = case["n"], case["rasters"]
d = sys.argv[2] if len(sys.argv) > 2 else tempfile.mkdtemp(prefix="tdx16k_")
os.makedirs(d, exist_ok=True)
f = lambda s: os.path.join(d, s)
ctx = T.Context(0)
dem = ctx.synth_dem(n, seed=case["seed"]).cpu().numpy()
check_digest("dem", dem, R["dem"])
T.write_raster(f("dem.tif"), dem, case["nodata"], geotransform=(0.0, case["dx"], 0.0, case["dy"] * n, 0.0, -case["dy"]), lzw=False)
del dem, ctx
os.environ["TAUDEM_AMD_COMPRESS"] = "NONE"
BIN = os.path.join(ROOT, "taudem_amd", "bin")
out = {}
def run(tool, *args):
    t0 = time.time()
    r = subprocess.run([os.path.join(BIN, tool), "--gpus", str(n_ranks), *args], capture_output=True, text=True, timeout=1800)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out[tool] = round(time.time() - t0, 1)
run("pitremove", "-z", f("dem.tif"), "-fel", f("fel.tif"))
run("d8flowdir", "-fel", f("fel.tif"), "-p", f("p.tif"), "-sd8", f("sd8.tif"))
run("aread8", "-p", f("p.tif"), "-ad8", f("ad8.tif"))
run("dinfflowdir", "-fel", f("fel.tif"), "-ang", f("ang.tif"), "-slp", f("slp.tif"))
run("areadinf", "-ang", f("ang.tif"), "-sca", f("sca.tif"))
for name, dt in (("fel", np.float32), ("p", np.int16), ("sd8", np.float32), ("ad8", np.float32), ("slp", np.float32), ("ang", np.float32), ("sca", np.float32)):
    a, _ = T.read_raster(f(name + ".tif"), dt)
    check_digest(f"{name} ({n_ranks} strips)", a, R[name])
    os.remove(f(name + ".tif"))
print(json.dumps({"check": f"16384^2 pipeline as {n_ranks} row strips through the CLI: all seven rasters have the restatement's SHA-256", "tool_wall_seconds_incl_file_io": out}))
