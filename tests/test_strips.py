"""Row strips across ranks (taudem_amd/distributed.py): protocol tests on CPU (gloo, world size 2 and 3)
and, on the GPU box, full PitRemove -> D8FlowDir -> AreaD8 runs with 2-4 ranks sharing the GPU, compared
bit-for-bit with the CPU oracle by tests/strip_worker.py."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "strip_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(nproc, extra, timeout=600):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), WORKER] + extra
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "1"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"{' '.join(cmd)}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-6000:]}"
    return r.stdout


def test_partition_rows_matches_linearpart():
    from taudem_amd.distributed import partition_rows

    assert partition_rows(10, 3) == [(0, 3), (3, 6), (6, 10)]      # remainder to the last rank (src/linearpart.h:133-134)
    assert partition_rows(8, 1) == [(0, 8)]
    assert partition_rows(65536, 8)[3] == (3 * 8192, 4 * 8192)


@pytest.mark.parametrize("nproc", [2, 3])
def test_stripcomm_protocol_gloo(nproc):
    _launch(nproc, ["--protocol"], timeout=300)


@pytest.mark.gpu
@pytest.mark.parametrize("nproc,case", [(2, "plain"), (3, "short_strips"), (2, "holes"), (4, "big"), (1, "plain"), (4, "wide"), (2, "plain+dinf"), (3, "holes+dinf"), (4, "short_strips+dinf")])
def test_strips_bit_exact_vs_oracle(nproc, case):
    out = _launch(nproc, ["--case", case], timeout=900)
    assert f"{nproc} ranks bit-exact vs oracle" in out
