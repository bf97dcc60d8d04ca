import os
import socket
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")

# The two-rank data-parallel check (tests/ddp_two_ranks_worker.py) runs in two CHILD processes that share cuda:0.  A process
# that has initialised the GPU may not start programs on the GPU boxes' terms, so the children are started here, at session
# start, before any test (or any import a test module makes) has touched the device; tests/test_ddp_two_ranks_gpu.py waits for
# their reports.  Three processes on the card at once (the limit is six).
DDP2 = {"procs": [], "dir": None, "skipped": None}


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _wants_gpu_tests(config):
    m = config.getoption("-m") or ""
    if "not gpu" in m:
        return False
    args = [str(a) for a in config.args]
    if any(a.rsplit("::", 1)[0].endswith(".py") for a in args):  # single files: only when the two-rank test's file is among them
        return any("test_ddp_two_ranks_gpu" in a for a in args)
    return True


def pytest_sessionstart(session):
    config = session.config
    if os.environ.get("PYTEST_XDIST_WORKER") or not _wants_gpu_tests(config):
        DDP2["skipped"] = "session does not select the GPU tests"
        return
    if not os.path.exists("/dev/kfd") or not os.path.exists(os.path.join(ROOT, "s2t_amd", "lib", "libs2t_hip.so")):
        DDP2["skipped"] = "no GPU device node / no built library"
        return
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tempfile.mkdtemp(prefix="s2t_ddp2_")
    DDP2["dir"] = out
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    worker = os.path.join(ROOT, "tests", "ddp_two_ranks_worker.py")
    for rank in range(2):
        log = open(os.path.join(out, "rank%d.log" % rank), "w")
        DDP2["procs"].append(subprocess.Popen([sys.executable, worker, str(rank), "2", str(port), out], env=env, stdout=log,
                                              stderr=subprocess.STDOUT, cwd=ROOT))


def pytest_sessionfinish(session, exitstatus):
    for p in DDP2["procs"]:  # the exact children started above, if a run was cut short
        if p.poll() is None:
            p.kill()
            p.wait()


@pytest.fixture(scope="session")
def ddp_two_ranks():
    return DDP2


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
