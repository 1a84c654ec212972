import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) GPU")
    # Test segments are small: by its size rule the library would keep their joined disjunctions on
    # 32-bit tiles and the paired-tile kernels (k_join_score<kJKHalf>, k_join_rescore) would only
    # run in case_paired_tiles.  Pair whatever the size, so that every plain-disjunction case
    # checks them against the oracle; the 32-bit tiles keep their coverage through
    # case_paired_tiles (unpaired run), every conjunction / min-match / deleted-docs case and
    # IRS_HIP_JOIN_HALF=0 runs of this suite.
    os.environ.setdefault("IRS_HIP_JOIN_HALF", "1")


def _sim_stale(so: Path) -> bool:
    if not so.exists():
        return True
    deps = list((ROOT / "iresearch_amd" / "csrc").rglob("*.h")) + \
        list((ROOT / "iresearch_amd" / "csrc").glob("*.hip")) + \
        list((ROOT / "tests" / "sim").glob("*.h")) + [ROOT / "include" / "irs_hip.h",
                                                        ROOT / "tests" / "sim" / "sim_switch.S"]
    return any(d.stat().st_mtime > so.stat().st_mtime for d in deps)


@pytest.fixture(scope="session")
def simlib():
    """The product sources compiled against the CPU fiber emulator (tests/sim):
    exercises the real host code and kernel logic on the GPU-less build box."""
    from iresearch_amd import _lib
    so = ROOT / "tests" / "sim" / "libirs_hip_sim.so"
    # (pytest-xdist: every worker has its own session — one of them builds, the others wait)
    import fcntl
    with open(ROOT / "tests" / "sim" / ".build.lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if _sim_stale(so):
            subprocess.run([str(ROOT / "tests" / "sim" / "build_sim.sh")], check=True)
    return _lib.bind(C.CDLL(str(so)))


@pytest.fixture(scope="session")
def gpulib():
    """The product library on a real GPU; fails (does not skip) without one."""
    from iresearch_amd import _lib
    L = _lib.lib()
    buf = C.create_string_buffer(64)
    rc = L.irs_hip_device_arch(0, buf, 64)
    assert rc == 0, "no usable HIP device: %s" % L.irs_hip_strerror(rc).decode()
    assert buf.value.decode().startswith("gfx950"), buf.value
    return L
