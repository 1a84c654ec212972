"""Explicit in-tree builds (no JIT cache): every shared object lands next to its
sources so that it travels with the repo snapshot to the GPU box.

  libirs_synth.so  host-only synthetic segment builder   (g++)
  libirs_hip.so    the product: HIP kernels + C-ABI        (hipcc, gfx950)
"""
from __future__ import annotations

import os
import shutil
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent
REPO = ROOT.parent

SYNTH_SRC = [ROOT / "index" / "synth_index.cpp", ROOT / "index" / "synth_dict.cpp"]
SYNTH_HDR = [ROOT / "index" / "synth_index.h"]
SYNTH_LIB = ROOT / "index" / "libirs_synth.so"

HIP_DIR = ROOT / "csrc"
HIP_LIB = HIP_DIR / "libirs_hip.so"


def _stale(target: Path, deps) -> bool:
    if not target.exists():
        return True
    t = target.stat().st_mtime
    return any(Path(d).stat().st_mtime > t for d in deps)


def _run(cmd):
    proc = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError(
            "build failed: %s\n%s\n%s" % (" ".join(map(str, cmd)), proc.stdout, proc.stderr)
        )
    return proc


def build_synth(force: bool = False) -> Path:
    if force or _stale(SYNTH_LIB, SYNTH_SRC + SYNTH_HDR):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall",
              "-o", SYNTH_LIB, *SYNTH_SRC])
    return SYNTH_LIB


def hip_sources():
    return sorted(HIP_DIR.glob("*.hip"))


def hip_deps():
    return hip_sources() + sorted(HIP_DIR.glob("*.h")) + sorted((HIP_DIR / "hip").glob("*.h")) + sorted((REPO / "include").glob("*.h"))


def hipcc_path() -> str | None:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    return None


def build_hip(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950 (cross-compiles without a GPU)."""
    if not (force or _stale(HIP_LIB, hip_deps())):
        return HIP_LIB
    hipcc = hipcc_path()
    if hipcc is None:
        if HIP_LIB.exists():
            return HIP_LIB
        raise RuntimeError("hipcc not found and %s is not prebuilt" % HIP_LIB)
    # -ffp-contract=off: BM25 is evaluated with the reference's operation order
    # and no FMA fusion, so per-term scores are bit-identical to the CPU path.
    _run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-pthread",
          "-ffp-contract=off", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
          "-I", REPO / "include", "-I", HIP_DIR, "-I", HIP_DIR / "hip", "-o", HIP_LIB, *hip_sources()])
    return HIP_LIB


def build_all(force: bool = False):
    build_synth(force)
    build_hip(force)
