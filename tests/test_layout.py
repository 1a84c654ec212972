"""Repo contracts: the C-ABI library exports what include/irs_hip.h declares; the
product never touches the oracle or the emulator."""
import ctypes as C
import re
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "irs_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(irs_hip_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_exports_every_declared_symbol():
    """No compute calls here (no GPU on the CPU tier): load + symbol table only."""
    from iresearch_amd import _build, _lib
    so = _build.build_hip()
    L = C.CDLL(str(so))
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), n
    assert sorted(_lib.SYMBOLS) == names
    assert _lib.bind(L).irs_hip_abi_version() == 12
    assert _lib.bind(L).irs_hip_strerror(-2) == b"corrupt postings data"
    out = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True)
    exported = set(re.findall(r" T (irs_hip_[a-z0-9_]+)", out.stdout))
    assert set(names) <= exported


def test_device_code_is_gfx950_only():
    from iresearch_amd import _build
    so = _build.build_hip()
    data = so.read_bytes()
    assert b"gfx950" in data
    for other in (b"gfx90a", b"gfx942", b"gfx1100", b"sm_"):
        assert other not in data, other


def test_without_a_gpu_the_product_fails_loudly():
    """segment_open must return EHIP (not a CPU result) when no gfx950 is usable."""
    import numpy as np
    import pytest
    import torch

    from iresearch_amd import _lib, search, synth
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    seg = synth.build_segment(2000, 16)
    with pytest.raises(_lib.IrsHipError) as e:
        search.SegmentReader.from_synth(seg)
    assert e.value.status == _lib.EHIP


def test_product_never_imports_oracle_or_emulator():
    bad = []
    for p in list((ROOT / "iresearch_amd").rglob("*")) + list((ROOT / "include").rglob("*")):
        if p.suffix not in (".py", ".h", ".hip", ".cpp", ".c"):
            continue
        t = p.read_text()
        if re.search(r"^\s*(import|from)\s+oracle\b", t, flags=re.M) or "liboracle" in t:
            bad.append(str(p))
        if re.search(r'#include\s+[<"](oracle|hip_sim)', t) or "libirs_hip_sim" in t:
            bad.append(str(p))
    assert not bad, bad
    # only the allowed places load the oracle
    users = []
    for p in ROOT.glob("*.py"):
        if re.search(r"^\s*import oracle\b|^\s*from oracle\b", p.read_text(), flags=re.M):
            users.append(p.name)
    assert set(users) <= {"bench.py", "__graft_entry__.py"}, users
