"""The C++ host layer above the C ABI (iresearch_amd/cpp/irs_hip.hpp): tests/cpp/test_host.cpp
is compiled with g++ and run — against the CPU emulator build of the product sources here,
against libirs_hip.so on a GPU box.  The program itself checks every result against the
oracle's C API."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _build_and_run(tmp_path, abi_lib: Path, extra=(), source="test_host", args=()):
    import oracle
    from iresearch_amd import _build
    synth = _build.build_synth()
    orc = oracle.build()
    exe = tmp_path / source
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall",
           "-I", str(ROOT / "include"), "-I", str(ROOT / "iresearch_amd" / "cpp"),
           "-I", str(ROOT / "iresearch_amd" / "index"), "-I", str(ROOT / "oracle"),
           str(ROOT / "tests" / "cpp" / (source + ".cpp")), "-o", str(exe),
           str(abi_lib), str(synth), str(orc), "-pthread",
           "-Wl,-rpath," + str(abi_lib.parent), "-Wl,-rpath," + str(Path(synth).parent),
           "-Wl,-rpath," + str(Path(orc).parent), *extra]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([str(exe), *args], capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
    assert source + " OK" in run.stdout


def test_cpp_sharded_search_two_processes(simlib, tmp_path):
    """search_sharded of the C++ host layer as TWO processes (one per rank) on the emulator
    build: segments sharded, the all-gather through irs_hip_comm_* / irs_hip_topk_allgather
    (a shared-memory stand-in for RCCL here), merge on every rank — against the one-process
    search() over all segments."""
    from iresearch_amd import _build
    abi_lib, synth = Path(simlib._name), _build.build_synth()
    exe = tmp_path / "test_sharded"
    cmd = ["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-Wall",
           "-I", str(ROOT / "include"), "-I", str(ROOT / "iresearch_amd" / "cpp"),
           "-I", str(ROOT / "iresearch_amd" / "index"),
           str(ROOT / "tests" / "cpp" / "test_sharded.cpp"), "-o", str(exe),
           str(abi_lib), str(synth), "-pthread",
           "-Wl,-rpath," + str(abi_lib.parent), "-Wl,-rpath," + str(Path(synth).parent)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    id_file = tmp_path / "comm_id"
    procs = [subprocess.Popen([str(exe), "2", str(r), str(id_file)], stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in (0, 1)]
    for r, p in enumerate(procs):
        text, _ = p.communicate(timeout=600)
        assert p.returncode == 0, text[-3000:]
        assert "test_sharded OK: rank %d" % r in text


def test_cpp_host_layer_on_the_emulator(simlib, tmp_path):
    _build_and_run(tmp_path, Path(simlib._name))


@pytest.mark.parametrize("positions", [0, 1])
def test_cpp_segment_from_file_bytes_on_the_emulator(simlib, tmp_path, positions):
    """a segment opened from `.doc` (+ `.pos`), the term dictionary and
    the columnstore files only (tests/cpp/test_files.cpp), then BASELINE config 2 on it."""
    _build_and_run(tmp_path, Path(simlib._name), source="test_files",
                   args=["60000", str(positions)])


@pytest.mark.gpu
def test_cpp_segment_from_file_bytes_on_the_gpu(gpulib, tmp_path):
    """... at config 2's own size: OR-of-2 BM25 top-100 on a 1 M-doc segment read from files."""
    from iresearch_amd import _build
    rocm = "/opt/rocm/lib"
    _build_and_run(tmp_path, Path(_build.HIP_LIB),
                   ["-Wl,-rpath," + rocm, "-Wl,-rpath-link," + rocm, "-Wl,--allow-shlib-undefined"],
                   source="test_files", args=["1000000", "0"])


def test_cpp_segment_of_several_fields_on_the_emulator(simlib, tmp_path):
    """SURVEY.md §8 f3, the host half: three fields of ONE segment (FREQ|POS + Norm2, no FREQ,
    FREQ without norms) opened from `.doc` / `.pos` / `.tm` / `.ti` / `.sm` / columnstore bytes
    and a field name — features, norm column, counts and block roots all come from the term
    index and the segment meta (tests/cpp/test_segment.cpp)."""
    _build_and_run(tmp_path, Path(simlib._name), source="test_segment", args=["40000"])


@pytest.mark.gpu
def test_cpp_segment_of_several_fields_on_the_gpu(gpulib, tmp_path):
    from iresearch_amd import _build
    rocm = "/opt/rocm/lib"
    _build_and_run(tmp_path, Path(_build.HIP_LIB),
                   ["-Wl,-rpath," + rocm, "-Wl,-rpath-link," + rocm, "-Wl,--allow-shlib-undefined"],
                   source="test_segment", args=["1000000"])


@pytest.mark.gpu
def test_cpp_host_layer_on_the_gpu(gpulib, tmp_path):
    from iresearch_amd import _build
    rocm = "/opt/rocm/lib"
    _build_and_run(tmp_path, Path(_build.HIP_LIB),
                   ["-Wl,-rpath," + rocm, "-Wl,-rpath-link," + rocm, "-Wl,--allow-shlib-undefined"])
