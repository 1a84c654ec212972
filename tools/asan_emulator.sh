#!/bin/sh
# Dev utility (CPU only): the product sources compiled against the fiber emulator WITH
# AddressSanitizer, then a spread of the shared test bodies run through the C ABI.  The
# emulator's "device" buffers are plain malloc blocks, so a kernel reading or writing past a
# staged buffer (beyond the documented zero padding) is reported.
set -e
cd "$(dirname "$0")/.."
OUT=${TMPDIR:-/tmp}/libirs_hip_asan.so
g++ -O1 -g -std=c++17 -fPIC -shared -pthread -ffp-contract=off -fsanitize=address \
  -fno-omit-frame-pointer -I include -I iresearch_amd/csrc -I tests/sim -I iresearch_amd/csrc/hip \
  -Wno-unknown-pragmas -o "$OUT" -x c++ iresearch_amd/csrc/irs_hip.hip -x assembler tests/sim/sim_switch.S
LD_PRELOAD=$(gcc -print-file-name=libasan.so) \
ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python - "$OUT" <<'PY'
import ctypes as C
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import cases
from iresearch_amd import _lib
L = _lib.bind(C.CDLL(sys.argv[1]))
for layout in (1, 0):
    cases.case_decode_positions(L, layout)
    cases.case_decode_sizes(L, layout)
    cases.case_header_chain(L, layout, 40_000)
    cases.case_bit_union(L, layout)
    cases.case_queries_ragged(L, layout)
    cases.case_phrase_reference_vectors(L, layout)
cases.case_phrase_ragged(L)
cases.case_phrase_fuzz(L, 6, 3)
cases.case_queries_all_scorers(L, 20_000, 128, 1, ks=(10,))
cases.case_multi_segment_batch(L, sizes=(9_000, 3_000, 20_000), max_rank=128, k=50)
cases.case_phrase_multi_segment(L, sizes=(6000, 2000, 9000))
cases.case_wand_data(L, 1)
cases.case_no_norms(L)
cases.case_wide_norms(L, 2)
cases.case_pilot_misled(L)
cases.case_merge_ties(L)
cases.case_boolean_reference_vectors(L)
cases.case_wand_equals_exhaustive(L, num_docs=30_000, max_rank=128, ks=(10,))
cases.case_decode_reference_packed(L, 1)
cases.case_errors(L)
cases.case_phrase_errors(L)
cases.case_merge_types(L)
cases.case_min_score_pushdown(L)
cases.case_many_items(L, 20_000)
cases.case_zero_boost(L)
cases.case_legacy_norms(L)
cases.case_paths_agree(L, num_docs=40_000, max_rank=128)
cases.case_join_edge_blocks(L, 1)
cases.case_join_counts(L, num_docs=40_000, max_rank=128)
cases.case_accumulator_switch(L)
cases.case_shared_threshold(L, sizes=(30_000, 13_000, 40_000), max_rank=128)
cases.case_shared_threshold_misled(L)
cases.case_doc_mask(L, 1, num_docs=30_000, max_rank=128)
cases.case_conj_sparse_lead(L, 1, n_docs=120_000)
print("asan emulator run: clean")
PY
# ... and the C++ host readers (header only: instrumented with the test binary) over a segment of
# three fields incl. damaged `.ti` / `.sm` / `.tm` files whose checksums were recomputed
python - <<'PY'
import subprocess, sys
sys.path.insert(0, ".")
import oracle
from iresearch_amd import _build
synth, orc = _build.build_synth(), oracle.build()
sim = "tests/sim/libirs_hip_sim.so"
exe = "/tmp/test_segment_asan"
subprocess.run(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address", "-fno-omit-frame-pointer",
                "-ffp-contract=off", "-I", "include", "-I", "iresearch_amd/cpp", "-I", "iresearch_amd/index",
                "-I", "oracle", "tests/cpp/test_segment.cpp", "-o", exe, sim, str(synth), str(orc), "-pthread",
                "-Wl,-rpath," + str(__import__("pathlib").Path(sim).resolve().parent),
                "-Wl,-rpath," + str(__import__("pathlib").Path(str(synth)).resolve().parent),
                "-Wl,-rpath," + str(__import__("pathlib").Path(str(orc)).resolve().parent)], check=True)
out = subprocess.run([exe, "20000", "600"], capture_output=True, text=True,
                     env={"ASAN_OPTIONS": "detect_leaks=0", "PATH": "/usr/bin:/bin"})
assert out.returncode == 0, out.stderr[-3000:]
print(out.stdout.strip().splitlines()[-2])
print("asan host readers: clean")
PY
