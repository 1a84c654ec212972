"""Generates tests/golden/codec_golden.npz with the REFERENCE's own compiled codec
(oracle/_ref, built from /root/reference/core/utils/bit_packing.cpp and
/root/reference/external/simdcomp/src/simdbitpacking.c).  Run in the build
container only:  python tests/golden/make_golden.py

Contents (all little-endian uint32):
  literal        the 128 values of tests/utils/bit_packing_tests.cpp:102-114
  random_b{bits} 128 random values < 2^bits (seeded)
  scalar_lit_b{bits}, simd4_lit_b{bits}   packed words of `literal & mask(bits)` /
                                          `literal` (simd4 packs without mask, so
                                          inputs are pre-masked) for bits 1..32
  scalar_rnd_b{bits}, simd4_rnd_b{bits}   packed words of random_b{bits}
The text of no reference source file is stored — only inputs and outputs.
"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
import oracle  # noqa: E402

LITERAL = [
    14410, 21766, 15994, 29493, 20819, 31650, 28103, 27900, 24340, 19822, 31073, 22825, 22494,
    6121, 20200, 28354, 25256, 220, 2, 393, 805, 18232, 956, 21480, 20565, 20500, 4324, 16372,
    1064, 9878, 13639, 18301, 31582, 18341, 30711, 25801, 16556, 23070, 7921, 20539, 23571,
    27043, 1344, 21307, 25545, 27844, 4796, 31011, 2238, 21070, 30090, 16475, 4683, 16839,
    14253, 17744, 5467, 19431, 17022, 24011, 21970, 24, 21009, 20131, 4494, 5110, 2991, 8127,
    21700, 20629, 31195, 20423, 24248, 15917, 31151, 8090, 24170, 11403, 30484, 11747, 23682,
    7179, 21741, 28313, 22102, 26759, 4385, 19645, 17771, 5977, 22946, 2705, 15188, 26374,
    25695, 25673, 4866, 8263, 1860, 24033, 12528, 14297, 11673, 20979, 446, 31576, 18451,
    13478, 11380, 6490, 2785, 3307, 1912, 382, 24139, 11483, 16582, 29873, 31287, 18465, 24084,
    29421, 18341, 21654, 3290, 19579]


def main():
    R = oracle.ref()
    assert R is not None, "oracle/_ref is not built (needs /root/reference)"
    lit = np.array(LITERAL, np.uint32)
    assert lit.size == 126
    lit = np.concatenate([lit, np.zeros(2, np.uint32)])  # items_required(126) == 128
    rng = np.random.default_rng(20260926)
    out = {"literal": lit}
    for bits in range(1, 33):
        mask = np.uint32(0xFFFFFFFF if bits == 32 else (1 << bits) - 1)
        rnd = rng.integers(0, 1 << bits, 128, dtype=np.uint64).astype(np.uint32)
        out["random_b%d" % bits] = rnd
        for name, vals in (("lit", lit & mask), ("rnd", rnd)):
            for lay, fn in (("scalar", R.ref_pack_scalar), ("simd4", R.ref_pack_simd4)):
                packed = np.zeros(4 * bits, np.uint32)
                fn(np.ascontiguousarray(vals).ctypes.data, bits, packed.ctypes.data)
                out["%s_%s_b%d" % (lay, name, bits)] = packed
    np.savez_compressed(Path(__file__).with_name("codec_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
