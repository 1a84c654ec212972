#!/usr/bin/env python3
"""Static instruction mix of one kernel from `hipcc -S --cuda-device-only` output.

  python tools/isa_stats.py out.s k_score [--blocks] [--dump LABEL]

Prints, for the first function whose mangled name contains the pattern: totals per
instruction class (VALU / SALU / LDS / VMEM-global / VMEM-flat / SMEM / waitcnt / nop /
branch), register and spill figures from the kernel descriptor comments, and with
--blocks the same per basic block (biggest first).  --dump LABEL prints one block.
The numbers are what the wavefront ISSUES per pass over a block: the static
counterpart of the SQ_INSTS_* counters under profiles/.
"""
from __future__ import annotations

import re
import sys
from collections import Counter, OrderedDict


def classify(op: str) -> str:
    if op.startswith("s_waitcnt"):
        return "wait"
    if op == "s_nop":
        return "nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "branch"
    if op == "s_barrier":
        return "barrier"
    if op.startswith(("s_load", "s_buffer_load", "s_store", "s_dcache")):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("flat_"):
        return "flat"
    if op.startswith(("global_", "buffer_", "scratch_")):
        return "vmem"
    if op.startswith("v_"):
        return "valu"
    return "other"


def main() -> int:
    if len(sys.argv) < 3:
        print(__doc__)
        return 2
    path, pat = sys.argv[1], sys.argv[2]
    want_blocks = "--blocks" in sys.argv
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    lines = open(path).read().splitlines()
    start = None
    for i, ln in enumerate(lines):
        m = re.match(r"^(_Z\w+):", ln)
        if m and pat in m.group(1):
            start = i
            name = m.group(1)
            break
    if start is None:
        print("no function matching", pat)
        return 1
    blocks: "OrderedDict[str, list[str]]" = OrderedDict()
    cur = "entry"
    blocks[cur] = []
    meta = {}
    end = start
    for i in range(start + 1, len(lines)):
        ln = lines[i]
        if ln.startswith(".Lfunc_end"):
            end = i
            break
        m = re.match(r"^(\.LBB\d+_\d+):", ln)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        s = ln.strip()
        if not s or s.startswith((";", ".", "//")):
            continue
        blocks[cur].append(s)
    for i in range(end, min(end + 200, len(lines))):
        m = re.match(r"^;\s*(\w[\w ]*?):\s*(\S+)", lines[i])
        if m:
            meta[m.group(1)] = m.group(2)
        m = re.match(r"^\s*\.(sgpr_spill_count|vgpr_spill_count|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s*(\d+)", lines[i])
        if m:
            meta[m.group(1)] = m.group(2)
    total = Counter()
    per = {}
    for lab, ins in blocks.items():
        c = Counter(classify(x.split()[0]) for x in ins)
        c["readlane"] = sum(1 for x in ins if x.startswith(("v_readlane", "v_readfirstlane")))
        per[lab] = c
        total.update(c)
    print(name)
    print(" ".join("%s=%s" % kv for kv in sorted(meta.items())
                   if kv[0] in ("NumVgprs", "NumSgprs", "NumAgprs", "ScratchSize", "Occupancy",
                                "sgpr_spill_count", "vgpr_spill_count", "LDSByteSize",
                                "TotalNumVgprs", "codeLenInByte")))
    keys = ["valu", "salu", "lds", "vmem", "flat", "smem", "wait", "nop", "branch", "barrier", "readlane"]
    print("TOTAL  " + "  ".join("%s=%d" % (k, total[k]) for k in keys))
    if want_blocks:
        order = sorted(per, key=lambda l: -sum(per[l][k] for k in keys if k != "readlane"))
        for lab in order[:40]:
            c = per[lab]
            n = sum(c[k] for k in keys if k != "readlane")
            print("%-12s n=%4d  " % (lab, n) + "  ".join("%s=%d" % (k, c[k]) for k in keys if c[k]))
    if dump:
        for x in blocks.get(dump, []):
            print("   ", x)
    return 0


if __name__ == "__main__":
    sys.exit(main())
