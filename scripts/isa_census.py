"""Instruction census of one kernel in a hipcc -S listing: counts per instruction class, for the whole kernel and per
basic block (label), so that the hot loops can be compared instruction by instruction between variants.
usage: isa_census.py file.s <substring of the mangled kernel name> [--blocks]"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"): return "mfma"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_accvgpr"): return "acc_mov"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    blocks = "--blocks" in sys.argv
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*:", l) and key in l)
    tot, per, cur = collections.Counter(), collections.OrderedDict(), "entry"
    ops = collections.Counter()
    for l in lines[start + 1:]:
        if l.startswith("\t.end_amdhsa_kernel") or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"^(\.LBB\S+):", l)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"^\t([a-z_0-9]+)", l)
        if not m or l.startswith("\t."):
            continue
        op = m.group(1)
        c = classify(op)
        tot[c] += 1
        ops[op] += 1
        per.setdefault(cur, collections.Counter())[c] += 1
    print("total", dict(tot))
    print("top ops", ops.most_common(25))
    if blocks:
        for k, v in per.items():
            if sum(v.values()) >= 12:
                print(k, dict(v))


main()
