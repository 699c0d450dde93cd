#!/usr/bin/env python3
"""Checks the generated code of a kernel that issues LDS reads by hand (inline `ds_read_*`, as
mpc_amd/csrc/box_dense.hip does): between such a read and the `s_waitcnt lgkmcnt(N)` that
guarantees its arrival no instruction may read or write its destination registers -- the compiler
takes the registers for written as soon as the read is issued, so a copy or a spill it places there
would move stale data.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o box_dense.s box_dense.hip
    python scripts/lds_hazard_check.py box_dense.s

A forward data-flow analysis over the basic blocks of the listing.  The state is, per vector
register with a hand-issued read in flight, the least number of LDS operations issued after that
read on any path (LDS operations of one wave complete in order, so `lgkmcnt(N)` has retired a read
once N or more were issued after it; scalar memory operations, which share the counter and complete
out of order, are not counted -- that errs on the strict side).  Prints every violation and exits
with status 1 if there is one."""
import re
import sys

CAP = 16  # lgkmcnt holds 0..15: sixteen or more later operations are as good as sixteen


def vregs(text):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", text):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def parse(path):
    """Basic blocks of the listing: [(label, [(line number, text, issued by hand)], successors)]"""
    blocks, cur, label, in_asm = [], [], "<entry>", False
    for no, raw in enumerate(open(path), 1):
        line = raw.strip()
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        if not raw.startswith("\t") or line.startswith((";", ".")) or not line:
            continue
        cur.append((no, line, in_asm))
        if line.startswith(("s_branch", "s_cbranch", "s_endpgm", "s_setpc")):
            blocks.append((label, cur))
            label, cur = "<after %d>" % no, []
    blocks.append((label, cur))
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    out = []
    for i, (lab, ins) in enumerate(blocks):
        succ = []
        last = ins[-1][1] if ins else ""
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\d+_\d+)", last)
        if m and m.group(1) in index:
            succ.append(index[m.group(1)])
        if not last.startswith(("s_branch", "s_endpgm", "s_setpc")) and i + 1 < len(blocks):
            succ.append(i + 1)
        out.append((lab, ins, succ))
    return out


def transfer(state, ins, report):
    state = dict(state)
    hand_reads = 0
    for no, text, by_hand in ins:
        op = text.split()[0]
        if op.startswith("ds_"):
            operands = text[len(op):].split(",")
            reads = op.startswith("ds_read")
            dst = vregs(operands[0]) if reads else set()
            src = vregs(",".join(operands[1:] if reads else operands))
            for r in sorted((src | (set() if by_hand else dst)) & state.keys()):
                report.add((no, "v%d is still in flight: %s" % (r, text)))
            for r in state:
                state[r] = min(state[r] + 1, CAP)
            if by_hand and reads:
                hand_reads += 1
                for r in dst:
                    state[r] = 0
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", text)
            if m:
                n = int(m.group(1))
                state = {r: d for r, d in state.items() if d < n}
            continue
        if op == "s_barrier" or op.startswith(("s_", "buffer_", "global_", "flat_")) and not vregs(text):
            continue
        for r in sorted(vregs(text) & state.keys()):
            report.add((no, "v%d is still in flight: %s" % (r, text)))
    return state, hand_reads


def main(path):
    blocks = parse(path)
    entry = [None] * len(blocks)
    entry[0] = {}
    work, report, hand_reads = [0], set(), 0
    counted = set()
    while work:
        b = work.pop()
        out, n = transfer(entry[b], blocks[b][1], report)
        if b not in counted:
            counted.add(b)
            hand_reads += n
        for s in blocks[b][2]:
            if entry[s] is None:
                entry[s] = dict(out)
                work.append(s)
                continue
            merged = dict(entry[s])
            for r, d in out.items():
                merged[r] = min(merged.get(r, CAP), d)
            if merged != entry[s]:
                entry[s] = merged
                work.append(s)
    for no, what in sorted(report):
        print("%s:%d: %s" % (path, no, what))
    print("%d hand-issued LDS reads, %d violations" % (hand_reads, len(report)))
    return hand_reads, len(report)


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1])[1] else 0)
