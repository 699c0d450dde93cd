"""Opcode classes of one kernel of a gfx950 assembly listing (hipcc --cuda-device-only -S), by
loop depth, to see what the VALU instructions that are not fp64 arithmetic are.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -o zspace.s mpc_amd/csrc/zspace_solve.hip
    python scripts/isa_histogram.py zspace.s 'zspace_kernelILi4ELb1ELb0ELb0E'

Classes: fp64 arithmetic (fma/add/mul/min/max/trans), integer/logic VALU, lane traffic
(v_readlane / v_writelane / v_readfirstlane, split into SGPR spill slots -- VGPRs that are only
ever written by v_writelane with a constant lane -- and deliberate broadcasts), DPP moves,
v_mov, v_cndmask, compares, LDS, VMEM, SALU, SMEM, waits/branches.
The loop depth of a basic block is taken from the compiler's own loop comments
('Loop Header: Depth=d', 'in Loop: Header=... Depth=d').
"""
import collections
import re
import sys


def classify(op, args, spill_vgprs):
    if op.startswith(("v_fma_f64", "v_add_f64", "v_mul_f64", "v_min_f64", "v_max_f64", "v_rcp_f64", "v_rsq_f64",
                      "v_sqrt_f64", "v_div_", "v_fmac_f64", "v_ldexp_f64", "v_frexp", "v_trig", "v_fract_f64",
                      "v_rndne_f64", "v_floor_f64", "v_ceil_f64", "v_cvt_")):
        return "fp64 arithmetic (+cvt)"
    if op.startswith(("v_readlane", "v_writelane")):
        m = re.findall(r"v(\d+)", args)
        if m and int(m[0] if op.startswith("v_writelane") else m[-1]) in spill_vgprs:
            return "lane traffic: SGPR spill slots"
        return "lane traffic: broadcasts (readlane/writelane)"
    if op.startswith("v_readfirstlane"):
        return "lane traffic: broadcasts (readlane/writelane)"
    if "dpp" in args or op.endswith("_dpp"):
        return "DPP moves / DPP-fused ops"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"):
        return "v_mov"
    if op.startswith("v_cndmask"):
        return "v_cndmask (selects)"
    if op.startswith("v_cmp") or op.startswith("v_cmpx"):
        return "v_cmp (compares)"
    if op.startswith("v_"):
        return "integer / logic VALU"
    if op.startswith("ds_"):
        return "LDS"
    if op.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "VMEM"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"):
        return "SMEM"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep")):
        return "waits / nops"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_swappc")):
        return "branches"
    if op.startswith("s_"):
        return "SALU"
    return "other"


def main():
    path, pattern = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\S*%s\S*:" % re.escape(pattern), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end + 1]
    ins = []
    depth = 0
    for l in body:
        m = re.search(r"Loop Header: Depth=(\d+)", l) or re.search(r"in Loop: Header=\S+ Depth=(\d+)", l)
        s = l.strip()
        if re.match(r"^\.LBB\d+_\d+:", s) or s.startswith("; %bb."):
            depth = int(m.group(1)) if m else 0
        elif m and s.startswith(";"):
            depth = int(m.group(1))
        if not s or s.startswith((";", ".", "_Z")) or s.endswith(":"):
            continue
        s = s.split(";")[0].strip()
        if not s:
            continue
        parts = s.split(None, 1)
        ins.append((depth, parts[0], parts[1] if len(parts) > 1 else ""))
    # VGPRs used as SGPR spill slots: written by v_writelane from an SGPR at a constant lane,
    # and never written by anything else
    wl = collections.Counter()
    other_writes = set()
    for d, op, args in ins:
        regs = re.findall(r"\bv(\d+)\b", args)
        if op.startswith("v_writelane") and re.match(r"v\d+, s\d+, \d+", args):
            wl[int(regs[0])] += 1
        elif op.startswith("v_") and regs and not op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
            other_writes.add(int(regs[0]))
    spill = {v for v, n in wl.items() if n >= 4 and v not in other_writes}
    by = collections.defaultdict(collections.Counter)
    for d, op, args in ins:
        by[min(d, 3)][classify(op, args, spill)] += 1
    classes = sorted({c for cnt in by.values() for c in cnt})
    print("kernel %s: %d instructions; SGPR spill VGPRs: %s" % (pattern, len(ins), sorted(spill)))
    print("%-48s %8s %8s %8s %8s %8s" % ("class", "depth 0", "depth 1", "depth 2", "depth 3+", "all"))
    for c in classes:
        row = [by[d][c] for d in range(4)]
        print("%-48s %8d %8d %8d %8d %8d" % (c, *row, sum(row)))
    tot = [sum(by[d].values()) for d in range(4)]
    print("%-48s %8d %8d %8d %8d %8d" % ("total", *tot, sum(tot)))
    valu = [sum(n for c, n in by[d].items() if c not in ("LDS", "VMEM", "SMEM", "SALU", "branches", "waits / nops", "other"))
            for d in range(4)]
    print("%-48s %8d %8d %8d %8d %8d" % ("  of which VALU", *valu, sum(valu)))


if __name__ == "__main__":
    main()
