#!/usr/bin/env python3
"""Static instruction mix of one gfx950 kernel, per loop of its assembly listing.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -disable-machine-licm --cuda-device-only -S -gline-tables-only \
        bio_ik_amd/csrc/bioik_hip.hip -o /tmp/k.s
    python tools/asm_loop_mix.py /tmp/k.s [_Z12k_solve_lean9SolveArgs] [rows]

Every basic block is charged to the innermost loop the assembler's comments place it in; a row is one loop: VALU instructions, then
the classes (fma / mul / add = FP64; imul = 32-bit integer multiplies, 2.4 issue slots each on gfx950; mov, cnd = v_cndmask, cmp, lane =
v_readlane / v_writelane, vint = every other VALU instruction; salu, smem, lds, wait = s_waitcnt, scratch) and the source lines
(file, line rounded to 10) most of the loop's instructions come from.  The counts are static: weigh them with the trip counts.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith(("v_fma", "v_fmac")):
        return "fma"
    if op.startswith("v_mul_f64"):
        return "mul"
    if op.startswith("v_add_f64"):
        return "add"
    if op.startswith(("v_mov", "v_accvgpr")):
        return "mov"
    if op.startswith("v_cndmask"):
        return "cnd"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith(("v_mul_lo", "v_mul_hi", "v_mad_u64")):
        return "imul"
    if op.startswith("v_"):
        return "vint"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("scratch_"):
        return "scratch"
    return "other"


VALU = ("fma", "mul", "add", "mov", "cnd", "cmp", "lane", "vint", "imul")
SHOWN = VALU + ("salu", "smem", "lds", "wait", "scratch")


def main():
    path = sys.argv[1]
    kernel = sys.argv[2] if len(sys.argv) > 2 else "_Z12k_solve_lean9SolveArgs"
    rows = int(sys.argv[3]) if len(sys.argv) > 3 else 25
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(kernel + ":"))
    end = next(i for i, l in enumerate(lines) if i > start and l.startswith(".Lfunc_end"))
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', l)
        if m:
            files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
    header, loc = ("top", 0), None
    count = collections.defaultdict(collections.Counter)
    source = collections.defaultdict(collections.Counter)
    for i in range(start, end):
        t = lines[i]
        m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", t)
        if m:
            comment = m.group(2) or ""
            j = i + 1
            while j < end and lines[j].lstrip().startswith(";"):
                comment += lines[j]
                j += 1
            own = re.search(r"This (?:Inner )?Loop Header: Depth=(\d+)", comment)
            inside = re.search(r"in Loop: Header=(BB\d+_\d+) Depth=(\d+)", comment)
            header = (m.group(1)[2:], int(own.group(1))) if own else ((inside.group(1), int(inside.group(2))) if inside else ("top", 0))
            continue
        t = t.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m:
            loc = (files.get(int(m.group(1))), int(m.group(2)) // 10 * 10)
            continue
        if not t or t[0] in ";." or t.split()[0].endswith(":"):
            continue
        count[header][classify(t.split()[0])] += 1
        if loc:
            source[header][loc] += 1
    total = collections.Counter()
    for c in count.values():
        total += c
    print("kernel", kernel, {k: total[k] for k in SHOWN if total[k]})
    ranked = sorted(count.items(), key=lambda kv: -sum(kv[1][x] for x in VALU))
    for h, c in ranked[:rows]:
        print(sum(c[x] for x in VALU), h, {k: c[k] for k in SHOWN if c[k]}, source[h].most_common(3))


if __name__ == "__main__":
    main()
