#!/usr/bin/env python3
"""Static view of a kernel's hot loop from its gfx950 assembly (tools/asm_x4.sh writes build/asm/x4_nc.s):
for every basic block with >= MIN MFMAs print the instruction mix and the number of instructions issued between
consecutive MFMAs (the microarch guide's budget: <= 5 single-issue fillers per v_mfma_f32_32x32x16 gap).
usage: python tools/x4_loop_stats.py [file.s] [--min 32] [--dump N]"""
import argparse
import collections
import re

ap = argparse.ArgumentParser()
ap.add_argument("file", nargs="?", default="tiny-flash-attention_amd/build/asm/x4_nc.s")
ap.add_argument("--min", type=int, default=32)
ap.add_argument("--dump", type=int, default=-1, help="print the instructions of hot block number N")
a = ap.parse_args()

blocks, cur, name = [], [], "entry"
for line in open(a.file):
    t = line.split(";")[0].rstrip()
    if not t.strip():
        continue
    if re.match(r"^\.?[A-Za-z_0-9$.]+:", t.strip()) and not t.startswith("\t"):
        if cur:
            blocks.append((name, cur))
        name, cur = t.strip().rstrip(":"), []
        continue
    if t.startswith("\t.") or t.strip().startswith("."):
        continue
    cur.append(t.strip())
if cur:
    blocks.append((name, cur))


def kind(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_accvgpr"): return "accvgpr"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith("ds_"): return "lds"
    if op.startswith("buffer_") or op.startswith("global_"): return "vmem"
    if op.startswith("v_exp") or op.startswith("v_log") or op.startswith("v_rcp"): return "trans"
    if op.startswith("v_"): return "valu"
    if op.startswith("s_waitcnt"): return "waitcnt"
    if op.startswith("s_nop"): return "nop"
    if op.startswith("s_barrier"): return "barrier"
    if op.startswith("s_"): return "salu"
    return "other"


hot = 0
for name, ins in blocks:
    ops = [i.split()[0] for i in ins]
    n_mfma = sum(1 for o in ops if o.startswith("v_mfma"))
    if n_mfma < a.min:
        continue
    mix = collections.Counter(kind(o) for o in ops)
    gaps, g = [], 0
    seen = False
    for o in ops:
        if o.startswith("v_mfma"):
            if seen:
                gaps.append(g)
            seen, g = True, 0
        elif seen:
            g += 1
    valu_ops = collections.Counter(o for o in ops if kind(o) in ("valu", "trans"))
    print(f"[{hot}] block {name}: {len(ins)} instructions, {n_mfma} MFMAs; mix {dict(mix)}")
    print(f"     fillers per MFMA gap: mean {sum(gaps) / max(1, len(gaps)):.2f}, max {max(gaps) if gaps else 0}, histogram {dict(sorted(collections.Counter(gaps).items()))}")
    print(f"     VALU ops: {dict(valu_ops.most_common(14))}")
    if a.dump == hot:
        for i in ins:
            print("        " + i)
    hot += 1
tot = collections.Counter()
for name, ins in blocks:
    for i in ins:
        tot[kind(i.split()[0])] += 1
print("whole kernel:", dict(tot))
