"""Static picture of a kernel's hot basic blocks from hipcc's assembly (-save-temps .s file).
usage: python tools/loop_stats.py file.s [substring of the kernel's mangled name ...] [--min-mfma 16]
For every basic block with at least --min-mfma MFMAs: instruction count, MFMAs, VALU, transcendental, SALU, s_waitcnt, s_nop,
LDS reads / writes, LDS-DMA / buffer ops, and v_readlane / v_writelane (SGPR spill traffic — must be 0 inside a tile loop)."""
import re, sys

def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_mfma = 16
    for i, a in enumerate(sys.argv):
        if a == "--min-mfma":
            min_mfma = int(sys.argv[i + 1])
            args = [x for x in args if x != sys.argv[i + 1]]
    path, pats = args[0], args[1:]
    kern, blocks, cur = None, {}, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1)
            cur = "entry"
            blocks[kern] = {cur: []}
            continue
        if kern is None:
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur = m.group(1)
            blocks[kern][cur] = []
            continue
        if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
            kern = None
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
            continue
        op = t.split()[0]
        if re.match(r"^[a-z_0-9]+$", op):
            blocks[kern][cur].append(op)
    for k, bl in blocks.items():
        if pats and not any(p in k for p in pats):
            continue
        rows = []
        for name, ops in bl.items():
            n = sum(1 for o in ops if o.startswith("v_mfma"))
            if n < min_mfma:
                continue
            c = lambda f: sum(1 for o in ops if f(o))
            trans = c(lambda o: o.split("_e32")[0].split("_e64")[0] in ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32"))
            valu = c(lambda o: o.startswith("v_") and not o.startswith("v_mfma") and not o.startswith("v_readlane") and not o.startswith("v_writelane") and not o.startswith("v_readfirstlane"))
            rows.append((name, len(ops), n, valu - trans, trans, c(lambda o: o.startswith("s_") and o not in ("s_waitcnt", "s_nop", "s_barrier")),
                         c(lambda o: o == "s_waitcnt"), c(lambda o: o == "s_nop"), c(lambda o: o.startswith("ds_read")), c(lambda o: o.startswith("ds_write")),
                         c(lambda o: o.startswith("buffer_") or o.startswith("global_")), c(lambda o: o.startswith("v_readlane") or o.startswith("v_writelane") or o.startswith("v_readfirstlane")),
                         c(lambda o: o.startswith("scratch_"))))
        print(f"{k[:110]}: {len(bl)} blocks, {sum(len(o) for o in bl.values())} instructions")
        for r in rows:
            print("   %-12s instr %4d  mfma %3d  valu %3d  trans %3d  salu %3d  waitcnt %2d  nop %2d  ds_read %3d  ds_write %2d  vmem %2d  lane-ops %2d  scratch %d   (%.2f non-MFMA per MFMA)" % (r + ((r[1] - r[2]) / r[2],)))

main()
