#!/usr/bin/env python3
"""Static hazard audit of inline-asm MFMAs in a gfx950 .s file: the compiler does not see that an asm statement is an MFMA,
so it does not keep the 2 wait states a VALU write of an MFMA A/B (or C) operand needs before the MFMA reads it.
Reports every v_mfma whose VGPR source registers are written by a VALU instruction fewer than NEED wait states earlier
(each instruction = 1 wait state, s_nop N = N+1).  usage: audit_mfma_hazard.py file.s [--need 2]"""
import re
import sys

need = 2
files = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--need" in sys.argv:
    need = int(sys.argv[sys.argv.index("--need") + 1])


def regs(tok):
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    if m:
        return {int(m.group(1))}
    return set()


bad = 0
for f in files:
    ins, asm_lines, inasm = [], set(), False
    for ln, line in enumerate(open(f), 1):
        if "#ASMSTART" in line:
            inasm = True
        if "#ASMEND" in line:
            inasm = False
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"):
            continue
        ins.append((ln, t))
        if inasm:
            asm_lines.add(ln)      # only inline-asm MFMAs are audited: hipcc pads the hazards of the builtin ones itself
    for idx, (ln, t) in enumerate(ins):
        if not t.startswith("v_mfma") or ln not in asm_lines:
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        src = set()
        for o in ops[1:]:
            src |= regs(o)
        ws, k = 0, idx - 1
        while k >= 0 and ws < need:
            pl, pt = ins[k]
            op = pt.split()[0]
            if op.startswith("s_nop"):
                ws += int(pt.split()[1]) + 1
            else:
                if op.startswith("v_") and not op.startswith("v_mfma") and not op.startswith("v_cmp") and not op.startswith("v_accvgpr_write"):
                    dst = [o.strip() for o in pt.split(None, 1)[1].split(",")][0]
                    hit = regs(dst) & src
                    if hit:
                        print(f"{f}:{ln}: {t[:70]}\n     <- line {pl}: {pt[:70]}  ({ws} wait states between)")
                        bad += 1
                ws += 1
            k -= 1
    # (2) an MFMA result (VGPR destination) read or overwritten by a non-MFMA instruction before the MFMA's passes are
    #     over: count an MFMA issued in between as 8 wait states (it cannot issue before the pipe frees), anything else as 1;
    #     the XDL-write -> VALU hazard of a 16-pass MFMA is 19 wait states
    NEED_RES = 19
    for idx, (ln, t) in enumerate(ins):
        if not t.startswith("v_mfma") or ln not in asm_lines:
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(",")]
        dst = regs(ops[0])
        if not dst:
            continue
        ws, k = 0, idx + 1
        while k < len(ins) and ws < NEED_RES:
            pl, pt = ins[k]
            op = pt.split()[0]
            if op.startswith("s_nop"):
                ws += int(pt.split()[1]) + 1
            elif op.startswith("v_mfma"):
                ws += 8
            elif op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm") or op.startswith("s_barrier"):
                break
            else:
                if op.startswith("v_") or op.startswith("ds_") or op.startswith("buffer_") or op.startswith("global_"):
                    toks = [o.strip() for o in pt.split(None, 1)[1].split(",")] if len(pt.split(None, 1)) > 1 else []
                    used = set()
                    for o in toks:
                        used |= regs(o.split()[0] if o else o)
                    if used & dst:
                        print(f"{f}:{pl}: {pt[:70]}\n     touches the result of line {ln}: {t[:60]} after only {ws} wait states")
                        bad += 1
                        break
                ws += 1
            k += 1
print(f"MFMA hazards: {bad}")
sys.exit(1 if bad else 0)
