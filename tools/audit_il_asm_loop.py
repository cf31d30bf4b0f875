#!/usr/bin/env python3
"""Static audit of the hand-scheduled tile loop (tfa_fwd_il_asm_loop.inc) on the DISASSEMBLED object — registers resolved, unlike the .s file, whose asm
text still carries the assembler symbols.  usage: audit_il_asm_loop.py file.o [kernel-name substring]   (exit status 1 on any finding)

Finds the statement (between its own labels: il_loop<N> .. il_exit<N>, ix_b.. .. ix_exit<N>, x4_loop<N> .. x4_exit<N>, dq_loop<N> .. dq_exit<N>, kv_r0b0<N> .. kv_exit<N>) in every kernel that has one and
checks, over the linear instruction sequence, over the steady-state bodies once more behind themselves (the back edge) and over every pair of exact bodies:
  1. every LDS-read destination is complete (s_waitcnt lgkmcnt, LDS returns in order) before an instruction reads it;
  2. a VALU write of an MFMA A/B/C operand is >= 2 instructions in front of the MFMA;
  3. an MFMA result is >= 12 wait states old when a non-MFMA instruction reads or overwrites it (8-pass MFMA; an instruction between counts 1, an MFMA
     between counts 8: behind a busy matrix pipe its issue alone takes that long);
  4. an LDS read never lands in a fragment buffer whose MFMA was issued fewer than 1 MFMA ago (the distance hipcc's own schedule keeps);
  5. the block is ONE basic block per tile (no branch targets inside), and reports its instruction mix."""
import collections
import os
import re
import shutil
import subprocess
import sys


def find_objdump():
    """llvm-objdump of the toolchain that built the object: $TFA_OBJDUMP, else next to $HIPCC / under $ROCM_PATH, else /opt/rocm, else PATH"""
    cands = [os.environ.get("TFA_OBJDUMP", "")]
    hipcc = os.environ.get("HIPCC", "")
    if hipcc:
        root = os.path.dirname(os.path.dirname(os.path.realpath(shutil.which(hipcc) or hipcc)))
        cands += [os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"), os.path.join(root, "llvm", "bin", "llvm-objdump")]
    for root in (os.environ.get("ROCM_PATH", ""), "/opt/rocm"):
        if root:
            cands.append(os.path.join(root, "lib", "llvm", "bin", "llvm-objdump"))
    cands.append(shutil.which("llvm-objdump") or "")
    for c in cands:
        if c and os.path.isfile(c) and os.access(c, os.X_OK):
            return c
    sys.exit("audit_il_asm_loop.py: no llvm-objdump found (looked at: " + ", ".join(c for c in cands if c) + "); set TFA_OBJDUMP or ROCM_PATH")


def regs(tok):
    tok = tok.strip()
    neg = tok.startswith("-")
    if neg:
        tok = tok[1:]
    m = re.match(r"^v\[(\d+):(\d+)\]$", tok)
    if m:
        return list(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"^v(\d+)$", tok)
    return [int(m.group(1))] if m else []


def parse(line):
    t = line.split("//")[0].strip()
    if not t:
        return None
    parts = t.split(None, 1)
    op = parts[0]
    ops = [o.strip() for o in parts[1].split(",")] if len(parts) > 1 else []
    ops = [o.split(" offset:")[0].split(" op_sel")[0] for o in ops]
    return op, ops


def check(loop, report_from=0):
    """findings (strings) of the linear scan of `loop`; only findings at instruction index >= report_from are returned (a sequence checked twice in a
    row — body n, then body n + 1 = the loop's back edge — reports the second copy only)"""
    out = []

    def say(idx, msg):
        if idx >= report_from:
            out.append(msg)

    pending = collections.deque()          # LDS reads in flight: destination registers, in issue order
    inflight = set()
    mfma_written = {}                      # reg -> index of the MFMA that last wrote it
    valu_written = {}                      # reg -> index of the VALU instruction that last wrote it
    mfma_read_ab = {}                      # reg -> count of MFMAs issued when an MFMA last read it as A/B
    n_mfma = 0
    for idx, (op, ops) in enumerate(loop):
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", " ".join(ops))
            if m:
                keep = int(m.group(1))
                while len(pending) > keep:
                    for r in pending.popleft():
                        inflight.discard(r)
            continue
        if op in ("s_barrier",) or op.startswith("s_"):
            continue
        dst = regs(ops[0]) if ops else []
        srcs = [r for o in ops[1:] for r in regs(o)]
        if op.startswith("ds_read"):
            addr = regs(ops[1])
            for r in addr:
                if r in inflight:
                    say(idx, f"   [1] {op} address v{r} still in flight (#{idx})")
            for r in dst:
                if r in mfma_read_ab and n_mfma - mfma_read_ab[r] < 1:
                    say(idx, f"   [4] {op} lands in v{r}, read by the MFMA issued just before (#{idx})")
            pending.append(dst)
            inflight.update(dst)
            continue
        if op.startswith("ds_write"):                  # (a store holds its place in the in-order LDS queue that lgkmcnt counts; it has no destination)
            for r in [x for o_ in ops for x in regs(o_)]:
                if r in inflight:
                    say(idx, f"   [1] {op} reads v{r} while its LDS read is in flight (#{idx})")
            pending.append([])
            continue
        if op.startswith("buffer_load"):
            srcs = regs(ops[0])
            dst = []
        for r in srcs + (dst if not op.startswith("v_mfma") else []):
            if r in inflight:
                say(idx, f"   [1] {op} touches v{r} while its LDS read is in flight (#{idx})")
        if op.startswith("v_mfma"):
            for r in srcs:
                if r in valu_written and idx - valu_written[r] < 3:      # >= 2 instructions strictly between
                    say(idx, f"   [2] {op} reads v{r} written by a VALU instruction {idx - valu_written[r] - 1} instructions earlier (#{idx})")
            for r in regs(ops[1]) + regs(ops[2]):
                mfma_read_ab[r] = n_mfma + 1
            n_mfma += 1
            for r in dst:
                mfma_written[r] = idx
            continue
        for r in srcs + dst:
            if r in mfma_written:
                ws = sum(8 if loop[k][0].startswith("v_mfma") else 1 for k in range(mfma_written[r] + 1, idx))
                if ws < 12:
                    say(idx, f"   [3] {op} touches v{r}, an MFMA result only {ws} wait states old (#{idx})")
        if op.startswith("v_") and not op.startswith("v_cmp"):
            for r in dst:
                valu_written[r] = idx

    return out


def main():
    allow_missing = "--allow-missing" in sys.argv          # (an A/B build with -DTFA_IL_USE_ASMLOOP=0 / -DTFA_X4_USE_ASMLOOP=0 has no loop to audit: the Makefile says so)
    if allow_missing:
        sys.argv.remove("--allow-missing")
    obj = sys.argv[1]
    pat = sys.argv[2] if len(sys.argv) > 2 else "fwd_kernel_"
    r = subprocess.run([find_objdump(), "-d", obj], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0 or not r.stdout:
        sys.exit(f"audit_il_asm_loop.py: llvm-objdump failed on {obj} (exit {r.returncode}): {r.stderr.strip()[:300]}")
    txt = r.stdout
    kernels, cur = collections.OrderedDict(), None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if m.group(1).startswith("_Z") or cur is None:      # (local labels of an asm statement — il_loop0, il_exit0 — belong to the kernel around them)
                cur = m.group(1)
                kernels[cur] = []
            else:
                kernels[cur].append(("label", [m.group(1)]))
            continue
        if cur is not None and line.startswith("\t"):
            p = parse(line)
            if p:
                kernels[cur].append(p)
    bad, found = 0, 0
    for name, ins in kernels.items():
        if pat not in name:
            continue
        # the loop lies between the asm statement's own labels: il_loop<N> .. il_exit<N> (lazy reference), the first ix_b.. label's exact_step .. ix_exit<N>
        labels = [(i, o[0]) for i, (op, o) in enumerate(ins) if op == "label"]
        first = next((i for i, n in labels if n.startswith("il_loop") or n.startswith("ix_b") or n.startswith("x4_loop") or n.startswith("dq_loop") or n.startswith("kv_r0")), None)
        last = next((i for i, n in labels if n.startswith("il_exit") or n.startswith("ix_exit") or n.startswith("x4_exit") or n.startswith("dq_exit") or n.startswith("kv_exit")), None)
        if first is None or last is None:
            continue
        if any(n.startswith("ix_b") for _, n in labels):
            first = max(i for i in range(first) if ins[i][0].startswith("v_xor_b32")) + 1     # the exact loop starts with the first tile's exact_step
        loop = [x for x in ins[first:last] if x[0] != "label"]
        found += 1
        n_tiles = sum(1 for op, _ in loop if op == "s_barrier")
        mix = collections.Counter()
        for op, _ in loop:
            k = ("mfma" if op.startswith("v_mfma") else "trans" if op.startswith("v_exp") else "valu" if op.startswith("v_") else "lds" if op.startswith("ds_")
                 else "vmem" if op.startswith("buffer_") else "waitcnt" if op == "s_waitcnt" else "barrier" if op == "s_barrier" else "branch" if "branch" in op else "salu")
            mix[k] += 1
        print(f"{name[:100]}\n   loop: {len(loop)} instructions, {n_tiles} tiles -> {len(loop) / n_tiles:.0f} per tile: " +
              "  ".join(f"{k} {v / n_tiles:.1f}" for k, v in sorted(mix.items())))
        finds = check(loop)
        # the back edge: the steady-state bodies once more behind themselves (lazy loop: il_loop .. the backward s_branch; x4: the same), and every body of the
        # exact loop behind every other (its four bodies follow each other in any order)
        segs, curseg = [], None
        for op_, o_ in ins[first:last]:
            if op_ == "label":
                curseg = [o_[0], []]
                segs.append(curseg)
            elif curseg is not None:
                curseg[1].append((op_, o_))
        steady = None
        for k_, sg in enumerate(segs):
            if sg[0].startswith("il_loop") or sg[0].startswith("x4_loop") or sg[0].startswith("dq_loop"):
                steady = list(sg[1])
                for sg2 in segs[k_ + 1:]:                  # (the dQ bodies carry labels of their own — a skipped request — that split nothing)
                    if not sg2[0].startswith("dq_b"):
                        break
                    steady += sg2[1]
                break
        if steady:
            cut = next((k for k, (op_, _) in enumerate(steady) if op_ == "s_branch"), len(steady) - 1) + 1
            finds += ["(back edge) " + f_ for f_ in check(steady[:cut] + steady[:cut], report_from=cut)]
        xb = [sg[1] for sg in segs if sg[0].startswith("ix_b")]
        for a_ in xb:
            for b_ in xb:
                finds += ["(body pair) " + f_ for f_ in check(a_ + b_, report_from=len(a_))]
        # the fused dK/dV statement: per role three bodies in rotation (kv_r<role>b<phase>; the labels in between — a skipped request — split nothing)
        for role in "01":
            kb = []
            for ph in "012":
                body, on = [], False
                for sg in segs:
                    if sg[0].startswith(f"kv_r{role}b{ph}") and not sg[0].startswith(f"kv_r{role}b{ph}n"):
                        on = True
                    elif on and sg[0].startswith("kv_r") and not sg[0].startswith(f"kv_r{role}b{ph}n"):
                        on = False
                    elif on and sg[0].startswith("kv_exit"):
                        on = False
                    if on:
                        body += sg[1]
                kb.append(body)
            if all(kb):
                for i_ in range(3):
                    a_, b_ = kb[i_], kb[(i_ + 1) % 3]
                    finds += [f"(role {role} phase {i_} -> {(i_ + 1) % 3}) " + f_ for f_ in check(a_ + b_, report_from=len(a_))]
        for f_ in finds:
            print(f_)
        bad += len(finds)
        branches = [i for i, (op, _) in enumerate(loop) if "branch" in op]
        print(f"   branches inside the loop: {len(branches)}; findings: {bad}")
    if not found:
        print("no hand-scheduled tile loop found in", obj)
        sys.exit(0 if allow_missing else 2)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
