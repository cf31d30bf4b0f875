#!/usr/bin/env python3
"""Generates tiny-flash-attention_amd/csrc/tfa_fwd_x4_asm_loop.inc: the steady-state tile loop of the 256-wide forward kernel (fwd_kernel_x4 with ONE 32-row
block per wave: head dim 256, four waves, one per SIMD) as hand-scheduled gfx950 assembly — the sibling of tools/gen_il_asm_loop.py.

    python tools/gen_x4_asm_loop.py > tiny-flash-attention_amd/csrc/tfa_fwd_x4_asm_loop.inc

Why: with one wave per SIMD every issued instruction costs the wave 4-5 cycles (docs/LABLOG.md L-9 item 3), and hipcc's schedule of this tile is ~590 instructions
for 64 MFMAs (2048 cycles of matrix pipe): the loop is issue-bound.  Among the 590: one VALU add per K fragment read (the K ring's buffer offset is a run-time
value), 64 + 21 scalar instructions, 36 s_nop, 12 v_readlane (SGPR spills), one s_waitcnt per MFMA.  Here:
  * the K ring (three buffers) and the V pair are unrolled: SIX tile bodies (K(t) in buffer t % 3, V(t) in buffer t % 2, S(t) in sA / sB by t % 2), every LDS
    address an immediate offset on one of two address sets (buffers 0/1: k-slot address + offset; buffer 2 lies beyond the 16-bit offset field: a second set);
  * P is formed in place in the S registers (element e = register e of the S pair; P slot s = registers 8s .. 8s+3), as in the il loop;
  * fragments travel in groups of four (one s_waitcnt per four MFMAs; eight buffers), and the stream does not stop at the tile boundary: the first K fragments
    of the next tile are requested behind the last PV MFMAs, in front of the barrier (hipcc's `kpre`): f0 / f1 are `kpre` on entry and on exit;
  * O lives in a[0:127], Q in a[128:191] (tfa_fwd_kernel_x4.h: hand-owned AccVGPRs) and are named literally; everything else is the compiler's choice
    (generic operands; single registers of tuples through assembler symbols parsed from the operand strings, see gen_il_asm_loop.py).
The loop is entered at a tile j with j % 6 == 0, runs while the next tile exists for the wave, is unmasked, K(j+3) exists and no row outgrew its reference;
it returns the first tile it did not process (any phase): S(j) in sA / sB by parity, its half-wave row maximum in ma / mb, the first two K fragments of
K(j+1) in f0 / f1.  Same arithmetic, same order as the compiler-scheduled body: bits identical.
"""
import os

DT = 8                                   # 32-column blocks that hold valid head-dim columns (DVB of tfa_fwd_kernel_x4.h: 5..8; set per text by build())
N1 = N2 = 4 * DT                         # K fragments (= S MFMAs) and V fragments (= PV MFMAs) per tile
NE, NE1, TAIL = 32, 20, 3 * DT - 1       # tfa_fwd_kernel_x4.h: x4_slot_of<N1, NE1, NE, RB*DT*3-1> with TFA_X4_NE1 = 40, RB = 1
DTL = 8                                  # the V tile's LDS layout always counts eight column blocks (the 256-wide image)
TILE = 32768
PPW = 8
DMA0, DMASTEP = 1, 1
QBASE = 128
MFMA = CVT = None
NBUF = 8 if os.environ.get("TFA_GEN_X4_QUAD", "1") == "1" else 4     # fragment buffers: 8 = fragments travel in QUADS (one s_waitcnt per four MFMAs), 4 = pairs
MAXFREE = False                          # set by build() while it writes a max-free text (bf16, round 6): no row maximum of S(t+1), a guard on the partial row sums (gen_il_asm_loop.py)
GUARD = "0x53800000"                     # 2^40

PARSED = {"sa0": "SA0", "sa1": "SA1", "sb0": "SB0", "sb1": "SB1", "l0": "L0", "l1": "L1", "l2": "L2", "l3": "L3",
          "f0": "F0", "f1": "F1", "f2": "F2", "f3": "F3", "ka": "KA", "kb": "KB"}
if NBUF == 8:
    PARSED.update({"f4": "F4", "f5": "F5", "f6": "F6", "f7": "F7"})


def parse_block(op, sym):
    return [f".set {sym}, 0", ".set _tfa_pd, 0", f'.irpc c, "%[{op}]"', ".ifc \\c, :", ".set _tfa_pd, 1", ".endif", ".if _tfa_pd == 0",
            ".irp d,0,1,2,3,4,5,6,7,8,9", ".ifc \\c, \\d", f".set {sym}, {sym}*10+\\d", ".endif", ".endr", ".endif", ".endr"]


def slot_of(n):
    return 1 + (n * N1 // NE1 if n < NE1 else N1 - 1 + (n - NE1) * TAIL // (NE - NE1))


def S(cur, e, n=1):
    base = ("SA" if cur == "a" else "SB") + ("0" if e < 16 else "1")
    off = e & 15
    return f"v[{base}+{off}]" if n == 1 else f"v[{base}+{off}:{base}+{off + n - 1}]"


def Sfull(cur, half):
    return f"%[s{cur}{half}]"


def frag(g, n=4, sub=0):
    b = f"F{g % NBUF}"
    return f"%[f{g % NBUF}]" if (n == 4 and sub == 0) else f"v[{b}+{sub}:{b}+{sub + n - 1}]"


def kaddr(ks, buf):
    """address register and immediate offset base of k-slot ks in K ring buffer buf"""
    if buf < 2:
        return ("%[kaddr]" if ks == 0 else f"v[KA+{ks}]"), buf * TILE
    return f"v[KB+{ks}]", 0


def frag_reads(t6, g):
    """ds_read instructions for fragment g of the tile with phase t6 (0..5); g >= 64 = the first K fragments of the NEXT tile (read in front of the barrier)"""
    if g >= N1 + N2:
        return frag_reads((t6 + 1) % 6, g - (N1 + N2))
    if g < N1:
        kt, ks = g & 1, g >> 1
        reg, off = kaddr(ks, (t6 + 1) % 3)               # part 1 reads K(t+1)
        return [f"ds_read_b128 {frag(g)}, {reg} offset:{off + kt * 16384}"]
    i = g - N1
    off = (t6 % 2) * TILE + (i // DT) * (2 * DTL * 512) + (i % DT) * 512        # part 2 reads V(t)
    return [f"ds_read_b64_tr_b16 {frag(g, 2, 0)}, %[va] offset:{off}", f"ds_read_b64_tr_b16 {frag(g, 2, 2)}, %[va] offset:{off + 256}"]


def body(t6):
    par = t6 % 2
    cur, nxt = ("a", "b") if par == 0 else ("b", "a")
    o = []
    a = o.append
    post = {}
    G = NBUF // 2                                          # fragments per group: the group behind the current one is in flight
    for g in range(N1 + N2):
        if g % G == 0:
            # fragment g+G is requested in FRONT of the wait + MFMA g, fragments g+G+1 .. g+2G-1 one behind each of the group's first MFMAs: a read never
            # lands in the buffer of the MFMA issued just before it, and one s_waitcnt serves G MFMAs
            rs = frag_reads(t6, g + G)
            o.extend(rs)
            a(f"s_waitcnt lgkmcnt({len(rs)})")
            post = {g + k: frag_reads(t6, g + G + 1 + k) for k in range(G - 1)}
        # LDS-DMA: V(t+1) pieces 0..7 -> V buffer par^1, K(t+3) pieces 0..7 -> K ring buffer t % 3, one piece behind each of MFMAs 1..16.  m0 in FRONT of the
        # slot's MFMA (the wait state an M0 write needs); source offset = lane offset (VGPR) + the tile's byte offset as the SCALAR offset (no VALU add; the
        # bounds check ignores it: only tiles wholly inside the key sequence are requested, the statement's jend sees to that)
        n = (g - DMA0) // DMASTEP if (g >= DMA0 and (g - DMA0) % DMASTEP == 0) else -1
        if 0 <= n < PPW:
            a(f"s_add_u32 m0, %[ldsw], {(3 + (par ^ 1)) * TILE + n * 1024}")
        elif PPW <= n < 2 * PPW:
            a(f"s_add_u32 m0, %[ldsw], {(t6 % 3) * TILE + (n - PPW) * 1024}")
        if g < N1:
            kt, ks = g & 1, g >> 1
            c = "0" if ks == 0 else Sfull(nxt, kt)
            a(f"{MFMA} {Sfull(nxt, kt)}, {frag(g)}, a[{QBASE + 4 * ks}:{QBASE + 4 * ks + 3}], {c}")
        else:
            i = g - N1
            ob = 16 * (i % DT)
            a(f"{MFMA} a[{ob}:{ob + 15}], {frag(g)}, {S(cur, 8 * (i // DT), 4)}, a[{ob}:{ob + 15}]")
        o.extend(post.get(g, []))
        if 0 <= n < PPW:
            a(f"buffer_load_dwordx4 %[vs{n}], %[vrs], %[voff] offen lds")
        elif PPW <= n < 2 * PPW:
            a(f"buffer_load_dwordx4 %[ks{n - PPW}], %[krs], %[koff] offen lds")
        for e in range(NE):
            if max(slot_of(e) - 2, 0) == g:
                a(f"v_fma_f32 {S(cur, e)}, {S(cur, e)}, %[sc], -%[mref]")
        for e in range(NE):
            if slot_of(e) - 1 == g:
                a(f"v_exp_f32 {S(cur, e)}, {S(cur, e)}")
        for e in range(NE):
            if slot_of(e) == g:
                a(f"v_add_f32 v[L{e & 3}], v[L{e & 3}], {S(cur, e)}")
                if e & 1:
                    s, k = e >> 3, (e & 7) >> 1
                    a(f"{CVT} {S(cur, 8 * s + k)}, {S(cur, e - 1)}, {S(cur, e)}")
        # row max of S(t+1), two elements per instruction, from two MFMAs behind the end of its chains (slot N1+2) to the last slot
        for q in range(16):
            if MAXFREE:
                break
            if N1 + 2 + (q * (N2 - 2)) // 16 == g:
                if q == 0:
                    a(f"v_max_f32 %[m{nxt}], {S(nxt, 0)}, {S(nxt, 1)}")
                else:
                    a(f"v_max3_f32 %[m{nxt}], %[m{nxt}], {S(nxt, 2 * q)}, {S(nxt, 2 * q + 1)}")
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    a("s_barrier")
    a("s_add_u32 %[j], %[j], 1")
    a("s_add_u32 %[koff], %[koff], %[kstr]")
    a("s_add_u32 %[voff], %[voff], %[vstr]")
    a("s_cmp_ge_i32 %[j], %[jend]")
    if MAXFREE:                                            # what has been summed, not what is about to be: a partial row sum beyond 2^40 leaves for a re-base; every exit forms the
        a(f"v_max3_f32 v[F{NBUF // 2}], v[L0], v[L1], v[L2]")    # maximum of the tile in hand on its way out (x4_x<parity>)
        a(f"v_max_f32 v[F{NBUF // 2}], v[F{NBUF // 2}], v[L3]")
        a(f"s_cbranch_scc1 x4_x{par ^ 1}%=")
        a(f"v_cmp_lt_f32 vcc, {GUARD}, v[F{NBUF // 2}]")
        a(f"s_cbranch_vccnz x4_x{par ^ 1}%=")
        return o
    a(f"v_mul_f32 v[F{NBUF // 2}], %[sc], %[m{nxt}]")   # (the second half of the fragment buffers is dead here; the first half holds the next tile's first K fragments)
    a("s_cbranch_scc1 x4_exit%=")
    a(f"v_cmp_gt_f32 vcc, v[F{NBUF // 2}], %[thr]")
    a("s_cbranch_vccnz x4_exit%=")
    return o


def build(dtype, dvb, maxfree=False):
    global MFMA, CVT, DT, N1, N2, TAIL, MAXFREE
    MAXFREE = bool(maxfree)
    assert not (maxfree and dtype != "bf16")
    DT = dvb
    N1 = N2 = 4 * DT
    TAIL = 3 * DT - 1
    for s in range(4):
        assert slot_of(8 * s + 7) < N1 + DT * s, "a P slot is packed too late for the PV MFMA that reads it"
    assert DMA0 + (2 * PPW - 1) * DMASTEP < N1, "the row-max register serves as nothing else while DMA pieces are in front of it"
    MFMA = "v_mfma_f32_32x32x16_bf16" if dtype == "bf16" else "v_mfma_f32_32x32x16_f16"
    CVT = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
    lines = []
    for op, sym in PARSED.items():
        lines.extend(parse_block(op, sym))
    a = lines.append
    for s in range(1, 2 * DT):
        a(f"v_xor_b32 v[KA+{s}], {s << 5}, %[kaddr]")
    a("v_add_u32 v[KB+0], 0x10000, %[kaddr]")
    for s in range(1, 2 * DT):
        a(f"v_add_u32 v[KB+{s}], 0x10000, v[KA+{s}]")
    a("v_add_f32 %[thr], 0x41000000, %[mref]")
    for g in range(2, NBUF // 2):                          # (quads: hipcc's kpre brings fragments 0 and 1 of the first tile; the loop asks for the rest of its first group)
        lines.extend(frag_reads(0, g))
    a("x4_loop%=:")
    for t6 in range(6):
        lines.extend(body(t6))
    a("s_branch x4_loop%=")
    if MAXFREE:
        for par in (0, 1):
            t = "a" if par == 0 else "b"
            a(f"x4_x{par}%=:")
            a(f"v_max_f32 %[m{t}], {S(t, 0)}, {S(t, 1)}")
            for q in range(1, 16):
                a(f"v_max3_f32 %[m{t}], %[m{t}], {S(t, 2 * q)}, {S(t, 2 * q + 1)}")
            if par == 0:
                a("s_branch x4_exit%=")
    a("x4_exit%=:")
    return lines, len(body(0))


def emit(name, lines, n_tile, what):
    out = [f"#define {name} \\"]
    for l in lines:
        esc = l.replace("\\", "\\\\").replace('"', '\\"')
        out.append(f'  "{esc}\\n\\t" \\')
    out.append('  ""')
    out.append(f"#define {name}_INSTR_PER_TILE {n_tile}    // {what}")
    return out


def main():
    out = ["// tfa_fwd_x4_asm_loop.inc — GENERATED by tools/gen_x4_asm_loop.py (do not edit; re-generate).  The steady-state tile loop of fwd_kernel_x4's 256-wide",
           "// instantiations (one 32-row block per wave, one wave per SIMD; 5..8 valid 32-column blocks = head dims 136..256) as hand-scheduled gfx950 assembly: six",
           "// tile bodies (K ring of three x V pair) each; at 8 blocks 346 instructions per tile — 64 MFMA, 128 softmax VALU + 16 row-max + 2, 96 LDS reads, 16 LDS-DMA,",
           "// 17 s_waitcnt — where hipcc's schedule is ~590.  Layout, rules and the reason: the generator's docstring.",
           f"#define TFA_X4_ASM_NBUF {NBUF}    // fragment buffers the statement must provide (f0 .. f{NBUF - 1})"]
    for dvb in (5, 6, 7, 8):
        lb, n = build("bf16", dvb)
        lh, _ = build("f16", dvb)
        out.extend(emit(f"TFA_X4_ASM_LOOP_V{dvb}", lb, n, f"bf16, {dvb} column blocks"))
        out.extend(emit(f"TFA_X4_ASM_LOOP_V{dvb}_F16", lh, n, f"fp16, {dvb} column blocks"))
        if os.environ.get("TFA_GEN_X4_MF", "0") == "1":    # the max-free texts: an ARM (round 6), not in the product file — priced at +0.7 .. +1.1 % (profiles/r06_x4_maxfree_arm.txt),
            lm, nm = build("bf16", dvb, maxfree=True)      # half of the il kernels' gain (a 256-wide tile has twice the MFMAs per softmax element), not worth the redo machinery there
            out.extend(emit(f"TFA_X4_ASM_LOOP_V{dvb}_MF", lm, nm, f"bf16, {dvb} column blocks, max-free"))
    print("\n".join(out))


if __name__ == "__main__":
    main()
