#!/usr/bin/env python3
"""Generates tiny-flash-attention_amd/csrc/tfa_fwd_il_asm_loop.inc: the steady-state tile loop of the il8 forward kernel (bf16, D = 128, the
headline instantiation) as ONE hand-written basic block per tile, with a fixed register map — what hipcc's schedule of the same work is
320 instructions per tile in three blocks plus glue (profiles/r05_loop_stats.txt).

    python tools/gen_il_asm_loop.py > tiny-flash-attention_amd/csrc/tfa_fwd_il_asm_loop.inc

The loop runs tiles j, j+1, ... of a pass while the next tile exists, needs no mask and no row of the wave has outgrown its reference exponent (the
lazy-reference rule of tfa_fwd_kernel_il.h); it leaves with j = the first tile it did not process, S(j) in sA (j even) / sB (j odd) and that tile's
half-wave row maximum in `mx`.  Same arithmetic, same order per partial sum as the compiler-scheduled body: bits identical (tools/ab_multi.py --check).

Registers: every operand is the compiler's choice (generic "v" / "s" constraints — fixed physical registers made hipcc copy 96 registers in and out and
spill 57: the first version).  The kernel is compiled with amdgpu_num_vgpr(96) = 192 allocator-owned registers; O lives in v[192:255], tfa_fwd_il_regs.h.
  sa0, sa1   S of even tiles, two 16-register MFMA accumulators (element e of a tile is register e of the pair); P is formed IN PLACE:
             x = fma(s, c, -mref) -> exp2 -> the packed 16-bit pair (2k, 2k+1) of P slot s lands in register 8s + k, so a slot's four
             registers are the PV MFMA's B operand without a move (hipcc's version needs 16 + ~10 more registers for pw / xs)
  sb0, sb1   S of odd tiles
  q0..q7     the eight k-slot fragments of the wave's rows (MFMA B operands of S^T = K Q^T)
  f0..f3     fragment buffers: K fragments by ds_read_b128, V fragments by two ds_read_b64_tr_b16, read in PAIRS two MFMAs ahead (one s_waitcnt per pair)
  l0..l3     four interleaved partial row sums;  mref (input), thr = mref + 8, tmp;  ma / mb: half-wave row maximum of the even / odd tile a body produced
  ka         K fragment addresses of k-slots 1..7 (kaddr ^ (slot << 5); slot 0 is kaddr itself);  va: V fragment base
  ks0, ks1, vs0, vs1   lane offsets of this wave's two K and two V LDS-DMA pieces; koff / voff (scalars): byte offset of the tile to request, advanced per
             tile and handed to the load as its scalar offset (which the bounds check ignores: only tiles wholly inside the key sequence are requested here)
Single registers of a tuple are reached through assembler symbols (SA0, ... KA) that .irpc blocks at the top parse out of the operand strings.

Shapes (second half of round 5): build(dtype, d, ppw) writes the text for a kernel width d (128: eight k-slots, four O column tiles, 16 KiB tiles; 64: four,
two, 8 KiB) and ppw LDS-DMA pieces per wave and tensor (tile bytes / 1 KiB / waves of a key-tile group: il8 2 / 1, il4 and the key-split kernels 4 / 2).  The
file carries TFA_IL_ASM_LOOP[_F16] (128 x 2, the headline), _EXACT[_F16] (128 x 2, exact running maximum), _D128_P4, _D64_P1, _D64_P2 (+ _F16); q4..q7,
ka5..ka7 and the higher ks / vs operands exist only in the shapes that have them (tfa_fwd_kernel_il.h: TFA_IL_ASM_LAZY_STMT_G).
"""
import sys

import os
N1, N2, DT = 16, 16, 4                                    # QK^T MFMAs, PV MFMAs per tile and 32-column tiles of O: set per text by build() (128 wide: 16, 16, 4; 64 wide: 8, 8, 2)
DS = 8                                                     # k-slots of 16 columns (Q fragments): D / 16
PPW = 2                                                    # LDS-DMA pieces per wave per tensor per tile: (64 * D * 2 / 1024) / waves of a key-tile group
NE1 = int(os.environ.get("TFA_GEN_NE1", "21"))             # softmax elements summed / packed behind the QK^T MFMAs (of 32); the rest behind the PV MFMAs
DMA0 = int(os.environ.get("TFA_GEN_DMA0", "0"))            # first of the four MFMA slots that carry an LDS-DMA piece
PRE = int(os.environ.get("TFA_GEN_PRE", "0"))              # softmax elements whose scale/subtract AND exp2 are issued in FRONT of the tile's first wait + MFMA (behind the
                                                           # barrier every wave waits ~100 cycles for its first K fragments: VALU work of the tile's own S fits there)
EXPD = int(os.environ.get("TFA_GEN_EXPD", "1"))            # an element's exp2 is issued EXPD slots, its scale/subtract 2 * EXPD slots ahead of its sum/pack slot
TILE = 16384                                               # bytes of a K / V tile in LDS: 64 * D * 2 (set per text)
MFMA = "v_mfma_f32_32x32x16_bf16"                          # set per dtype by main()
CVT = "v_cvt_pk_bf16_f32"
WITH_TAIL = int(os.environ.get("TFA_GEN_TAIL", "1"))   # the lazy-reference statements also carry the bodies behind the loop (round 6): dispatch, N / M / L per parity
XL = True                                                  # the lazy statements take %[xl] (early requests for a pair's second pass, round 6); cleared while the exact statement's text is written
NINF = "thr"                                               # the operand a masked tail body parks -inf in: thr (lazy statements), alpha (the exact statement; dead behind exact_step's re-base)
MAXFREE = False                                            # set by build() while it writes the max-free text (bf16 only): no row maximum of S(j+1), a guard on the partial row sums instead
GUARD = "0x53800000"                                       # 2^40: a partial row sum beyond it leaves the statement for a re-base (87 binary orders below fp32 overflow)
KSTEP = 1                                                  # tiles of the head between two tiles of a wave (key-split kernels: 2)
NBUF = int(os.environ.get("TFA_GEN_NBUF", "4"))    # experiment knob: fewer fragment buffers (WRONG results below 4 with this schedule: register-pressure probe only)

# operands whose register NUMBER the text needs (sub-registers of a tuple, or single registers used inside v[..] expressions): name -> asm symbol
PARSED = {"sa0": "SA0", "sa1": "SA1", "sb0": "SB0", "sb1": "SB1", "l0": "L0", "l1": "L1", "l2": "L2", "l3": "L3",
          "f0": "F0", "f1": "F1", "f2": "F2", "f3": "F3", "ka": "KA"}
KADDR = {0: "%[kaddr]", 1: "v[KA+0]", 2: "v[KA+1]", 3: "v[KA+2]", 4: "v[KA+3]", 5: "%[ka5]", 6: "%[ka6]", 7: "%[ka7]"}


def parse_block(op, sym):
    """assembler directives that set symbol `sym` to the number of the first register of inline-asm operand %[op] ("v[12:27]" or "v5")"""
    return [f".set {sym}, 0", ".set _tfa_pd, 0", f'.irpc c, "%[{op}]"', ".ifc \\c, :", ".set _tfa_pd, 1", ".endif", ".if _tfa_pd == 0",
            ".irp d,0,1,2,3,4,5,6,7,8,9", ".ifc \\c, \\d", f".set {sym}, {sym}*10+\\d", ".endif", ".endr", ".endif", ".endr"]


def slot_of_elem(e):
    return 1 + (e * N1 // NE1 if e < NE1 else N1 + (e - NE1) * (3 * DT - 1) // (32 - NE1))


def S(cur, e, n=1):
    """register(s) of S element e (0..31) of the tile set `cur` ('a' or 'b'): two 16-register tuples"""
    base = ("SA" if cur == "a" else "SB") + ("0" if e < 16 else "1")
    off = e & 15
    return f"v[{base}+{off}]" if n == 1 else f"v[{base}+{off}:{base}+{off + n - 1}]"


def Sfull(cur, half):
    return f"%[s{cur}{half}]"


def frag(g, n=4, sub=0):
    b = f"F{g % NBUF}"
    return f"%[f{g % NBUF}]" if (n == 4 and sub == 0) else f"v[{b}+{sub}:{b}+{sub + n - 1}]"


def frag_reads(g, par, buf=None):
    """ds_read instructions that bring fragment g (0..31) of the tile body into buffer g % 4 (buf: into the buffer of that index instead — bodies that leave fragments out)"""
    b = g if buf is None else buf
    if g < N1:
        kt, ks = g & 1, g >> 1
        off = (par ^ 1) * TILE + kt * (TILE // 2)             # key block kt = rows 32 kt .. of the tile
        return [f"ds_read_b128 {frag(b)}, {KADDR[ks]} offset:{off}"]
    i = g - N1
    off = (2 + par) * TILE + (i // DT) * (2 * DT * 512) + (i % DT) * 512
    return [f"ds_read_b64_tr_b16 {frag(b, 2, 0)}, %[va] offset:{off}",
            f"ds_read_b64_tr_b16 {frag(b, 2, 2)}, %[va] offset:{off + 256}"]


def mask_ko(e):
    """key offset inside the tile of S element e (0..31) of a lane with hi = 0: 32 * key block + (r & 3) + 8 * (r >> 2) — tfa_fwd_kernel_il.h: apply_mask"""
    tt, r = e >> 4, e & 15
    return 32 * tt + (r & 3) + 8 * (r >> 2)


def body(par, lbl, exact=False, resc=False, tail=False, mask=False, half=False):
    """One tile body.  tail: a body OUTSIDE the loop (round 6) — no loop control behind the barrier, the K(j+2) request only when %[ik] != 0 (a wave's
    last tiles: K(j+2) may lie behind the block's last tile); mask: S(j+1) is the wave's masked (diagonal / ragged) tile — element e becomes -inf where
    its key offset exceeds the lane's limit %[lim], two VALU per element in front of the row maximum that reads it; half (with mask): the masked tile's SECOND
    key block is hidden from every row of the wave (the diagonal's even waves: rows 0..31 of a 64-key tile see keys 0..31 at most) — its eight QK^T MFMAs and
    their fragment reads are left out, its sixteen S registers are set to -inf outright"""
    assert not half or (mask and tail)

    # the fragments this body reads, in order; fragment al[k] travels in buffer k % 4 and in pairs by k (a body that leaves fragments out keeps the pairing and the
    # one-MFMA distance between an MFMA and the next read into its buffer)
    al = [g for g in range(N1 + N2) if not (half and g < N1 and (g & 1))]
    ai = {g: k for k, g in enumerate(al)}

    def reads(k):
        return frag_reads(al[k], par, buf=k) if k < len(al) else []
    cur, nxt = ("a", "b") if par == 0 else ("b", "a")
    o = []
    a = o.append
    a(f"; ---- tile of parity {par}: S(j) in s{cur} -> P, O += P V(j); S(j+1) = K(j+1) Q^T -> s{nxt}")
    if mask:
        # the lane's mask limit for tile j + 1, in the (dead) maximum of tile j: lane & 31 (its row inside the wave) - 4 * (lane >> 5) (the half-wave's key offset,
        # apply_mask) + slim - 64 * (j + 1 - fmx), slim = the wave's first row + the causal shift - the first key of tile fmx (a scalar).  The tails only run on
        # blocks whose tiles all lie inside the keys, so the key-count bound of apply_mask never binds.  thr is scratch first, then holds -inf for the selects
        a(f"v_mbcnt_lo_u32_b32 %[{NINF}], -1, 0")
        a(f"v_mbcnt_hi_u32_b32 %[{NINF}], -1, %[{NINF}]")
        a("s_add_u32 %[ts], %[j], 1")
        a("s_sub_u32 %[ts], %[ts], %[fmx]")
        a(f"s_lshl_b32 %[ts], %[ts], {6 + (KSTEP - 1)}")
        a("s_sub_u32 %[ts], %[slim], %[ts]")
        a(f"v_and_b32 %[m{cur}], 31, %[{NINF}]")
        a(f"v_lshrrev_b32 %[{NINF}], 5, %[{NINF}]")
        a(f"v_lshlrev_b32 %[{NINF}], 2, %[{NINF}]")
        a(f"v_sub_u32 %[m{cur}], %[m{cur}], %[{NINF}]")
        a(f"v_add_u32 %[m{cur}], %[ts], %[m{cur}]")
        a(f"v_mov_b32 %[{NINF}], 0xff800000")
    # fragments travel in pairs: at an even slot g fragment g+2 is requested in FRONT of the wait + MFMA g and fragment g+3 BEHIND MFMA g, so that a read
    # never lands in the buffer of the MFMA issued just before it (one MFMA of distance, what hipcc's own schedule keeps) and one s_waitcnt serves two MFMAs
    for k in (0, 1):
        o.extend(reads(k))
    if PRE:
        o.extend(reads(2))                                 # (the slot-0 pre-read moves up as well: all three requests are out before the VALU work)
        for e in range(PRE):
            a(f"v_fma_f32 {S(cur, e)}, {S(cur, e)}, %[sc], -%[mref]")
        for e in range(PRE):
            a(f"v_exp_f32 {S(cur, e)}, {S(cur, e)}")
    post = []
    for g in range(N1 + N2):
        k = ai.get(g)                                      # None: a fragment (and MFMA) this body leaves out — the slot keeps its share of the DMA and softmax work
        if k is not None and k % 2 == 0:
            cnt = 0
            if k + 2 < len(al):
                rs = reads(k + 2)
                if not (PRE and k == 0):
                    o.extend(rs)
                cnt = len(rs)
            a(f"s_waitcnt lgkmcnt({cnt})")
            post = reads(k + 3)
        # LDS-DMA pieces behind the first four MFMAs: V(j+1) -> V buffer par^1, K(j+2) -> K buffer par.  m0 is written in FRONT of the slot's MFMA (which
        # is the wait state an M0 write needs before the load reads it); the source offset is the piece's lane offset (VGPR) plus the tile's byte offset
        # as the instruction's SCALAR offset — no VALU add.  The scalar offset takes no part in the descriptor's bounds check: the loop only requests tiles
        # that lie wholly inside the key sequence (the host side of the statement limits jend), lanes of chunks beyond the head dim stay out of range by themselves
        gd = g - DMA0
        if 0 <= gd < PPW:
            a(f"s_add_u32 m0, %[ldsw], {(2 + (par ^ 1)) * TILE + gd * 1024}")
        elif PPW <= gd < 2 * PPW:
            a(f"s_add_u32 m0, %[ldsw], {par * TILE + (gd - PPW) * 1024}")
        if g < N1:
            kt, ks = g & 1, g >> 1
            c = "0" if ks == 0 else Sfull(nxt, kt)
            if k is not None:
                a(f"{MFMA} {Sfull(nxt, kt)}, {frag(k)}, %[q{ks}], {c}")
        else:
            i = g - N1
            ob = 192 + 16 * (i % DT)
            # a pack that wrote this MFMA's P operand needs two instructions in front of the MFMA (the max-free texts have no row maximum left to fill the gap)
            real = [l for l in o if not (l.startswith(";") or l.endswith(":"))]
            since = next((k for k, l in enumerate(reversed(real)) if l.startswith(CVT)), 99)
            if since < 2:
                a(f"s_nop {1 - since}")
            a(f"{MFMA} v[{ob}:{ob + 15}], {frag(k)}, {S(cur, 8 * (i // DT), 4)}, v[{ob}:{ob + 15}]")
        if k is not None and k % 2 == 0:
            o.extend(post)
        if 0 <= gd < PPW:
            a(f"buffer_load_dwordx4 %[vs{gd}], %[vrs], %[voff] offen lds")
        elif PPW <= gd < 2 * PPW:
            if tail:                                       # (behind the loop tile j may be the block's last but one: K(j+2) exists only while j + 2 < nt)
                a("s_add_u32 %[ts], %[j], 2")
                a("s_cmp_ge_i32 %[ts], %[nt]")
                a(f"s_cbranch_scc1 {lbl}_{('mh' if half else 'm') if mask else 'n'}{par}nok{gd}%=")
            a(f"buffer_load_dwordx4 %[ks{gd - PPW}], %[krs], %[koff] offen lds")
            if tail:
                a(f"{lbl}_{('mh' if half else 'm') if mask else 'n'}{par}nok{gd}%=:")
        if resc and g < N1 and N1 * 4 == 16 * DT:          # the re-basing body: O *= alpha rides behind the QK^T MFMAs, one register quad per MFMA
            for k in range(4):
                a(f"v_mul_f32 v{192 + 4 * g + k}, v{192 + 4 * g + k}, %[alpha]")
        # this slot's share of tile j's softmax: scale/subtract two slots ahead of an element's own slot, exp2 one ahead, sum + pack in it
        for e in range(PRE, 32):
            if max(slot_of_elem(e) - 2 * EXPD, 0) == g:
                a(f"v_fma_f32 {S(cur, e)}, {S(cur, e)}, %[sc], -%[mref]")
        for e in range(PRE, 32):
            if max(slot_of_elem(e) - EXPD, 0) == g:
                a(f"v_exp_f32 {S(cur, e)}, {S(cur, e)}")
        for e in range(32):
            if slot_of_elem(e) == g:
                a(f"v_add_f32 v[L{e & 3}], v[L{e & 3}], {S(cur, e)}")
                if e & 1:
                    s, k = e >> 3, (e & 7) >> 1
                    a(f"{CVT} {S(cur, 8 * s + k)}, {S(cur, e - 1)}, {S(cur, e)}")
        if g >= N1:                                        # row max of S(j+1): sixteen pairs of elements over the N2 slots of part 2
            for q in range(16):
                if q * N2 // 16 == g - N1:
                    if half and q >= 8:                    # the hidden key block: -inf outright (whoever takes the tile over — the last-tile body, the burst path — reads it so)
                        for e in (2 * q, 2 * q + 1):
                            a(f"v_mov_b32 {S(nxt, e)}, 0xff800000")
                        continue
                    if mask:                               # (both MFMA chains of S(j+1) are >= one MFMA old here: the distance the unmasked body's maximum keeps)
                        for e in (2 * q, 2 * q + 1):
                            a(f"v_cmp_le_i32 vcc, {mask_ko(e)}, %[m{cur}]")
                            a(f"v_cndmask_b32 {S(nxt, e)}, %[{NINF}], {S(nxt, e)}, vcc")    # (a literal next to vcc is two constant-bus reads: -inf sits in thr for the length of this body)
                    if MAXFREE:
                        continue
                    if q == 0:
                        a(f"v_max_f32 %[m{nxt}], {S(nxt, 0)}, {S(nxt, 1)}")
                    else:
                        a(f"v_max3_f32 %[m{nxt}], %[m{nxt}], {S(nxt, 2 * q)}, {S(nxt, 2 * q + 1)}")
    a("s_waitcnt vmcnt(0)")
    a("s_barrier")
    a("s_add_u32 %[j], %[j], 1")
    a("s_add_u32 %[koff], %[koff], %[kstr]")
    a("s_add_u32 %[voff], %[voff], %[vstr]")
    if tail:                                               # on to the dispatch of the tile just produced (parity par ^ 1)
        if mask and not MAXFREE and NINF == "thr":
            a("v_add_f32 %[thr], 0x41000000, %[mref]")     # (thr held -inf for the mask)
        a(f"s_branch {lbl}_d{par ^ 1}%=")
        return o
    a("s_cmp_ge_i32 %[j], %[jend]")
    if not exact and MAXFREE:
        # no maximum to test: the guard is on what has been SUMMED — a partial row sum beyond 2^40 (some P of this row was that large) leaves for a re-base
        a("v_max3_f32 v[F0], v[L0], v[L1], v[L2]")        # (the fragment buffers are dead behind the last MFMA)
        a("v_max_f32 v[F0], v[F0], v[L3]")
        a(f"s_cbranch_scc1 {lbl}_d{par ^ 1}%=")
        a(f"v_cmp_lt_f32 vcc, {GUARD}, v[F0]")
        a(f"s_cbranch_vccnz {lbl}_x{par ^ 1}%=")
    elif not exact:
        a(f"v_mul_f32 v[F0], %[sc], %[m{nxt}]")            # (the fragment buffers are dead behind the last MFMA)
        a(f"s_cbranch_scc1 {lbl}_d{par ^ 1}%=" if WITH_TAIL else f"s_cbranch_scc1 {lbl}_exit%=")   # the loop's range ends here: the tile just produced goes to the dispatch
        a("v_cmp_gt_f32 vcc, v[F0], %[thr]")
        a(f"s_cbranch_vccnz {lbl}_exit%=")
    else:
        a(f"s_cbranch_scc1 {lbl}_d{par ^ 1}%=" if WITH_TAIL else f"s_cbranch_scc1 {lbl}_exit%=")
        o.extend(exact_step(nxt, lbl))
    return o


def last_body(par, lbl, half=False):
    """A wave's LAST tile of a pass (round 6; hipcc's burst-structured `slow` before): S(j) in the set of parity `par` -> P, O += P V(j), nothing else — no
    S(j+1).  The softmax of P slot s + 1 rides behind the DT PV MFMAs of slot s (scale/subtract behind the first, exp2 behind the second, the sums and
    the packs behind the rest), slot 0's in front of the first MFMA; V fragments in pairs as in the loop.  The wave still asks for its pieces of the
    tiles the block's other waves go on to read: V(j+1) when %[iv] != 0, K(j+2) when %[ik] != 0.  Same operations in the same order per partial sum as
    `slow`: bits identical."""
    cur = "a" if par == 0 else "b"
    o = []
    a = o.append
    NS = 2 if half else 4                                  # P slots that can hold anything but zeros (half: the tile's second key block is hidden from the whole wave)
    NM = NS * DT                                           # PV MFMAs
    a(f"; ---- last tile of a wave, parity {par}: S(j) in s{cur} -> P, O += P V(j)" + (" (first key block only)" if half else ""))

    def vreads(i):
        return frag_reads(N1 + i, par)

    def soft(s, stage):
        es = range(8 * s, 8 * s + 8)
        if stage == 0:
            return [f"v_fma_f32 {S(cur, e)}, {S(cur, e)}, %[sc], -%[mref]" for e in es]
        if stage == 1:
            return [f"v_exp_f32 {S(cur, e)}, {S(cur, e)}" for e in es]
        if stage == 2:
            return [f"v_add_f32 v[L{e & 3}], v[L{e & 3}], {S(cur, e)}" for e in es]
        return [f"{CVT} {S(cur, 8 * s + k)}, {S(cur, 8 * s + 2 * k)}, {S(cur, 8 * s + 2 * k + 1)}" for k in range(4)]

    # the four stages of a slot's softmax over the DT MFMAs of the slot before it
    def share(s, k):
        if DT == 4:
            return soft(s, k)
        return soft(s, 0) + soft(s, 1) if k == 0 else soft(s, 2) + soft(s, 3)

    for i in (0, 1):
        o.extend(vreads(i))
    # the wave's LDS-DMA pieces first: nothing of this tile depends on them, and the other waves' next tiles do
    a("s_add_u32 %[ts], %[j], 1")
    for i in range(PPW):                                   # (s_add_u32 writes SCC: the test comes behind it; test + branch are the M0 write's wait states)
        a(f"s_add_u32 m0, %[ldsw], {(2 + (par ^ 1)) * TILE + i * 1024}")
        a("s_cmp_ge_i32 %[ts], %[nt]")
        a(f"s_cbranch_scc1 {lbl}_l{'h' if half else ''}{par}nov{i}%=")
        a(f"buffer_load_dwordx4 %[vs{i}], %[vrs], %[voff] offen lds")
        a(f"{lbl}_l{'h' if half else ''}{par}nov{i}%=:")
    a("s_add_u32 %[ts], %[j], 2")
    for i in range(PPW):
        a(f"s_add_u32 m0, %[ldsw], {par * TILE + i * 1024}")
        a("s_cmp_ge_i32 %[ts], %[nt]")
        a(f"s_cbranch_scc1 {lbl}_l{'h' if half else ''}{par}nok{i}%=")
        a(f"buffer_load_dwordx4 %[ks{i}], %[krs], %[koff] offen lds")
        a(f"{lbl}_l{'h' if half else ''}{par}nok{i}%=:")
    for st in range(4):
        o.extend(soft(0, st))
    post = []
    for i in range(NM):
        if i % 2 == 0:
            cnt = 0
            if i + 2 < NM:
                rs = vreads(i + 2)
                o.extend(rs)
                cnt = len(rs)
            elif DT == 2:
                a("s_nop 0")                               # (64 wide: the last slot's packs sit right behind MFMA NM - 3; with no read left the wait alone is one wait state of the two)
            a(f"s_waitcnt lgkmcnt({cnt})")
            post = vreads(i + 3) if i + 3 < NM else []
        ob = 192 + 16 * (i % DT)
        a(f"{MFMA} v[{ob}:{ob + 15}], {frag(N1 + i)}, {S(cur, 8 * (i // DT), 4)}, v[{ob}:{ob + 15}]")
        if i % 2 == 0:
            o.extend(post)
        if i // DT + 1 < NS:
            o.extend(share(i // DT + 1, i % DT))
    if XL:                                                 # (xl == 2: the next pass's first requests went out in front of this tile — nothing of THIS pass is in flight, and they must stay)
        a("s_cmp_eq_u32 %[xl], 2")
        a(f"s_cbranch_scc1 {lbl}_l{'h' if half else ''}{par}nw%=")
    a("s_waitcnt vmcnt(0)")
    if XL:
        a(f"{lbl}_l{'h' if half else ''}{par}nw%=:")
    a("s_barrier")
    a("s_add_u32 %[j], %[j], 1")
    a(f"s_branch {lbl}_exit%=")
    return o


def tail_blocks_exact(lbl):
    """The exact-running-max statement's tails (variant 38): the dispatch d<p> advances the running maximum to tile j as the loop's bodies do (exact_step: mref, alpha,
    l *= alpha) and, when some row of the wave moved, multiplies O by alpha in ONE burst (a tile or two per pass: no interleaved twin of every tail body); then
    the same three bodies as the lazy statement — n<p>, m<p>, l<p> — which read mref as it stands."""
    o = []
    a = o.append
    for par in (0, 1):
        t = "a" if par == 0 else "b"
        a(f"{lbl}_d{par}%=:")
        a("s_cmp_lt_i32 %[fmx], 0")
        a(f"s_cbranch_scc1 {lbl}_exit%=")
        es = exact_step(t, lbl)
        cut = next(k for k, l in enumerate(es) if l.startswith("s_cbranch_vccz"))
        o.extend(es[:cut])
        a(f"s_cbranch_vccz {lbl}_t{par}%=")
        o.extend(es[cut + 1:-1])                           # the row sums take the factor (s_nop + 4 v_mul); no branch to a loop body
        for r in range(192, 192 + 16 * DT):
            a(f"v_mul_f32 v{r}, v{r}, %[alpha]")
        a(f"{lbl}_t{par}%=:")
        a("s_add_u32 %[ts], %[j], 1")
        a("s_cmp_ge_i32 %[ts], %[nact]")
        a(f"s_cbranch_scc1 {lbl}_l{par}%=")
        a("s_cmp_ge_i32 %[ts], %[fmx]")
        a(f"s_cbranch_scc1 {lbl}_m{par}%=")
        a(f"{lbl}_n{par}%=:")
        o.extend(body(par, lbl, tail=True))
        a(f"{lbl}_m{par}%=:")
        o.extend(body(par, lbl, tail=True, mask=True))
        a(f"{lbl}_l{par}%=:")
        o.extend(last_body(par, lbl))
    return o


def tail_blocks(lbl):
    """Behind the loop, in the same statement (round 6): per parity p of the tile j whose S the wave holds — the DISPATCH d<p> (leave for the compiler-scheduled
    paths when the tails are off for this pass (fmx < 0) or a row of tile j has outgrown its reference; else j is the wave's last tile -> l<p>, tile j + 1 is
    masked -> m<p>, or -> n<p>), and the three bodies: n<p> / m<p> run tile j like the loop's body (m: S(j+1) masked against the lane's limit) and go on to
    d<p^1>; l<p> runs the last tile and leaves with j = nact."""
    o = []
    a = o.append
    for par in (0, 1):
        t = "a" if par == 0 else "b"
        a(f"{lbl}_d{par}%=:")
        if MAXFREE:
            a("v_max3_f32 v[F0], v[L0], v[L1], v[L2]")
            a("v_max_f32 v[F0], v[F0], v[L3]")
            a("s_cmp_lt_i32 %[fmx], 0")
            a(f"s_cbranch_scc1 {lbl}_x{par}%=")
            a(f"v_cmp_lt_f32 vcc, {GUARD}, v[F0]")
            a(f"s_cbranch_vccnz {lbl}_x{par}%=")
        else:
            a(f"v_mul_f32 v[F0], %[sc], %[m{t}]")
            a("s_cmp_lt_i32 %[fmx], 0")
            a(f"s_cbranch_scc1 {lbl}_exit%=")
            a("v_cmp_gt_f32 vcc, v[F0], %[thr]")
            a(f"s_cbranch_vccnz {lbl}_exit%=")
        a("s_add_u32 %[ts], %[j], 1")
        a("s_cmp_ge_i32 %[ts], %[nact]")
        a(f"s_cbranch_scc1 {lbl}_lq{par}%=")
        a("s_cmp_ge_i32 %[ts], %[fmx]")
        a(f"s_cbranch_scc1 {lbl}_mq{par}%=")
        a(f"{lbl}_n{par}%=:")
        o.extend(body(par, lbl, tail=True))
        # a masked tile whose SECOND key block no row of the wave sees (31 + slim - 64 (tile - fmx) < 32: the diagonal's even waves) takes the half forms:
        # eight QK^T MFMAs, sixteen mask pairs, and — in the last-tile body — sixteen softmax elements and eight PV MFMAs less (1.5 % of a causal launch's MFMAs)
        a(f"{lbl}_mq{par}%=:")
        a("s_sub_u32 %[ts], %[ts], %[fmx]")
        a(f"s_lshl_b32 %[ts], %[ts], {6 + (KSTEP - 1)}")
        a("s_sub_u32 %[ts], %[slim], %[ts]")
        a("s_cmp_le_i32 %[ts], 0")
        a(f"s_cbranch_scc1 {lbl}_mh{par}%=")
        a(f"{lbl}_m{par}%=:")
        o.extend(body(par, lbl, tail=True, mask=True))
        a(f"{lbl}_mh{par}%=:")
        o.extend(body(par, lbl, tail=True, mask=True, half=True))
        a(f"{lbl}_lq{par}%=:")                             # xl == 1: leave in front of the wave's last tile (the caller sends the next pass's first requests, then comes back with xl == 2)
        a("s_cmp_eq_u32 %[xl], 1")
        a(f"s_cbranch_scc1 {lbl}_x{par}%=" if MAXFREE else f"s_cbranch_scc1 {lbl}_exit%=")
        a("s_cmp_lt_i32 %[j], %[fmx]")                     # (an unmasked last tile — non-causal passes — has both key blocks)
        a(f"s_cbranch_scc1 {lbl}_l{par}%=")
        a("s_sub_u32 %[ts], %[j], %[fmx]")
        a(f"s_lshl_b32 %[ts], %[ts], {6 + (KSTEP - 1)}")
        a("s_sub_u32 %[ts], %[slim], %[ts]")
        a("s_cmp_le_i32 %[ts], 0")
        a(f"s_cbranch_scc1 {lbl}_lh{par}%=")
        a(f"{lbl}_l{par}%=:")
        o.extend(last_body(par, lbl))
        a(f"{lbl}_lh{par}%=:")
        o.extend(last_body(par, lbl, half=True))
    if MAXFREE:
        # leaving with a tile in hand: the compiler-scheduled paths want its half-wave row maximum (the statement itself never formed it)
        for par in (0, 1):
            t = "a" if par == 0 else "b"
            a(f"{lbl}_x{par}%=:")
            a(f"v_max_f32 %[m{t}], {S(t, 0)}, {S(t, 1)}")
            for q in range(1, 16):
                a(f"v_max3_f32 %[m{t}], %[m{t}], {S(t, 2 * q)}, {S(t, 2 * q + 1)}")
            if par == 0:
                a(f"s_branch {lbl}_exit%=")
    return o


def exact_step(t, lbl):
    """The exact running maximum advanced to the tile whose S is in set `t` (half-wave row maximum in m<t>): both half-waves' maxima combined
    (v_permlane32_swap), nref = max(mref, max * c); alpha = exp2(mref - nref); mref = nref; l *= alpha; then on to that tile's body — the one that
    also re-bases O when some row of the wave moved (alpha != 1 somewhere), the plain one otherwise.  Scratch: the (dead) fragment buffers."""
    par = 0 if t == "a" else 1
    return [f"v_mov_b32 v[F0], %[m{t}]", f"v_mov_b32 v[F0+1], %[m{t}]", "s_nop 1", "v_permlane32_swap_b32 v[F0], v[F0+1]",
            "v_max_f32 v[F0], v[F0], v[F0+1]", "v_mul_f32 v[F0], %[sc], v[F0]", "v_max_f32 v[F0], %[mref], v[F0]",
            "v_cmp_neq_f32 vcc, v[F0], %[mref]", "v_sub_f32 v[F0+1], %[mref], v[F0]", "v_mov_b32 %[mref], v[F0]", "v_exp_f32 %[alpha], v[F0+1]",
            f"s_cbranch_vccz {lbl}_b{par}n%=",
            # some row moved: the row sums take the factor here, O inside the body
            "s_nop 0", "v_mul_f32 v[L0], v[L0], %[alpha]", "v_mul_f32 v[L1], v[L1], %[alpha]", "v_mul_f32 v[L2], v[L2], %[alpha]", "v_mul_f32 v[L3], v[L3], %[alpha]",
            f"s_branch {lbl}_b{par}r%="]


def emit(name, lines, n_tile, what):
    out = [f"#define {name} \\"]
    for l in lines:
        if l.startswith(";"):
            continue
        esc = l.replace("\\", "\\\\").replace('"', '\\"')
        out.append(f'  "{esc}\\n\\t" \\')
    out.append('  ""')
    out.append(f"#define {name}_INSTR_PER_TILE {n_tile}    // {what}")
    return out


def build(dtype, d=128, ppw=2, maxfree=False):
    """(lines of the lazy-reference loop, lines of the exact-running-max loop, instructions per tile of each body) for one 16-bit type, kernel width d
    (128 or 64) and ppw LDS-DMA pieces per wave and tensor (8-wave kernel: 2 / 1; 4-wave and key-split kernels: 4 / 2)"""
    global MFMA, CVT, N1, N2, DT, DS, TILE, PPW, MAXFREE
    MAXFREE = bool(maxfree)
    assert not (maxfree and (dtype != "bf16" or not WITH_TAIL)), "max-free: bf16 only (a 16-bit P with fp32's exponent range), with the tails"
    DS, DT, TILE, PPW = d // 16, d // 32, 64 * d * 2, ppw
    N1, N2 = 2 * DS, 4 * DT
    for s_ in range(4):
        assert slot_of_elem(8 * s_ + 7) < N1 + DT * s_, "a P slot is packed too late for the PV MFMA that reads it"
    assert DMA0 + 2 * PPW <= N1 + N2
    MFMA = "v_mfma_f32_32x32x16_bf16" if dtype == "bf16" else "v_mfma_f32_32x32x16_f16"
    CVT = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
    head = []
    for op, sym in PARSED.items():
        head.extend(parse_block(op, sym))
    for s in range(1, DS):
        head.append(f"v_xor_b32 {KADDR[s]}, {s << 5}, %[kaddr]")
    # ---- the lazy-reference loop (the headline kernel)
    lines = list(head)
    a = lines.append
    if not MAXFREE:
        a("v_add_f32 %[thr], 0x41000000, %[mref]")
    if WITH_TAIL:                                          # entered at any even tile without a pending re-base: tiles below jend take the loop, the rest the dispatch
        a("s_bitcmp1_b32 %[j], 0")                         # (an odd tile — S in sb — only ever comes in for the dispatch: the wave's last tile behind the early requests)
        a("s_cbranch_scc1 il_d1%=")
        a("s_cmp_ge_i32 %[j], %[jend]")
        a("s_cbranch_scc1 il_d0%=")
    a("il_loop%=:")
    lines.extend(body(0, "il"))
    lines.extend(body(1, "il"))
    a("s_branch il_loop%=")
    if WITH_TAIL:
        lines.extend(tail_blocks("il"))
    a("il_exit%=:")
    n_tile = sum(1 for l in body(0, "x") if not l.startswith(";"))
    # ---- the exact-running-max loop (VF_IL_EXACT, variant 38): per parity a plain body and one that also re-bases O; every body ends in the
    # exact_step of the tile it produced, which picks the next body
    MAXFREE = False                                        # (the exact-running-max text below is what it is for)
    global NINF, XL
    NINF, XL = "alpha", False
    xl = list(head)
    a = xl.append
    if WITH_TAIL:
        a("s_cmp_ge_i32 %[j], %[jend]")
        a("s_cbranch_scc1 ix_d0%=")
    xl.extend(exact_step("a", "ix"))
    for par in (0, 1):
        for resc in (False, True):
            a(f"ix_b{par}{'r' if resc else 'n'}%=:")
            xl.extend(body(par, "ix", exact=True, resc=resc))
    if WITH_TAIL:
        xl.extend(tail_blocks_exact("ix"))
    a("ix_exit%=:")
    NINF, XL = "thr", True
    n_x = sum(1 for l in body(0, "x", exact=True, resc=True) if not l.startswith(";"))
    n_xn = sum(1 for l in body(0, "x", exact=True, resc=False) if not l.startswith(";"))
    return lines, xl, n_tile, n_xn, n_x


def main():
    lines, xl, n_tile, n_xn, n_x = build("bf16")
    lines_h, xl_h, _, _, _ = build("f16")
    out = []
    out.append("// tfa_fwd_il_asm_loop.inc — GENERATED by tools/gen_il_asm_loop.py (do not edit; re-generate).  The steady-state tile loop of fwd_kernel_il's")
    out.append("// 128-wide 8-wave instantiations (the headline: bf16, lazy row reference; its fp16 twin) as hand-scheduled gfx950 assembly: ONE basic block of "
               f"{n_tile} instructions per tile")
    out.append("// (32 MFMA, 128 + 2 VALU, 48 LDS reads, 4 LDS-DMA, 17 s_waitcnt, 12 scalar) where hipcc's schedule of the same work is ~320 in three blocks plus glue;")
    out.append(f"// and the same for the exact-running-max instantiation (variant 38): {n_xn} instructions per tile, {n_x} in the body that also re-bases O.")
    out.append("// Registers are the COMPILER's choice (generic constraints): the text reaches single registers of a tuple through assembler symbols that the")
    out.append("// leading .irpc blocks parse out of the operand strings (\"v[12:27]\" -> 12).  Rules and layout: the generator's docstring.")
    out.extend(emit("TFA_IL_ASM_LOOP", lines, n_tile, "bf16, lazy row reference"))
    out.extend(emit("TFA_IL_ASM_LOOP_F16", lines_h, n_tile, "fp16, lazy row reference"))
    out.extend(emit("TFA_IL_ASM_LOOP_EXACT", xl, n_x, "bf16, exact running maximum, the re-basing body"))
    out.extend(emit("TFA_IL_ASM_LOOP_EXACT_F16", xl_h, n_x, "fp16, exact running maximum, the re-basing body"))
    # the max-free texts (bf16 only; every shape): no row maximum, a guard on the partial row sums
    for d, ppw, stem in ((128, 2, ""), (128, 4, "_D128_P4"), (64, 1, "_D64_P1"), (64, 2, "_D64_P2")):
        l_, _, n_, _, _ = build("bf16", d, ppw, maxfree=True)
        out.extend(emit(f"TFA_IL_ASM_LOOP{stem}_MF", l_, n_, f"bf16, {d} wide, {ppw} pieces, max-free"))
    # the other tile shapes of fwd_kernel_il (lazy row reference only): 128 wide with 4 pieces per wave (the 4-wave and the key-split kernels), 64 wide with
    # 1 piece (8 waves) or 2 (4-wave / key-split)
    for d, ppw in ((128, 4), (64, 1), (64, 2)):
        for dt in ("bf16", "f16"):
            l_, _, n_, _, _ = build(dt, d, ppw)
            out.extend(emit(f"TFA_IL_ASM_LOOP_D{d}_P{ppw}" + ("" if dt == "bf16" else "_F16"), l_, n_, f"{dt}, {d} wide, {ppw} DMA pieces per wave and tensor"))
    print("\n".join(out))


if __name__ == "__main__":
    main()
