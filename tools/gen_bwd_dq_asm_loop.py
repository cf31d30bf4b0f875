#!/usr/bin/env python3
"""Generates tiny-flash-attention_amd/csrc/tfa_bwd_dq_asm_loop.inc: the unmasked tiles of the backward's dQ launch (bwd_kernel<BWD_DQ>, 128 wide, 8 waves) as ONE
hand-scheduled basic block per tile — the forward's generator (tools/gen_il_asm_loop.py) applied to the kernel whose compiler schedule reads a fragment pair,
waits for it and issues its two MFMAs (matrix pipe 0.42 busy at 2.2 GHz: not the power cap, the schedule; profiles/r06_pmc_bwd_dq_cfg3.txt).

    python tools/gen_bwd_dq_asm_loop.py > tiny-flash-attention_amd/csrc/tfa_bwd_dq_asm_loop.inc

A tile = 64 keys against the wave's 32 resident query rows, two halves t of 32 keys:
  GI(t)   S_t = K_t Q^T, dP_t = V_t dO^T      16 MFMAs, the two chains alternating; A operands by ds_read_b128 from the row-major K / V images
  EW(t)   P = exp2(S c - lse2), dS = P dP' (dP' = dP - delta: the dP chain's first MFMA takes -delta of the lane's row as its C operand), packed IN PLACE (the pair (2k, 2k+1) of 16-key slot s lands in register 8 s + k of S_t: the slot's
          four registers are GII's B operand without a move): 56 VALU
  GII(t)  dQ^T += K_t^T dS_t                   8 MFMAs (two slots x four 32-column blocks), A operands by ds_read_b64_tr_b16 from the transposed K image
Order in a tile: GI(0) | GI(1) with EW(0) in its shadow | GII(0) with most of EW(1) | GII(1) with the rest — every VALU instruction has an MFMA in front of it.
Fragments travel as in the forward loop: four 4-register buffers in rotation, requested in pairs two MFMAs ahead, one s_waitcnt per pair.  The tile's first six
MFMA slots carry the wave's six LDS-DMA pieces of tile u + 1 (K, V and transposed-K images of the other stage); one s_waitcnt vmcnt(0) + s_barrier per tile, the
protocol of the compiler-scheduled loop around it (tfa_bwd_kernel.h), so a wave may leave for that loop — ragged and inactive tiles — at any tile.
Behind the unmasked tiles (u < %[uend]) the statement runs the wave's diagonal tiles (u < %[mend]) in MASKED bodies: S becomes -inf where the key lies behind the
lane's row, two VALU per score in front of its scale / subtract (the compare result in an SGPR pair).  Every request is predicated on "tile u + 1 exists"
(vcc, set at the head of a tile), so the block's last tile runs inside as well.

LDS (bytes; the compiler-scheduled loop of this instantiation uses the same map): row-major images at (2 stage + image) * 0x4000 (image 0 K, 1 V) — every
ds_read_b128 offset fits the 16-bit immediate —, transposed K images at 0x10000 + stage * 0x4000 (the address operand `vat` carries the 0x10000).
Registers are the compiler's choice; single registers of a tuple are reached through assembler symbols parsed out of the operand strings (the forward's way).
"""
import os
import sys

ABL = os.environ.get("TFA_GEN_DQ_ABL", "").split(",")        # timing-only ablations (WRONG results): nolds (no fragment waits), novalu, nobar, nodma
ABL_NOWAIT = int(os.environ.get("TFA_GEN_DQ_NOWAIT", "0"))   # timing-only ablation (WRONG results): the tile's own LDS-DMA pieces stay in flight across the barrier
DS, DT = 8, 4                                              # k-steps of 16 columns, 32-column blocks of dQ (128 wide)
MFMA = "v_mfma_f32_32x32x16_bf16"
CVT = "v_cvt_pk_bf16_f32"
NG = 48                                                    # MFMAs per tile: GI(0) 0..15, GI(1) 16..31, GII(0) 32..39, GII(1) 40..47

PARSED = {"s0": "S0", "p0": "P0", "s1": "S1", "p1": "P1", "f0": "F0", "f1": "F1", "f2": "F2", "f3": "F3", "ka": "KA"}
KADDR = {0: "%[kaddr]", 1: "v[KA+0]", 2: "v[KA+1]", 3: "v[KA+2]", 4: "v[KA+3]", 5: "%[ka5]", 6: "%[ka6]", 7: "%[ka7]"}


def parse_block(op, sym):
    return [f".set {sym}, 0", ".set _tfa_pd, 0", f'.irpc c, "%[{op}]"', ".ifc \\c, :", ".set _tfa_pd, 1", ".endif", ".if _tfa_pd == 0",
            ".irp d,0,1,2,3,4,5,6,7,8,9", ".ifc \\c, \\d", f".set {sym}, {sym}*10+\\d", ".endif", ".endr", ".endif", ".endr"]


def frag(k, n=4, sub=0):
    return f"%[f{k % 4}]" if (n == 4 and sub == 0) else f"v[F{k % 4}+{sub}:F{k % 4}+{sub + n - 1}]"


def frag_reads(g, par):
    if g < 32:
        t, sl, which = g >> 4, (g & 15) >> 1, g & 1
        off = par * 0x8000 + which * 0x4000 + t * 0x2000
        return [f"ds_read_b128 {frag(g)}, {KADDR[sl]} offset:{off}"]
    i = g - 32
    off = par * 0x4000 + (i // DT) * (2 * DT * 512) + (i % DT) * 512
    return [f"ds_read_b64_tr_b16 {frag(g, 2, 0)}, %[vat] offset:{off}", f"ds_read_b64_tr_b16 {frag(g, 2, 2)}, %[vat] offset:{off + 256}"]


def R(sym, e, n=1):
    return f"v[{sym}+{e}]" if n == 1 else f"v[{sym}+{e}:{sym}+{e + n - 1}]"


def ew_slots(mask=False):
    """slot (MFMA index of the tile) -> element-wise instructions issued behind that MFMA"""
    sl = {g: [] for g in range(NG)}
    for t, (base, first) in enumerate(((16, lambda e: 1 + e * 13 // 16), (32, lambda e: (1 + e * 4 // 8) if e < 8 else (3 + (e - 8) * 6 // 8)))):
        s_, p_ = f"S{t}", f"P{t}"
        for e in range(16):
            g0 = base + first(e)
            if mask:                                       # key offset ko of element e inside the tile; the lane's row sees keys up to %[lim]
                ko = 32 * t + (e & 3) + 8 * (e >> 2)
                sl[g0].append(f"v_cmp_le_i32 %[msk], {ko}, %[lim]")
                sl[g0].append(f"v_cndmask_b32 {R(s_, e)}, %[ninf], {R(s_, e)}, %[msk]")
            sl[g0].append(f"v_fma_f32 {R(s_, e)}, {R(s_, e)}, %[sc], -%[l2]")
            sl[g0 + 1].append(f"v_exp_f32 {R(s_, e)}, {R(s_, e)}")
            sl[g0 + 2].append(f"v_mul_f32 {R(s_, e)}, {R(s_, e)}, {R(p_, e)}")
            if e & 1:
                s, k = e >> 3, (e & 7) >> 1
                sl[g0 + 2].append(f"{CVT} {R(s_, 8 * s + k)}, {R(s_, e - 1)}, {R(s_, e)}")
    return sl


def body(par, mask=False):
    """one tile.  mask: a diagonal tile of the wave — S becomes -inf where the key lies behind the lane's row (two VALU per element in front of its scale /
    subtract; the compare result in an SGPR pair: vcc carries the tile's "tile u + 1 exists" flag for the LDS-DMA pieces)"""
    o = []
    a = o.append
    a(f"; ---- tile of stage {par}" + (" (masked)" if mask else ""))
    ew = ew_slots(mask)
    tag = f"dq_{'m' if mask else 'b'}{par}"
    a("s_add_u32 %[ts], %[u], 1")                          # the block's last tile requests nothing: vcc = (u + 1 < nu), every request branches on it
    a("s_cmp_lt_i32 %[ts], %[nu]")
    a("s_cselect_b64 vcc, -1, 0")
    dma = []                                               # (m0 immediate, source operand, descriptor, scalar offset) of the six pieces of tile u + 1 -> stage par ^ 1
    for img, (src, rs, so) in enumerate((("ks", "krs", "koff"), ("vs", "vrs", "voff"), ("ts", "krs", "koff"))):
        for i in range(2):
            base = (2 * (par ^ 1) + img) * 0x4000 if img < 2 else 0x10000 + (par ^ 1) * 0x4000
            dma.append((base + i * 1024, f"%[{src}{i}]", f"%[{rs}]", f"%[{so}]"))
    for k in (0, 1):
        o.extend(frag_reads(k, par))
    post = []
    for g in range(NG):
        if g % 2 == 0:
            cnt = 0
            if g + 2 < NG:
                rs = frag_reads(g + 2, par)
                o.extend(rs)
                cnt = len(rs)
            if "nolds" not in ABL:
                a(f"s_waitcnt lgkmcnt({cnt})")
            post = frag_reads(g + 3, par) if g + 3 < NG else []
        if g < len(dma):
            a(f"s_add_u32 m0, %[ldsw], {dma[g][0]}")       # (the MFMA behind it is the wait state an M0 write needs in front of the load that reads it)
        if g < 32:
            t, sl, which = g >> 4, (g & 15) >> 1, g & 1
            dst = f"%[{'sp'[which]}{t}]"
            c0 = "%[dinit]" if which else "0"               # dP starts at -delta of the lane's row (sixteen registers that never change): no subtraction per element
            a(f"{MFMA} {dst}, {frag(g)}, %[{'qd'[which]}{sl}], {c0 if sl == 0 else dst}")
        else:
            i = g - 32
            slot, d = i // DT, i % DT                      # 16-key slot 0..3 of the tile, 32-column block of dQ
            real = [l for l in o if not (l.startswith(";") or l.endswith(":"))]
            since = next((k for k, l in enumerate(reversed(real)) if l.startswith(CVT)), 99)
            if since < 2:                                  # a pack that wrote this MFMA's B operand needs two instructions in front of the MFMA
                a(f"s_nop {1 - since}")
            a(f"{MFMA} %[acc{d}], {frag(g)}, {R('S' + str(slot >> 1), 8 * (slot & 1), 4)}, %[acc{d}]")
        if g % 2 == 0:
            o.extend(post)
        if g < len(dma) and "nodma" not in ABL:
            a(f"s_cbranch_vccz {tag}nd{g}%=")
            a(f"buffer_load_dwordx4 {dma[g][1]}, {dma[g][2]}, {dma[g][3]} offen lds")
            a(f"{tag}nd{g}%=:")
        if "novalu" not in ABL:
            o.extend(ew[g])
    a("s_waitcnt vmcnt(6) lgkmcnt(0)" if ABL_NOWAIT else "s_waitcnt vmcnt(0) lgkmcnt(0)")
    if "nobar" not in ABL:
        a("s_barrier")
    a("s_add_u32 %[u], %[u], 1")
    a("s_add_u32 %[koff], %[koff], %[kstr]")
    a("s_add_u32 %[voff], %[voff], %[vstr]")
    if mask:
        a("v_subrev_u32 %[lim], 64, %[lim]")               # the next tile's keys are 64 further on
        a("s_cmp_ge_i32 %[u], %[mend]")
        a("s_cbranch_scc1 dq_exit%=")
        a(f"s_branch dq_m{par ^ 1}%=")
    else:
        a("s_cmp_ge_i32 %[u], %[uend]")
        a(f"s_cbranch_scc1 dq_d{par ^ 1}%=")               # the unmasked range ends here: on to the masked tiles (or out)
    return o


def build(dtype):
    global MFMA, CVT
    MFMA = "v_mfma_f32_32x32x16_bf16" if dtype == "bf16" else "v_mfma_f32_32x32x16_f16"
    CVT = "v_cvt_pk_bf16_f32" if dtype == "bf16" else "v_cvt_pk_f16_f32"
    lines = []
    for op, sym in PARSED.items():
        lines.extend(parse_block(op, sym))
    for s in range(1, DS):
        lines.append(f"v_xor_b32 {KADDR[s]}, {s << 5}, %[kaddr]")
    a = lines.append
    a("v_mov_b32 %[ninf], 0xff800000")
    a("s_cmp_ge_i32 %[u], %[uend]")                        # (entered at tile 0: no unmasked tile at all -> the masked ones)
    a("s_cbranch_scc1 dq_d0%=")
    a("dq_loop%=:")
    lines.extend(body(0))
    lines.extend(body(1))
    a("s_branch dq_loop%=")
    for par in (0, 1):                                     # behind the unmasked range: masked tiles while u < mend, alternating stages
        a(f"dq_d{par}%=:")
        a("s_cmp_ge_i32 %[u], %[mend]")
        a("s_cbranch_scc1 dq_exit%=")
        a(f"s_branch dq_m{par}%=")
    for par in (0, 1):
        a(f"dq_m{par}%=:")
        lines.extend(body(par, mask=True))
    a("dq_exit%=:")
    return lines, sum(1 for l in body(0) if not l.startswith(";"))


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


def main():
    lb, n = build("bf16")
    lh, _ = build("f16")
    out = ["// tfa_bwd_dq_asm_loop.inc — GENERATED by tools/gen_bwd_dq_asm_loop.py (do not edit; re-generate).  The unmasked tiles of the backward's dQ launch",
           f"// (bwd_kernel<BWD_DQ>, 128 wide, 8 waves) as hand-scheduled gfx950 assembly: ONE basic block of {n} instructions per 64-key tile (48 MFMA, 112 VALU,",
           "// 64 LDS reads, 6 LDS-DMA).  Layout, schedule and the register rules: the generator's docstring."]
    out.extend(emit("TFA_BWD_DQ_ASM_LOOP", lb, n, "bf16"))
    out.extend(emit("TFA_BWD_DQ_ASM_LOOP_F16", lh, n, "fp16"))
    print("\n".join(out))


if __name__ == "__main__":
    main()
