#!/usr/bin/env python3
"""Generates tiny-flash-attention_amd/csrc/tfa_bwd_kv_asm_loop.inc: the unmasked iterations of the backward's fused dK/dV launch (bwd_kv_kernel, 128 wide, four key
groups x two roles) as hand-scheduled gfx950 assembly — the dQ launch's generator (tools/gen_bwd_dq_asm_loop.py) applied to the two-role kernel.

    python tools/gen_bwd_kv_asm_loop.py > tiny-flash-attention_amd/csrc/tfa_bwd_kv_asm_loop.inc

ONE statement carries both roles (a scalar branch on %[role] at its head: one register assignment for the kernel) and, per role, three bodies — iteration `it` runs
body it mod 3 (three tile stages: the stage of tile it is it mod 3), entered at any phase (%[ph]) and left at any iteration (%[it1]):
  role 0 (tile it,     stage b):        x_t = Q_t K^T -> P = exp2(x c - lse2) -> 16 bit IN PLACE -> P buffer b (ds_write) ; dV^T += dO_t^T P
  role 1 (tile it - 1, stage (b+2)%3):  x_t = dO_t V^T -> dS = P (x - delta), P from P buffer (b+2)%3          ; dK^T += Q_t^T dS
Per half t of 32 query rows: GI(t) 8 MFMAs (A: ds_read_b128 rows of the unified row-major image, XOR swizzle u_swz), EW(t) 56 VALU, GII(t) 8 MFMAs (A: two
ds_read_b64_tr_b16 of the same image).  Order: GI(0) | GI(1) + EW(0) | GII(0) + EW(1) | GII(1) + the rest.  Fragments: four buffers in rotation, requested two MFMAs
ahead; every s_waitcnt lgkmcnt count comes from a model of the in-order LDS queue (statistics, P reads and P writes travel in the same queue).
The first four MFMA slots carry the wave's LDS-DMA pieces of tile it + 1 (stage (b+1)%3), waves 0 and 1 also its row statistics (LSE, delta: one dword per lane).

LDS map (bytes; tfa_bwd_kv_kernel.h, the tensor-major map of this instantiation): Q images at s * 0x4000, dO images at 0xc000 + s * 0x4000 — a role's image base
is part of its address registers, so every fragment offset fits the 16-bit immediate —, P buffers at 0x18000 + p * 0x4000 + key group * 0x1000, statistics at
0x24000 + s * 512 (+ 256: delta).
"""
import os
import sys

ABL = os.environ.get("TFA_GEN_KV_ABL", "").split(",")
DS, DT = 8, 4
MFMA = "v_mfma_f32_32x32x16_bf16"
CVT = "v_cvt_pk_bf16_f32"
BF = True
NG = 32                                                    # MFMAs per tile and role: GI(0) 0..7, GI(1) 8..15, GII(0) 16..23, GII(1) 24..31

PARSED = {"x0": "X0", "x1": "X1", "f0": "F0", "f1": "F1", "f2": "F2", "f3": "F3", "ka": "KA", "t1": "T1", "t2": "T2",
          "st": "ST", "tm0": "TM0", "pp0": "PP0", "pp1": "PP1", "pp2": "PP2", "pp3": "PP3"}


def parse_block(op, sym):
    return [f".set {sym}, 0", ".set _tfa_pd, 0", f'.irpc c, "%[{op}]"', ".ifc \\c, :", ".set _tfa_pd, 1", ".endif", ".if _tfa_pd == 0",
            ".irp d,0,1,2,3,4,5,6,7,8,9", ".ifc \\c, \\d", f".set {sym}, {sym}*10+\\d", ".endif", ".endr", ".endif", ".endr"]


def kaddr(sl):
    """address register of k-step sl"""
    return {0: "%[kaddr]", 1: "v[KA+0]", 2: "v[KA+1]", 3: "v[KA+2]", 4: "v[KA+3]", 5: "%[ka5]", 6: "%[ka6]", 7: "%[ka7]"}[sl]


def R(sym, e, n=1):
    return f"v[{sym}+{e}]" if n == 1 else f"v[{sym}+{e}:{sym}+{e + n - 1}]"


class Q:
    """the wave's in-order LDS queue: every ds instruction is pushed with a tag; wait(tag) emits the s_waitcnt that leaves only younger operations outstanding"""

    def __init__(self, out):
        self.q, self.o = [], out

    def push(self, tag, line):
        self.q.append(tag)
        self.o.append(line)

    def wait(self, tag):
        if tag not in self.q:
            return
        last = max(i for i, t in enumerate(self.q) if t == tag)
        n = len(self.q) - 1 - last
        if "nolds" not in ABL:
            self.o.append(f"s_waitcnt lgkmcnt({min(n, 15)})")
        self.q = self.q[last + 1:] if n <= 15 else []


def frag_reads(g, role, st):
    """the LDS reads of fragment g (tile in stage st) into buffer g % 4.  The role's image base is part of the address registers (tensor-major map: the
    three images of a tensor are 0x4000 apart), so every offset fits the 16-bit immediate"""
    b = g % 4
    if g < 16:
        t, sl = g >> 3, g & 7
        return [f"ds_read_b128 %[f{b}], {kaddr(sl)} offset:{st * 0x4000 + t * 0x2000}"]
    i = g - 16
    slot, d = i // DT, i % DT
    off = st * 0x4000 + slot * 0x1000
    return [f"ds_read_b64_tr_b16 v[F{b}+0:F{b}+1], v[T1+{d}] offset:{off}", f"ds_read_b64_tr_b16 v[F{b}+2:F{b}+3], v[T2+{d}] offset:{off}"]


def ew_plan(role, mask=False):
    """slot -> the element-wise work of both halves behind the MFMAs.  Three steps per element: A (the only reader of the row statistics: scale + subtract, or
    subtract) early — the sixteen statistics registers serve both halves, half 1's are read once half 0's step A is through —, B (exp2 / the 16-bit P unpacked
    into one of eight rotating temporaries) and C (the product; the pack of a pair, the P hand-off of a complete slot) spread over the slots up to their MFMA"""
    sl = {g: [] for g in range(NG + 1)}
    firsts = (lambda e: 9 + e * 3 // 8 if e < 8 else 11 + (e - 8) * 5 // 8, lambda e: 17 + e * 4 // 8 if e < 8 else 20 + (e - 8) * 5 // 8)
    for t in (0, 1):
        X = f"X{t}"
        for e in range(16):
            g0 = firsts[t](e)
            gA = (9 if t == 0 else 17) + e // 4
            assert gA <= g0
            if role == 0:
                if mask:                                   # query offset qo of element e inside the tile; the lane's key is hidden from queries below %[lim]
                    qo = 32 * t + (e & 3) + 8 * (e >> 2)
                    sl[gA].append(f"v_cmp_ge_i32 %[msk], {qo}, %[lim]")
                    sl[gA].append(f"v_cndmask_b32 {R(X, e)}, %[ninf], {R(X, e)}, %[msk]")
                sl[gA].append(f"v_mul_f32 {R('ST', e)}, 0x3fb8aa3b, {R('ST', e)}")
                sl[gA].append(f"v_fma_f32 {R(X, e)}, {R(X, e)}, %[sc], -{R('ST', e)}")
                sl[g0 + 1].append(f"v_exp_f32 {R(X, e)}, {R(X, e)}")
            else:
                P = f"PP{2 * t + (e >> 3)}"
                pw = R(P, (e & 7) >> 1)
                # the unpacked P: half 0 in four rotating temporaries (element e + 4 unpacks in the slot element e's product is formed, behind it), half 1 in
                # the registers half 0's P arrived in (dead by then)
                tm = R("TM0", e & 3) if t == 0 else R("PP0" if (e & 7) < 4 else "PP1", e & 3)
                if t == 0 and e + 4 < 16:
                    assert firsts[0](e + 4) >= g0 + 1
                sl[gA].append(f"v_sub_f32 {R(X, e)}, {R(X, e)}, {R('ST', e)}")
                if BF:
                    sl[g0 + 1].append(f"v_and_b32 {tm}, 0xffff0000, {pw}" if e & 1 else f"v_lshlrev_b32 {tm}, 16, {pw}")
                else:
                    if e & 1:
                        sl[g0 + 1].append(f"v_lshrrev_b32 {tm}, 16, {pw}")
                        sl[g0 + 1].append(f"v_cvt_f32_f16 {tm}, {tm}")
                    else:
                        sl[g0 + 1].append(f"v_cvt_f32_f16 {tm}, {pw}")
                sl[g0 + 2].append(f"v_mul_f32 {R(X, e)}, {tm}, {R(X, e)}")
            if e & 1:
                s, k = e >> 3, (e & 7) >> 1
                sl[g0 + 2].append(f"{CVT} {R(X, 8 * s + k)}, {R(X, e - 1)}, {R(X, e)}")
                if role == 0 and (e & 7) == 7:             # the slot's four registers are complete: hand them to role 1
                    sl[g0 + 2].append(("pwrite", t, s))
    return sl


def body(role, b, mask=False):
    """iteration body of phase b = it mod 3.  mask (role 0): the wave's diagonal tiles — S is set to -inf where the query lies before the lane's key (two VALU per
    element in front of its scale / subtract; the compare result in an SGPR pair: vcc carries the tile's "tile it + 1 exists" flag for the LDS-DMA pieces)"""
    o = []
    a = o.append
    st = b if role == 0 else (b + 2) % 3                   # stage (and P buffer) of this wave's tile
    sn = (b + 1) % 3                                       # stage of tile it + 1: the DMA's target
    a(f"; ---- role {role}, phase {b}: tile in stage {st}, requests into stage {sn}")
    q = Q(o)
    ew = ew_plan(role, mask)
    tag = f"kv_r{role}{'m' if mask else 'b'}{b}"
    # the last iterations request nothing (tile it + 1 does not exist): vcc = (it + 1 < nu), every request branches on it
    a("s_add_u32 %[ts], %[it], 1")
    a("s_cmp_lt_i32 %[ts], %[nu]")
    a("s_cselect_b64 vcc, -1, 0")
    dma = [((3 * img + sn) * 0x4000 + i * 1024, f"%[{'qd'[img]}s{i}]", f"%[{'qd'[img]}rs]", f"%[{'qd'[img]}off]") for img in (0, 1) for i in (0, 1)]
    for k in (0, 1):
        for l in frag_reads(k, role, st):
            q.push(("f", k), l)
    for g in range(NG):
        if g % 2 == 0:
            if g + 2 < NG:
                for l in frag_reads(g + 2, role, st):
                    q.push(("f", g + 2), l)
            q.wait(("f", g + 1))
        if g < len(dma):
            a(f"s_add_u32 m0, %[ldsw], {dma[g][0]}")
        if g < 16:
            t, sl = g >> 3, g & 7
            a(f"{MFMA} %[x{t}], %[f{g % 4}], %[r{sl}], {'0' if sl == 0 else f'%[x{t}]'}")
        else:
            i = g - 16
            slot, d = i // DT, i % DT
            real = [l for l in o if not (l.startswith(";") or l.endswith(":"))]
            since = next((k for k, l in enumerate(reversed(real)) if l.startswith(CVT)), 99)
            if since < 2:
                a(f"s_nop {1 - since}")
            a(f"{MFMA} %[acc{d}], %[f{g % 4}], {R('X' + str(slot >> 1), 8 * (slot & 1), 4)}, %[acc{d}]")
        if g % 2 == 0 and g + 3 < NG:
            for l in frag_reads(g + 3, role, st):
                q.push(("f", g + 3), l)
        if g < len(dma) and "nodma" not in ABL:
            a(f"s_cbranch_vccz {tag}nd{g}%=")
            a(f"buffer_load_dwordx4 {dma[g][1]}, {dma[g][2]}, {dma[g][3]} offen lds")
            a(f"{tag}nd{g}%=:")
        if g in (1, 13):
            # the tile's row statistics (this lane's 16 rows of a half: four runs of four): half 0's behind the second MFMA, half 1's once half 0's were used
            t = 0 if g == 1 else 1
            for g4 in range(4):
                q.push(("st", t), f"ds_read_b128 {R('ST', 4 * g4, 4)}, %[sta] offset:{st * 512 + t * 128 + g4 * 32}")
        if g == 1:
            if role == 1:                                  # ... and P of both halves
                for j in range(4):
                    q.push(("pp", j >> 1), f"ds_read_b128 %[pp{j}], %[pxa] offset:{st * 0x4000 + j * 1024}")
        if g == 4 and role == 0:
            # waves 0 and 1 also fetch the statistics of tile it + 1 (LSE / delta: one dword per lane)
            a("s_cmp_eq_u32 %[stq], 0")
            a(f"s_cbranch_scc1 {tag}ns%=")
            a(f"s_cbranch_vccz {tag}ns%=")
            a(f"s_add_u32 m0, %[ldsst], {sn * 512}")
            a("s_nop 0")
            a("v_mbcnt_lo_u32_b32 v[TM0+0], -1, 0")        # (lane * 4, in a temporary that is idle this early in the tile)
            a("v_mbcnt_hi_u32_b32 v[TM0+0], -1, v[TM0+0]")
            a("v_lshlrev_b32 v[TM0+0], 2, v[TM0+0]")
            a("buffer_load_dword v[TM0+0], %[strs], %[stoff] offen lds")
            a(f"{tag}ns%=:")
        for item in ew[g]:
            if "novalu" in ABL:
                continue
            if isinstance(item, tuple):
                _, t, s = item
                q.push(("pw",), f"ds_write_b128 %[pxa], {R(f'X{t}', 8 * s, 4)} offset:{st * 0x4000 + (2 * t + s) * 1024}")
                continue
            # the first use of a half's statistics / P waits for them (the queue model knows what is still in flight)
            for t in (0, 1):
                if "ST+" in item and ("st", t) in q.q and g >= (9, 17)[t]:
                    q.wait(("st", t))
                for j in (0, 1):
                    if f"PP{2 * t + j}+" in item and ("pp", t) in q.q:
                        q.wait(("pp", t))
            a(item)
    a("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if "nobar" not in ABL:
        a("s_barrier")
    a("s_add_u32 %[it], %[it], 1")
    a("s_add_u32 %[qoff], %[qoff], %[qstr]")
    a("s_add_u32 %[doff], %[doff], %[dstr]")
    a("s_add_u32 %[stoff], %[stoff], 256")
    if mask:
        a("v_subrev_u32 %[lim], 64, %[lim]")               # the next tile's queries are 64 further on
    a("s_cmp_ge_i32 %[it], %[it1]")
    a("s_cbranch_scc1 kv_exit%=")
    if mask:                                               # on to the next phase: its masked body while the wave's tiles still cross the diagonal
        a("s_cmp_ge_i32 %[it], %[um]")
        a(f"s_cbranch_scc1 kv_r0b{(b + 1) % 3}%=")
        a(f"s_branch kv_r0m{(b + 1) % 3}%=")
    return o


def build(dtype):
    global MFMA, CVT, BF
    BF = dtype == "bf16"
    MFMA = "v_mfma_f32_32x32x16_bf16" if BF else "v_mfma_f32_32x32x16_f16"
    CVT = "v_cvt_pk_bf16_f32" if BF else "v_cvt_pk_f16_f32"
    lines = []
    for op, sym in PARSED.items():
        lines.extend(parse_block(op, sym))
    a = lines.append
    for s in range(1, DS):
        a(f"v_xor_b32 {kaddr(s)}, {s << 5}, %[kaddr]")
    for d in range(DT):
        a(f"v_xor_b32 v[T1+{d}], {d << 6}, %[ta1]")
        a(f"v_xor_b32 v[T2+{d}], {d << 6}, %[ta2]")
    a("v_mov_b32 %[ninf], 0xff800000")
    a("s_cmp_eq_u32 %[role], 1")
    a("s_cbranch_scc1 kv_r1%=")
    for role in (0, 1):
        if role:
            a("kv_r1%=:")
        else:                                              # role 0 starts in its masked bodies while it < um
            a("s_cmp_ge_i32 %[it], %[um]")
            a("s_cbranch_scc1 kv_r0u%=")
            a("s_cmp_eq_u32 %[ph], 1")
            a("s_cbranch_scc1 kv_r0m1%=")
            a("s_cmp_eq_u32 %[ph], 2")
            a("s_cbranch_scc1 kv_r0m2%=")
            for b in range(3):
                a(f"kv_r0m{b}%=:")
                lines.extend(body(0, b, mask=True))
            a("kv_r0u%=:")
        a("s_cmp_eq_u32 %[ph], 1")
        a(f"s_cbranch_scc1 kv_r{role}b1%=")
        a("s_cmp_eq_u32 %[ph], 2")
        a(f"s_cbranch_scc1 kv_r{role}b2%=")
        for b in range(3):
            a(f"kv_r{role}b{b}%=:")
            lines.extend(body(role, b))
        a(f"s_branch kv_r{role}b0%=")
    a("kv_exit%=:")
    n = [sum(1 for l in body(r, 0) if not l.startswith(";") and not l.endswith(":")) for r in (0, 1)]
    return lines, n


def emit(name, lines, n, what):
    out = [f"#define {name} \\"]
    for l in lines:
        if l.startswith(";"):
            continue
        esc = l.replace("\\", "\\\\").replace('"', '\\"')
        out.append(f'  "{esc}\\n\\t" \\')
    out.append('  ""')
    out.append(f"#define {name}_INSTR_PER_TILE_ROLE0 {n[0]}    // {what}")
    out.append(f"#define {name}_INSTR_PER_TILE_ROLE1 {n[1]}")
    return out


def main():
    lb, n = build("bf16")
    lh, nh = build("f16")
    out = ["// tfa_bwd_kv_asm_loop.inc — GENERATED by tools/gen_bwd_kv_asm_loop.py (do not edit; re-generate).  The unmasked iterations of the fused dK/dV launch",
           f"// (bwd_kv_kernel, 128 wide, four key groups x two roles) as hand-scheduled gfx950 assembly: {n[0]} (role 0) / {n[1]} (role 1) instructions per tile and wave",
           "// for its 32 MFMAs.  Layout, schedule and the register rules: the generator's docstring."]
    out.extend(emit("TFA_BWD_KV_ASM_LOOP", lb, n, "bf16"))
    out.extend(emit("TFA_BWD_KV_ASM_LOOP_F16", lh, nh, "fp16"))
    print("\n".join(out))


if __name__ == "__main__":
    main()
