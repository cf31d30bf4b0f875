// tfa_bwd_kv_kernel.h — dK and dV in ONE launch: S and dP are computed once each (4 GEMM units where the two
// single-gradient launches of tfa_bwd_kernel.h execute 5: S twice).  gfx950.
//
// A wave that owned both accumulator sets would need 128 accumulator + 64 resident-fragment registers before any working
// state — more than the 256 a wave has with two waves per SIMD.  So the work of one block of resident keys is split by ROLE
// between the two waves that share a SIMD (waves w and w+4 of the workgroup, same 32 keys):
//     role 0 (waves 0-3):  S^T = Qtile . K^T   ->  P = exp2(S*scale - LSE)  ->  dV^T += dOtile^T . P        resident: K rows
//     role 1 (waves 4-7):  dP^T = dOtile . V^T ->  dS = P o (dP - delta)    ->  dK^T += Qtile^T . dS        resident: V rows
// P goes from role 0 to role 1 through LDS as the 16-bit values the dV product consumes, in the lane-private register layout
// (same lane = same key in both waves: a straight copy, no transposition).  Role 1 runs ONE TILE BEHIND role 0, so the copy is
// ordered by the tile loop's one barrier and neither wave waits for the other inside a tile:
//     iteration it:   role 0 works on tile it (writes P(it) into exchange buffer it&1),
//                     role 1 works on tile it-1 (reads P(it-1) from buffer (it-1)&1);  LDS-DMA fills tile it+1.
// Three tile stages of two images each (Q and dO, row-major with the u_swz XOR swizzle of tfa_bwd_kernel.h: one image
// serves the b128 row reads of GEMM-I and the transpose reads of GEMM-II), KG key groups x 2 roles per workgroup (KG = 6: 192
// resident keys, twelve waves, three per SIMD — the default; KG = 4: eight waves, the workspace form), every layout the forward's
// (verified on hardware).  Deterministic: no atomics, fixed summation order.
// GQA: the streamed sequence runs over the G query heads of the K/V head, as in tfa_bwd_kernel.h.
#pragma once
#include "tfa_bwd_kernel.h"
#if defined(TFA_BWD_KV_ASM_INC)
#include TFA_BWD_KV_ASM_INC       // an A/B arm's text (tools/gen_bwd_kv_asm_loop.py with TFA_GEN_KV_* set), never the product build
#else
#include "tfa_bwd_kv_asm_loop.inc"
#endif

#if !defined(TFA_BWD_KV_USE_ASM)
#define TFA_BWD_KV_USE_ASM 1     // 0: the compiler-scheduled iteration body everywhere (the A/B arm of the hand-scheduled iterations, tools/gen_bwd_kv_asm_loop.py)
#endif
// The hand-scheduled unmasked iterations of the fused dK/dV launch: ONE statement for both roles (one register assignment), every operand the compiler's choice.
#define TFA_BWD_KV_ASM_STMT(TEXT) \
  asm volatile(TEXT \
  : [acc0] "+v"(acc[0]), [acc1] "+v"(acc[1]), [acc2] "+v"(acc[2]), [acc3] "+v"(acc[3]), [it] "+s"(a_it), [qoff] "+s"(a_qoff), [doff] "+s"(a_doff), [stoff] "+s"(a_stoff), [lim] "+v"(a_lim), [ts] "=&s"(a_ts), [msk] "=&s"(a_msk), [ninf] "=&v"(a_ninf), \
  [x0] "=&v"(ax0), [x1] "=&v"(ax1), [f0] "=&v"(af0), [f1] "=&v"(af1), [f2] "=&v"(af2), [f3] "=&v"(af3), \
  [ka] "=&v"(aka), [ka5] "=&v"(aka5), [ka6] "=&v"(aka6), [ka7] "=&v"(aka7), \
  [t1] "=&v"(at1), [t2] "=&v"(at2), [st] "=&v"(ast), [tm0] "=&v"(atm0), \
  [pp0] "=&v"(app0), [pp1] "=&v"(app1), [pp2] "=&v"(app2), [pp3] "=&v"(app3) \
  : [r0] "v"(rf[0]), [r1] "v"(rf[1]), [r2] "v"(rf[2]), [r3] "v"(rf[3]), [r4] "v"(rf[4]), [r5] "v"(rf[5]), [r6] "v"(rf[6]), [r7] "v"(rf[7]), \
  [kaddr] "v"(a_kaddr), [ta1] "v"(a_ta1), [ta2] "v"(a_ta2), [pxa] "v"(a_pxa), [sta] "v"(a_sta), \
  [qs0] "v"(src[0][0]), [qs1] "v"(src[0][1]), [ds0] "v"(src[1][0]), [ds1] "v"(src[1][1]), \
  [sc] "s"(a_sc), [qrs] "s"(q_rs), [drs] "s"(do_rs), [strs] "s"(a_strs), [ldsw] "s"(a_ldsw), [ldsst] "s"(a_ldsst), [qstr] "s"(a_qstr), [dstr] "s"(a_dstr), \
  [it1] "s"(a_it1), [ph] "s"(a_ph), [role] "s"(a_role), [stq] "s"(a_stq), [nu] "s"(a_nu), [um] "s"(a_um) \
  : "m0", "vcc", "scc", "memory")

namespace tfa {

// WS: role 1 also writes dS (16 bit, without the softmax scale) to BArgs::ws as dS^T[b, query head][key block][query tile][128
// keys][64 queries] (Nk, Nq rounded up to 128 / 256) — for bwd_dq_ws_kernel (tfa_bwd_dq_kernel.h), which turns it into dQ with ONE
// GEMM instead of recomputing S and dP.  Every (128-key block, 64-query tile) pair this kernel visits is written completely
// (fully masked 32-key pieces as zeros); pairs it does not visit are entirely above the causal diagonal and the consumer
// never reads them.
// BIG: (b,h) slices of 2 GiB and more — windowed descriptors as in bwd_kernel (one per streamed tile and image, one over the
// workgroup's resident rows, one over its gradient rows); its own instantiation, without the workspace form.
// DVB: 32-wide column blocks that can hold valid head-dim columns (tfa_bwd_kernel.h): 3 for head dims up to 96 in the 128-wide kernel, 1 for
// head dims up to 32 in the 64-wide one; the GEMM loops, resident fragments and LDS reads skip the zero columns.
template <typename T, int D, bool CAUSAL, bool F32OUT, bool WS = false, int KG = 4, bool BIG = false, int DVB = D / 32>
__global__ __launch_bounds__(KG * 128, KG == 6 ? 3 : 2) void bwd_kv_kernel(const BArgs p) {
  static_assert(!(BIG && WS), "the workspace form keeps one descriptor per slice");
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr int NW = 2 * KG;                       // KG key groups x 2 roles (KG = 6: three waves per SIMD, 168 registers each)
  constexpr int NDMA = 8;                          // waves that issue LDS-DMA pieces
  static_assert(KG == 4 || (KG == 6 && !WS), "the workspace layout is blocked by 128 keys");
  constexpr int BMK = KG * 32;                     // resident keys per workgroup
  constexpr int BN = 64;                           // streamed query rows per tile
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int PIECES = TILE_BYTES / 1024;
  constexpr int PPW = PIECES / NDMA;
  static_assert(DVB >= 1 && DVB <= D / 32, "valid 32-column blocks of a D-wide kernel");
  constexpr int DS = 2 * DVB;
  constexpr int DT = DVB;
  constexpr int NSTAGE = 3;
  constexpr int STAGE_BYTES = 2 * TILE_BYTES;      // image 0: Q tile, image 1: dO tile
  constexpr int PX_BYTES = 32 * BN * 2;            // one key group's P tile, 16 bit
  constexpr int NPX = KG == 4 ? 3 : 2;             // P exchange buffers: one per tile stage (indexed like the stages: the hand-scheduled iterations' three bodies), or by tile parity
  // the hand-scheduled unmasked iterations (tfa_bwd_kv_asm_loop.inc): the 128-wide eight-wave launch, no workspace, no windows, all four column blocks
  constexpr bool ASMKV = TFA_BWD_KV_USE_ASM != 0 && D == 128 && KG == 4 && !WS && !BIG && DVB == 4;
  static_assert(PPW >= 1 && PPW * NDMA == PIECES, "");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  // image offsets: stage-major (Q, dO of a stage side by side), or — next to the hand-scheduled iterations — tensor-major (the three Q images, then the three
  // dO images): every fragment offset of a role then fits the 16-bit immediate of its LDS reads (the role's image base sits in the address registers)
  auto q_img = [](int stage) -> int { return ASMKV ? stage * TILE_BYTES : stage * STAGE_BYTES; };
  auto do_img = [](int stage) -> int { return ASMKV ? (NSTAGE + stage) * TILE_BYTES : stage * STAGE_BYTES + TILE_BYTES; };
  // the resident rows' landing slices (the prologue below): images tile 0 does not use — stages 1 and 2
  auto slice_off = [](int w) -> int { return ASMKV ? (w < 4 ? TILE_BYTES + w * (32 * D * 2) : (NSTAGE + 1) * TILE_BYTES + (w - 4) * (32 * D * 2)) : STAGE_BYTES + w * (32 * D * 2); };
  char* const pbuf = smem + NSTAGE * STAGE_BYTES;  // [NPX][KG][PX_BYTES]
  constexpr int ST_OFF = NSTAGE * STAGE_BYTES + NPX * KG * PX_BYTES;   // per stage: LSE of the tile's 64 query rows (256 B), delta (256 B)
  const char* const sbuf = smem + ST_OFF;

#if defined(TFA_BWD_TRACE)
  const unsigned long long tw_entry = __builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int kg = wave % KG;                        // key group: rows r0 + 32*kg ..
  const int role = wave / KG;                      // 0: S, P, dV     1: dP, dS, dK
  // The two waves of a SIMD run the burst-structured tile body (GEMM-I, element-wise, GEMM-II) in step behind the tile barrier: matrix bursts collide,
  // element-wise bursts leave the pipe idle.  A static issue priority for the younger wave (role 1 in the fused launch) lets it win the first
  // burst and pulls the two out of step: +0.3 .. +0.8 percent at D = 128 over five shapes, -2.6 percent at D = 64 (profiles/r04b_bwd_priority_ab.txt)
  if (D == 128 && (role == 1)) asm volatile("s_setprio 2" ::: "memory");
  const int qi = lane & 31;
  const int hi = lane >> 5;

  const int G = p.H / p.Hk;
  int rb, bhr;
  {
    const int nbh = p.B * p.Hk, id = blockIdx.x;   // a (b, K/V head) stays on one XCD; causal: the block with the most tiles first
    if ((nbh & 7) == 0) {
      const int x = id & 7, q8 = id >> 3;
      bhr = x + 8 * (q8 / p.nrb);
      rb = q8 % p.nrb;
    } else {
      bhr = id / p.nrb;
      rb = id % p.nrb;
    }
  }
  const int b = bhr / p.Hk;
  const int hr = bhr - b * p.Hk;
  const int shift = p.Nk - p.Nq;
  const int r0 = rb * BMK;
  const int wave_row0 = r0 + kg * 32;
  const int my_row = wave_row0 + qi;

  // ---- streamed tile range of this block (query tiles that see key r0 or later ones) -----------------------------
  int t_begin = 0;
  const int t_end = (p.Nq + BN - 1) / BN;
  if (CAUSAL) {
    const int qmin = r0 - shift;                   // first query row that sees key r0
    t_begin = qmin > 0 ? qmin / BN : 0;
    if (t_begin > t_end) t_begin = t_end;
  }
  const int ntl = t_end - t_begin;                 // tiles per streamed head
  const int nu = ntl * G;                          // tiles of the flat (head-major) sequence

  // ---- per-lane DMA source offsets of the two images ---------------------------------------------------------------
  int src[2][PPW];
  int tile_stride[2];
#pragma unroll
  for (int img = 0; img < 2; ++img) {
    const int sn = (int)(img ? p.dout.s_n : p.q.s_n);
    tile_stride[img] = BN * sn * 2;
#pragma unroll
    for (int i = 0; i < PPW; ++i) {
      const int pc = wave * PPW + i;
      const int row = pc * (1024 / (D * 2)) + lane / CPR;
      const int cpos = lane % CPR;
      const int ch = cpos ^ u_swz<D>(row);
      src[img][i] = ch * 8 < p.dv ? row * sn * 2 + (ch << 4) : (int)TFA_OOB;
    }
  }
  // The streamed (b, query head) changes every ntl tiles (G > 1 only): descriptors live in SGPRs and are rebuilt at head
  // boundaries, tile positions are counted — a division, two 64-bit base computations and two descriptors per tile and wave
  // were 3.8 SALU instructions per MFMA (PMC: profiles/r03_pmc_bwd_kv_cfg3.txt before this change)
  auto head_rsrc = [&](const BTensor& x, int g) {
    const T* base = reinterpret_cast<const T*>(x.p) + b * x.s_b + (hr * G + g) * x.s_h;
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, x.bytes, 0x00020000);
  };
  // per tile-row statistics (LSE for role 0, delta for role 1) travel WITH the tile, by LDS-DMA (one dword per lane = the tile's 64
  // rows; waves 0 and 1 issue them): loaded from global memory by every lane they sat in the same in-order vmcnt queue as the next
  // tile's DMA pieces, so the first use of a statistic waited for that whole tile to land.  Out-of-range rows read as 0.
  auto stat_rsrc = [&](const float* base, int g) {
    const long long so = (long long)(b * p.H + hr * G + g) * p.Nq;
    return __builtin_amdgcn_make_buffer_rsrc((void*)(base + so), 0, (unsigned)p.Nq * 4u, 0x00020000);
  };
  auto lse_rs = stat_rsrc(p.lse, 0), dl_rs = stat_rsrc(p.delta, 0);
  auto q_rs = head_rsrc(p.q, 0), do_rs = head_rsrc(p.dout, 0);
  auto head_base = [&](const BTensor& x, int g) { return reinterpret_cast<const T*>(x.p) + b * x.s_b + (hr * G + g) * x.s_h; };
  int jt_d = t_begin, g_d = 0;                     // position of the NEXT tile to request
  auto dma_next = [&](int stage) {
    if (KG != 4 && wave >= NDMA) return;             // (KG = 6: the tile's pieces are issued by the first eight waves)
    if (wave < 2) lds_dma4_m0(wave ? dl_rs : lse_rs, lds_base + ST_OFF + stage * 512 + wave * 256, (jt_d * BN + lane) * 4);
    if constexpr (BIG) {
      const auto qw = rsrc_at(head_base(p.q, g_d), p.q.full, (unsigned long long)jt_d * (unsigned)tile_stride[0]);
      const auto dw = rsrc_at(head_base(p.dout, g_d), p.dout.full, (unsigned long long)jt_d * (unsigned)tile_stride[1]);
#pragma unroll
      for (int i = 0; i < PPW; ++i) lds_dma16_m0_fresh(qw, lds_base + q_img(stage) + (wave * PPW + i) * 1024, src[0][i]);
#pragma unroll
      for (int i = 0; i < PPW; ++i) lds_dma16_m0_fresh(dw, lds_base + do_img(stage) + (wave * PPW + i) * 1024, src[1][i]);
      if (++jt_d == t_end) { jt_d = t_begin; if (++g_d < G) { lse_rs = stat_rsrc(p.lse, g_d); dl_rs = stat_rsrc(p.delta, g_d); } }
      return;
    }
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      lds_dma16_m0(q_rs, lds_base + q_img(stage) + (wave * PPW + i) * 1024, src[0][i] + jt_d * tile_stride[0]);
#pragma unroll
    for (int i = 0; i < PPW; ++i)
      lds_dma16_m0(do_rs, lds_base + do_img(stage) + (wave * PPW + i) * 1024, src[1][i] + jt_d * tile_stride[1]);
    if (++jt_d == t_end) {
      jt_d = t_begin;
      if (++g_d < G) { q_rs = head_rsrc(p.q, g_d); do_rs = head_rsrc(p.dout, g_d); lse_rs = stat_rsrc(p.lse, g_d); dl_rs = stat_rsrc(p.delta, g_d); }
    }
  };

  // ---- resident fragments: role 0 holds its K rows, role 1 its V rows ------------------------------------------------
  X8 rf[DS];
  {
    const BTensor& x = role ? p.v : p.k;
    const T* b1 = reinterpret_cast<const T*>(x.p) + b * x.s_b + hr * x.s_h;
    auto rs1 = BIG ? rsrc_at(b1, x.full, (unsigned long long)r0 * (unsigned long long)x.s_n * 2ull) : __builtin_amdgcn_make_buffer_rsrc((void*)b1, 0, x.bytes, 0x00020000);
    if constexpr (KG == 4) {
      // Through LDS (the forward's first prologue, VF_IL_QLDS): the wave's 32 rows arrive by LDS-DMA as whole 1 KiB pieces — 64 cache lines per wave
      // instead of 256 32-byte segments — in its slice of stages 1 and 2 (idle until iteration 0 requests tile 1, behind the barrier below),
      // chunk position XOR u_swz(row); the fragments are read back once tile 0 has been requested
      constexpr int RPP = 1024 / (D * 2);            // rows per piece
      const unsigned slice = __builtin_amdgcn_readfirstlane(lds_base + slice_off(wave));
      int lanex = lane;                              // (through an empty asm: this one-off address math shares nothing with the tile loop's)
      asm volatile("" : "+v"(lanex));
#pragma unroll
      for (int i = 0; i < 32 / RPP; ++i) {
        const int row = i * RPP + lanex / CPR, cpos = lanex % CPR;
        const int ch = cpos ^ u_swz<D>(row);
        lds_dma16_m0(rs1, slice + i * 1024, ch * 8 < p.dv ? (wave_row0 - (BIG ? r0 : 0) + row) * (int)x.s_n * 2 + (ch << 4) : (int)TFA_OOB);
      }
    } else {
      const int off1 = (my_row - (BIG ? r0 : 0)) * (int)x.s_n * 2 + hi * 16;
#pragma unroll
      for (int s = 0; s < DS; ++s) rf[s] = __builtin_bit_cast(X8, __builtin_amdgcn_raw_buffer_load_b128(rs1, (2 * s + hi) * 8 < p.dv ? off1 + s * 32 : (int)TFA_OOB, 0, 0));
    }
  }

  f32x16 acc[DT];
#pragma unroll
  for (int d = 0; d < DT; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;

  const int k_rd_base = qi * (D * 2);
  const int k_rd_swz = u_swz<D>(qi);
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  // transpose reads of the row-major image: this lane addresses 4 consecutive d (8 bytes) of tile row 16*sl + tr_row (first read)
  // and 16*sl + tr_row + 8 (second read); chunk = 4*dtile + tr_clo, XOR-swizzled by the row (tfa_bwd_kernel.h, UNI)
  const int tr_row = 4 * hi + (i16 >> 2);
  const int tr_clo = 2 * g16 + ((i16 & 3) >> 1);
  const int tr_byte = ((i16 & 3) & 1) * 8;
  const int tr_s1 = u_swz<D>(tr_row), tr_s2 = u_swz<D>(tr_row + 8);
  const int tr_b1 = tr_row * (D * 2) + tr_byte, tr_b2 = (tr_row + 8) * (D * 2) + tr_byte;
  const float sc = p.scale_log2;

  // WS: this lane's 16 dS values of a 32-query half (queries qh + 8*g4 + 4*hi + 0..3, g4 = 0..3) -> dS^T[key my_row][...]
  auto ws_store = [&](int g, int qh, const X8 (&y)[2]) {
    // BLOCKED layout: the 128 keys x 64 queries of one (key block, query tile) pair are 16 KiB contiguous (key rows of 128 bytes),
    // pairs ordered [key block][query tile] — a wave's stores of a tile stay inside one 4 KiB span instead of striding over 32
    // rows 2*Nq bytes apart (a power-of-two stride: measured 1.2 TB/s), and the consumer's [64 keys][64 queries] sub-blocks are
    // contiguous.  The two lanes of a key hold interleaved runs of 4 queries (4*hi + 8*g4 ..): one v_permlane32_swap per dword
    // regroups them into 16 consecutive queries per lane (lower half-wave: queries 0-15 of the half, upper: 16-31), so a lane
    // stores 2 x 16 bytes instead of 4 x 8.
    T* const slab = reinterpret_cast<T*>(p.ws) + (long long)(b * p.H + hr * G + g) * p.ws_nk * p.ws_nq;
    auto w_rs = __builtin_amdgcn_make_buffer_rsrc((void*)slab, 0, (unsigned)((long long)p.ws_nk * p.ws_nq * 2), 0x00020000);
    const int jt = qh >> 6;
    const int off = (((rb * (p.ws_nq >> 6) + jt) * BMK + kg * 32 + qi) * BN + (qh & 63) + 16 * hi) * 2;
    u32x4 A = __builtin_bit_cast(u32x4, y[0]), B = __builtin_bit_cast(u32x4, y[1]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      unsigned a_ = A[i], b_ = B[i];
      asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a_), "+v"(b_));
      A[i] = a_; B[i] = b_;
    }
    const u32x4 w0 = {A[0], A[1], B[0], B[1]}, w1 = {A[2], A[3], B[2], B[3]};
    __builtin_amdgcn_raw_buffer_store_b128(w0, w_rs, off, 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(w1, w_rs, off + 16, 0, 0);
  };

  if (nu > 0) dma_next(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if constexpr (KG == 4) {
    const char* const sl = smem + slice_off(wave);
    int qix = qi, hix = hi;
    asm volatile("" : "+v"(qix), "+v"(hix));
#pragma unroll
    for (int s = 0; s < DS; ++s) rf[s] = __builtin_bit_cast(X8, lds_read_b128(sl, qix * (D * 2) + (((2 * s + hix) ^ u_swz<D>(qix)) << 4)));
  }
#pragma unroll
  for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(rf[s]));
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (every wave has its fragments out of stages 1 and 2: iteration 0 refills them)

  int jt_c = t_begin, g_c = 0;                       // this wave's tile: position inside the head, head
  auto next_tile = [&]() {
    if (++jt_c == t_end) { jt_c = t_begin; ++g_c; }
  };
#if defined(TFA_BWD_TRACE)   // debug build: cycles each wave spends waiting at the end of an iteration (memory, then barrier)
  unsigned long long tw_mem = 0, tw_bar = 0;
  const unsigned long long tw_t0 = __builtin_amdgcn_s_memtime();
#endif
  // Start value of the GEMM-I accumulators.  A role-0 lane whose key lies behind the last one (the padding of a ragged last key block: K and V
  // read as zeros there) starts at -inf, so that its P is 0 — not exp(-LSE), which overflows 16 bits once a row's LSE is below -11 (fp16) and
  // would put inf/NaN into the dS workspace and, through 0 * inf, into that query's dQ.  Sixteen registers that never change: the first MFMA of
  // a half reads them as its C operand where it used to read the inline constant 0 — no instruction is added to the tile body.
  f32x16 xinit;
  {
    const float v0 = (role == 0 && my_row >= p.Nk) ? -INFINITY : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) xinit[r] = v0;
    // (keep it a register array, not re-materialised per use — except next to the hand-scheduled iterations, which need the sixteen registers and leave
    //  the compiler-scheduled body a handful of iterations: there it is one register, copied per half)
    if constexpr (!ASMKV) {
#pragma unroll
      for (int r = 0; r < 16; ++r) asm volatile("" : "+v"(xinit[r]));
    }
  }
  int st_next = 1;                                   // stage of tile it+1
  int st_mine = role ? NSTAGE - 1 : 0;               // stage of this wave's tile (role 1: tile it-1)
  // ---- the hand-scheduled iterations a_it0 .. a_it1 - 1 of this wave, per streamed head: from its first active tile (role 0: it, masked bodies for the diagonal
  //      tiles below a_um; role 1: it - 1, P arrives masked) to the head's last request-free point — every tile requested from inside the statement belongs to the
  //      SAME head (the descriptors change at a head's end: the compiler-scheduled iterations do that) and lies wholly inside the query rows (the LDS-DMA pieces take
  //      the tile's byte offset as the SCALAR offset, outside the descriptor's bounds check).  Every key of the block inside the sequence (the -inf start of a
  //      padded key's S is the compiler-scheduled body's).
  int a_it0 = 0, a_it1 = 0, a_um = 0, a_head = 0;
  int a_ua = 0, a_uml = 0;                           // the wave's first active tile / first tile without a mask, inside a head's tile sequence
  const bool a_ok = ASMKV && r0 + BMK <= p.Nk;
  auto a_range = [&](int h) {                        // the range of streamed head h
    const int base = h * ntl;
    a_it0 = base + a_ua + role;
    a_um = base + a_uml;
    if (p.Nq % BN != 0) { int l = ntl - 1; const int whole = p.Nq / BN - t_begin - 1; l = whole < l ? whole : l; a_it1 = base + l; }
    else a_it1 = h == G - 1 ? nu + role : base + ntl - 1;
    a_it0 = __builtin_amdgcn_readfirstlane(a_it0);
    a_it1 = __builtin_amdgcn_readfirstlane(a_it1);
    a_um = __builtin_amdgcn_readfirstlane(a_um);
  };
  if constexpr (ASMKV) {
    if (a_ok) {
      if (CAUSAL) {
        const int v = wave_row0 - shift;               // the first query that sees the wave's first key
        const int ja = v > 0 ? v / BN : 0;             // first tile with a query that sees a key of the wave
        const int jm = v + 31 > 0 ? (v + 31 + BN - 1) / BN : 0;   // first tile whose 64 queries all see the wave's 32 keys
        a_ua = ja > t_begin ? ja - t_begin : 0;
        a_uml = jm > t_begin ? jm - t_begin : 0;
      }
      a_range(0);
    } else a_it0 = -1;                                 // (never reached)
  }
#pragma nounroll
  for (int it = 0; it <= nu; ++it) {
    if constexpr (ASMKV) {
      if (it == a_it0 && a_it1 <= a_it0) { if (++a_head < G) a_range(a_head); else a_it0 = -1; }   // (nothing of this head for the statement)
      if (it == a_it0 && a_it1 > a_it0) {
        f32x16 ax0, ax1, ast;
        u32x4 af0, af1, af2, af3, aka, at1, at2, app0, app1, app2, app3, atm0;
        unsigned aka5, aka6, aka7;
        // (the role's images: GEMM-I reads rows of Q (role 0) / dO (role 1), GEMM-II the transposed dO (role 0) / Q (role 1))
        const unsigned a_kaddr = lds_base + (role ? do_img(0) : q_img(0)) + (unsigned)k_rd_base + ((unsigned)(hi ^ k_rd_swz) << 4);
        const unsigned a_ta1 = lds_base + (role ? q_img(0) : do_img(0)) + (unsigned)tr_b1 + ((unsigned)(tr_clo ^ tr_s1) << 4);
        const unsigned a_ta2 = lds_base + (role ? q_img(0) : do_img(0)) + (unsigned)tr_b2 + ((unsigned)(tr_clo ^ tr_s2) << 4);
        const unsigned a_pxa = lds_base + NSTAGE * STAGE_BYTES + kg * PX_BYTES + lane * 16;
        const unsigned a_sta = lds_base + ST_OFF + role * 256 + hi * 16;
        const unsigned a_ldsw = __builtin_amdgcn_readfirstlane(lds_base + wave * (PPW * 1024));
        const unsigned a_ldsst = __builtin_amdgcn_readfirstlane(lds_base + ST_OFF + (wave & 1) * 256);
        const int a_qstr = __builtin_amdgcn_readfirstlane(tile_stride[0]), a_dstr = __builtin_amdgcn_readfirstlane(tile_stride[1]);
        const float a_sc = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, sc)));
        const int a_role = __builtin_amdgcn_readfirstlane(role), a_stq = __builtin_amdgcn_readfirstlane(wave < 2 ? 1 : 0), a_ph = __builtin_amdgcn_readfirstlane(it % 3);
        const auto a_strs = wave ? dl_rs : lse_rs;
        int a_it = it, a_qoff = __builtin_amdgcn_readfirstlane(jt_d * tile_stride[0]), a_doff = __builtin_amdgcn_readfirstlane(jt_d * tile_stride[1]);
        int a_stoff = __builtin_amdgcn_readfirstlane(jt_d * (BN * 4));
        const int a_nu = __builtin_amdgcn_readfirstlane(nu);
        int a_lim = my_row - shift - jt_c * BN - 4 * hi;   // role 0's masked bodies: queries of the tile below this offset do not see the lane's key
        int a_ts;
        unsigned long long a_msk;
        float a_ninf;
        if constexpr (std::is_same<T, __bf16>::value) { TFA_BWD_KV_ASM_STMT(TFA_BWD_KV_ASM_LOOP); }
        else { TFA_BWD_KV_ASM_STMT(TFA_BWD_KV_ASM_LOOP_F16); }
        const int n = a_it - it;                       // iterations done: the streams' positions and the stage rotation move with them
        jt_d += n; jt_c += n;
        if (jt_d >= t_end) {                           // the statement requested the head's last tile: on to the next head's descriptors (dma_next's own step)
          jt_d = t_begin;
          if (++g_d < G) { q_rs = head_rsrc(p.q, g_d); do_rs = head_rsrc(p.dout, g_d); lse_rs = stat_rsrc(p.lse, g_d); dl_rs = stat_rsrc(p.delta, g_d); }
        }
        st_next = (st_next + n) % NSTAGE; st_mine = (st_mine + n) % NSTAGE;
        it = a_it;
        if (++a_head < G) a_range(a_head); else a_it0 = -1;
        if (it > nu) break;                            // (role 1 ran the last iteration inside the statement)
      }
    }
    if (it + 1 < nu) dma_next(st_next);              // tile it+1; that stage held tile it-2: role 1 left it at the last barrier
    const int u = it - role;
    if (u >= 0 && u < nu) {
      const int g = g_c, jt = jt_c;
      const int row0 = jt * BN;                      // first query row of the tile
      const char* img_q = smem + q_img(st_mine);
      const char* img_do = smem + do_img(st_mine);
      const char* img1 = role ? img_do : img_q;      // GEMM-I operand rows (role 0: Q, role 1: dO)
      const char* imgt = role ? img_q : img_do;      // GEMM-II transposed operand (role 0: dO, role 1: Q)
      char* const px = pbuf + ((NPX == 3 ? st_mine : (u & 1)) * KG + kg) * PX_BYTES;

      // this wave's 32 keys x 64 queries: anything masked?  everything masked?
      const bool need_mask = CAUSAL && (row0 < wave_row0 + 31 - shift);
      const bool active = !CAUSAL || (row0 + BN - 1 >= wave_row0 - shift);

      bool stored = false;
      if (WS && role == 1 && !active) {              // a fully masked piece of a visited (block, tile) pair: zeros
        X8 z[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int e = 0; e < 8; ++e) z[j][e] = (T)0.f;
        ws_store(g, row0, z);
        ws_store(g, row0 + 32, z);
        stored = true;
      }
      if (active) {
        // (next to the hand-scheduled iterations the fragment-address pieces are re-derived HERE from the lane id, behind an empty asm: a dozen loop-invariant
        //  registers that would otherwise stay live across the statement, which needs them for its own operands; the other instantiations hoist them as before)
        int lane_l = lane;
        if constexpr (ASMKV) asm volatile("" : "+v"(lane_l));
        const int qi_l = lane_l & 31, hi_l = lane_l >> 5;
        const int k_rd_base = qi_l * (D * 2), k_rd_swz = u_swz<D>(qi_l);
        const int i16_l = lane_l & 15, g16_l = (lane_l >> 4) & 1;
        const int tr_row = 4 * hi_l + (i16_l >> 2), tr_clo = 2 * g16_l + ((i16_l & 3) >> 1), tr_byte = ((i16_l & 3) & 1) * 8;
        const int tr_s1 = u_swz<D>(tr_row), tr_s2 = u_swz<D>(tr_row + 8);
        const int tr_b1 = tr_row * (D * 2) + tr_byte, tr_b2 = (tr_row + 8) * (D * 2) + tr_byte;
        // statistics of the 64 tile rows first (both halves): no load is issued between the dS stores below and the barrier
        // statistics of the tile's rows from the stage's LDS copy: this lane's 16 rows of half t are 4 runs of 4 consecutive rows
        float stv[2][16];
        const char* const st_img = sbuf + st_mine * 512 + role * 256;
        auto load_stats = [&](int t) {
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(st_img + (32 * t + 8 * g4 + 4 * hi) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) stv[t][4 * g4 + e] = a[e];
          }
        };
        X8 keep[2][2];                                  // WS: dS of both halves, stored behind the tile's last MFMA
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          // ---- GEMM-I over the 32 tile rows of half t: S (role 0) or dP (role 1) -----------------------------------
          load_stats(t);
          f32x16 x;
#pragma unroll
          for (int r = 0; r < 16; ++r) x[r] = xinit[r];   // 0, or -inf for a role-0 lane whose key lies behind the last one (see xinit)
#pragma unroll
          for (int sl = 0; sl < DS; ++sl) {
            const int off = k_rd_base + t * 32 * (D * 2) + (((2 * sl + hi) ^ k_rd_swz) << 4);
            x = E::mfma(__builtin_bit_cast(X8, lds_read_b128(img1, off)), rf[sl], x);
          }
          X8 pk[2];
          if (role == 0) {
            if (need_mask) {
              const int limq = my_row - shift - row0 - 4 * hi;     // query offsets below this do not see the lane's key
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const int qo = 32 * t + (r & 3) + 8 * (r >> 2);
                if (qo < limq) x[r] = -INFINITY;
              }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) pk[r >> 3][r & 7] = (T)fast_exp2(fmaf(x[r], sc, -stv[t][r] * 1.4426950408889634f));
            // hand P to the role-1 wave of this key group (it reads it in the next iteration)
#pragma unroll
            for (int j = 0; j < 2; ++j) lds_write_b128(px, (t * 2 + j) * 1024 + lane * 16, __builtin_bit_cast(u32x4, pk[j]));
          } else {
            X8 pp[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) pp[j] = __builtin_bit_cast(X8, lds_read_b128(px, (t * 2 + j) * 1024 + lane * 16));
#pragma unroll
            for (int r = 0; r < 16; ++r) pk[r >> 3][r & 7] = (T)((float)pp[r >> 3][r & 7] * (x[r] - stv[t][r]));
            if (WS) { keep[t][0] = pk[0]; keep[t][1] = pk[1]; }
          }
          // ---- GEMM-II for the two 16-row slots of this half: dV^T += dO^T . P  /  dK^T += Q^T . dS ----------------
#pragma unroll
          for (int sl = 2 * t; sl < 2 * t + 2; ++sl)
#pragma unroll
            for (int d = 0; d < DT; ++d) {
              const int c = 4 * d + tr_clo;
              const s16x4 lo = lds_read_tr16_b64(imgt + sl * 16 * (D * 2) + tr_b1 + ((c ^ tr_s1) << 4));
              const s16x4 hh = lds_read_tr16_b64(imgt + sl * 16 * (D * 2) + tr_b2 + ((c ^ tr_s2) << 4));
              const s16x8 vf = __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7);
              acc[d] = E::mfma(__builtin_bit_cast(X8, vf), pk[sl - 2 * t], acc[d]);
            }
        }
        if (WS && role == 1) {
          ws_store(g, row0, keep[0]);
          ws_store(g, row0 + 32, keep[1]);
          stored = true;
        }
      }
      // WS: the 4 dS stores of this tile are the wave's YOUNGEST vector-memory operations (vmcnt counts stores and retires in
      // issue order): leave them in flight across the barrier — everything older, the LDS-DMA pieces of tile it+1 included, is done
      next_tile();
      if (WS && stored) {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        st_next = st_next + 1 == NSTAGE ? 0 : st_next + 1;
        st_mine = st_mine + 1 == NSTAGE ? 0 : st_mine + 1;
        continue;
      }
    }
#if defined(TFA_BWD_TRACE)
    {
      const unsigned long long a0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      const unsigned long long a1 = __builtin_amdgcn_s_memtime();
      asm volatile("s_barrier" ::: "memory");
      const unsigned long long a2 = __builtin_amdgcn_s_memtime();
      tw_mem += a1 - a0; tw_bar += a2 - a1;
    }
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    st_next = st_next + 1 == NSTAGE ? 0 : st_next + 1;
    st_mine = st_mine + 1 == NSTAGE ? 0 : st_mine + 1;
  }

#if defined(TFA_BWD_TRACE)
  const unsigned long long tw_t1 = __builtin_amdgcn_s_memtime();
#endif
  // ---- epilogue: role 0 writes dV, role 1 writes dK * scale.  acc[dt][r] = grad[row my_row][32*dt + (r&3) + 8*(r>>2) + 4*hi] ----
  const float osc = role ? p.scale : 1.f;
  void* const gp = role ? p.grad : p.grad2;
  const long long gsb = role ? p.gs_b : p.g2s_b, gsh = role ? p.gs_h : p.g2s_h;
  const int gsn = (int)(role ? p.gs_n : p.g2s_n);
  const unsigned gbytes = role ? p.g_bytes : p.g2_bytes;
  const unsigned long long gfull = role ? p.g_full : p.g2_full;
  if (F32OUT) {
    float* gb = reinterpret_cast<float*>(gp) + b * gsb + hr * gsh;
    auto g_rs = BIG ? rsrc_at(gb, gfull, (unsigned long long)r0 * (unsigned long long)gsn * 4ull) : __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, gbytes, 0x00020000);
    const int goff = (my_row - (BIG ? r0 : 0)) * gsn * 4 + hi * 16;
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        f32x4 v4 = {acc[d][4 * g4 + 0] * osc, acc[d][4 * g4 + 1] * osc, acc[d][4 * g4 + 2] * osc, acc[d][4 * g4 + 3] * osc};
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), g_rs, d * 32 + g4 * 8 + hi * 4 < p.dv ? goff + (d * 32 + g4 * 8) * 4 : (int)TFA_OOB, 0, 0);
      }
  } else {
    // A lane holds 4-element pieces of ONE gradient row spread over 16 register groups: stored directly that is 16 eight-byte stores per lane,
    // 32 different rows per instruction — 7.5 k cycles behind the tile loop (tools/trace_bwd_kv.py), a twentieth of a workgroup's life.  As in the
    // forward's epilogue (tfa_fwd_il_epilogue.inc) the wave transposes its 32 x D tile through its own slice of the (now idle) tile stages —
    // 16-byte chunk index XOR row — and writes whole rows: 1 KiB contiguous per store instruction.  Every wave is behind the loop's last barrier.
    T* gb = reinterpret_cast<T*>(gp) + b * gsb + hr * gsh;
    auto g_rs = BIG ? rsrc_at(gb, gfull, (unsigned long long)r0 * (unsigned long long)gsn * 2ull) : __builtin_amdgcn_make_buffer_rsrc((void*)gb, 0, gbytes, 0x00020000);
    typedef __attribute__((ext_vector_type(4))) T t4;
    static_assert(NW * 32 * D * 2 <= NSTAGE * STAGE_BYTES, "one 32 x D slice per wave inside the tile stages");
    // (the lane ids go through an empty asm: nothing below is computed in front of the tile loop and kept live across it)
    int qix = qi, lanex = lane, hix = hi;
    asm volatile("" : "+v"(qix), "+v"(lanex), "+v"(hix));
    char* const ow = smem + wave * (32 * D * 2);
    constexpr int CH = D / 8;                        // 16-byte chunks per row
    const int osw = (CH == 16) ? (qix & 15) : (qix & 7);
#pragma unroll
    for (int d = 0; d < DT; ++d)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        t4 v4 = {(T)(acc[d][4 * g4 + 0] * osc), (T)(acc[d][4 * g4 + 1] * osc), (T)(acc[d][4 * g4 + 2] * osc), (T)(acc[d][4 * g4 + 3] * osc)};
        const int c = d * 4 + g4;
        *reinterpret_cast<u32x2*>(ow + qix * (D * 2) + ((c ^ osw) << 4) + hix * 8) = __builtin_bit_cast(u32x2, v4);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private slice: no barrier needed
    constexpr int RPI = 64 / CH;                     // rows per store instruction (4 at D = 128, 8 at D = 64)
#pragma unroll
    for (int i = 0; i < 32 / RPI; ++i) {
      const int r = i * RPI + lanex / CH, cpos = lanex % CH;
      const int c = cpos ^ ((CH == 16) ? (r & 15) : (r & 7));
      const u32x4 v = *reinterpret_cast<const u32x4*>(ow + r * (D * 2) + (cpos << 4));
      __builtin_amdgcn_raw_buffer_store_b128(v, g_rs, c * 8 < p.dv ? (wave_row0 - (BIG ? r0 : 0) + r) * gsn * 2 + (c << 4) : (int)TFA_OOB, 0, 0);
    }
  }
#if defined(TFA_BWD_TRACE)
  if (p.tr != nullptr && lane == 0) {                  // tiles | cycles from kernel entry to the tile loop << 16 | cycles behind the loop << 40
    unsigned long long* tr = reinterpret_cast<unsigned long long*>(p.tr) + ((size_t)blockIdx.x * NW + wave) * 4;
    const unsigned long long pro = (tw_t0 - tw_entry) & 0xffffffull, epi = (__builtin_amdgcn_s_memtime() - tw_t1) & 0xffffffull;
    tr[0] = tw_t1 - tw_t0; tr[1] = tw_mem; tr[2] = tw_bar; tr[3] = (unsigned long long)nu | (pro << 16) | (epi << 40);
  }
#endif
}

}  // namespace tfa
