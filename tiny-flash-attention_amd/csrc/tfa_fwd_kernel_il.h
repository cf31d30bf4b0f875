// tfa_fwd_kernel_il.h — forward kernel with an ISSUE-INTERLEAVED tile loop (gfx950).
//
// Why (tools/probe_issue.hip, tools/probe_overlap.hip, profiles/r01_pmc_default_cfg3.txt):
//   * one SIMD has ONE VALU issue port; a v_mfma_f32_32x32x16 keeps the matrix pipe busy for 32 cycles but
//     the port only for ~4-8, so ~5 ordinary VALU instructions (a v_exp_f32 counts as 2) issued by the SAME
//     wave right behind an MFMA run for free in its shadow;
//   * a wave that streams MFMAs back to back re-arms the port the moment it frees, so VALU work of the
//     OTHER wave on that SIMD is starved (head-of-line), and s_setprio does not change that;
//   * in the burst-structured kernels (QK^T burst, softmax burst, PV burst) the PMC counters show
//     time ~= MFMA time + VALU time: nothing overlaps.
// So the loop below is software-pipelined inside each wave and the order is pinned MFMA by MFMA with
// sched_barrier(0): every MFMA is followed by its share of the softmax VALU work of ANOTHER tile.
//
//   iteration j:   part 1   S(j+1) = K(j+1) Q^T   (2*D/16 MFMAs)  each followed by exp2/sum/pack of ~1.25 elements of tile j
//                  part 2   O += P(j) V(j)        (4*D/32 MFMAs)  each followed by 1 element of tile j (first 3/4) and
//                                                                  one v_max3 of the row max of S(j+1)
//   P(j) slot s (16 keys) is complete before the first PV MFMA that consumes it.  K tiles run one tile
//   ahead of V tiles in LDS (iteration j reads K(j+1) and V(j)); two K and two V buffers, refilled by
//   LDS-DMA at the top of the iteration that follows their last read; one workgroup barrier per tile.
//   The loop is unrolled by two so buffer addresses are immediates and S(j)/S(j+1) swap without copies.
//   Tiles that need masking (causal diagonal, ragged tail) run the same body with the mask applied
//   between the two parts (template flag, real branch: never if-converted into the steady state).
//
// Three forms are dispatched (tfa_api.hip:pick_variant): 8 waves x 32 rows = 256-row blocks, one workgroup per CU (il8, the
// default); 4 waves = 128-row blocks, two workgroups per CU (il4, medium grids); and the KEY-SPLIT form for grids of at most
// one 128-row block per CU (VF_IL_KSPLIT: 8 waves on one 128-row block, the two groups of four waves take the even / odd
// KV tiles and merge through LDS).  il8 and il4 also exist with per-tile buffer descriptors for head slices of 2 GiB and
// more (VF_IL_WINDOWED).
#pragma once
#include <type_traits>
#include "tfa_fwd_kernel_dma.h"

namespace tfa {


constexpr int VF_IL = 32768;        // issue-interleaved kernel (this file)
constexpr int VF_IL_DMASPREAD = 65536;   // issue the LDS-DMA pieces between MFMAs of part 1 instead of at the top
constexpr int VF_IL_EPI = 262144;        // 16-bit O leaves through a separate LDS region as whole rows (16-byte coalesced stores)
constexpr int VF_IL_EPI_INPLACE = 1048576;   // with EPI: the epilogue slices live in the (idle) tile buffers instead of a separate
                                             // region — 64 KiB total, so two 4-wave workgroups still fit a CU
constexpr int VF_IL_PREF = 524288;       // the next pass's K(0)/V(0)/K(1)/Q are requested BEFORE this pass's epilogue
constexpr int VF_IL_SEAM = 2097152;      // causal pairs: the heavy pass's last two iterations already request the light pass's K(0), K(1), V(0)
                                         // (same head, same K/V: the tile stream simply continues across the seam) and its Q
                                         // fragments are requested before the epilogue: the second prologue finds everything on chip
constexpr int VF_IL_KSPLIT = 1 << 25;    // small non-causal grids: the 8 waves work on ONE 128-row query block — waves 0-3 ("group 0") take the even
                                         // KV tiles, waves 4-7 the odd ones, each group with its own K/V ring in LDS; group 1 hands its O, m, l
                                         // to group 0 through LDS at the end.  Two waves per SIMD where 128-row workgroups alone would leave one.
constexpr int VF_IL_IDLE = 1 << 26;      // waves whose 32 rows all lie behind the last query row skip the tile work (decode-like problems: one
                                         // query block with one valid wave).  Its own instantiation: the flag costs the causal headline 0.5 %
constexpr bool IL_DECODE_NT = true;      // decode instantiations (VF_IL_IDLE: one query block per head): K/V tiles nobody else reads are
                                         // loaded non-temporal — MHA decode 6.3 -> 6.6 TB/s, packed GQA 5.9 -> 7.0 TB/s (profiles/r02_decode_nt_ab.txt)
constexpr int IL_OSTORE_AUX = 2;         // cache policy of the O row stores: nt — O is written once and never re-read, the XCD's L2 is better spent on
                                         // the K/V tiles every query block of the head re-reads (cfg3: +1.8 %, others +-0; profiles/r02_ostore_ab.txt)
constexpr int VF_IL_WINDOWED = 1 << 24;  // K/V tiles through per-tile descriptors (rsrc_at): a (b,h) slice may exceed 2 GiB.  ~6 % slower (a fresh
                                         // descriptor per tile: ~14 SALU + the SGPR->VMEM wait states), so only launched when needed
constexpr int VF_IL_PREF2 = 1 << 29;     // paired causal blocks: the light pass's K(0)/V(0)/K(1)/Q are requested behind the heavy pass's tile loop, in FRONT of
                                         // its O stores, and waited for with a COUNTED vmcnt behind them — vmcnt retires in issue order, so the light prologue
                                         // never waits for the stores' write acknowledges (round 4: light prologue 6.9 k -> 3.1 k cycles, +0.8 % on the headline;
                                         // earlier requests — inside the loop — cost the loop more registers than they save: docs/LABLOG.md L-9)
constexpr int VF_IL_QLDS = 1 << 28;      // the FIRST prologue of a workgroup brings Q in through LDS: a wave's 32 rows by LDS-DMA into its slice of the (idle)
                                         // epilogue region — whole 1 KiB pieces, 64 cache lines per wave instead of 256 32-byte segments — and the eight
                                         // fragments back with ds_read_b128 (the K tile's swizzle).  Later passes load Q as before (registers, PREF2)
constexpr int VF_IL_EXACT = 1 << 30;     // the reference's rounding points (TFA_FWD_EXACT_MAX): every tile is exponentiated against the EXACT running row maximum
                                         // (flash_attention.cu:263-316, main_torch_only.py:240-260) instead of the lazily re-based reference below.  Same
                                         // issue-interleaved body; a tile in which some row of the wave saw a new maximum runs the instantiation of the body
                                         // that also multiplies O by exp2(old - new): 64 v_mul, four behind each QK^T MFMA (round 5)
constexpr int VF_IL_DMASTAGGER = 131072; // with DMASPREAD: the upper half of the waves issues its pieces behind the first PV MFMAs,
                                         // so the two waves of a SIMD never sit in an LDS-DMA issue stall at the same time

}  // namespace tfa
#include "tfa_fwd_il_regs.h"
#if defined(TFA_IL_ASM_INC)        // an A/B arm of the generated loop (tools/gen_il_asm_loop.py with TFA_GEN_* knobs; tools/r5_arm.sh)
#include TFA_IL_ASM_INC
#else
#include "tfa_fwd_il_asm_loop.inc"
#endif
// the two asm statements of the hand-scheduled loops (operands = the kernel's locals; the text differs per dtype: tfa_fwd_il_tile_loop.inc picks it)
#define TFA_IL_ASM_Q8 [q0] "v"(qf[0]), [q1] "v"(qf[1]), [q2] "v"(qf[2]), [q3] "v"(qf[3]), [q4] "v"(qf[4]), [q5] "v"(qf[5]), [q6] "v"(qf[6]), [q7] "v"(qf[7])
#define TFA_IL_ASM_Q4 [q0] "v"(qf[0]), [q1] "v"(qf[1]), [q2] "v"(qf[2]), [q3] "v"(qf[3])
#define TFA_IL_ASM_SRC1 [ks0] "v"(k_src[0]), [vs0] "v"(v_src[0])
#define TFA_IL_ASM_SRC2 [ks0] "v"(k_src[0]), [ks1] "v"(k_src[1]), [vs0] "v"(v_src[0]), [vs1] "v"(v_src[1])
#define TFA_IL_ASM_SRC4 [ks0] "v"(k_src[0]), [ks1] "v"(k_src[1]), [ks2] "v"(k_src[2]), [ks3] "v"(k_src[3]), [vs0] "v"(v_src[0]), [vs1] "v"(v_src[1]), [vs2] "v"(v_src[2]), [vs3] "v"(v_src[3])
// (QOPS: the Q fragments of the kernel's width — 8 at 128, 4 at 64; SRCOPS: the lane offsets of the wave's DMA pieces — 1, 2 or 4 per tensor)
// Round 6: the lazy-reference statement also carries the bodies BEHIND the loop (dispatch + pinned / masked / last-tile body per parity, tools/gen_il_asm_loop.py:
// tail_blocks): [nact] the wave's tile count, [fmx] its first masked tile (-1: tails off — leave at jend as before), [nt] the block's tile count, [slim] the scalar part of the
// lanes' mask limit for tile fmx (apply_mask's: the wave's first row + the causal shift - the tile's first key), [ts] a scratch scalar, [xl] 0 / 1 = leave in front of the wave's last tile / 2 = run
// that tile without the closing vmcnt wait (the early requests for a pair's second pass, below).  One statement, one register assignment: as statements of their own the tails made hipcc move
// whole accumulator tuples through scratch between them (480 bytes per lane)
#define TFA_IL_ASM_LAZY_STMT_G(TEXT, QOPS, SRCOPS) \
  asm volatile(TEXT \
  : [sa0] "+v"(sA[0]), [sa1] "+v"(sA[1]), [sb0] "+v"(sB[0]), [sb1] "+v"(sB[1]), \
  [l0] "+v"(l4[0]), [l1] "+v"(l4[1]), [l2] "+v"(l4[2]), [l3] "+v"(l4[3]), [ma] "+v"(mA), [mb] "+v"(mB), [j] "+s"(j), \
  [koff] "+s"(koff), [voff] "+s"(voff), [ts] "=&s"(ts), \
  [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2), [f3] "=&v"(f3), [ka] "=&v"(ka), [ka5] "=&v"(ka5), [ka6] "=&v"(ka6), [ka7] "=&v"(ka7), [thr] "=&v"(thr) \
  : QOPS, [mref] "v"(mref), [kaddr] "v"(k_rd_addr), [va] "v"(vaddr), SRCOPS, \
  [sc] "s"(sc), [krs] "s"(k_rs), [vrs] "s"(v_rs), [ldsw] "s"(ldsw), [kstr] "s"(k_tile_stride), [vstr] "s"(v_tile_stride), [jend] "s"(jend), \
  [nact] "s"(nact_s), [fmx] "s"(fmx), [nt] "s"(nt_s), [slim] "s"(slim), [xl] "s"(xl) \
  : TFA_O_CLOB0, TFA_O_CLOB1, TFA_O_CLOB2, TFA_O_CLOB3, "m0", "vcc", "scc", "memory")
#define TFA_IL_ASM_LAZY_STMT(TEXT) TFA_IL_ASM_LAZY_STMT_G(TEXT, TFA_IL_ASM_Q8, TFA_IL_ASM_SRC2)
#define TFA_IL_ASM_EXACT_STMT(TEXT) \
  asm volatile(TEXT \
  : [sa0] "+v"(sA[0]), [sa1] "+v"(sA[1]), [sb0] "+v"(sB[0]), [sb1] "+v"(sB[1]), \
  [l0] "+v"(l4[0]), [l1] "+v"(l4[1]), [l2] "+v"(l4[2]), [l3] "+v"(l4[3]), [ma] "+v"(mA), [mb] "+v"(mB), [mref] "+v"(mref), [j] "+s"(j), \
  [koff] "+s"(koff), [voff] "+s"(voff), [ts] "=&s"(ts), \
  [f0] "=&v"(f0), [f1] "=&v"(f1), [f2] "=&v"(f2), [f3] "=&v"(f3), [ka] "=&v"(ka), [ka5] "=&v"(ka5), [ka6] "=&v"(ka6), [ka7] "=&v"(ka7), [alpha] "=&v"(alpha) \
  : [q0] "v"(qf[0]), [q1] "v"(qf[1]), [q2] "v"(qf[2]), [q3] "v"(qf[3]), [q4] "v"(qf[4]), [q5] "v"(qf[5]), [q6] "v"(qf[6]), [q7] "v"(qf[7]), \
  [kaddr] "v"(k_rd_addr), [va] "v"(vaddr), [ks0] "v"(k_src[0]), [ks1] "v"(k_src[1]), [vs0] "v"(v_src[0]), [vs1] "v"(v_src[1]), \
  [sc] "s"(sc), [krs] "s"(k_rs), [vrs] "s"(v_rs), [ldsw] "s"(ldsw), [kstr] "s"(k_tile_stride), [vstr] "s"(v_tile_stride), [jend] "s"(jend), \
  [nact] "s"(nact_s), [fmx] "s"(fmx), [nt] "s"(nt_s), [slim] "s"(slim) \
  : TFA_O_CLOB0, TFA_O_CLOB1, TFA_O_CLOB2, TFA_O_CLOB3, "m0", "vcc", "scc", "memory")
#if !defined(TFA_IL_USE_ASMTAIL)
#define TFA_IL_USE_ASMTAIL 1     // 0: the compiler-scheduled bodies outside the loop (the A/B arm of the round-6 tail bodies)
#endif
#if !defined(TFA_IL_USE_EARLY)
#define TFA_IL_USE_EARLY 0       // 1: the early requests for a pair's second pass (round 6, issue_early below) — an A/B arm, measured and NOT shipped: the light prologue
                                 // halves (3.1 k -> 1.5 k cycles) and epilogue + light pass lose 4.9 k, but the heavy pass's last iteration pays 2.5 k for the requests
                                 // themselves (a wave's eight Q loads are 256 32-byte segments on the CU's one address path, whenever they are issued) and the
                                 // launch comes out 0.3-1 % SLOWER in five spellings (profiles/r06_early_requests_ab.txt).  Default: PREF2's place, in the epilogue
#endif
#if !defined(TFA_IL_USE_MAXFREE)
#define TFA_IL_USE_MAXFREE 1     // 0: bf16 keeps the lazily re-based row reference with its per-tile row maximum (the A/B arm of the round-6 max-free rule)
#endif
#if !defined(TFA_IL_USE_ASMLOOP)
#define TFA_IL_USE_ASMLOOP 1     // 0: the compiler-scheduled body everywhere (the A/B arm of the hand-scheduled steady state, tools/r5_arm.sh)
#endif

// Timing probe (never in the product build): -DTFA_IL_PAD=n -DTFA_IL_PADKIND=k puts n extra do-nothing instructions behind every MFMA of the
// fast path — k = 0 s_nop 0, 1 SALU (s_mov_b32 to a dead register), 2 VALU (v_mov_b32 to a dead register), 3 s_waitcnt with counts nothing reaches.
// The slope d(time)/d(instructions) per class is what an instruction diet of the tile body can buy (tools/r5_pad.sh, profiles/r05_pad_slope.txt)
#if defined(TFA_IL_PAD)
#define TFA_IL_PAD1_0 asm volatile("s_nop 0");
#define TFA_IL_PAD1_1 { int pad_s_; asm volatile("s_mov_b32 %0, 0" : "=s"(pad_s_)); }
#define TFA_IL_PAD1_2 { int pad_v_; asm volatile("v_mov_b32 %0, 0" : "=v"(pad_v_)); }
#define TFA_IL_PAD1_3 asm volatile("s_waitcnt vmcnt(63) lgkmcnt(15)");
#define TFA_IL_PADCAT_(k) TFA_IL_PAD1_##k
#define TFA_IL_PADCAT(k) TFA_IL_PADCAT_(k)
#define TFA_IL_PAD_HERE { _Pragma("unroll") for (int pad_i_ = 0; pad_i_ < TFA_IL_PAD; ++pad_i_) TFA_IL_PADCAT(TFA_IL_PADKIND) }
#else
#define TFA_IL_PAD_HERE
#endif

namespace tfa {

// AB: timing-only ablation bits of the fast path (results are wrong when set; tools/ablate_il.py)
// (ILAB_TRACE is not an ablation: the instantiation that writes the per-workgroup cycle stamps of tfa_debug_set_trace.  The stamps' scalar
//  state — four 64-bit counters live from the first instruction to the last — cost every workgroup ~130 v_readlane / v_writelane, so the
//  kernels the library dispatches are compiled without it and a traced twin is launched when a trace buffer is set: tfa_fwd_inst.inc)
constexpr int ILAB_TRACE = 256;
#define P_TRACE ((AB & ILAB_TRACE) ? p.trace : (unsigned long long*)nullptr)
constexpr int ILAB_NOEXP = 1, ILAB_NODMA = 2, ILAB_NOBARRIER = 4, ILAB_NOMAX = 8, ILAB_NOQK = 16, ILAB_NOPV = 32, ILAB_NOKREAD = 64, ILAB_NOVREAD = 128;

// DVB: 32-wide column blocks that can hold valid head-dim columns (default: all of them).  D / 32 - 1 is instantiated for the two main
// kernels: head dims up to 96 (128 wide) / up to 32 (64 wide) skip the MFMAs, fragment reads and O registers of the empty last block;
// the LDS tiles and every address keep the full width.
template <typename T, int D, int NW, bool CAUSAL, bool F32OUT, int VF, int AB = 0, int DVB = D / 32>
__global__ __launch_bounds__(NW * 64, 2) __attribute__((amdgpu_num_vgpr(96))) void fwd_kernel_il(const KArgs p) {
  using E = Elem<T>;
  using X8 = typename E::x8;
  constexpr bool KSPLIT = (VF & VF_IL_KSPLIT) != 0;
  constexpr int NWG = KSPLIT ? NW / 2 : NW;        // waves per query block (KSPLIT: per key-tile group)
  constexpr int BM = NWG * 32;
  static_assert(!KSPLIT || ((VF & VF_IL_EPI_INPLACE) && !(VF & (VF_IL_SEAM | VF_IL_PREF | VF_IL_WINDOWED))), "KSPLIT: in-place epilogue");
  constexpr int KSTEP = KSPLIT ? 2 : 1;            // a wave's tile t is tile KSTEP*t + grp of the head
  constexpr int BN = 64;
  constexpr int CPR = D / 8;
  constexpr int TILE_BYTES = BN * D * 2;
  constexpr int PIECES = TILE_BYTES / 1024;
  constexpr int PPW = PIECES / NWG;                // DMA pieces per wave per tensor per tile
  static_assert(DVB >= 1 && DVB <= D / 32, "valid 32-column blocks of a D-wide kernel");
  constexpr int DS = 2 * DVB;                      // k-slots that are multiplied
  constexpr int DT = DVB;                          // 32-column tiles of O that exist
  constexpr int DT_L = D / 32;                     // ... as the V tile's LDS layout counts them
  constexpr int N1 = 2 * DS;                       // QK^T MFMAs per tile
  constexpr int N2 = 4 * DT;                       // PV MFMAs per tile
  // (settled by same-process A/Bs, all within +-0.7 %: read-ahead 3 or 4, 19 or 24 elements in part 1, a uniform element-to-slot
  //  map, inline-asm QK^T MFMAs, one key block after the other in part 1, the causal wave permutation — profiles/r02_*_ab.txt)
  constexpr int NE1 = 21;                          // softmax elements (of 32 per lane) summed/packed during part 1
  constexpr int PFK = 2, PFV = 2;                  // fragment read-ahead, in MFMAs
  // MFMA slot (0..N1+N2-1) in which softmax element e (0..31) is summed and packed; its exp2 is issued one slot and its
  // scale/subtract two slots earlier.  P slot s (elements 8s..8s+7) feeds PV MFMAs N1+DT*s.., so it must be packed in
  // an EARLIER slot than N1+DT*s (also the distance the asm MFMA needs after a VALU write of its operand).
  // (the body that also re-bases O — VF_IL_EXACT — carries 4 v_mul per MFMA in part 1 on top; moving elements behind the PV MFMAs to even the two parts
  //  out — 10, 12 or 14 elements in part 1 — measured +-0.4 %, like every other placement question here: the VALU total is what counts, round 5)
  auto slot_of_elem = [](int e) constexpr -> int {
    return 1 + (e < NE1 ? e * N1 / NE1 : N1 + (e - NE1) * (3 * DT - 1) / (32 - NE1));
  };
  static_assert(slot_of_elem(7) < N1 && slot_of_elem(15) < N1 + DT && slot_of_elem(23) < N1 + 2 * DT && slot_of_elem(31) < N1 + 3 * DT,
                "a P slot is packed too late for the PV MFMA that reads it");
  // QK^T MFMA i works on key block KT(i) with k-slot KS(i): the two key blocks alternate
#define KT(i) ((i) & 1)
#define KS(i) ((i) >> 1)
  constexpr bool PAIR = CAUSAL && (VF & VF_PAIR);
  static_assert(PPW >= 1 && PPW * NWG == PIECES, "tile does not split into whole DMA pieces per wave");

  extern __shared__ __attribute__((aligned(1024))) char smem[];
  // A workgroup's start-up is bound by instruction issue, and the older wave of a SIMD wins every issue slot it can use: the younger wave's first
  // memory requests used to go out ~4 k cycles after the older one's (tools/trace_prologue.py).  Until its requests are out a wave runs at
  // priority 1, afterwards at 0: the wave that still has to ask for its data goes first.
  asm volatile("s_setprio 1" ::: "memory");
  const int wave_id = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = KSPLIT ? wave_id / NWG : 0;      // key-tile group of this wave (KSPLIT), its own four tile buffers
  char* const gsm = smem + grp * 4 * TILE_BYTES;
  char* const kl = gsm;                            // K buffers 0,1
  char* const vl = gsm + 2 * TILE_BYTES;           // V buffers 0,1
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)gsm;

  unsigned long long t_start = 0, t_pro = 0, t_loop = 0, rt_start = 0;
#if defined(TFA_IL_TRACEPRO)
  const unsigned long long tp_entry = __builtin_amdgcn_s_memtime();      // before the first kernel argument has arrived
#endif
  if (P_TRACE) { rt_start = __builtin_amdgcn_s_memrealtime(); t_start = __builtin_amdgcn_s_memtime(); }
  const int tid = threadIdx.x;
  const int lane = tid & 63;
#if defined(TFA_IL_TRACEPRO)
  // start-up stamps k = 0..7 of every wave, stored at once (they sit in front of the first LDS-DMA: no counted vmcnt wait sees the stores)
#define TP_STAMP(k, dep)                                                                                                         \
  do {                                                                                                                             \
    asm volatile("" ::"s"(dep));                                                                                                   \
    if (P_TRACE) {                                                                                                                 \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                                  \
      P_TRACE[(size_t)gridDim.x * 8 + (size_t)gridDim.x * 32 + ((size_t)blockIdx.x * 8 + wave_id) * 8 + (k)] = t_; /* (every lane, same value) */ \
    }                                                                                                                              \
  } while (0)
  if (P_TRACE) P_TRACE[(size_t)gridDim.x * 8 + (size_t)gridDim.x * 32 + ((size_t)blockIdx.x * 8 + wave_id) * 8 + 0] = tp_entry;
#else
#define TP_STAMP(k, dep)
#endif
  const int wave = KSPLIT ? wave_id % NWG : wave_id;   // index inside the query block: rows, DMA pieces, epilogue slice
  // the 32-row block of a wave (a permutation that evens out the diagonal block's tiles per SIMD — {0,1,2,3,7,6,5,4} — measured
  // neutral: the per-tile barrier sets the diagonal's wall time whatever the map; profiles/r02_window_ab2.txt)
  const int wrow = wave;
  const int qi = lane & 31;
  const int hi = lane >> 5;

  // work item -> (b, h, hk, wi).  Divisions by launch constants through the host's magic numbers (FastDiv), one branch-free form for the three
  // orders (KArgs::rr ..): this decode is on the critical path of every workgroup's first memory request, and the two waves of a SIMD run it
  // one behind the other.  GQA: the G query heads of a K/V head stay on ONE XCD (its K/V tiles are fetched into one L2 instead of G)
  int bh, wi, b, h, hk;
  {
    const int id = blockIdx.x;
    const int x = p.rr ? (id & 7) : 0, s = p.rr ? (id >> 3) : id;
    const int sq = fd_div(s, p.fd_wa), r = s - sq * p.wa;
    const int kg = p.rr ? x + 8 * sq : sq;
    const int rq = fd_div(r, p.fd_nwork);
    wi = r - rq * p.nwork;
    b = fd_div(kg, p.fd_wd);
    h = (kg - b * p.wd) * p.wg + rq;
#if defined(TFA_IL_DECODE_WIMAJOR)
    // measurement arm (never in the product; MHA with B * H a multiple of 8 only): on each XCD the work items in WORK-ITEM-major order — all heads of the XCD run
    // their item 0, then item 1, .. — instead of head-major (a head's items back to back): profiles/r06_decode_order_pmc.txt
    if (p.rr && p.wg == 1) {
      const int hx = p.nbh >> 3;
      wi = s / hx;
      const int kg2 = x + 8 * (s - wi * hx);
      b = kg2 / p.wd;
      h = kg2 - b * p.wd;
    }
#endif
    hk = fd_div(h, p.fd_g);
    bh = b * p.H + h;
  }
  __builtin_assume(b >= 0 && h >= 0 && hk >= 0);     // (64-bit stride products without the sign terms)
  TP_STAMP(1, bh + wi + hk);                           // work item decoded
  const int shift = p.shift;

  const T* qbase = reinterpret_cast<const T*>(p.q) + b * p.qs_b + h * p.qs_h;
  const T* kbase = reinterpret_cast<const T*>(p.k) + b * p.ks_b + hk * p.ks_h;
  const T* vbase = reinterpret_cast<const T*>(p.v) + b * p.vs_b + hk * p.vs_h;
  // Q/O: one descriptor per query block (rsrc_at, once per pass).  K/V: one per slice, or — VF_IL_WINDOWED — one per tile.
  constexpr bool WIN = (VF & VF_IL_WINDOWED) != 0;
  // KSPLIT: group g sees the key sequence through a strided view — its tile j is tile 2j+g of the head
  unsigned long long k_bytes = p.k_bytes, v_bytes = p.v_bytes;
  if (KSPLIT) {
    const unsigned long long ko = (unsigned long long)grp * BN * p.ks_n * 2, vo = (unsigned long long)grp * BN * p.vs_n * 2;
    kbase += grp * BN * p.ks_n;
    vbase += grp * BN * p.vs_n;
    k_bytes = k_bytes > ko ? k_bytes - ko : 0;
    v_bytes = v_bytes > vo ? v_bytes - vo : 0;
  }
  // descriptor of a Q / O slice: the whole slice (the host guarantees < 2 GiB for the instantiations that are not WINDOWED), or from `off` on
  auto slice_rsrc = [&](const void* base, unsigned long long bytes, unsigned long long off) {
    if constexpr (WIN) return rsrc_at(base, bytes, off);
    else return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (unsigned)bytes, 0x00020000);
  };
  auto k_rs = __builtin_amdgcn_make_buffer_rsrc((void*)kbase, 0, WIN ? 0u : (unsigned)k_bytes, 0x00020000);
  auto v_rs = __builtin_amdgcn_make_buffer_rsrc((void*)vbase, 0, WIN ? 0u : (unsigned)v_bytes, 0x00020000);

  int k_src[PPW], v_src[PPW];
#pragma unroll
  for (int i = 0; i < PPW; ++i) {
    const int pc = wave * PPW + i;
    {
      const int row = pc * (1024 / (D * 2)) + lane / CPR;
      const int cpos = lane % CPR;
      const int kch = cpos ^ k_swz<D>(row);            // source chunk of this lane; chunks beyond the valid head dim read as zeros
      k_src[i] = kch * 8 < p.dv ? row * (int)p.ks_n * 2 + (kch << 4) : (int)TFA_OOB;
    }
    {
      const int o = pc * 1024 + lane * 16;
      const int sub = o >> 9, R = (o >> 6) & 7, pcs = (o >> 4) & 3;
      const int dt = sub % DT_L, sh = sub / DT_L;
      const int key = 16 * (sh >> 1) + 4 * (sh & 1) + 8 * (R >> 2) + (R & 3);
      v_src[i] = (dt * 4 + pcs) * 8 < p.dv ? key * (int)p.vs_n * 2 + ((dt * 4 + pcs) << 4) : (int)TFA_OOB;
    }
  }
  const int k_tile_stride = (KSPLIT ? 2 : 1) * BN * (int)p.ks_n * 2;
  const int v_tile_stride = (KSPLIT ? 2 : 1) * BN * (int)p.vs_n * 2;

  // (query heads that share a K/V head share its tiles in L2: no streaming hint then; nor for a K/V cache that the 256 MB memory-side
  //  cache can keep until the next decode step — the host sets KArgs::kv_stream from 768 MiB on: profiles/r03_decode_nt_ab.txt)
  TP_STAMP(2, (int)(size_t)qbase + (int)(size_t)kbase + (int)(size_t)vbase);   // base pointers
  const bool kv_private = p.H == p.Hk && p.kv_stream != 0;
  auto dma_k1 = [&](int t, int buf, int i) {
    if constexpr (WIN) lds_dma16_m0_fresh(rsrc_at(kbase, k_bytes, (unsigned long long)t * (unsigned)k_tile_stride), lds_base + buf * TILE_BYTES + (wave * PPW + i) * 1024, k_src[i]);
    else if ((VF & VF_IL_IDLE) && IL_DECODE_NT && kv_private) lds_dma16_m0_nt(k_rs, lds_base + buf * TILE_BYTES + (wave * PPW + i) * 1024, k_src[i] + t * k_tile_stride);
    else lds_dma16_m0(k_rs, lds_base + buf * TILE_BYTES + (wave * PPW + i) * 1024, k_src[i] + t * k_tile_stride);
  };
  auto dma_v1 = [&](int t, int buf, int i) {
    if constexpr (WIN) lds_dma16_m0_fresh(rsrc_at(vbase, v_bytes, (unsigned long long)t * (unsigned)v_tile_stride), lds_base + (2 + buf) * TILE_BYTES + (wave * PPW + i) * 1024, v_src[i]);
    else if ((VF & VF_IL_IDLE) && IL_DECODE_NT && kv_private) lds_dma16_m0_nt(v_rs, lds_base + (2 + buf) * TILE_BYTES + (wave * PPW + i) * 1024, v_src[i] + t * v_tile_stride);
    else lds_dma16_m0(v_rs, lds_base + (2 + buf) * TILE_BYTES + (wave * PPW + i) * 1024, v_src[i] + t * v_tile_stride);
  };
  auto dma_k = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_k1(t, buf, i);
  };
  auto dma_v = [&](int t, int buf) {
#pragma unroll
    for (int i = 0; i < PPW; ++i) dma_v1(t, buf, i);
  };

  // K fragment of k-slot sl: 16-byte chunk (2*sl + hi) ^ swz of row qi.  2*sl and hi do not share bits, so the LDS
  // address is (row base + ((hi ^ swz) << 4)) ^ (sl << 5): one register and one v_xor per read pair (smem is 1 KiB aligned).
  unsigned k_rd_addr = lds_base + qi * (D * 2) + ((hi ^ k_swz<D>(qi)) << 4);
  asm volatile("" : "+v"(k_rd_addr));
  const int i16 = lane & 15, g16 = (lane >> 4) & 1;
  const int v_rd_base = (hi * DT_L << 9) + ((i16 >> 2) << 6) + (g16 << 5) + ((i16 & 3) << 3);
  const float sc = p.scale_log2;
  int nt_total = 0, n_slow = 0;
  unsigned long long tw_wait = 0, tw_bar = 0;   // TFA_IL_TRACEWAIT debug sums

#if defined(TFA_IL_TRACEPRO)
  asm volatile("" ::"v"(k_src[0]), "v"(v_src[PPW - 1]));
#endif
  TP_STAMP(3, wi);                                     // lane offsets of the DMA pieces (in front of the first request)
  const int npass = PAIR ? ((p.nmb - 1 - wi) != wi ? 2 : 1) : 1;
  constexpr bool EPI = (VF & VF_IL_EPI) != 0;
  constexpr bool PREF2 = PAIR && (VF & VF_IL_PREF2) != 0;
  constexpr bool QLDS = (VF & VF_IL_QLDS) != 0;
  constexpr bool EXACT = (VF & VF_IL_EXACT) != 0;
  // MAX-FREE row reference (round 6; bf16 instantiations that carry the hand-scheduled statement): bf16 has fp32's exponent range, so P = exp2(s*c - mref)
  // needs no running maximum for RANGE — mref stays the row's maximum over its FIRST key tile, the 16 v_max3 per tile are gone (-11.5 % of the tile's VALU
  // work; +2.9 % on the headline) and what guards the fp32 sums is a test of what has been summed: a partial row sum beyond 2^40 leaves the statement and the
  // wave re-bases by an exact power of two (mref += e, O and l *= 2^-e: no rounding, so the re-base moves no result bit).  A tile that lifts a row
  // sum beyond 2^64 (a score 2^64 above everything the row had seen: the products P*v could overflow with it) is not repaired but the pass REDONE: the
  // workgroup agrees through one LDS word behind the tile loop, streams the block's K tiles once more for the rows' TRUE maxima (S = K Q^T and its row
  // maximum, nothing else), leaves them in LDS and runs the pass again with mref seeded by them (P <= 1: it cannot happen twice) — rare, 1.5 passes extra,
  // correct, and no second rule in the tile bodies (a max-tracking twin of every body beside the max-free one made hipcc spill 90 registers).
  // Stated domain: |v| * Nk < 2^63.  fp16 keeps the maximum (its P overflows at 2^16).  oracle/oracle.py: tiled_emulation_first_tile restates the rule.
  constexpr bool MAXFREE_BASE = std::is_same<T, __bf16>::value && !EXACT && TFA_IL_USE_MAXFREE && TFA_IL_USE_ASMLOOP && (D == 128 || D == 64) && DVB == D / 32 &&
                                (AB & ~ILAB_TRACE) == 0 && !(VF & (VF_IL_WINDOWED | VF_IL_IDLE | VF_IL_DMASTAGGER | VF_IL_SEAM)) && (PPW == 1 || PPW == 2 || PPW == 4);
  constexpr bool MAXFREE = MAXFREE_BASE;
  static_assert(!MAXFREE || !((VF & VF_IL_PREF) && !(VF & VF_IL_PREF2)), "max-free: a redone pass re-issues its own first requests (PREF2 or no prefetch)");
  // behind everything else in LDS: one word "some row of this block needs the pass redone", then one float per query row of the block (the seeds of a redone pass)
  constexpr int REDO_OFF = (KSPLIT ? 8 : 4) * TILE_BYTES + (((VF & VF_IL_EPI) && !(VF & VF_IL_EPI_INPLACE)) ? NW * 32 * D * 2 : 0);
  constexpr int SEED_OFF = REDO_OFF + 16;
  bool seeded = false;                                // this pass is a redone one: its rows take their reference from the seeds in LDS
  if (MAXFREE && threadIdx.x == 0) *reinterpret_cast<volatile int*>(smem + REDO_OFF) = 0;   // (ordered before its first read by every pass's barriers)
  static_assert(!QLDS || ((VF & VF_IL_EPI) && !(VF & (VF_IL_EPI_INPLACE | VF_IL_KSPLIT | VF_IL_WINDOWED | VF_IL_IDLE | VF_IL_SEAM))), "QLDS: a wave-private slice of the separate epilogue region");
  // store instructions of O per pass and wave (the epilogue's three forms: fp32 direct, 16-bit rows through LDS, 16-bit direct)
  constexpr int NST_EPI = F32OUT ? 4 * DT : ((VF & VF_IL_EPI) ? 32 / (64 / (D / 8)) : 4 * DT);
  constexpr bool PREF = (VF & VF_IL_PREF) != 0 || PREF2;
  static_assert(!PREF2 || ((VF & VF_IL_EPI) && !(VF & (VF_IL_EPI_INPLACE | VF_IL_SEAM | VF_IL_KSPLIT))), "PREF2: separate epilogue region, no seam streaming");
  constexpr bool SEAM = PAIR && (VF & VF_IL_SEAM) != 0;
  static_assert(!(SEAM && PREF), "SEAM includes the Q prefetch");
  static_assert(!(SEAM && (VF & VF_IL_EPI_INPLACE)), "the in-place epilogue would overwrite the streamed tiles");
  bool seam_in = false;                              // this pass's first tiles and Q were requested by the previous pass
  X8 qf[DS];
  auto own_tiles = [&](int ntg) -> int { return KSPLIT ? (ntg - grp + 1) >> 1 : ntg; };   // this wave's share of ntg tiles of the head
  auto key0_of = [&](int t) -> int { return (KSTEP * t + grp) * BN; };                        // first key of the wave's tile t
  auto block_of = [&](int pass) -> int {
    if (PAIR) return pass == 0 ? (p.nmb - 1 - wi) : wi;
    return CAUSAL ? (p.nmb - 1 - wi) : wi;
  };
  // requests for the start of query block mbx: K(0), V(0), K(1) by LDS-DMA and this lane's Q fragments
  // (round 6: what the per-pass code needs of the kernel arguments is read again from the kernarg segment through a laundered pointer — see the epilogue)
  typedef __attribute__((address_space(4))) const KArgs kargs_c;
  auto fresh_args = [&]() -> kargs_c* {
    kargs_c* a_ = (kargs_c*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(a_));
    return a_;
  };
  auto issue_prologue = [&](int mbx, bool with_dma) {
    kargs_c& p = *fresh_args();                        // (shadows the kernel's `p` inside this lambda on purpose)
    const int q0x = mbx * BM;
    int kve = p.Nk;
    if (CAUSAL) {
      const int lim = (((VF & VF_IL_IDLE) && p.row_mod > 0) ? p.row_mod : q0x + BM) + shift;
      kve = lim < kve ? lim : kve;
    }
    const int ntx = own_tiles(kve > 0 ? (kve + BN - 1) / BN : 0);
    if (with_dma && ntx > 0) dma_k(0, 0);
    if (with_dma && ntx > 0) dma_v(0, 0);
    int one = 1;                                       // (opaque: K(1)'s source offsets are otherwise shared with the first prologue's and live across the tile loop)
    asm volatile("" : "+s"(one));
    if (with_dma && ntx > 1) dma_k(one, 1);
    auto q_rs = slice_rsrc(qbase, p.q_bytes, (unsigned long long)q0x * (unsigned long long)p.qs_n * 2ull);
    // (the half-wave index goes through an empty asm: otherwise hipcc hoists the eight per-k-slot offsets out of the pass loop, where they stay live
    //  across the tile loop — eight registers the loop does not have, round 5 — and the lane id is re-derived with v_mbcnt instead of kept from the start)
    const int lane_p = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));   // (the lane id re-derived, like the epilogue does)
    int hix = lane_p >> 5;
    asm volatile("" : "+v"(hix));
    const int qoff = ((WIN ? 0 : q0x) + wrow * 32 + (lane_p & 31)) * (int)p.qs_n * 2 + hix * 16;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(q_rs, (2 * s + hix) * 8 < p.dv ? qoff + s * 32 : (int)TFA_OOB, 0, 0);
      qf[s] = __builtin_bit_cast(X8, t);
    }
  };
  // EARLY requests for the second pass of a causal pair (round 6; VERDICT r05 item 1's "seam"): in the LAST iteration of the heavy pass both K buffers and V
  // buffer 0 are idle (nt is even: V(nt-1) sits in buffer 1) and a wave's Q registers are dead (its last tile has no S(j+1)) — so every wave sends the light
  // block's K(0), V(0), K(1) pieces and its Q rows THERE, a tile time before the epilogue would (PREF2), and nothing of that iteration waits for them.  The Q
  // loads are inline asm into the registers in place ("+v"): hipcc's wait-count model does not see them (it would put a vmcnt(0) in front of the next pass's first
  // MFMA, docs/LABLOG.md L-9 item 7); the epilogue's counted vmcnt does the waiting (one store more is younger than the requests: the LSE store)
  auto issue_early = [&](int mbx) {
    kargs_c& p = *fresh_args();
    dma_k(0, 0);
    dma_v(0, 0);
    int one = 1;
    asm volatile("" : "+s"(one));
    dma_k(one, 1);
    const int q0x = mbx * BM;
    auto q_rs = slice_rsrc(qbase, p.q_bytes, (unsigned long long)q0x * (unsigned long long)p.qs_n * 2ull);
    const int lane_p = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    int hix = lane_p >> 5;
    asm volatile("" : "+v"(hix));
    const int qoff = ((WIN ? 0 : q0x) + wrow * 32 + (lane_p & 31)) * (int)p.qs_n * 2 + hix * 16;
    asm volatile("s_nop 4" ::: "memory");              // (a descriptor SALU code has just written -> VMEM: 5 wait states, and nothing pads an asm)
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const int off = (2 * s + hix) * 8 < p.dv ? qoff + s * 32 : (int)TFA_OOB;
      asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen" : "+v"(qf[s]) : "v"(off), "s"(q_rs) : "memory");
    }
  };
  // ---- QLDS: this wave's 32 rows of query block mbx into its slice of the epilogue region; the fragments out of it
  const unsigned q_slice = lds_base + 4 * TILE_BYTES + wave * (32 * D * 2);
  auto issue_q_dma = [&](int mbx) {
    constexpr int RPP = 1024 / (D * 2);                // rows per 1 KiB piece
    auto q_rs = slice_rsrc(qbase, p.q_bytes, 0ull);
    const int r0 = __builtin_amdgcn_readfirstlane(mbx * BM + wave * 32);   // (uniform on purpose: hipcc otherwise carries it in a VGPR into the pass loop)
#pragma unroll
    for (int i = 0; i < 32 / RPP; ++i) {
      const int row = i * RPP + lane / CPR, cpos = lane % CPR;
      const int kch = cpos ^ k_swz<D>(row);            // the K tile's image: position cpos of a row holds its chunk cpos ^ swz(row)
      const int off = kch * 8 < p.dv ? (r0 + row) * (int)p.qs_n * 2 + (kch << 4) : (int)TFA_OOB;
      lds_dma16_m0(q_rs, __builtin_amdgcn_readfirstlane(q_slice) + i * 1024, off);
    }
  };
  auto read_q_lds = [&]() {
    typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
#pragma unroll
    for (int s = 0; s < DS; ++s) {
      const unsigned a = q_slice + qi * (D * 2) + ((((2 * s + hi) ^ k_swz<D>(qi))) << 4);
      qf[s] = __builtin_bit_cast(X8, *reinterpret_cast<lds_u32x4*>(a));
    }
  };
  // a workgroup's first requests: K(0), the Q rows, then V(0) and K(1) — the first barrier needs only K(0) and Q, so the wait in front of it
  // (wait_first) leaves the two younger tiles in flight behind S(0) = K(0) Q^T; the barrier in front of the tile loop waits for them
  int first_late = 0;                                  // 0: nothing issued behind Q, 1: V(0), 2: V(0) and K(1)
  auto issue_first = [&](int mbx) {
    const int q0x = mbx * BM;
    int kve = p.Nk;
    if (CAUSAL) { const int lim = q0x + BM + shift; kve = lim < kve ? lim : kve; }
    const int ntx = kve > 0 ? (kve + BN - 1) / BN : 0;
    if (ntx > 0) dma_k(0, 0);
    issue_q_dma(mbx);
    if (ntx > 0) dma_v(0, 0);
    if (ntx > 1) dma_k(1, 1);
    first_late = ntx > 1 ? 2 : (ntx > 0 ? 1 : 0);
  };
  auto wait_first = [&]() {                            // vmcnt retires in issue order: K(0) and Q are the oldest requests
    if (first_late == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PPW) : "memory");
    else if (first_late == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  const int tr_pass = (p.dbg & 128) ? 1 : 0;
#if defined(TFA_IL_TRACEPRO)
  unsigned long long tp_issue = 0, tp_kv = 0, tp_q = 0, tp_bar = 0;
  const unsigned long long tp_t0 = t_start;
#endif
  if (PREF2) {                                         // the first pass's requests, complete before the loop (see the pass prologue)
    if (QLDS) issue_first(block_of(0));
    else issue_prologue(block_of(0), true);
    __builtin_amdgcn_sched_barrier(0);                 // (the loop's hoisted invariants stay behind the requests)
    asm volatile("s_setprio 0" ::: "memory");
    o_zero<DT>();                                      // (work that needs no data goes in front of the wait)
#if defined(TFA_IL_TRACEPRO)
    // debug build (tools/trace_prologue.py): when were the first requests out, when had K(0)/V(0)/K(1) landed (the 3 * PPW DMA pieces are the
    // OLDEST requests: vmcnt retires in order), when the Q fragments
    if (P_TRACE) {
      tp_issue = __builtin_amdgcn_s_memtime();
      if (QLDS) {                                      // K(0) and Q are the oldest requests here: one stamp for both
        wait_first();
        tp_kv = tp_q = __builtin_amdgcn_s_memtime();
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(DS) : "memory");
        tp_kv = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tp_q = __builtin_amdgcn_s_memtime();
      }
    }
#endif
    if (QLDS) { wait_first(); read_q_lds(); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < DS; ++s) asm volatile("" : "+v"(qf[s]));
  }
#pragma nounroll
  for (int pass = 0; pass < npass; ++pass) {
    const int mb = block_of(pass);
    if (P_TRACE && pass == 1 && tr_pass == 1) t_start = __builtin_amdgcn_s_memtime();
    const int q0 = mb * BM;
    // query POSITIONS of the wave's rows (for the causal mask): rows themselves, or — packed GQA heads, decode instantiation
    // only — row % row_mod, in which case a wave's rows span every position 0 .. row_mod-1
    const int rmod = ((VF & VF_IL_IDLE) && CAUSAL) ? p.row_mod : 0;
    int kv_end = p.Nk;
    if (CAUSAL) {
      const int lim = (rmod > 0 ? rmod : q0 + BM) + shift;
      kv_end = lim < kv_end ? lim : kv_end;
    }
    const int ntg = kv_end > 0 ? (kv_end + BN - 1) / BN : 0;   // tiles of the head this block visits
    const int nt_own = own_tiles(ntg);
    // iterations of the WORKGROUP (barriers, DMA): KSPLIT: group 0's count — group 1 may have one tile less (nact below)
    const int nt = KSPLIT ? (ntg + 1) >> 1 : ntg;
    nt_total += nt;

    const int wave_row0 = q0 + wrow * 32;
    // (the 4-wave and key-split instantiations re-derive the lane's row here: `qi` kept live from the kernel's first instructions cost their first prologue a spill)
    const int my_row = wave_row0 + ((NW == 4 || KSPLIT) ? (int)(__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) & 31u) : qi);
    const int pos_lo = rmod > 0 ? 0 : wave_row0, pos_hi = rmod > 0 ? rmod - 1 : wave_row0 + 31;   // positions of the wave's rows: [pos_lo, pos_hi]
    const int my_pos = rmod > 0 ? my_row % rmod : my_row;
    const int last_g = CAUSAL ? ((pos_hi + shift) >= 0 ? (pos_hi + shift) / BN : -1) : (ntg - 1);   // last tile of the head the wave's rows see
    const int wave_last_tile = KSPLIT ? (last_g >= grp ? (last_g - grp) >> 1 : -1) : last_g;

    float l4[4] = {0.f, 0.f, 0.f, 0.f};              // row sum of P, four interleaved partial sums carried across the tiles

    auto k_frag = [&](int kbuf, int i) -> X8 {   // fragment of QK^T MFMA i: key block i/DS, k-slot i%DS
      typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
      const unsigned a = (k_rd_addr ^ (KS(i) << 5)) + kbuf * TILE_BYTES + KT(i) * 32 * (D * 2);
      return __builtin_bit_cast(X8, *reinterpret_cast<lds_u32x4*>(a));
    };
    auto k_frag_rt = [&](unsigned kb_bytes, int i) -> X8 {   // same, K buffer chosen at run time
      typedef __attribute__((address_space(3))) const u32x4 lds_u32x4;
      const unsigned a = (k_rd_addr ^ (KS(i) << 5)) + kb_bytes + KT(i) * 32 * (D * 2);
      return __builtin_bit_cast(X8, *reinterpret_cast<lds_u32x4*>(a));
    };
    auto v_frag = [&](const char* vb, int i) -> X8 {   // fragment of PV MFMA i: key slot i/DT, d tile i%DT
      const char* a = vb + v_rd_base + ((i / DT) * 2 * DT_L << 9) + ((i % DT) << 9);
      s16x4 lo = lds_read_tr16_b64(a);
      s16x4 hh = lds_read_tr16_b64(a + 256);
      return __builtin_bit_cast(X8, __builtin_shufflevector(lo, hh, 0, 1, 2, 3, 4, 5, 6, 7));
    };
    auto needs_mask = [&](int t) -> bool {
      const int key0 = key0_of(t);
      bool nm = (key0 + BN > p.Nk);
      if (CAUSAL) nm = nm || (key0 + BN - 1 > pos_lo + shift);
      return nm;
    };
    auto apply_mask = [&](int t, f32x16 (&s)[2]) {
      int lim = p.Nk - 1;
      if (CAUSAL) { const int c = my_pos + shift; lim = c < lim ? c : lim; }
      lim -= key0_of(t) + 4 * hi;
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ko = 32 * tt + (r & 3) + 8 * (r >> 2);
          if (ko > lim) s[tt][r] = -INFINITY;
        }
    };
    // Reference exponent of the row ("mref", log2 domain) instead of the exact running max: P = exp2(s*sc - mref)
    // where mref is the row's scaled running max as of the last re-base.  A wave re-bases (every row takes its current
    // running max, O and l are multiplied by exp2(old - new)) only when some row max has outgrown its mref by more
    // than 2^8, so P <= 2^8: nothing can overflow in the fp32 sums or the 16-bit P, and O / l and the LSE are
    // unchanged mathematically.  That takes the O rescale (64 VALU per tile on ~60% of the tiles at N=4096 with the
    // exact-max rule, since SOME row of a wave nearly always sees a new max) out of the steady state; it lives in
    // the slow path below.  oracle/oracle.py:tiled_emulation_lazy restates this rule for the parity tests.
    float mref = -1e30f;
    auto l_part_max = [&]() -> float { return fmaxf(fmaxf(l4[0], l4[1]), fmaxf(l4[2], l4[3])); };
    auto trigger = [&](float mloc) -> bool {
      if constexpr (MAXFREE) return __any(l_part_max() > 0x1p40f);          // max-free: what has been summed, not what is about to be
      else return __any(mloc * sc > mref + 8.f);
    };
    // (mloc may be the max over only this half-wave's 32 keys of the tile: the trigger is an OR over all lanes anyway;
    //  the re-base itself combines the two halves first so that both lanes of a row keep the same reference)
    // EXACT: the running maximum itself is the reference — nref = max(mref, tile max) for every row, every tile; rows whose maximum did not
    // move get alpha = exp2(0) = 1 (the multiplication is the identity), and a wave none of whose rows moved skips it
    auto exact_step = [&](float mloc, float& alpha) -> bool {
      const float x = pair_max(mloc) * sc;
      const float nref = fmaxf(mref, x);
      const bool moved = __any(nref != mref);
      alpha = fast_exp2(mref - nref);
      mref = nref;
      return moved;
    };
    auto rescale_if_needed = [&](float mloc) {
      if constexpr (EXACT) {
        float alpha;
        if (exact_step(mloc, alpha)) {
#pragma unroll
          for (int i = 0; i < 4; ++i) l4[i] *= alpha;
          o_scale<DT>(alpha);
        }
      } else
      if constexpr (MAXFREE) {
        if (__any(l_part_max() > 0x1p40f)) {
          // re-base by the row sum itself: e = floor(log2(l)) from the exponent field, alpha = 2^-e — exact in O and l (a sum that is inf / NaN already gets
          // some factor: the pass is redone behind the loop anyway)
          const float lt = pair_sum((l4[0] + l4[1]) + (l4[2] + l4[3]));
          if (!(lt <= 0x1p64f)) *reinterpret_cast<volatile int*>(smem + REDO_OFF) = 1;   // beyond repair: the pass is redone behind the loop
          int e = (int)((__builtin_bit_cast(unsigned, lt) >> 23) & 0xffu) - 127;
          e = e < 0 ? 0 : (e > 126 ? 126 : e);
          const float alpha = __builtin_bit_cast(float, (unsigned)(127 - e) << 23);
#pragma unroll
          for (int i = 0; i < 4; ++i) l4[i] *= alpha;
          o_scale<DT>(alpha);
          mref += (float)e;
        }
      } else
      if (__any(mloc * sc > mref + 8.f)) {
        const float x = pair_max(mloc) * sc;
        const float nref = fmaxf(mref, x);
        const float alpha = fast_exp2(mref - nref);
#pragma unroll
        for (int i = 0; i < 4; ++i) l4[i] *= alpha;
        o_scale<DT>(alpha);
        mref = nref;
      }
    };

    // one softmax element of the current tile: e in [0,32): P slot e>>3 (16 keys), position e&7.  The empty asm
    // statements pin each result inside the MFMA slot it was written in (IR passes otherwise re-associate the row
    // sum into packed adds and sink the row max to the end of the block, keeping all 32 exponentials live).
    typedef __attribute__((ext_vector_type(2))) T t2;
    auto soft_elem = [&](int e, const f32x16 (&s)[2], float msc, float (&lsum)[4], unsigned (&pw)[16], float& ev_hold) {
      const int slot = e >> 3, t = slot >> 1, r = (slot & 1) * 8 + (e & 7);
      const float ev = fast_exp2(fmaf(s[t][r], sc, -msc));
      lsum[e & 3] += ev;
      asm volatile("" : "+v"(lsum[e & 3]));
      if (e & 1) {
        const t2 w = {(T)ev_hold, (T)ev};
        pw[e >> 1] = __builtin_bit_cast(unsigned, w);
        asm volatile("" : "+v"(pw[e >> 1]));
      } else {
        ev_hold = ev;
      }
    };
    // The same element split into three stages that the fast path issues in three DIFFERENT MFMA slots (fma two slots
    // ahead, exp2 one slot ahead, sum/pack in the element's own slot): a wave issues in order, so a chain
    // fma -> exp -> add inside one slot would stall on every result; staged, all VALU work of a slot is independent.
    auto st_fma = [&](int e, const f32x16 (&s)[2], float msc, float (&xs)[32]) {
      const int slot = e >> 3, t = slot >> 1, r = (slot & 1) * 8 + (e & 7);
      xs[e] = fmaf(s[t][r], sc, -msc);
      asm volatile("" : "+v"(xs[e]));
    };
    auto st_exp = [&](int e, float (&xs)[32]) {
      xs[e] = fast_exp2(xs[e]);
      asm volatile("" : "+v"(xs[e]));
    };
    auto st_sum = [&](int e, float (&xs)[32], float (&lsum)[4], unsigned (&pw)[16]) {
      lsum[e & 3] += xs[e];
      asm volatile("" : "+v"(lsum[e & 3]));
      if (e & 1) {
        const t2 w = {(T)xs[e - 1], (T)xs[e]};
        pw[e >> 1] = __builtin_bit_cast(unsigned, w);
        asm volatile("" : "+v"(pw[e >> 1]));
      }
    };
    auto p_frag = [&](const unsigned (&pw)[16], int slot) -> X8 {
      const u32x4 w = {pw[4 * slot], pw[4 * slot + 1], pw[4 * slot + 2], pw[4 * slot + 3]};
      return __builtin_bit_cast(X8, w);
    };

#include "tfa_fwd_il_pass_prologue.inc"
#include "tfa_fwd_il_tile_loop.inc"
    if (MAXFREE && redo_pass) {
      // (rare) this query block again.  First its rows' TRUE maxima: the block's K tiles once more through the (free) K buffers, S = K Q^T masked as in the
      // pass, the row maximum, nothing else; into LDS, one float per row.  Then the epilogue below runs as for any pass — what it stores for this block is
      // garbage the redone pass overwrites (same wave, same addresses, program order) — with THIS block as the "next" one: its requests, its O = 0; the pass
      // counter steps back, and the pass that follows takes its row references from the seeds (prologue): P <= 1 throughout, it cannot happen twice
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave is behind the tile loop and has read the word
      *reinterpret_cast<volatile int*>(smem + REDO_OFF) = 0;
      if (early_done) {                                // (the early requests put the NEXT block's Q rows in the registers: this block's again, and the epilogue asks anew)
        issue_prologue(mb, false);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int s_ = 0; s_ < DS; ++s_) asm volatile("" : "+v"(qf[s_]));
        early_done = false;
      }
      float mxr = -INFINITY;
      if (nt > 0) dma_k(0, 0);
#pragma nounroll
      for (int t = 0; t < nt; ++t) {
        if (t + 1 < nt) { dma_k(t + 1, (t + 1) & 1); asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PPW) : "memory"); }
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");
        if (t < nact && !idle_wave) {
          float mt;
          qk_burst(t & 1, t, sA, mt);
          mxr = fmaxf(mxr, mt);
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");              // buffer t & 1 is refilled two tiles on
      }
      {
        const int l_ = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        reinterpret_cast<volatile float*>(smem + SEED_OFF)[wave_id * 32 + (l_ & 31)] = mxr * sc;   // (both half-waves hold the row's maximum: qk_burst pairs them)
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      seeded = true;
      --pass;
    }
    // (the epilogue's "is there another pass, and for which block": a block to be redone is its own successor)
    const bool more_passes = (MAXFREE && redo_pass) || pass + 1 < npass;
    const int next_mb = (MAXFREE && redo_pass) ? mb : block_of(pass + 1);
#include "tfa_fwd_il_epilogue.inc"
  }

  if (P_TRACE) {
    __builtin_amdgcn_s_waitcnt(0);
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (tid == 0) {
      unsigned long long* t = P_TRACE + (size_t)blockIdx.x * 8;
      t[0] = t_start; t[1] = t_pro; t[2] = t_loop; t[3] = t_end;
#if defined(TFA_IL_TRACEWAIT)
      t[1] = t_start + tw_wait; t[2] = t_start + tw_wait + tw_bar;   // debug: "prologue" = memory waits, "loop" = barrier waits of wave 0
#endif
      t[4] = (unsigned long long)nt_total | ((unsigned long long)n_slow << 32);   // wave 0's slow-path tiles in the high half
      t[5] = (unsigned long long)__builtin_amdgcn_s_getreg(63508) | ((unsigned long long)__builtin_amdgcn_s_getreg(63492) << 32);   // XCC_ID | HW_ID << 32
#if defined(TFA_IL_TRACEPRO)
      t[4] = ((tp_issue - t_start) & 0xffffffffull) | ((tp_kv - t_start) << 32);
      t[5] = (t[5] & 0xffffffffull) | ((tp_q - t_start) << 32);
#endif
      t[6] = __builtin_amdgcn_s_memrealtime() - rt_start;   // 100 MHz ticks over the same span as t[3] - t[0] shader cycles
      t[7] = ((unsigned long long)bh << 32) | (unsigned)wi;
    }
#if defined(TFA_IL_TRACEPRO)
    if (lane == 0) {                                   // per wave: start, requests out, Q landed, at the first barrier
      unsigned long long* tw = P_TRACE + (size_t)gridDim.x * 8 + ((size_t)blockIdx.x * 8 + wave_id) * 4;
      tw[0] = tp_t0; tw[1] = tp_issue; tw[2] = tp_q; tw[3] = tp_bar;
    }
#endif
  }
}

#undef KT
#undef P_TRACE
#undef TP_STAMP
#undef KS

}  // namespace tfa
