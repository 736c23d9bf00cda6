// Family "q": the software-pipelined one-wave-per-SIMD kernel (hgemm_kernel_sp.hpp) with an EARLY-A
// operand split: two barriers per K-step instead of three, and ~1.5x the LDS-DMA flight time.
//
// Why (round-2 measurements): family "s" and hipBLASLt's kernel of the same geometry execute
// the same instruction counts; the 5-10 % gap on the compute-bound shapes is wave time parked at the sync
// points (SQ_WAIT_ANY 8-11 % vs 4-10 %).  In "s" the A pieces of a tile fly for only 66-108 MFMA slots
// (the A region of a stage is released late, after the slice-1 A fragments were read in interval A of the
// step that consumes the stage), and a K-step has three barriers (X1, Y1, Y2).
//
// Here the A fragments of BOTH K=32 slices of tile t+1 are read one K-step early (during K-step t, into a
// third fragment register set: +32 VGPRs), which frees the A region of its stage half a K-step earlier,
// and every sync point pairs one "landed" wait with one "region free" wait:
//
//   K-step t, stage s = t & 1 (tile t in stage s, tile t+1 in stage s^1); T = FM*FN MFMA slots per interval
//   interval A(t): MFMAs slice 0 = X x U
//       [0, RS*FN)        ds_read B fragments slice 1 of tile t            -> V     (from B[s])
//       P                 lgkmcnt(0) + vmcnt: A(t+1) landed; barrier.  B[s] is free.
//       (P, T)            LDS-DMA: B pieces of tile t+2 -> B[s];  ds_read A fragments slice 1 of tile t+1 -> Z (A[s^1])
//   interval B(t): MFMAs slice 1 = Y x V
//       [0, RS*FM)        ds_read A fragments slice 0 of tile t+1          -> X     (from A[s^1])
//       Q                 lgkmcnt(0) + vmcnt: B(t+1) landed; barrier.  A[s^1] is free.
//       (Q, T)            ds_read B fragments slice 0 of tile t+1 -> U (B[s^1]);  LDS-DMA: A pieces of tile t+3 -> A[s^1]
//   next K-step: slice-1 A set Y <-> Z swap roles (the loop is unrolled by two K-steps).
//
// Stream order ... B(t+1), A(t+2), B(t+2), A(t+3) ...: both waits are vmcnt(NJA + NJB) (two younger
// half-tiles stay in flight).  Flight time: B pieces >= ~100 slots (A(t) after P -> Q of K-step t+1), A pieces
// >= ~165 slots (B(t) after Q -> P of K-step t+2), against 66 / 127 in family "s".  LDS: the same two 64 KiB
// stages; registers: X, Y, Z (A) + U, V (B) = 160 fragment VGPRs + 256 accumulator AGPRs.
// Everything else (LDS image, swizzle, AGPR-resident accumulators, persistent item walk with the DMA streams
// crossing work items, epilogues) is family "s"; the A and the B stream now have their own cursor, because
// the A stream runs a K-step ahead of the B stream and crosses into the next work item earlier.
#pragma once

#include "hgemm_kernel_sp.hpp"

// Slot-plan knobs (experiment builds: HGEMM_LIB_SUFFIX=x HGEMM_EXTRA_HIPFLAGS="-DHGEMM_SQ_SLACK=10" python build.py).
#ifndef HGEMM_SQ_SLACK
#define HGEMM_SQ_SLACK 6      // MFMA slots between the last leading fragment read and the sync point
#endif
#ifndef HGEMM_SQ_SLACK32
#define HGEMM_SQ_SLACK32 4    // the same for the 32x32x16 members (their slots are 32 cycles long)
#endif
#ifndef HGEMM_SQ_RS64
#define HGEMM_SQ_RS64 2       // leading reads every N slots when an interval has >= 64 slots
#endif
// (Round 3 had an experiment knob here, HGEMM_SQ_XSTAGGER: a per-XCD K stagger of every work item's K walk.  It computed wrong
// results for every non-square member -- profiles/withdrawn/r03_check_q_xstagger_variant_FAILS.log -- and was removed from the
// source in round 4: no switch of this header selects a kernel that has not passed `hgemm_tune check`.  The stagger that the
// lock-step one-tile-per-CU plans lack comes from the plans themselves now: split-K / stream-K parts of a tile start their K
// walks at different offsets.)
#ifndef HGEMM_SQ_QORDER
#define HGEMM_SQ_QORDER 0     // behind Q: 0 = B-fragment reads lead the A pieces, 1 = the pieces lead
#endif
#ifndef HGEMM_SQ_SPREAD
#define HGEMM_SQ_SPREAD 1     // 1: the LDS-DMA pieces of a half-tile are spread over a whole interval's worth of slots: the
                              // first D of them go out behind the sync point that frees their region (as before), the last
                              // E = NJ - D in the pre-sync window of the NEXT interval, between its leading fragment reads.
                              // 0: round-2 plan (all pieces in the ~2/3 of an interval behind the sync point, one every 4 slots
                              // from each of the four waves at once).  Why: round-3 timeline ablation -- the pieces cost 180 of
                              // the 214 cycles a K-step spends beyond its 2048 MFMA cycles (tools/lab/gpu_round3_a.sh):
                              // the CU's one address path takes ~20 cycles per piece, four waves x one piece per 64 cycles
                              // saturates it behind every sync point while it idles in front of the next one.
#endif
#ifndef HGEMM_SQ_GAPS
#define HGEMM_SQ_GAPS 1       // spread plan only.  One wave per SIMD: whatever sits between two MFMAs shares the 16 cycles of the
                              // first one (about four issue slots, a VMEM or DS instruction takes two).  Round-3 timeline
                              // ablation: the K-step costs 2284 cycles against 2070 for the bare MFMA stream, 180 of the
                              // difference vanish with the LDS-DMA pieces although spreading them changes nothing -- what costs
                              // is the gap that holds buffer_load + s_add m0 + s_add soffset, the one with ds_read between two
                              // compiler-placed s_waitcnt, and the sync point's seven instructions.  1: (a) a piece is TWO asm
                              // statements in two different gaps (s_add m0 one gap ahead of the buffer_load; per-piece lane
                              // offsets instead of scalar ones, so no third instruction); (b) one lgkmcnt(0) the compiler can
                              // see (builtin) near the end of every interval, after which it places no waits of its own in
                              // front of the next interval's MFMAs; (c) the sync point's vmcnt wait sits one gap ahead of the
                              // lgkmcnt wait + barrier.  The vendor kernel's loop has the same one-instruction-per-gap shape.
#endif
#ifndef HGEMM_SQ_ABL
#define HGEMM_SQ_ABL 0        // measurement builds only (results are garbage): drop parts of the K loop to price them with the
                              // timeline stamps: 1 no s_barrier, 2 no vmcnt wait, 4 no lgkmcnt(0) at the sync points,
                              // 8 no LDS-DMA pieces, 16 no fragment reads
#endif

namespace hgemm_mi355x {

// KT = BK=64 sub-tiles per pipeline stage.  KT = 1: a stage holds K = 64, an interval is one K=32 MFMA slice.
// KT = 2: a stage holds K = 128 as two complete BK=64 images ([A rows][B rows] each, same swizzle), an interval
// is one sub-tile (two K=32 slices): twice the MFMA slots between sync points, which is what lets a 128x128
// tile (16 MFMA tiles per wave) amortise its two barriers per stage.
//
// MI = 32 ("_m32"): the same schedule on v_mfma_f32_32x32x16_f16.  An interval (K = 32) is then TWO k = 16 MFMA
// slices of FM x FN = 4 x 4 tiles: half the MFMA issues (each twice as long), the same eight ds_read_b128 per
// operand and interval, half the operand-register reads per flop.  The slot plan is the generic one with 32-cycle
// slots (RS = 1, its own slack).  KT = 1 only.
template <int BM_, int BN_, int WM_, int WN_, int KT_ = 1, int MI_ = 16>
struct CfgSQ : Cfg<BM_, BN_, WM_, WN_, MI_, 2> {
  using Base = Cfg<BM_, BN_, WM_, WN_, MI_, 2>;
  static constexpr int KT  = KT_;
  static constexpr int KS  = (MI_ == 16) ? 1 : 2;          // MFMA k-slices per K = 32
  static constexpr int SL  = KT * KS;                      // MFMA k-slices per interval
  static constexpr int ACC = (MI_ == 16) ? 4 : 16;         // accumulator registers per MFMA tile
  static constexpr int SUB_BYTES   = Base::STAGE_BYTES;    // one BK=64 image: (BM + BN) rows x 128 B
  static constexpr int STAGE_BYTES = KT * SUB_BYTES;
  static constexpr int LDS_BYTES   = 2 * STAGE_BYTES;
  static constexpr int NFA = Base::FM * SL, NFB = Base::FN * SL;   // fragment reads per operand per interval
  static constexpr int T   = Base::FM * Base::FN * SL;     // MFMA slots per interval
  static constexpr int PA  = Base::NI_A / Base::NW;        // LDS-DMA pieces per wave per operand per sub-tile
  static constexpr int PB  = Base::NJ - PA;
  static constexpr int NJA = KT * PA, NJB = KT * PB;       // ... per stage
  static constexpr int RS  = (T >= 64) ? HGEMM_SQ_RS64 : 1; // one leading fragment read every RS slots
  static constexpr int SLACK = (MI_ == 32) ? HGEMM_SQ_SLACK32 : (T >= 64) ? HGEMM_SQ_SLACK : (T >= 32 ? 6 : 2);
  static constexpr int P   = RS * NFB + SLACK;             // slot of interval A that carries sync P
  static constexpr int Q   = RS * NFA + SLACK;             // slot of interval B that carries sync Q
  // behind a sync point a DMA piece and a fragment read alternate, one item every ST slots
  static constexpr int STA = (T - P - 1) / (NJB + NFA) >= 2 ? 2 : 1;
  static constexpr int STB = (T - Q - 1) / (NJA + NFB) >= 2 ? 2 : 1;
  // spread plan: late pieces per operand (issued in the pre-sync window of the following interval: A's in interval A,
  // in front of P; B's in interval B, in front of Q) in proportion to that window's share of the interval
  static constexpr int EA = !HGEMM_SQ_SPREAD ? 0 : ((NJA * P + T / 2) / T < NJA ? (NJA * P + T / 2) / T : NJA - 1);
  static constexpr int EB = !HGEMM_SQ_SPREAD ? 0 : ((NJB * Q + T / 2) / T < NJB ? (NJB * Q + T / 2) / T : NJB - 1);
  static constexpr int DA = NJA - EA, DB = NJB - EB;       // early pieces (behind the sync point that frees the region)
  static_assert(KT == 1 || KT == 2, "one or two BK=64 sub-tiles per stage");
  static_assert(MI_ == 16 || (MI_ == 32 && KT == 1), "the 32x32x16 members hold K = 64 per stage");
  static_assert(Base::NI % Base::NW == 0 && Base::NI_A % Base::NW == 0, "every wave owns whole A and B pieces");
  static_assert(P + 1 + STA * (NJB + NFA) <= T && Q + 1 + STB * (NJA + NFB) <= T, "slot plan does not fit the interval");
  static_assert(Base::FM * Base::FN * ACC <= 256, "accumulators live in a0..a255");
  static_assert(LDS_BYTES + 64 <= 160 * 1024, "LDS budget");
  // AGPRs the kernel descriptor reserves: all 256 (one wave per SIMD owns the file), or exactly the accumulators when the LDS
  // footprint admits two workgroups per CU (sp_reserve_agprs)
  // WGS = 2: "two-resident" member (round 5) -- its two stages fit twice into the CU's 160 KiB, its waves into half a SIMD's
  // register file: the plain-epilogue kernels then declare the stages and nothing else (no vote word, no staged epilogue), and the
  // host launches 512 persistent workgroups.  Measured: the two workgroups of a CU do NOT end up with one's
  // epilogue under the other's K loop (both stay in their XCD's lock-step) -- what they buy is two LDS-DMA and two MFMA streams per
  // CU that overlap without a schedule of ours, which beats the register-staged family on the skinny shapes.
  static constexpr int WGS = (2 * LDS_BYTES <= 160 * 1024) ? 2 : 1;
  static constexpr int AGPRS = WGS == 2 ? Base::FM * Base::FN * ACC : 256;
};

#if defined(__HIP_DEVICE_COMPILE__)
// after sync P (interval A): item 2i = B piece i, item 2i+1 = A-fragment read i (pieces first: they need the flight time)
// after sync Q (interval B): item 2i = B-fragment read i (needed at the top of the next interval), item 2i+1 = A piece i
template <class CFG>
struct SqPlan {
  static constexpr int FM = CFG::NFA, FN = CFG::NFB, T = CFG::T;   // fragment reads per operand, MFMA slots
  // interval A
  static constexpr int a_piece_at(int n) {   // B piece index issued behind slot n of interval A, or -1
    for (int i = 0; i < CFG::NJB; ++i) if (CFG::P + 1 + CFG::STA * a_item_of_piece(i) == n) return i;
    return -1;
  }
  static constexpr int a_read_at(int n) {    // A-fragment (slice 1, next tile) read index behind slot n, or -1
    for (int i = 0; i < FM; ++i) if (CFG::P + 1 + CFG::STA * a_item_of_read(i) == n) return i;
    return -1;
  }
  // interleave two lists of possibly different length: alternate while both last, then the rest
  static constexpr int a_item_of_piece(int i) { return i < FM ? 2 * i : FM + i; }
  static constexpr int a_item_of_read(int i) { return i < CFG::NJB ? 2 * i + 1 : CFG::NJB + i; }
  // interval B
  static constexpr int b_item_of_read(int i) { return i < CFG::NJA ? 2 * i + HGEMM_SQ_QORDER : CFG::NJA + i; }
  static constexpr int b_item_of_piece(int i) { return i < FN ? 2 * i + 1 - HGEMM_SQ_QORDER : FN + i; }
  static constexpr int b_read_at(int n) {
    for (int i = 0; i < FN; ++i) if (CFG::Q + 1 + CFG::STB * b_item_of_read(i) == n) return i;
    return -1;
  }
  static constexpr int b_piece_at(int n) {
    for (int i = 0; i < CFG::NJA; ++i) if (CFG::Q + 1 + CFG::STB * b_item_of_piece(i) == n) return i;
    return -1;
  }
  // ---- spread plan (HGEMM_SQ_SPREAD): PH = 0 interval A (sync slot P), 1 interval B (sync slot Q) -------------------
  // late piece k of E, in front of sync slot S: centred in its 1/E share of [0, S), on an odd slot when the leading reads
  // sit on the even ones
  // (one-instruction-per-gap form: M0 is written one slot ahead, so slot 0 is out; the vmcnt wait of the sync point sits
  // at slot S - 2 and counts on every late piece having been issued: nothing later than S - 3)
  static constexpr int late_slot(int k, int E, int S) {
    int x = ((2 * k + 1) * S) / (2 * E);
    if (CFG::RS == 2) x |= 1;
    const int lo = 1 + k, hi = S - 3 - (E - 1 - k);
    return x < lo ? lo : x > hi ? hi : x;
  }
  // trailing read i of N behind sync slot S: every second slot when they fit, else every slot
  static constexpr int trail_step(int N, int S) { return S + 1 + 2 * (N - 1) < T ? 2 : 1; }
  static constexpr int trail_slot(int i, int N, int S) { return S + 1 + trail_step(N, S) * i; }
  // early piece k of D behind sync slot S: centred in its 1/D share of (S, T), off the trailing reads' slots
  static constexpr int early_slot(int k, int D, int S) {
    int x = S + 1 + ((2 * k + 1) * (T - S - 1)) / (2 * D);
    if (((x - S) & 1) != 0) ++x;
    return x < T ? x : T - 1;
  }
  template <int PH> static constexpr int sync_slot() { return PH == 0 ? CFG::P : CFG::Q; }
  template <int PH> static constexpr int n_late() { return PH == 0 ? CFG::EA : CFG::EB; }
  template <int PH> static constexpr int n_early() { return PH == 0 ? CFG::DB : CFG::DA; }
  template <int PH> static constexpr int n_trail() { return PH == 0 ? FM : FN; }
  template <int PH> static constexpr int late_at(int n) {     // late piece (index inside the late group) at slot n, or -1
    for (int k = 0; k < n_late<PH>(); ++k) if (late_slot(k, n_late<PH>(), sync_slot<PH>()) == n) return k;
    return -1;
  }
  template <int PH> static constexpr int early_at(int n) {
    for (int k = 0; k < n_early<PH>(); ++k) if (early_slot(k, n_early<PH>(), sync_slot<PH>()) == n) return k;
    return -1;
  }
  template <int PH> static constexpr int trail_at(int n) {
    for (int i = 0; i < n_trail<PH>(); ++i) if (trail_slot(i, n_trail<PH>(), sync_slot<PH>()) == n) return i;
    return -1;
  }
  template <int PH> static constexpr bool slots_ok() {        // every item has its own slot inside the interval
    const int S = sync_slot<PH>();
    for (int k = 0; k < n_late<PH>(); ++k) {
      if (late_slot(k, n_late<PH>(), S) > S - 3 || late_slot(k, n_late<PH>(), S) < 1) return false;
      if (k && late_slot(k, n_late<PH>(), S) <= late_slot(k - 1, n_late<PH>(), S)) return false;
    }
    for (int k = 0; k < n_early<PH>(); ++k) {
      if (early_slot(k, n_early<PH>(), S) >= T || early_slot(k, n_early<PH>(), S) <= S) return false;
      if (k && early_slot(k, n_early<PH>(), S) <= early_slot(k - 1, n_early<PH>(), S)) return false;
    }
    return trail_slot(n_trail<PH>() - 1, n_trail<PH>(), S) < T;
  }
  static_assert(!HGEMM_SQ_SPREAD || (slots_ok<0>() && slots_ok<1>()), "spread slot plan does not fit the interval");
};

// The spread plan of one interval as tables indexed by the slot (the K loop's body is unrolled over the slots: a table
// look-up with a constant index folds, the search loops of SqPlan::*_at would each be unrolled T times first and blow
// the unroller's size budget).  -1 = nothing in this slot.
template <class CFG, int PH>
struct SqSlots {
  signed char late[CFG::T + 1], early[CFG::T + 1], trail[CFG::T + 1];
  constexpr SqSlots() : late(), early(), trail() {
    using PL = SqPlan<CFG>;
    for (int n = 0; n <= CFG::T; ++n) {
      late[n] = (signed char)PL::template late_at<PH>(n);
      early[n] = (signed char)PL::template early_at<PH>(n);
      trail[n] = (signed char)PL::template trail_at<PH>(n);
    }
  }
};

// LDS-DMA piece `idx` (0 .. KT*P_op - 1) of operand OP for the stage at `stage`: sub-tile idx / P_op, 8-row
// block idx % P_op of this wave; the source advances 128 B per sub-tile.
// Addressing (round 3): the per-lane offsets (lane's row of the wave's q-th 8-row block, swizzled chunk) are computed ONCE
// per kernel -- they depend on the lane and the leading dimension only -- and the descriptor starts at the tile's first
// row and ENDS WITH THE MATRIX, so rows past the M / N edge are out of range and arrive as zeros.  Round 2 clamped the
// rows instead, which made the offsets depend on the work item: 16 VGPRs recomputed at every item seam with ~160 VALU
// operations and held twice by the register allocator around the seams.
template <class CFG, int OP>
__device__ __forceinline__ void sq_issue_piece(__amdgpu_buffer_rsrc_t rs, const uint32_t (&voff)[OP == 0 ? CFG::PA : CFG::PB], char* stage,
                                               int wave, int idx, uint32_t kbyte) {
  constexpr int POP = OP == 0 ? CFG::PA : CFG::PB;
  const int sub = idx / POP, q = idx % POP, p = (OP == 0 ? 0 : CFG::PA) + q;
  lds_void_t* dst = (lds_void_t*)(stage + sub * CFG::SUB_BYTES + (wave + p * CFG::NW) * 1024);
  // (the sub-tile's 128 B go into the scalar offset: the instruction's immediate offset would also be added to
  // the LDS address of an LDS-DMA)
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, dst, 16, voff[q], kbyte + sub * ROW_BYTES, 0, HGEMM_DMA_AUX);
}

// HGEMM_SQ_GAPS form of a piece: the LDS destination goes into M0 in one gap, the load is issued in the next.  M0 is
// the compiler's register; it has no use for it inside the K loops once every LDS-DMA there is one of these pairs (the
// builtin form, which sets M0 itself, is only used outside them), and tests/test_build_audit.py checks on the ISA that
// M0 writes and LDS-DMA loads strictly alternate in every MFMA loop.  An MFMA always sits between the two statements, which
// covers the one wait state an LDS-DMA needs behind an M0 write.
// `cont`: the previous piece of this window was idx - 1 in the same sub-tile, so M0 only moves on by one row block of the
// four waves (one instruction instead of an address computation + a move).
// Both statements name M0 as clobbered (round 5): the compiler then models the write -- it may merge two identical M0
// initialisations of its own only when nothing in between defines M0 -- instead of the ISA audit being the only guard.  clang
// warns that M0 is a reserved register it will not preserve across the statement: not preserving it is the point.
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
template <class CFG, int OP>
__device__ __forceinline__ void sq_piece_m0(uint32_t wave_stage_lds, int idx, bool cont) {
  constexpr int POP = OP == 0 ? CFG::PA : CFG::PB;
  const int sub = idx / POP, p = (OP == 0 ? 0 : CFG::PA) + idx % POP;
  if (cont && idx % POP != 0) {
    asm volatile("s_add_u32 m0, m0, %0" ::"n"(CFG::NW * 1024) : "m0");
  } else {
    // (the sum as an "s" operand: the piece index is a constant only after unrolling, too late for an immediate constraint)
    asm volatile("s_mov_b32 m0, %0" ::"s"(wave_stage_lds + (uint32_t)(sub * CFG::SUB_BYTES + p * CFG::NW * 1024)) : "m0");
  }
}
#pragma clang diagnostic pop
template <class CFG, int OP>
__device__ __forceinline__ void sq_piece_load(__amdgpu_buffer_rsrc_t rs, const uint32_t (&voff)[OP == 0 ? CFG::PA : CFG::PB], int idx,
                                              uint32_t kbyte) {
  constexpr int POP = OP == 0 ? CFG::PA : CFG::PB;
  const int sub = idx / POP, q = idx % POP;
  static_assert(HGEMM_DMA_AUX == 0, "cache-policy experiments use the builtin form");
  asm volatile("buffer_load_dwordx4 %0, %1, %2 offen lds" ::"v"(voff[q]), "s"(rs), "s"(kbyte + sub * ROW_BYTES) : "memory");
}
// every register of the three fragment sets the next MFMAs read is an operand of one asm statement: whatever copies the
// compiler owes them are done in front of it, and its s_nop covers the VALU -> MFMA-source wait states
template <int NA, int NB>
__device__ __forceinline__ void sq_settle(f16x8 (&x)[NA], f16x8 (&y)[NA], f16x8 (&u)[NB]) {
#pragma unroll
  for (int r = 0; r < NA; ++r) asm volatile("" : "+v"(x[r]), "+v"(y[r]));
#pragma unroll
  for (int r = 0; r < NB; ++r) asm volatile("" : "+v"(u[r]));
  asm volatile("s_nop 1");
}
constexpr int kWaitLgkm0 = 0xC07F;   // s_waitcnt lgkmcnt(0) as the builtin's immediate (vmcnt 63, expcnt 7 = no wait)

// One interval.  PHASE 0 = A(t): MFMAs af x bf; leading reads -> lead (B fragments of the second half of tile t);
// behind P: B pieces of tile t+2 and trailing reads -> trail (A fragments of the second half of tile t+1).
// PHASE 1 = B(t): leading reads -> lead (A fragments of the first half of tile t+1); behind Q: trailing reads ->
// trail (B fragments of the first half of tile t+1) and A pieces of tile t+3.
// Spread plan: the pieces behind the sync point are the FIRST D of the half-tile ("early", descriptor / cursor rs_e,
// kbyte_e, stage_e); in front of it go the LAST E pieces of the half-tile whose region the previous sync point freed
// ("late": interval A: A pieces of tile t+2 into A[s]; interval B: B pieces of tile t+2 into B[s]; rs_l, kbyte_l, stage_l).
// Fragment read r of a set: slice r / F reads from src0 / src1, row block r % F.
template <class CFG, int PHASE>
__device__ __forceinline__ void sq_interval(const f16x8 (&af)[CFG::NFA], const f16x8 (&bf)[CFG::NFB],
                                            f16x8 (&lead)[PHASE == 0 ? CFG::NFB : CFG::NFA], const char* lead0, const char* lead1,
                                            f16x8 (&trail)[PHASE == 0 ? CFG::NFA : CFG::NFB], const char* trail0, const char* trail1,
                                            int wave, const uint32_t (&voffA)[CFG::PA], const uint32_t (&voffB)[CFG::PB],
                                            __amdgpu_buffer_rsrc_t rs_l, char* stage_l, uint32_t kbyte_l,
                                            __amdgpu_buffer_rsrc_t rs_e, char* stage_e, uint32_t kbyte_e) {
  using PL = SqPlan<CFG>;
  constexpr int FM = CFG::FM, FN = CFG::FN, T = CFG::T, RS = CFG::RS;
  constexpr int NLEAD = PHASE == 0 ? CFG::NFB : CFG::NFA, FLEAD = PHASE == 0 ? FN : FM, FTRAIL = PHASE == 0 ? FM : FN;
  constexpr int OP_E = PHASE == 0 ? 1 : 0, OP_L = PHASE == 0 ? 0 : 1;          // operand of the early / late pieces
  constexpr int D_L = PHASE == 0 ? CFG::DA : CFG::DB;                          // late pieces are D_L .. NJ_op - 1 of their half-tile
  constexpr int S = PHASE == 0 ? CFG::P : CFG::Q;
  constexpr bool GAPS = HGEMM_SQ_SPREAD && HGEMM_SQ_GAPS;
  // LDS byte address of this wave's first 1-KiB block of the destination stages (GAPS: M0 = this + a constant per piece)
  const uint32_t lds_l = (uint32_t)(uintptr_t)(lds_void_t*)stage_l + (uint32_t)wave * 1024u;
  const uint32_t lds_e = (uint32_t)(uintptr_t)(lds_void_t*)stage_e + (uint32_t)wave * 1024u;
  (void)lds_l; (void)lds_e;
#pragma unroll
  for (int n = 0; n < T; ++n) {
    const int u = n / (FM * FN), i = (n / FN) % FM, j = n % FN;   // MFMA k-slice, accumulator tile (i, j)
    if (!GAPS && n == S) {
      // every fragment read of the region about to be refilled has RETURNED (LDS returns in order, and the
      // leading reads were the last ones issued), and my pieces of the half-tile the trailing reads are
      // about to consume have landed; two younger half-tiles may stay in flight
      if (!(HGEMM_SQ_ABL & 4)) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (!(HGEMM_SQ_ABL & 2)) wait_vmcnt<CFG::NJA + CFG::NJB>();
      if (!(HGEMM_SQ_ABL & 1)) sp_sync();
    }
    if (GAPS && n == S) {   // (the vmcnt wait went out one gap earlier)
      __builtin_amdgcn_sched_barrier(0);
      if (!(HGEMM_SQ_ABL & 4)) __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
      if (!(HGEMM_SQ_ABL & 1)) sp_sync();
    }
    sp_mfma_mi<CFG::MI>(i * FN + j, bf[u * FN + j], af[u * FM + i]);
    if (!(HGEMM_SQ_ABL & 16) && n % RS == 0 && n / RS < NLEAD) {
      const int r = n / RS;
      lead[r] = *(const f16x8*)((r / FLEAD ? lead1 : lead0) + (r % FLEAD) * CFG::MI * ROW_BYTES);
    }
    if constexpr (GAPS) {
      // gap n (behind MFMA n): at most one memory instruction besides a leading read; the M0 write of a piece one gap early
      constexpr SqSlots<CFG, PHASE> tab{};
      const int l = (HGEMM_SQ_ABL & 8) ? -1 : tab.late[n], l1 = (HGEMM_SQ_ABL & 8) ? -1 : tab.late[n + 1];
      const int e = (HGEMM_SQ_ABL & 8) ? -1 : tab.early[n], e1 = (HGEMM_SQ_ABL & 8) ? -1 : tab.early[n + 1];
      const int r = (HGEMM_SQ_ABL & 16) ? -1 : tab.trail[n];
      if (l >= 0) { if constexpr (OP_L == 0) sq_piece_load<CFG, 0>(rs_l, voffA, D_L + l, kbyte_l); else sq_piece_load<CFG, 1>(rs_l, voffB, D_L + l, kbyte_l); }
      if (e >= 0) { if constexpr (OP_E == 0) sq_piece_load<CFG, 0>(rs_e, voffA, e, kbyte_e); else sq_piece_load<CFG, 1>(rs_e, voffB, e, kbyte_e); }
      if (r >= 0) trail[r] = *(const f16x8*)((r / FTRAIL ? trail1 : trail0) + (r % FTRAIL) * CFG::MI * ROW_BYTES);
      if (l1 >= 0) sq_piece_m0<CFG, OP_L>(lds_l, D_L + l1, l1 > 0);
      if (e1 >= 0) sq_piece_m0<CFG, OP_E>(lds_e, e1, e1 > 0);
      if (n == S - 2 && !(HGEMM_SQ_ABL & 2)) wait_vmcnt<CFG::NJA + CFG::NJB>();
      if (n == T - 3) {   // every read of this interval has long returned: from here on the compiler knows it, too
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_waitcnt(kWaitLgkm0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if constexpr (HGEMM_SQ_SPREAD) {
      constexpr SqSlots<CFG, PHASE> tab{};
      const int l = (HGEMM_SQ_ABL & 8) ? -1 : tab.late[n];
      const int e = (HGEMM_SQ_ABL & 8) ? -1 : tab.early[n];
      const int r = (HGEMM_SQ_ABL & 16) ? -1 : tab.trail[n];
      if (l >= 0) { if constexpr (OP_L == 0) sq_issue_piece<CFG, 0>(rs_l, voffA, stage_l, wave, D_L + l, kbyte_l); else sq_issue_piece<CFG, 1>(rs_l, voffB, stage_l, wave, D_L + l, kbyte_l); }
      if (r >= 0) trail[r] = *(const f16x8*)((r / FTRAIL ? trail1 : trail0) + (r % FTRAIL) * CFG::MI * ROW_BYTES);
      if (e >= 0) { if constexpr (OP_E == 0) sq_issue_piece<CFG, 0>(rs_e, voffA, stage_e, wave, e, kbyte_e); else sq_issue_piece<CFG, 1>(rs_e, voffB, stage_e, wave, e, kbyte_e); }
    } else if (PHASE == 0) {
      const int r = (HGEMM_SQ_ABL & 16) ? -1 : PL::a_read_at(n), p = (HGEMM_SQ_ABL & 8) ? -1 : PL::a_piece_at(n);
      if (p >= 0) { if constexpr (OP_E == 0) sq_issue_piece<CFG, 0>(rs_e, voffA, stage_e, wave, p, kbyte_e); else sq_issue_piece<CFG, 1>(rs_e, voffB, stage_e, wave, p, kbyte_e); }
      if (r >= 0) trail[r] = *(const f16x8*)((r / FTRAIL ? trail1 : trail0) + (r % FTRAIL) * CFG::MI * ROW_BYTES);
    } else {
      const int r = (HGEMM_SQ_ABL & 16) ? -1 : PL::b_read_at(n), p = (HGEMM_SQ_ABL & 8) ? -1 : PL::b_piece_at(n);
      if (r >= 0) trail[r] = *(const f16x8*)((r / FTRAIL ? trail1 : trail0) + (r % FTRAIL) * CFG::MI * ROW_BYTES);
      if (p >= 0) { if constexpr (OP_E == 0) sq_issue_piece<CFG, 0>(rs_e, voffA, stage_e, wave, p, kbyte_e); else sq_issue_piece<CFG, 1>(rs_e, voffB, stage_e, wave, p, kbyte_e); }
    }
  }
}
#endif  // __HIP_DEVICE_COMPILE__

// One operand's LDS-DMA stream cursor: descriptor of the current work item's tile rows and the K position inside it.
// OP 0 = A, OP 1 = B.  Descriptors must be PROVABLY uniform (readfirstlane on the base pointer) or hipcc wraps every
// LDS-DMA in a waterfall loop (see hgemm_kernel_sp.hpp).  The range ends with the operand's last row (see sq_issue_piece).
#define SQ_LOAD_ITEM(OP, ITEM)                                                                                 \
  do {                                                                                                         \
    /* the A stream crosses into an item first; the B stream reuses its tile coordinates (the raster map     \
       costs several integer divisions: computed once per item, kept in SGPRs) */                            \
    if (nxt_item != (ITEM)) {                                                                                  \
      const TileCoord itc = map_logical(g, walk.base + walk.first + (ITEM) * walk.stride, BM, BN);             \
      nxt_item = (ITEM);                                                                                       \
      nxt_m0 = __builtin_amdgcn_readfirstlane(itc.m0); nxt_n0 = __builtin_amdgcn_readfirstlane(itc.n0);        \
      nxt_kb = __builtin_amdgcn_readfirstlane(itc.k_begin * 2);                                                \
      /* stages of K = 64 * KT (host: K chunk % (64 KT) == 0); ktail variant: WHOLE stages of the item's K range */       \
      nxt_nk = __builtin_amdgcn_readfirstlane(KTAIL ? sq_k_items(g, itc) / (BK * CFG::KT) : itc.nk / CFG::KT);           \
    }                                                                                                          \
    const uintptr_t addr = (OP) == 0 ? reinterpret_cast<uintptr_t>(g.A + (size_t)nxt_m0 * g.lda)               \
                                     : reinterpret_cast<uintptr_t>(g.Bt + (size_t)nxt_n0 * g.ldb);             \
    const uintptr_t uni = ((uintptr_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(addr >> 32)) << 32) |     \
                          (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)addr);                       \
    const unsigned long long rows_ = (OP) == 0 ? (unsigned long long)(g.M - nxt_m0) : (unsigned long long)(g.N - nxt_n0); \
    const unsigned long long rem_ = rows_ * (unsigned long long)((OP) == 0 ? g.lda : g.ldb) * 2ull;            \
    const unsigned range_ = (unsigned)__builtin_amdgcn_readfirstlane((int)(rem_ > 0xFFFFFFFFull ? 0xFFFFFFFFu : (unsigned)rem_)); \
    if ((OP) == 0) rsA = __builtin_amdgcn_make_buffer_rsrc((void*)uni, 0, (int)range_, 0x00020000);            \
    else           rsB = __builtin_amdgcn_make_buffer_rsrc((void*)uni, 0, (int)range_, 0x00020000);            \
    /* kstagger variant: this XCD's workgroups enter the item's K range stag_ stages in and wrap behind its last stage;   \
       BOTH streams of an item take the same rotation (the cursor is all that knows about it: whatever is issued for a      \
       stream position reads (descriptor, kbyte) of the cursor, early and late pieces alike) */                            \
    const int stag_ = STAG ? (int)(((uint32_t)(blockIdx.x & (NUM_XCD - 1)) * (uint32_t)nxt_nk) >> 3) : 0;                   \
    cur[OP].kbyte0 = (uint32_t)nxt_kb; cur[OP].kwrap = nxt_nk - stag_;                                         \
    cur[OP].kbyte = (uint32_t)nxt_kb + (uint32_t)stag_ * (uint32_t)(CFG::KT * ROW_BYTES);                      \
    cur[OP].item = (ITEM); cur[OP].kt = 0; cur[OP].nk = nxt_nk;                                                \
  } while (0)

// one stage on inside the current item
#define SQ_STEP_CURSOR(OP)                                                                   \
  do {                                                                                       \
    ++cur[OP].kt;                                                                            \
    cur[OP].kbyte += CFG::KT * ROW_BYTES;                                                    \
    if (STAG && cur[OP].kt == cur[OP].kwrap) cur[OP].kbyte = cur[OP].kbyte0;   /* (kwrap == nk without a stagger: never reached) */ \
  } while (0)

// Move a stream one K-step on; past the last step of the last item it stays put (the branch-free DMA then
// re-reads that valid tile into a region nobody consumes).
#define SQ_ADVANCE(OP)                                        \
  do {                                                        \
    if (cur[OP].kt + 1 < cur[OP].nk) {                        \
      SQ_STEP_CURSOR(OP);                                     \
    } else if (cur[OP].item + 1 < walk.count) {               \
      SQ_LOAD_ITEM(OP, cur[OP].item + 1);                     \
    }                                                         \
  } while (0)

// One pipeline step (K = 64 * KT) on stage (step & 1).  YS = second-half A fragments of the current tile,
// ZS = where the next tile's go.  Fragment source of interval h, slice u: KT = 1: the stage image, K = 32 half h
// (MI = 32: its k = 16 slice u); KT = 2: sub-tile image h, K = 32 half u.
#define SQ_FRAG(STAGE, OPOFF, H, U) ((STAGE) + (CFG::KT == 2 ? (H) * CFG::SUB_BYTES : 0) + (OPOFF) + \
                                     foff[CFG::KT == 2 ? (U) : (H)][CFG::KT == 2 ? 0 : (U) % CFG::KS])
// ADV0 / ADV1 move the A / B stream one stage on.  The A stream moves between the two intervals: interval A still issues
// the late pieces of the A tile the cursor points at, interval B the early pieces of the next one.  The B stream moves
// behind interval B (early pieces in interval A, late ones in interval B, same tile).
#define SQ_K_STEP(YS, ZS, ADV0, ADV1)                                                                           \
  do {                                                                                                          \
    char* st  = smem + (step & 1) * CFG::STAGE_BYTES;                                                           \
    char* nst = smem + ((step + 1) & 1) * CFG::STAGE_BYTES;                                                     \
    sq_interval<CFG, 0>(fX, fU, fV, SQ_FRAG(st, b_base_off, 1, 0), SQ_FRAG(st, b_base_off, 1, 1),               \
                        ZS, SQ_FRAG(nst, a_base_off, 1, 0), SQ_FRAG(nst, a_base_off, 1, 1), wave, voffA, voffB, \
                        rsA, st, cur[0].kbyte, rsB, st, cur[1].kbyte);                                          \
    ADV0;                                                                                                       \
    sq_interval<CFG, 1>(YS, fV, fX, SQ_FRAG(nst, a_base_off, 0, 0), SQ_FRAG(nst, a_base_off, 0, 1),             \
                        fU, SQ_FRAG(nst, b_base_off, 0, 0), SQ_FRAG(nst, b_base_off, 0, 1), wave, voffA, voffB, \
                        rsB, st, cur[1].kbyte, rsA, nst, cur[0].kbyte);                                         \
    ADV1;                                                                                                       \
    ++step;                                                                                                     \
  } while (0)

// K range of a work item in the ktail variant: the LAST split runs to K (the host cuts K into chunks of whole stages, counted
// on floor(K / stage): the remainder rides on the last split)
__device__ __forceinline__ int sq_k_items(const GemmArgs& g, const TileCoord& tc) {
  return (tc.split + 1 == g.splits ? g.K : tc.k_begin + g.k_chunk) - tc.k_begin;
}

// EPI_: SP_EPI_NARROW / SP_EPI_WIDE / SP_EPI_SLAB / SP_EPI_FUSED (as family "s"), + EPI_KTAIL: the variant for K % (64 KT) != 0
// (K % 8 == 0, MI = 16 members): the pipeline walks the whole stages of every work item, the remaining < 64 KT elements are
// accumulated by direct_k_tail between the item's last K-step and its epilogue.  At that point the fragment sets X, Y, U
// already hold the NEXT item's first tile (the streams cross item seams), Z and V are dead: the tail's fragments take their
// registers.  Its loads queue behind the LDS-DMA pieces in flight; waiting for them (vmcnt is in order) only lands the
// next item's tiles early.
// + EPI_KSTAGGER (round 5): the variant for the one-round, lock-step plans (one tile per CU, every workgroup at the same K offset
// of rows 16-32 KiB apart: which HBM channels collide depends on the box, DESIGN.md sections 4.12 / 6.5): the workgroups of XCD x
// walk every work item's stages in the order x nk / 8, ..., nk - 1, 0, ..., x nk / 8 - 1.  Inside an XCD the workgroups stay in
// lock-step (they share operand panels in its L2), the eight XCDs are nk / 8 stages apart.  Round 3's knob of the same intent
// offset the two streams separately and was wrong for every non-square member; here the rotation lives in the stream cursor
// (SQ_LOAD_ITEM / SQ_STEP_CURSOR), which both streams of an item derive from the same (XCD, stage count).  The summation order of
// a tile then depends on the XCD that computes it: exact on 0/1 inputs, deterministic per plan (raster group included) on N(0,1).
template <class CFG, int EPI_>
__global__ void __launch_bounds__(CFG::THREADS) hgemm_tn_sq_kernel(const GemmArgs g) {
  prefetch_kernargs<sizeof(GemmArgs)>();
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int EPI = EPI_ & 7;
  constexpr bool KTAIL = (EPI_ & EPI_KTAIL) != 0;
  constexpr bool STAG = (EPI_ & EPI_KSTAGGER) != 0;
  static_assert(!KTAIL || CFG::MI == 16, "the K tail is built from 16x16x32 fragments");
  static_assert(!(KTAIL && STAG), "the kstagger variant takes whole stages only (the host drops the flag when K has a tail)");
  constexpr int BM = CFG::BM, BN = CFG::BN, FM = CFG::FM, FN = CFG::FN, NJ = CFG::NJ, MI = CFG::MI;
  constexpr int NFA = CFG::NFA, NFB = CFG::NFB;
  constexpr int NQ = CFG::ACC / 4;              // f32x4 quads per accumulator tile

  // the two stages + one word for the single-launch split-K vote (ONE LDS object, see hgemm_kernel_sp.hpp)
  constexpr bool STAGED = EPI == SP_EPI_WIDE && CFG::WGS == 1 && sp_staged_ok<CFG>(CFG::LDS_BYTES);   // + 4 KiB per wave for the epilogue
#ifdef HGEMM_TIMELINE
  constexpr int TL_EXTRA = 64;   // measurement build: eight more stamp slots behind everything else
  constexpr int FLAG_BYTES = 64;
#else
  constexpr int TL_EXTRA = 0;
  constexpr int FLAG_BYTES = (CFG::WGS == 2 && EPI != SP_EPI_FUSED) ? 0 : 64;   // (two-resident members: exactly the stages, see CfgSQ::WGS)
#endif
  constexpr int STAGED_BYTES = STAGED ? CFG::NW * SP_STAGED_BYTES_PER_WAVE : 0;
  __shared__ __attribute__((aligned(1024))) char smem[CFG::LDS_BYTES + FLAG_BYTES + STAGED_BYTES + TL_EXTRA];

  const int tid  = threadIdx.x;
  // (measurement build: stamps go to the 64 scratch bytes behind the stages; slot 0 doubles as the fused vote word,
  // so the timeline is only meaningful for the plain epilogues)
  HGEMM_TL_STAMP(smem + CFG::LDS_BYTES, 0, tid);
  HGEMM_TL_REALTIME(smem + CFG::LDS_BYTES, 1, tid);
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave_m = wave / CFG::WN;
  const int wave_n = wave % CFG::WN;

  const ItemWalk walk = persistent_walk(g.items);
  if (walk.count == 0) return;

  // ---- phase control of the persistent walk (round 5; plan flags, wave-uniform, outside every loop) ------------------------------
  // What it is for: the workgroups of an XCD walk their items in lock-step -- they share operand panels
  // in its L2 -- so all 32 CUs of the XCD reach their epilogues together and 32 x BM x BN x 2 B of C meet the XCD's one path into
  // the fabric at once (16384^2 x 256: ~11k cycles per epilogue of a 256 x 256 tile = 0.7 TB/s per XCD), while that path idles
  // during the K loops: the time of the small-K / large-MN class is the SUM of a store phase and an MFMA phase.
  //   flags bit 3 (HGEMM_PLAN_PHASE_OFFSET): every second workgroup of an XCD enters its walk half an item period late, so half of
  //     the XCD's CUs store while the other half multiplies.  The wait is a sleep in front of the first LDS-DMA piece, once per launch.
  //     Bit 5 (HGEMM_PLAN_PHASE_OFFSET4): four groups a quarter period apart; both bits: eight groups an eighth apart.
  //   flags bit 4 (HGEMM_PLAN_WAVE_PRIORITY), two-resident members: the wave in the odd hardware slot of its SIMD raises its
  //     priority for good, so the two workgroups of a CU stop sharing the matrix pipe evenly (and reaching their epilogues
  //     together): one's K loop runs at full rate and its epilogue under the other's K loop.
  if (g.flags & (8 | 32)) {   // (bit 5, HGEMM_PLAN_PHASE_OFFSET4: four phase groups a quarter period apart instead of two; both bits: eight)
    const int j = (int)(blockIdx.x >> 3);   // index of the workgroup inside its XCD
    const int groups = (g.flags & (8 | 32)) == (8 | 32) ? 8 : (g.flags & 32) ? 4 : 2, grp = j & (groups - 1);
    if (grp != 0 && walk.count > 1) {
      // spacing of two neighbouring groups: an equal share of the item period, but no more than an epilogue that has the XCD's
      // fabric path to itself takes anyway (~BM x BN / 6 cycles when all 32 CUs store at once): with a long K the point is only
      // that the groups' epilogues do not coincide, and a group that trails by a few K-steps still finds its panels in the L2
      const int nk0 = g.k_chunk / (BK * CFG::KT);
      const int period = nk0 * 2 * CFG::T * (CFG::MI == 32 ? 32 : 16) + CFG::BM * CFG::BN / 12;   // shader cycles, roughly
      const int spacing = min(period / groups, CFG::BM * CFG::BN / 6);
#pragma clang loop unroll(disable)
      for (int c = 0; c < spacing * grp; c += 1024) __builtin_amdgcn_s_sleep(16);
    }
  }
  if constexpr (CFG::WGS == 2) {
    if (g.flags & 16) {
      unsigned hw;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      if (hw & 1u) __builtin_amdgcn_s_setprio(2);
    }
  }

  // fragment lane mapping (hgemm_kernel_sp.hpp): MI = 16: row lane & 15, 16-B chunk 4h + (lane >> 4);
  // MI = 32: row lane & 31, chunk 4h + 2u + (lane >> 5); the image's swizzle is keyed on (row >> 1) & 7
  const int lr = lane & (MI - 1), lq = lane / MI, sw = (lr >> 1) & 7;
  int foff[2][CFG::KS];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int u = 0; u < CFG::KS; ++u) foff[h][u] = lr * ROW_BYTES + (((h * 4 + u * 2 + lq) ^ sw) << 4);
  const int a_base_off = wave_m * CFG::TM * ROW_BYTES;
  const int b_base_off = BM * ROW_BYTES + wave_n * CFG::TN * ROW_BYTES;

  sp_reserve_agprs<CFG::AGPRS>();

  // ---- the two LDS-DMA streams (A: three tiles ahead of the MFMAs, B: two) ---------------------------------
  __amdgpu_buffer_rsrc_t rsA, rsB;
  // per-lane source offsets: lane's row of the wave's q-th 8-row block (row (wave + q * NW) * 8 + lane / 8; the 16-B chunk
  // is the LDS image's swizzle, keyed on the row block's parity = the wave's, NW being even): the same for every work
  // item of the launch
  static_assert(CFG::NW % 2 == 0, "the swizzle key of a wave's row blocks must not depend on the piece");
  const uint32_t chunk0 = (uint32_t)((lane & 7) ^ (((wave & 1) << 2) | (lane >> 4)));
  uint32_t voffA[CFG::PA], voffB[CFG::PB];
#pragma unroll
  for (int q = 0; q < CFG::PA; ++q) voffA[q] = ((uint32_t)((wave + q * CFG::NW) * 8 + (lane >> 3)) * (uint32_t)g.lda + chunk0 * 8u) * 2u;
#pragma unroll
  for (int q = 0; q < CFG::PB; ++q) voffB[q] = ((uint32_t)((wave + q * CFG::NW) * 8 + (lane >> 3)) * (uint32_t)g.ldb + chunk0 * 8u) * 2u;
  struct Cursor { uint32_t kbyte; int item, kt, nk; uint32_t kbyte0; int kwrap; } cur[2];   // (kbyte0, kwrap: kstagger variant only)
  int nxt_item = -1, nxt_m0 = 0, nxt_n0 = 0, nxt_kb = 0, nxt_nk = 0;   // tile coordinates of the item the streams enter next
  SQ_LOAD_ITEM(0, 0);
  SQ_LOAD_ITEM(1, 0);
  HGEMM_TL_STAMP(smem + CFG::LDS_BYTES + 64 + STAGED_BYTES, 0, tid);   // arguments read, coordinates and offsets computed
  // prologue: A(0), B(0) -> stage 0, A(1), B(1) -> stage 1
  // The accumulators are cleared BETWEEN the pieces: round-3 timeline, 4096^3: the 32 pieces take ~3.7k cycles to issue
  // (the address path accepts a cold piece every ~100 cycles; the first tile has landed 150 cycles after the last piece is
  // out) and the 256 v_accvgpr_write another ~1k behind them -- the VALU work fits into the issue stalls.
  constexpr int NZ = FM * FN * NQ, NPIECE = 2 * (CFG::NJA + CFG::NJB);
  int zi = 0, pi = 0;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int p = 0; p < CFG::NJA; ++p) {
      sq_issue_piece<CFG, 0>(rsA, voffA, smem + s * CFG::STAGE_BYTES, wave, p, cur[0].kbyte);
      for (++pi; zi < (pi * NZ) / NPIECE; ++zi) sp_zero_acc(zi);
    }
#pragma unroll
    for (int p = 0; p < CFG::NJB; ++p) {
      sq_issue_piece<CFG, 1>(rsB, voffB, smem + s * CFG::STAGE_BYTES, wave, p, cur[1].kbyte);
      for (++pi; zi < (pi * NZ) / NPIECE; ++zi) sp_zero_acc(zi);
    }
    SQ_ADVANCE(0);
    SQ_ADVANCE(1);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (; zi < NZ; ++zi) sp_zero_acc(zi);
  HGEMM_TL_STAMP(smem + CFG::LDS_BYTES + 64 + STAGED_BYTES, 1, tid);   // both tiles issued, accumulators cleared
  wait_vmcnt<CFG::NJA + CFG::NJB>();   // tile 0 landed (tile 1 may fly)
  __builtin_amdgcn_s_barrier();
  HGEMM_TL_STAMP(smem + CFG::LDS_BYTES + 64 + STAGED_BYTES, 2, tid);   // tile 0 landed for every wave

  // fragment sets: X = A first half, Y / Z = A second half (alternating), U = B first half, V = B second half
  f16x8 fX[NFA], fY[NFA], fZ[NFA], fU[NFB], fV[NFB];
#pragma unroll
  for (int r = 0; r < NFA; ++r) {
    fX[r] = *(const f16x8*)(SQ_FRAG(smem, a_base_off, 0, r / FM) + (r % FM) * MI * ROW_BYTES);
    fY[r] = *(const f16x8*)(SQ_FRAG(smem, a_base_off, 1, r / FM) + (r % FM) * MI * ROW_BYTES);
  }
#pragma unroll
  for (int r = 0; r < NFB; ++r) fU[r] = *(const f16x8*)(SQ_FRAG(smem, b_base_off, 0, r / FN) + (r % FN) * MI * ROW_BYTES);
  // the A region of stage 0 is consumed: the early pieces of A(2) go there (sync = the "Q" of a virtual K-step -1); its late
  // pieces follow in interval A of the first K-step, which then moves the A stream on
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(kWaitLgkm0);   // (the builtin form: the compiler then places no waits of its own in the K loop)
  sp_sync();
#pragma unroll
  for (int p = 0; p < CFG::DA; ++p) sq_issue_piece<CFG, 0>(rsA, voffA, smem, wave, p, cur[0].kbyte);

  int step = 0;               // global K-step of this workgroup's stream: stage = step & 1
  HGEMM_TL_STAMP(smem + CFG::LDS_BYTES, 2, tid);
#pragma clang loop unroll(disable)
  for (int item = 0; item < walk.count; ++item) {
    const TileCoord tc = map_logical(g, walk.base + walk.first + item * walk.stride, BM, BN);
    const int nk = __builtin_amdgcn_readfirstlane(KTAIL ? sq_k_items(g, tc) / (BK * CFG::KT) : tc.nk / CFG::KT);
    HGEMM_TL_STAMP(smem + CFG::LDS_BYTES, 3, tid);
    // hot loop, two K-steps per trip: both streams stay inside this work item (in K-step t the A stream moves on to tile
    // t+3 and the B stream to tile t+3 behind it: the trip's last move, to t+4, must stay inside the item; t + 5 < nk keeps
    // one more step of margin, as the round-2 plan needed), so moving them on is a scalar add
    int t = 0;
#pragma clang loop unroll(disable)
    for (; t + 5 < nk; t += 2) {
      SQ_K_STEP(fY, fZ, SQ_STEP_CURSOR(0), SQ_STEP_CURSOR(1));
      SQ_K_STEP(fZ, fY, SQ_STEP_CURSOR(0), SQ_STEP_CURSOR(1));
    }
    // last (up to) six K-steps: the streams may cross into the next work item
#pragma clang loop unroll(disable)
    for (; t + 1 < nk; t += 2) {
      SQ_K_STEP(fY, fZ, SQ_ADVANCE(0), SQ_ADVANCE(1));
      SQ_K_STEP(fZ, fY, SQ_ADVANCE(0), SQ_ADVANCE(1));
    }
    if (t < nk) {   // odd K-step count: one more step, then put the next tile's slice-1 fragments back into Y
      // This block is a control-flow merge (entered from the prologue, the hot loop or the tail loop) and ends with a register
      // copy: hipcc may move fragment registers around it with VALU instructions.  The MFMAs are asm statements, so its hazard
      // recognizer does not see them: an MFMA that reads a VGPR within two wait states of a VALU write gets the OLD value
      // (found with a 192 x 192 member, DESIGN.md section 4.8: the compiler moved A fragment 5 from v[2:5] to v[0:3] right in
      // front of its first MFMA -- one output tile wrong whenever the K-step count was odd).  sq_settle makes every fragment
      // set live in its final registers here and spends the two wait states; tests/test_build_audit.py checks the condition on
      // every MFMA of every persistent kernel.
      sq_settle(fX, fY, fU);
      SQ_K_STEP(fY, fZ, SQ_ADVANCE(0), SQ_ADVANCE(1));
#pragma unroll
      for (int r = 0; r < NFA; ++r) fY[r] = fZ[r];
      sq_settle(fX, fY, fU);
    }
    if constexpr (KTAIL) {
      const int k_items = sq_k_items(g, tc), k_full = nk * (BK * CFG::KT);
      if (k_full < k_items) {   // (wave-uniform)
        __builtin_amdgcn_sched_barrier(0);
        direct_k_tail<FM, FN>(g, tc.m0, tc.n0, wave_m * CFG::TM, wave_n * CFG::TN, tc.k_begin + k_full, tc.k_begin + k_items, lane,
                              [](int i, int j, const f16x8& b, const f16x8& a) __attribute__((always_inline)) { sp_mfma(i * FN + j, b, a); });
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // ---- epilogue of this work item (as family "s"): unit by unit (MI = 16: one fragment row, MI = 32: one tile) ---
    HGEMM_TL_STAMP(smem + CFG::LDS_BYTES, 4, tid);
    sp_mfma_drain();
    const bool rezero = item + 1 < walk.count;
    const __amdgpu_buffer_rsrc_t rsP = fused_rsrc(g);
    const int m_wave = tc.m0 + wave_m * CFG::TM, n_wave = tc.n0 + wave_n * CFG::TN;
    (void)rsP; (void)m_wave; (void)n_wave;
    if constexpr (STAGED) {
      __builtin_amdgcn_sched_barrier(0);
      sp_epilogue_staged<CFG>(g, m_wave, n_wave, rezero, smem + CFG::LDS_BYTES + 64 + wave * SP_STAGED_BYTES_PER_WAVE);
    } else if constexpr (MI == 16) {
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        __builtin_amdgcn_sched_barrier(0);
        f32x4 row[FN];
#pragma unroll
        for (int j = 0; j < FN; ++j) row[j] = sp_read_acc(i * FN + j);
        if (rezero) {
#pragma unroll
          for (int j = 0; j < FN; ++j) sp_zero_acc(i * FN + j);
        }
        if constexpr (EPI == SP_EPI_FUSED) {
#pragma unroll
          for (int j = 0; j < FN; ++j) fused_store(rsP, fused_off<CFG::THREADS>(tc.item, BM * BN, i * FN + j, tid), row[j]);
        } else {
          if (!HGEMM_DBG(g, 2))
            store_tile_row<16, FN, CFG::TM, CFG::TN, EPI == SP_EPI_SLAB, EPI == SP_EPI_SLAB ? -1 : EPI>(g, tc, wave_m, wave_n, lane, i, row);
        }
      }
    } else {   // one 32x32 tile (16 registers) live at a time, as family "s"
#pragma unroll
      for (int x = 0; x < FM * FN; ++x) {
        __builtin_amdgcn_sched_barrier(0);
        const int i = x / FN, j = x % FN;
        f32x4 qd[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) qd[q] = sp_read_acc(x * 4 + q);
        if (rezero) {
#pragma unroll
          for (int q = 0; q < 4; ++q) sp_zero_acc(x * 4 + q);
        }
        if constexpr (EPI == SP_EPI_FUSED) {
#pragma unroll
          for (int q = 0; q < 4; ++q) fused_store(rsP, fused_off<CFG::THREADS>(tc.item, BM * BN, x * 4 + q, tid), qd[q]);
        } else if constexpr (EPI == SP_EPI_SLAB) {
          const int m = m_wave + i * 32 + (lane & 31);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int n = n_wave + j * 32 + 8 * q + 4 * (lane >> 5);
            if (m < g.M && n < g.N) *(f32x4*)(tc.slab + (size_t)(m - tc.m0) * tc.slab_ld + (n - tc.n0)) = qd[q];
          }
        } else {
          if (!HGEMM_DBG(g, 2)) sp_store_tile32<EPI>(g, m_wave + i * 32 + (lane & 31), n_wave + j * 32, lane, qd);
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (EPI == SP_EPI_FUSED) {
      if (fused_publish_and_vote(g, tc.tile, (volatile unsigned*)(smem + CFG::LDS_BYTES), tid)) {
        const int tiles = g.tiles_m * g.tiles_n;
        if constexpr (MI == 16) {
          // Round 6: the combine used to be FM x splits dependent round trips to the memory side of the fabric (a fragment row's
          // quads of ONE slab, wait, add, next slab; next row) -- 8 of them for the 2-way split of 512 x 4096 x 4096 on q128x128,
          // on top of an epilogue that no MFMA covers.  Now the first UQ slabs of a fragment row are requested together (up to 8 loads)
          // and the NEXT row's go out before this row is added up and stored: one round trip + FM issue slots for splits <= UQ.
          // Slabs are added in split order as before (slab 0 initialises); loads of u >= splits re-read the last slab and their sum is
          // discarded by a select (no branch: the compiler's waits stay counted).
          // (register budget: this row's UQ x FN quads + the next row's must fit beside the fragment sets without the compiler
          // reaching for the AGPRs that hold the accumulators -- tests/test_build_audit.py; at most 8-12 quads per batch)
          constexpr int UQ = FN <= 2 ? 4 : 2;
          if constexpr (FN < 8 && FM * FN <= 32) {
            const int S = g.splits;
            auto issue = [&](int i, f32x4 (&buf)[UQ][FN]) __attribute__((always_inline)) {
#pragma unroll
              for (int u = 0; u < UQ; ++u) {
                const int su = min(u, S - 1);
#pragma unroll
                for (int j = 0; j < FN; ++j) buf[u][j] = fused_load(rsP, fused_off<CFG::THREADS>(su * tiles + tc.tile, BM * BN, i * FN + j, tid));
              }
            };
            f32x4 nxt[UQ][FN];
            issue(0, nxt);
#pragma unroll
            for (int i = 0; i < FM; ++i) {
              f32x4 cur[UQ][FN];
#pragma unroll
              for (int u = 0; u < UQ; ++u)
#pragma unroll
                for (int j = 0; j < FN; ++j) cur[u][j] = nxt[u][j];
              if (i + 1 < FM) issue(i + 1, nxt);
              f32x4 row[FN];
#pragma unroll
              for (int j = 0; j < FN; ++j) row[j] = cur[0][j];
#pragma unroll
              for (int u = 1; u < UQ; ++u)
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                  const f32x4 t = row[j] + cur[u][j];
                  row[j] = (u < S) ? t : row[j];
                }
              for (int sidx = UQ; sidx < S; ++sidx) {   // deeper splits than UQ (rare on these tiles): one slab at a time, as before
#pragma unroll
                for (int j = 0; j < FN; ++j) row[j] += fused_load(rsP, fused_off<CFG::THREADS>(sidx * tiles + tc.tile, BM * BN, i * FN + j, tid));
              }
              store_tile_row<16, FN, CFG::TM, CFG::TN, false, -1>(g, tc, wave_m, wave_n, lane, i, row);
            }
          } else {
            // the members with more than 128 accumulator registers (256 x 256, 256 x 192, 192 x 256, 128 x 256) have no room for a
            // second batch of quads in flight in their fused + K-tail variants (the compiler would park VGPRs in the accumulators'
            // AGPRs: tests/test_build_audit.py): they keep the round-2 walk, one slab of one fragment row per round trip
#pragma unroll 1
            for (int i = 0; i < FM; ++i) {
              f32x4 row[FN];
              for (int sidx = 0; sidx < g.splits; ++sidx) {
#pragma unroll
                for (int j = 0; j < FN; ++j) {
                  const f32x4 v = fused_load(rsP, fused_off<CFG::THREADS>(sidx * tiles + tc.tile, BM * BN, i * FN + j, tid));
                  row[j] = (sidx == 0) ? v : row[j] + v;
                }
              }
              store_tile_row<16, FN, CFG::TM, CFG::TN, false, -1>(g, tc, wave_m, wave_n, lane, i, row);
            }
          }
        } else {
#pragma unroll 1
          for (int x = 0; x < FM * FN; ++x) {
            const int i = x / FN, j = x % FN;
            f32x4 qd[4];
            for (int sidx = 0; sidx < g.splits; ++sidx) {
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                const f32x4 v = fused_load(rsP, fused_off<CFG::THREADS>(sidx * tiles + tc.tile, BM * BN, x * 4 + q, tid));
                qd[q] = (sidx == 0) ? v : qd[q] + v;
              }
            }
            sp_store_tile32<-1>(g, m_wave + i * 32 + (lane & 31), n_wave + j * 32, lane, qd);
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    HGEMM_TL_STAMP(smem + CFG::LDS_BYTES, 5, tid);
  }
  wait_vmcnt<0>();  // redundant tail pieces must not outlive the workgroup's LDS allocation
#ifdef HGEMM_TIMELINE
  HGEMM_TL_STAMP(smem + CFG::LDS_BYTES, 6, tid);
  HGEMM_TL_REALTIME(smem + CFG::LDS_BYTES, 7, tid);
  if (g.timeline != nullptr && tid == 0) {
    // words 0..7: entry, entry (100 MHz clock), pipeline primed, last item's loop start, its last MFMA, its last store
    // issued, everything acknowledged, exit (100 MHz clock); 8: XCC id << 32 | HW_ID; 9: K-steps walked; 10: work items
    unsigned long long* tl = g.timeline + (size_t)blockIdx.x * HGEMM_TL_WORDS;
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
#pragma unroll
    for (int w = 0; w < 8; ++w) tl[w] = hgemm_tl_get(smem + CFG::LDS_BYTES, w);
#pragma unroll
    for (int w = 0; w < 3; ++w) tl[11 + w] = hgemm_tl_get(smem + CFG::LDS_BYTES + 64 + STAGED_BYTES, w);   // 11..13: head split
    tl[8] = ((unsigned long long)xcc << 32) | hw; tl[9] = (unsigned long long)step; tl[10] = (unsigned long long)walk.count;
  }
#endif
#endif  // __HIP_DEVICE_COMPILE__
}

#undef SQ_LOAD_ITEM
#undef SQ_ADVANCE
#undef SQ_STEP_CURSOR
#undef SQ_K_STEP
#undef SQ_FRAG

}  // namespace hgemm_mi355x
